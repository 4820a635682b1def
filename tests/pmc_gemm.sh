# PMC deep-dive on the GEMM / attention micro-kernels (separate passes, counters only)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc2/counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/set$i -- python3 $GRAFT_REPO_ROOT/tests/bench_kernels.py --quick > $GRAFT_REPO_ROOT/gpurun_out/pmc2/set$i.log 2>&1
  echo "set$i rc=$?"
done
