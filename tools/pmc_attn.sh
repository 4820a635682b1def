# PMC counters of the self-attention kernel at the bench shape (separate passes): bash tools/pmc_attn.sh [arm]
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; cd /tmp; export TMPDIR=/tmp
ARM=${1:-2}
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i + 1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$i -- python3 $GRAFT_REPO_ROOT/tests/ab_attn.py $ARM > $GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$i.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob("gpurun_out/pmc_attn_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "attn_mfma_hd64" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-28s %14.0f per launch (%d launches)" % (k, v / n, n))
PY
