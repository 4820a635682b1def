# final measurements of a round: bash tools/final_measure.sh <tag>   (GPU box; outputs under gpurun_out/<tag>_*)
tag=${1:-final}
: "${GRAFT_REPO_ROOT:=$(pwd)}"; export GRAFT_REPO_ROOT
mkdir -p gpurun_out
ok=1
step() { [ $ok -eq 1 ] || return; echo "== $1"; shift; "$@"; rc=$?; echo "rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then ok=0; fi; }
t_tests() { timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/${tag}_gpu_tests.log 2>&1; r=$?; tail -4 gpurun_out/${tag}_gpu_tests.log; cp gpurun_out/parity_report.json gpurun_out/${tag}_parity_report.json; return $r; }
t_ab() { timeout -k 10 600 python -m pytest tests/ab_bit_identity.py -q -m gpu -p no:cacheprovider > gpurun_out/${tag}_ab_tests.log 2>&1; r=$?; tail -3 gpurun_out/${tag}_ab_tests.log; return $r; }
t_bench() { timeout -k 10 900 python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; r=$?; tail -3 gpurun_out/${tag}_bench.err; return $r; }
t_b1() { timeout -k 10 300 python bench.py --batch 1 --steps 10 --warmup 2 --cpu-steps 0 --no-nar --no-fp8 --no-nq8 --no-vctk > gpurun_out/${tag}_bench_b1.json 2> gpurun_out/${tag}_bench_b1.err; }
t_vctk() { timeout -k 10 400 python bench.py --config vctk --steps 5 --warmup 2 --cpu-steps 0 > gpurun_out/${tag}_bench_vctk_b32.json 2> gpurun_out/${tag}_bench_vctk.err; }
t_micro() { timeout -k 10 400 python tests/bench_kernels.py > gpurun_out/${tag}_microbench.txt 2> gpurun_out/${tag}_microbench.err; timeout -k 10 200 python tests/ab_fp8.py > gpurun_out/${tag}_ab_mx_fp8.txt 2>&1; }
t_prof() { cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-latency --no-nar --no-nq8 --no-vctk --no-fp8 --no-kernel-events --profile-iters 33 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1; r=$?; cd $GRAFT_REPO_ROOT; find gpurun_out/${tag}_prof -name "*kernel_trace*" -delete; return $r; }
t_prof1() { cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 5 --warmup 1 --cpu-steps 0 --no-latency --no-nar --no-fp8 --no-nq8 --no-vctk --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof1.log 2>&1; r=$?; cd $GRAFT_REPO_ROOT; find gpurun_out/${tag}_prof1 -name "*kernel_trace*" -delete; return $r; }
t_pmc() { cd /tmp && export TMPDIR=/tmp; r=0
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-latency --no-nar --no-nq8 --no-vctk --no-fp8 --profile-iters 3 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$c.log 2>&1 || r=$?
    find $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$c -name "*kernel_trace*" -delete
  done; cd $GRAFT_REPO_ROOT; return $r; }
for s in ${STAGES:-tests ab bench b1 vctk micro prof prof1 pmc}; do step $s t_$s; done
