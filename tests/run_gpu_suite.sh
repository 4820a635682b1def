# GPU-box driver: kernel numerics, parity, smoke, bench, rocprof.  Usage: bash tests/run_gpu_suite.sh [stage...]
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
mkdir -p gpurun_out
STAGES="${@:-kernels parity smoke bench prof}"
ok=1
for st in $STAGES; do
  [ $ok -eq 1 ] || break
  case $st in
    kernels) timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_kernels.log 2>&1; rc=$?; tail -15 gpurun_out/pytest_kernels.log;;
    fold)    timeout -k 10 900 python -m pytest tests/test_gpu_fold.py tests/test_gpu_batch_sweep.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_fold.log 2>&1; rc=$?; tail -40 gpurun_out/pytest_fold.log;;
    all)     timeout -k 10 1150 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_all.log 2>&1; rc=$?; tail -40 gpurun_out/pytest_all.log;;
    benchq)  timeout -k 10 600 python bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-nar --no-nq8 --no-fp8 --no-vctk 2> gpurun_out/benchq.err | tee gpurun_out/benchq.json; rc=${PIPESTATUS[0]}; tail -5 gpurun_out/benchq.err;;
    profab)  cd /tmp && export TMPDIR=/tmp; rc=0
             for arm in ${PROF_ARMS:-ln_fold=1 ln_fold=0}; do
               tag=$(echo "$arm" | tr ',=' '__')
               timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-latency --no-nar --no-nq8 --no-fp8 --no-vctk --no-kernel-events --profile-iters 24 --tune "$arm" ${PROF_EXTRA:-} > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1 || rc=$?
               f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
               echo "== $arm"; python3 $GRAFT_REPO_ROOT/tools/show_kernel_stats.py "$f" 22
               cp "$f" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$tag.csv
             done; cd $GRAFT_REPO_ROOT;;
    parity)  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_parity.log 2>&1; rc=$?; tail -25 gpurun_out/pytest_parity.log;;
    nar)     timeout -k 10 600 python -m pytest tests/test_gpu_nar.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_nar.log 2>&1; rc=$?; tail -25 gpurun_out/pytest_nar.log;;
    smoke)   timeout -k 10 180 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; rc=$?; tail -3 gpurun_out/smoke.log;;
    bench)   timeout -k 10 900 python bench.py --steps 2 --warmup 1 2> gpurun_out/bench.err | tee gpurun_out/bench.json; rc=${PIPESTATUS[0]}; tail -5 gpurun_out/bench.err;;
    noev)    timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-kernel-events --cpu-steps 0 --no-vctk --no-nq8 --no-latency --no-fp8 2> gpurun_out/bench_noev.err | tee gpurun_out/bench_noev.json; rc=${PIPESTATUS[0]};;
    streams) rc=0; for k in 1 2 4; do timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-vctk --no-nq8 --no-latency --no-fp8 --streams $k 2>> gpurun_out/bench_streams.err | tee -a gpurun_out/bench_streams.json || rc=$?; done;;
    dp2)     timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --batch 4 --profile-iters 3 --cpu-steps 0 --no-vctk --no-nq8 --no-latency --no-nar --no-fp8 2> gpurun_out/bench_dp2.err | tee gpurun_out/bench_dp2.json; rc=${PIPESTATUS[0]}; tail -3 gpurun_out/bench_dp2.err;;
    micro)   timeout -k 10 600 python tests/bench_kernels.py > gpurun_out/kernels.txt 2> gpurun_out/kernels.err; rc=$?; cat gpurun_out/kernels.txt;;
    pmc)     cd /tmp && export TMPDIR=/tmp; rc=0
             for c in FETCH_SIZE WRITE_SIZE; do
               timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-vctk --no-nq8 --no-latency --no-nar --no-fp8 --profile-iters 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1 || rc=$?
               echo "pmc $c rc=$rc"
             done; cd $GRAFT_REPO_ROOT; find gpurun_out/pmc_* -name "*.csv" | head;;
    prof1)   cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 5 --warmup 1 --cpu-steps 0 --no-vctk --no-nq8 --no-latency > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1; rc=$?; cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/prof1.log;;
    prof)    cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-vctk --no-nq8 --no-latency --no-nar > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; rc=$?; cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/prof.log; find gpurun_out/prof -name "*stats*" | head;;
  esac
  echo "stage $st rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then ok=0; fi
done
