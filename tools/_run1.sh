mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 tools/probe_mx.hip -o /tmp/probe_mx > gpurun_out/probe_build.log 2>&1 && timeout -k 10 60 /tmp/probe_mx > gpurun_out/probe_mx.txt 2>&1
echo "probe rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_measured_paths.py -q -m gpu -p no:cacheprovider > gpurun_out/r3a_tests.log 2>&1; rc=$?; tail -15 gpurun_out/r3a_tests.log; echo "tests rc=$rc"
if [ $rc -le 1 ]; then
timeout -k 10 400 python bench.py --config vctk --steps 5 --warmup 2 --cpu-steps 0 > gpurun_out/r3a_vctk_b32.json 2> gpurun_out/r3a_vctk_b32.err; rc=$?; tail -2 gpurun_out/r3a_vctk_b32.err; echo "vctk32 rc=$rc"
fi
if [ $rc -le 1 ]; then
timeout -k 10 300 python bench.py --config vctk --batch 1 --steps 10 --warmup 2 --cpu-steps 0 > gpurun_out/r3a_vctk_b1.json 2> gpurun_out/r3a_vctk_b1.err; rc=$?; echo "vctk1 rc=$rc"
fi
if [ $rc -le 1 ]; then
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3a_vctk_prof -- python3 $GRAFT_REPO_ROOT/bench.py --config vctk --steps 1 --warmup 1 --cpu-steps 0 --no-latency --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/r3a_vctk_prof.log 2>&1; rc=$?; cd $GRAFT_REPO_ROOT; echo "prof rc=$rc"
find gpurun_out/r3a_vctk_prof -name "*kernel_trace*" -delete
fi
cat gpurun_out/probe_mx.txt
