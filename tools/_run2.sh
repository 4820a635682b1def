mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 tools/probe_mx.hip -o /tmp/probe_mx > gpurun_out/probe_build.log 2>&1 && timeout -k 10 120 /tmp/probe_mx > gpurun_out/probe_mx2.txt 2>&1
echo "probe rc=$?"; grep exp8 gpurun_out/probe_mx2.txt
timeout -k 10 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r3b_tests.log 2>&1; rc=$?; tail -8 gpurun_out/r3b_tests.log; echo "tests rc=$rc"
if [ $rc -le 1 ]; then
timeout -k 10 600 python -m pytest tests/ab_bit_identity.py -q -m gpu -p no:cacheprovider > gpurun_out/r3b_ab_tests.log 2>&1; rc=$?; tail -8 gpurun_out/r3b_ab_tests.log; echo "ab tests rc=$rc"
fi
if [ $rc -le 1 ]; then
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-nar > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; rc=$?; tail -2 gpurun_out/r3b_bench.err; echo "bench rc=$rc"
fi
