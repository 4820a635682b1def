mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider -k "mx or row_panel" > gpurun_out/r3d_mx_tests.log 2>&1; rc=$?; tail -30 gpurun_out/r3d_mx_tests.log; echo "mx tests rc=$rc"
if [ $rc -le 1 ]; then
timeout -k 10 300 python tests/ab_fp8.py > gpurun_out/r3d_ab_fp8.txt 2>&1; rc=$?; cat gpurun_out/r3d_ab_fp8.txt; echo "ab rc=$rc"
fi
if [ $rc -le 1 ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "fp8 or row_panel" > gpurun_out/r3d_fp8_parity.log 2>&1; rc=$?; tail -15 gpurun_out/r3d_fp8_parity.log; echo "fp8 parity rc=$rc"; cat gpurun_out/parity_report.json
fi
if [ $rc -le 1 ]; then
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-nar --no-latency > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; rc=$?; tail -2 gpurun_out/r3d_bench.err; echo "bench rc=$rc"
python -c "import json;d=json.load(open('gpurun_out/r3d_bench.json'));print(d['value'], d['fp8_fast_path'])"
fi
