mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider -k "mx" > gpurun_out/r3h_mx_tests.log 2>&1; rc=$?; tail -20 gpurun_out/r3h_mx_tests.log; echo "mx tests rc=$rc"
if [ $rc -le 1 ]; then
timeout -k 10 300 python tests/ab_fp8.py > gpurun_out/r3h_ab_fp8.txt 2>&1; rc=$?; cat gpurun_out/r3h_ab_fp8.txt; echo "ab rc=$rc"
fi
if [ $rc -le 1 ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "fp8" > gpurun_out/r3h_fp8_parity.log 2>&1; rc=$?; tail -15 gpurun_out/r3h_fp8_parity.log; echo "fp8 parity rc=$rc"; cat gpurun_out/parity_report.json
fi
