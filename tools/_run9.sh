mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bench_path.py tests/test_gpu_measured_paths.py tests/test_gpu_batch_sweep.py tests/test_gpu_parity.py -q -x -p no:cacheprovider > gpurun_out/r9_tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r9_tests.log
