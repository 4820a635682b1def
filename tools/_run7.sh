mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_batch_sweep.py -q -x -p no:cacheprovider -k "attention or utterance_zero" > gpurun_out/r7_tests_a.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r7_tests_a.log
