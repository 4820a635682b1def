mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_nq.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r3e_nq_tests.log 2>&1; rc=$?; tail -30 gpurun_out/r3e_nq_tests.log; echo "nq tests rc=$rc"
if [ $rc -le 1 ]; then
timeout -k 10 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider --deselect tests/test_gpu_nq.py > gpurun_out/r3e_tests.log 2>&1; rc=$?; tail -8 gpurun_out/r3e_tests.log; echo "tests rc=$rc"
fi
if [ $rc -le 1 ]; then
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-nar --no-latency > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err; rc=$?; tail -2 gpurun_out/r3e_bench.err; echo "bench rc=$rc"
python -c "import json;d=json.load(open('gpurun_out/r3e_bench.json'));print(d['value'], d['n_q8_extension'], d['fp8_fast_path']['speedup_fp8_vs_bf16_same_schedule'])"
fi
