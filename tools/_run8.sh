mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r8_gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r8_gpu_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --cpu-steps 0 --no-nar --no-nq8 --no-fp8 > gpurun_out/r8_bench.json 2> gpurun_out/r8_bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-steps 0 --no-nar --no-nq8 --no-fp8 --tune attn_query_groups=2 > gpurun_out/r8_bench_qg2.json 2> gpurun_out/r8_bench_qg2.err; echo "bench qg2 rc=$?"
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-steps 0 --no-nar --no-nq8 --no-fp8 > gpurun_out/r8_bench_b.json 2> gpurun_out/r8_bench_b.err; echo "bench rc=$?"
