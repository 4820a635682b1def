# interleaved arms of bench.py --tune on one box: bash tools/ab_tune_arms.sh "gemm_variant=0 gemm_variant=6" [rounds]
arms=${1:-"gemm_variant=0"}; rounds=${2:-2}
for i in $(seq $rounds); do
  for a in $arms; do
    timeout -k 10 300 python bench.py --tune "$a" --steps 6 --warmup 2 --cpu-steps 0 --no-latency --no-nar --no-fp8 --no-nq8 --no-vctk --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', round(d['value']), round(d['ms_per_step'],2))" || exit 1
  done
done
