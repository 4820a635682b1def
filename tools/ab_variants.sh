# interleaved arms of bench.py over library variants on one box: bash tools/ab_variants.sh "p00 p30 default" [rounds] [extra bench args]
# (variants: tools/build_variant.py; "default" = the product library)
names=${1:-"default"}; rounds=${2:-3}; shift 2 2>/dev/null
for i in $(seq $rounds); do
  for n in $names; do
    if [ "$n" = default ]; then unset D3PM_HIP_LIB; else export D3PM_HIP_LIB=$PWD/tts-with-diffusion-model_amd/lib/variants/libd3pm_$n.so; fi
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-steps 0 --no-latency --no-nar --no-fp8 --no-nq8 --no-vctk --no-kernel-events "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', round(d['value']), round(d['ms_per_step'],2), d.get('p50_utterance_latency_ms'))" || exit 1
  done
done
