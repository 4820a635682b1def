# A/B of d3pm_tuning fields on the headline workload: bash tools/ab_tune.sh "row_panel=0" "row_panel=3" ...   (one bench line per arm)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for arm in "$@"; do
  tag=$(echo "$arm" | tr ',=' '__')
  timeout -k 10 250 python bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-latency --no-nar --no-fp8 --tune "$arm" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || exit 1
  python - "$arm" gpurun_out/ab_$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
kc = d.get("kernel_classes", {})
n_it = d["steps"] * 6.0                                  # every 16th of the 99 iterations carries events: t = 96, 80, .., 16
per = {k: round(v["launches_timed"] * v["avg_launch_us"] / n_it, 1) for k, v in kc.items()}      # us per sampled iteration
print("%-10s %8.0f tok/s %7.1f ms/step  us/iteration: %s  sum %.0f" % (sys.argv[1], d["value"], d["ms_per_step"], per, sum(per.values())))
PY
done
