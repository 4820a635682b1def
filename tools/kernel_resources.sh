#!/bin/bash
# Register / scratch / occupancy of every kernel of one csrc file: bash tools/kernel_resources.sh d3pm_mfma_gemm_big.hip [grep-filter]
# (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed)
src="$(cd "$(dirname "$0")/.." && pwd)/tts-with-diffusion-model_amd/csrc/$1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden ${EXTRA_FLAGS:-} -x hip -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*) \[-Rpass", line) or re.search(r"remark: (.*) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t and cur is not None:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("d3pm::(anonymous namespace)::", "").replace("void ", "")
    if flt and not re.search(flt, n): continue
    print("%-70s VGPR %4s AGPR %4s scratch %4s occ %s" % (n[:70], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
' "$2"
