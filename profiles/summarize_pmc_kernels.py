"""Per-kernel SQ / TCC / GRBM counters from the separate rocprofv3 --pmc passes of tests/pmc_gemm.sh (counters only, one
set per pass, as MI355X_MICROARCH.md prescribes) -> one JSON with the derived fractions.

    python profiles/summarize_pmc_kernels.py gpurun_out/pmc2 > profiles/<name>_pmc_kernels.json
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.search(r"(gemm_mfma_\w+|attn_\w+|layernorm_\w+|posterior_sample_rows|final_sample_fused|gemm_fp8_\w+|linear_tiled|attention_rows)", name)
    base = m.group(1) if m else name[:40]
    epi = re.search(r"IDF16b?Li(\d+)E", name)
    return base + (f"<epi{epi.group(1)}>" if epi and base.startswith("gemm") else "")


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            c = agg[k][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    out = {}
    for k, cs in agg.items():
        if not re.match(r"(gemm|attn|layernorm|posterior|final)", k):
            continue
        v = {c: s / n for c, (n, s) in cs.items()}          # per-launch averages
        d = {"launches": max(n for n, _ in cs.values())}
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            d["frac_wait_any"] = v.get("SQ_WAIT_ANY", 0) / wc
            d["frac_wait_inst"] = v.get("SQ_WAIT_INST_ANY", 0) / wc
            d["frac_active"] = v.get("SQ_ACTIVE_INST_ANY", 0) / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
            # busy cycles are summed over the chip's shader engines; 4 SIMDs x 256 CUs issue MFMAs
            d["mfma_busy_cycles_per_simd"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
        if "GRBM_GUI_ACTIVE" in v:
            d["gui_active_cycles_per_xcd"] = v["GRBM_GUI_ACTIVE"] / 8.0
            if "mfma_busy_cycles_per_simd" in d:
                d["mfma_pipe_busy_frac"] = d["mfma_busy_cycles_per_simd"] / d["gui_active_cycles_per_xcd"]
        if "TCC_HIT_sum" in v:
            d["l2_hit_rate"] = v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0), 1)
        if "SQ_LDS_IDX_ACTIVE" in v:
            d["lds_bank_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v["SQ_LDS_IDX_ACTIVE"], 1)
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVES"):
            if c in v:
                d[c] = v[c]
        out[k] = d
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
