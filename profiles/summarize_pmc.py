"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files (two separate passes, as
MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2) into per-launch
L2-miss traffic per kernel class.  gfx950 corrections of that guide: both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads -> doubled.

    python profiles/summarize_pmc.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > profiles/<name>.json
"""
import collections
import csv
import glob
import json
import sys

import re

CLASSES = {"gemm": "gemm_mfma", "gemm_layernorm": "gemm_mfma_big", "gemm_mx": "gemm_mx_big", "attention": "attn", "layernorm": "layernorm",
           "sample": "posterior_sample", "embed": "embed_rows"}


def fused_gemm(name):
    """gemm_mfma_big<T, EPI, WM, WN, MODE, FUSE>: FUSE with a LayerNorm bit = the row-panel launches (projection + LayerNorms);
    FUSE = 1 alone is the dual out-projection on ordinary tiles (a plain GEMM launch)."""
    m = re.search(r"gemm_mfma_bigI\w+?Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EE", name)
    if m:
        return (int(m.group(5)) & ~1) != 0
    m = re.search(r"gemm_mfma_big<([^>]*)>", name)
    if m:
        ints = re.findall(r"\b\d+\b", m.group(1))
        return len(ints) >= 5 and int(ints[-1]) != 0
    return False


def classify(name):
    if "gemm_mfma" in name:
        return "gemm_layernorm" if fused_gemm(name) else "gemm"
    for cls, needle in CLASSES.items():
        if cls not in ("gemm", "gemm_layernorm") and needle in name:
            return cls
    return None


# per-kernel rows beside the classes (which attention kernel re-fetches)
KERNELS = {"attention_self_attn32p": "attn32p_hd64", "attention_cross_pair_attn32": "attn32_cross_hd64", "attention_self_16x16": "attn_mfma_hd64",
           "gemm_folded_ln": None, "gemm_row_moments": None}


def kernel_rows(name):
    rows = [k for k, needle in KERNELS.items() if needle and needle in name]
    m = re.search(r"gemm_mfma_bigI\w+?Li(\d+)E", name)
    if m:
        epi = int(m.group(1))
        if epi & 64:
            rows.append("gemm_folded_ln")
        if epi & 128:
            rows.append("gemm_row_moments")
    return rows


def load(directory, counter):
    import os
    path = max(glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        cls = classify(r["Kernel_Name"])
        if cls:
            agg[cls][0] += 1
            agg[cls][1] += float(r["Counter_Value"])
        for k in kernel_rows(r["Kernel_Name"]):
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for cls in list(CLASSES) + list(KERNELS):
        if fetch[cls][0] == 0:
            continue
        f = fetch[cls][1] / fetch[cls][0] * 1024.0
        w = write[cls][1] / max(write[cls][0], 1) * 1024.0
        out[cls] = {"launches_profiled": fetch[cls][0], "fetch_bytes_per_launch_raw": f,
                    "fetch_bytes_per_launch_corrected": 2.0 * f, "write_bytes_per_launch": w,
                    "traffic_bytes_per_launch": 2.0 * f + w}
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out["_build_id"] = bench.build_id()          # bench.py only reports this traffic beside timings of the same build
    out["_method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 1 "
                      "--warmup 0 --profile-iters 3 (B=32 libritts bf16); KiB -> bytes; FETCH_SIZE x2 (gfx950)")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
