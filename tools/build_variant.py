"""A/B builds of the product library with other compile-time constants:

    python tools/build_variant.py NAME -DD3PM_EPI_PRIO_BIG=0 [-D...]   ->  tts-with-diffusion-model_amd/lib/variants/libd3pm_NAME.so

Run an arm with D3PM_HIP_LIB=<that path> python bench.py ... (vall_e/vall_e/_hip.py loads the library the variable names).  The variants
are git-ignored build products like the library itself; they travel to the GPU box with the snapshot."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g

name, flags = sys.argv[1], tuple(sys.argv[2:])
out = os.path.join(g.PKG, "lib", "variants", f"libd3pm_{name}.so")
g._build_library(out, flags, force=True)
print(out)
