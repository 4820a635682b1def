import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tts-with-diffusion-model_amd")
for p in (PKG, ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from vall_e.vall_e import _hip
    return _hip.lib()


def same_platform_as_golden() -> bool:
    """Bit-exact float comparisons of transformer outputs only hold on the CPU the fixtures were made on."""
    from make_golden import fingerprint
    try:
        with open(os.path.join(GOLDEN, "FINGERPRINT.txt")) as f:
            return f.read().strip() == fingerprint()
    except OSError:
        return False


@pytest.fixture(scope="session", autouse=True)
def _parity_report_file():
    """The numbers the -m gpu tests measured (tests/util.py:REPORT) -> gpurun_out/parity_report.json."""
    yield
    import json
    from util import REPORT
    if REPORT:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
