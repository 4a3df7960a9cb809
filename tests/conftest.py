import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURES = os.path.join(ROOT, "tests", "fixtures")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no librdyn_hip.so (build artefacts are git-ignored): build it once (hipcc cross-compiles for
    gfx950 without a GPU, ~2 min) so that the C-ABI / ingest tests of the CPU suite do not depend on a previous build()."""
    import subprocess
    lib = os.path.join(ROOT, "rosdyn_amd", "librdyn_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.call(["make", "-C", os.path.join(ROOT, "rosdyn_amd", "csrc"), "-j", str(min(8, os.cpu_count() or 1))],
                        stdout=subprocess.DEVNULL)


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load_golden(name):
    import numpy as np
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: g[k] for k in g.files}
    d["urdf_path"] = os.path.join(FIXTURES, str(d["urdf"]))
    d["base"], d["tool"] = str(d["base"]), str(d["tool"])
    d["inputs"] = [str(x) for x in d["inputs"]] or None
    d["gravity"] = tuple(float(x) for x in d["gravity"])
    return d


@pytest.fixture(scope="session")
def fixtures_dir():
    return FIXTURES
