import os
import sys

import pytest

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  When a test hands torch
# device pointers to librays1.so both must share ONE runtime: importing torch first makes
# librays1's DT_NEEDED libamdhip64.so resolve to the copy torch already loaded.
try:
    import torch  # noqa: F401
except Exception:  # torch is plumbing for the device-pointer tests only
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# tools/entry_ab.sh and friends: run the GPU parity tests against another build of the library (an experiment's librays1_tuning.so)
if os.environ.get("R1_TEST_LIB"):
    from rays1bench_amd import binding as _binding
    _binding.set_lib_path(os.environ["R1_TEST_LIB"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
