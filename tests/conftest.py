import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # No test streams more than a few hundred MiB through the out-of-core engines.  Cap what they may page-lock on the
    # host (planner.host_available, csrc/tvdn_stream.hip): a test that asks for more by mistake is refused with an
    # error instead of taking the machine down (this is what cost round 2 its GPU).
    os.environ.setdefault("TVDN_HOST_LIMIT", "64G")
    # a fresh checkout has no built libraries (they are git-ignored): compile them once (hipcc cross-compiles
    # for gfx950 without a GPU; the oracle needs gcc only)
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "cytvdn_amd", "libtvdn_hip.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cytvdn_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libtvdn_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle"])


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); builds its C library on first use."""
    from oracle import oracle as orc

    orc.build()
    orc.set_threads(1)
    return orc
