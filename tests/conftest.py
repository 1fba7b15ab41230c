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
    # The tests move small NumPy arrays with torch (`x.cuda()`, `t.cpu()`): hipMemcpy from / to pageable memory, for which the
    # ROCm runtime pins the caller's pages in place and caches the pin by address and size.  When glibc trims the heap such a pin
    # dies with its pages, and the next copy from an array that lands on the same address takes the GPU down with a memory access
    # fault (once in nine suites: profiles/r06_abort_found.txt).  The library itself no longer hands pageable memory to the runtime
    # (csrc/tvdn_hostio.hip); for torch's own copies in the tests the heap is simply never trimmed and arrays up to 32 MiB stay in it.
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-1, 2 ** 31 - 1)      # M_TRIM_THRESHOLD
        libc.mallopt(-3, 32 << 20)         # M_MMAP_THRESHOLD (its maximum)
    except Exception:
        pass
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
