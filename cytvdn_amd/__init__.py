"""cytvdn_amd -- MI355X-native anisotropic TV denoising behind the cyTVDN API.

`import cytvdn_amd as tv` gives the names `import cyTVDN as tv` gives (reference
cyTVDN/__init__.py:1): denoise4D, denoise3D, check_memory and the kernel-level
accumulator_update_* / datacube_update_* / sum_square_error_* functions.  All arithmetic runs in
hand-written HIP kernels for gfx950 (libtvdn_hip.so, C ABI in include/tvdn.h); there is no CPU
fallback: compute calls raise when the library or the GPU is missing.

Beyond the reference's names: `plan_run` (HBM capacity planner: engine + slab count), `denoise_file` (memory-mapped
file to file, the I/O half of the reference's cyTVMPI), `cytvdn_amd.distributed.denoise_slabs` (one slab per GPU),
`warm_up` (optional: the one-time set-up of a process's first call, paid while the caller still loads its data).
"""
from .cubeio import denoise_file
from .driver import check_memory, denoise3D, denoise4D
from .planner import plan_run
from .kernels import (accumulator_update_3D, accumulator_update_3D_FISTA, accumulator_update_4D,
                      accumulator_update_4D_FISTA, datacube_update_3D, datacube_update_4D,
                      iso_accumulator_update_4D, iso_accumulator_update_4D_FISTA, sum_square_error_3D,
                      sum_square_error_4D)

__version__ = "0.6.0"


def warm_up(device: int = 0, background: bool = False):
    """Optional.  The first `denoise3D/4D` of a process pays 0.1-0.15 s once: the pinned staging lanes, the first stream of each
    priority class, the library's code object on the device, the allocator's canary (tvdn.h tvdn_warm_up).  A script that is
    about to read a cube from disk calls `warm_up(background=True)` first and pays it meanwhile; the returned thread (or None)
    can be joined, and nothing goes wrong if it is not.  The reference has no counterpart."""
    from . import _lib

    def go():
        _lib.check(_lib.lib().tvdn_warm_up(int(device)))

    if not background:
        go()
        return None
    import threading
    t = threading.Thread(target=go, name="tvdn-warm-up", daemon=True)
    t.start()
    return t

__all__ = [
    "denoise4D", "denoise3D", "check_memory",
    "accumulator_update_4D", "accumulator_update_4D_FISTA", "accumulator_update_3D", "accumulator_update_3D_FISTA",
    "datacube_update_4D", "datacube_update_3D", "sum_square_error_4D", "sum_square_error_3D",
    "iso_accumulator_update_4D", "iso_accumulator_update_4D_FISTA",
    "plan_run", "denoise_file", "warm_up",
]
