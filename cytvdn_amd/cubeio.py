"""Streaming ingest and egress of datacubes on disk.

The reference's MPI driver memory-maps its input (ncempy / py4DSTEM / h5py, cyTVDN/mpi.py:95-124), loads each
rank's tile as float32 (:217-239) and writes the result into a hand-built EMD v0.7 HDF5 file (:446-498).  The
counterpart here hands the engines the memory-mapped file itself when it holds the compute dtype (the library reads it in
place: staged through its pinned lanes, or page-locked), converts any other dtype block by block (:217-239 does it per tile),
and lets the result land in the output file's own memory map (.npy / raw) or writes it block by block (HDF5 / EMD):

    b_norm, delta_recon = denoise_file("scan.npy", "scan_denoised.npy", mu=[1, 1, .5, .5], iterations=50)

Formats: `.npy` (numpy.lib.format, memory-mapped both ways) always; `.h5` / `.hdf5` / `.emd` when h5py is importable
(absent from this image: the branch raises ImportError with that message), input dataset = `dataset` or the EMD v0.7
datacube path, output = the EMD v0.7 layout of mpi.py:446-491; raw binary (`.raw`, `.dat`, `.bin`) with explicit
`shape` and `dtype`.  The result is bit-identical to `denoise3D/4D` on the same cube loaded into memory
(tests/test_gpu_cubeio.py).
"""
from __future__ import annotations

import os

import numpy as np

EMD_DATA = "4DSTEM_experiment/data/datacubes/datacube_0/data"
_H5_EXT = (".h5", ".hdf5", ".emd")
_RAW_EXT = (".raw", ".dat", ".bin")


class LazyCube:
    """A cube that lives in a file: shape/dtype of an array, rows delivered on demand in the compute dtype."""

    def __init__(self, src, dtype, closer=None):
        self.src = src
        self.dtype = np.dtype(dtype)
        self.shape = tuple(int(s) for s in src.shape)
        self.ndim = len(self.shape)
        self.size = int(np.prod(self.shape))
        self.nbytes = self.size * self.dtype.itemsize
        self._closer = closer

    def read_rows(self, a: int, b: int) -> np.ndarray:
        """Rows [a, b) of axis 0, C-contiguous, converted to the compute dtype (mpi.py:217-239 does this per tile)."""
        blk = np.asarray(self.src[a:b])
        return np.ascontiguousarray(blk, dtype=self.dtype)

    def block_rows(self, target_bytes: int = 256 << 20) -> int:
        row = max(1, self.nbytes // max(1, self.shape[0]))
        return max(1, min(self.shape[0], target_bytes // row))

    def close(self):
        if self._closer is not None:
            self._closer()
            self._closer = None


def _h5py():
    try:
        import h5py
        return h5py
    except Exception as e:  # pragma: no cover - h5py is not in this image
        raise ImportError("HDF5 / EMD files need h5py, which is not installed; convert to .npy or install h5py") from e


def open_cube(path, dtype=np.float32, dataset=None, shape=None, file_dtype=None) -> LazyCube:
    """Memory-map the cube stored in `path` (never loads it)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return LazyCube(np.load(path, mmap_mode="r"), dtype)
    if ext in _RAW_EXT:
        if shape is None or file_dtype is None:
            raise ValueError("raw binary input needs shape= and file_dtype=")
        return LazyCube(np.memmap(path, dtype=np.dtype(file_dtype), mode="r", shape=tuple(shape)), dtype)
    if ext in _H5_EXT:
        h5 = _h5py()
        f = h5.File(path, "r")
        name = dataset
        if name is None:
            if EMD_DATA in f:
                name = EMD_DATA
            else:
                found = []
                f.visititems(lambda n, o: found.append(n) if isinstance(o, h5.Dataset) and o.ndim in (3, 4) else None)
                if not found:
                    f.close()
                    raise ValueError(f"no 3-D or 4-D dataset in {path}")
                name = found[0]
        return LazyCube(f[name], dtype, closer=f.close)
    raise NotImplementedError(f"Incompatible File type {ext!r} (supported: .npy, .raw/.dat/.bin, .h5/.hdf5/.emd)")


class CubeWriter:
    """Output cube on disk, filled row block by row block."""

    def __init__(self, path, shape, dtype):
        self.path, self.shape, self.dtype = path, tuple(int(s) for s in shape), np.dtype(dtype)
        ext = os.path.splitext(path)[1].lower()
        self._h5 = None
        if ext == ".npy":
            self.arr = np.lib.format.open_memmap(path, mode="w+", dtype=self.dtype, shape=self.shape)
        elif ext in _RAW_EXT:
            self.arr = np.memmap(path, dtype=self.dtype, mode="w+", shape=self.shape)
        elif ext in _H5_EXT:
            h5 = _h5py()
            self._h5 = h5.File(path, "w")
            self.arr = _emd_v07_layout(self._h5, self.shape, self.dtype)
        else:
            raise NotImplementedError(f"Incompatible File type {ext!r}")

    def write_rows(self, a: int, block: np.ndarray):
        self.arr[a:a + block.shape[0]] = block

    def close(self):
        if self._h5 is not None:
            self._h5.close()
            self._h5 = None
        elif hasattr(self.arr, "flush"):
            self.arr.flush()
        self.arr = None


def _emd_v07_layout(f, shape, dtype):
    """The group structure the reference writes by hand (EMD v0.7, cyTVDN/mpi.py:446-491); returns the data dataset."""
    top = f.create_group("4DSTEM_experiment")
    top.attrs.create("emd_group_type", 2)
    top.attrs.create("version_major", 0)
    top.attrs.create("version_minor", 7)
    top.create_group("metadata")
    data = top.create_group("data")
    cubes = data.create_group("datacubes")
    for g in ("counted_datacubes", "diffractionslices", "realslices", "pointlists", "pointlistarrays"):
        data.create_group(g)
    dc = cubes.create_group("datacube_0")
    dset = dc.create_dataset("data", shape, dtype=dtype)
    dc.attrs.create("emd_group_type", 1)
    dc.attrs.create("metadata", -1)
    names = ("R_x", "R_y", "Q_x", "Q_y") if len(shape) == 4 else ("R_x", "R_y", "E")
    for i, (n, nm) in enumerate(zip(shape, names)):
        d = dc.create_dataset(f"dim{i + 1}", (n,))
        d[...] = np.arange(0, n)
        d.attrs.create("name", np.bytes_(nm))
        d.attrs.create("units", np.bytes_("[pix]"))
    return dset


def denoise_file(input_path, output_path, mu, iterations=10, FISTA=True, stopping_relative_change=None,
                 BC_mode=2, lam=None, dtype=np.float32, dataset=None, shape=None, file_dtype=None, quiet=True,
                 device=None):
    """denoise3D / denoise4D (by the rank of the stored cube) from file to file.

    Arguments as `denoise4D` (cyTVDN/cyTVDN.py:19-31) with paths in place of the array; `dtype` is the compute dtype
    the stored values are converted to (the reference's MPI driver hard-codes float32, mpi.py:217-239).
    Returns (b_norm, delta_recon); the reconstruction is in `output_path`."""
    from . import driver
    dt = np.dtype(dtype)
    assert dt in (np.float32, np.float64), "datacube must be floating point datatype."
    # The output is created (truncated) while the input is still being read through its memory map: writing onto the
    # input -- same path, a symlink or a hard link to it -- would destroy the cube before a single row has been read.
    if os.path.exists(output_path) and os.path.exists(input_path) and os.path.samefile(input_path, output_path):
        raise ValueError(f"output {output_path!r} is the input file itself: denoise_file writes to a distinct file "
                         "(as the reference's MPI driver does, mpi.py:436-498)")
    src = open_cube(input_path, dt, dataset=dataset, shape=shape, file_dtype=file_dtype)
    try:
        nd = src.ndim
        if nd not in (3, 4):
            raise AssertionError("Bad number of dimensions...")
        mu = np.asarray(mu, dt)
        if lam is None:
            lam = mu * 1.0 / 32.0 if nd == 4 else mu / 16.0
        lam = np.asarray(lam, dt)
        if mu.shape != (nd,) or lam.shape != (nd,):
            raise ValueError(f"mu and lam must have {nd} entries")
        if nd == 3:
            lam_mu = (lam / mu).astype(dt)
            assert np.all(lam_mu <= (1.0 / 16.0)) & np.all(lam_mu > 0), "Parameters must satisfy 0 < λ/μ <= 1/16"
        # The engines see arrays: a stored cube of the compute dtype IS one (its memory map, read in place), any other is
        # converted block by block (mpi.py:217-239 converts per tile); the result lands in the output file's own memory map
        # where the format has one (.npy / raw), else it is written block by block (HDF5 / EMD).
        in_place = isinstance(src.src, np.ndarray) and src.src.dtype == dt and src.src.flags["C_CONTIGUOUS"]
        out_mapped = os.path.splitext(str(output_path))[1].lower() not in _H5_EXT
        _check_host_holds_the_copies(src.shape, dt, nd, FISTA, stopping_relative_change is not None, device,
                                     converted=not in_place, result_in_ram=not out_mapped)
        if in_place:
            x = src.src
        else:
            x = np.empty(src.shape, dt)
            step = src.block_rows()
            for a0 in range(0, src.shape[0], step):
                x[a0:a0 + step] = src.read_rows(a0, min(a0 + step, src.shape[0]))
        out = CubeWriter(output_path, src.shape, dt)
        try:
            target = out.arr if isinstance(out.arr, np.ndarray) and out.arr.dtype == dt else None
            res = driver._run(nd, x, mu, lam, iterations, FISTA, stopping_relative_change, None, BC_mode, quiet, device,
                              out=target)
            if target is None:
                step = src.block_rows()
                for a0 in range(0, src.shape[0], step):
                    out.write_rows(a0, res[0][a0:a0 + step])
        finally:
            out.close()
    finally:
        src.close()
    return res[1], res[2]


def _check_host_holds_the_copies(shape, dt, nd, fista, stop, device, converted: bool, result_in_ram: bool) -> None:
    """A stored cube of another dtype (uint16 detector counts, say) is converted into ONE host array of the compute dtype, and a
    result bound for HDF5 / EMD is held in RAM until it is written block by block: up to two more cubes of pageable memory that
    the streamed engine's own guard (tvdn_stream_host_need) knows nothing of -- it could pass, and the kernel then kill the
    process.  Add them to what the run page-locks and refuse BEFORE anything is allocated."""
    from . import planner
    cube = int(np.prod(shape)) * np.dtype(dt).itemsize
    extra = cube * (int(bool(converted)) + int(bool(result_in_ram)))
    if not extra:
        return
    avail = planner.host_available()
    if avail is None:
        return
    pinned = 0
    try:
        dev = int(device[0]) if isinstance(device, (list, tuple)) and len(device) else (0 if device is None else int(device))
        plan = planner.plan_run(tuple(shape), dt, bool(fista), 1, stop=bool(stop), device=dev)
        pinned = int(plan.get("host_bytes_per_rank") or 0)
    except Exception:
        pinned = 0                      # no GPU to ask (the run itself will say so): judge the copies alone
    if extra + pinned > planner.HOST_FRACTION * avail:
        what = " + ".join(w for w, on in (("the cube converted to %s" % np.dtype(dt).name, converted),
                                          ("the result held for the HDF5 writer", result_in_ram)) if on)
        raise MemoryError(f"denoise_file would hold {extra / 2 ** 30:.1f} GiB of host memory for {what}"
                          + (f" beside {pinned / 2 ** 30:.1f} GiB of page-locked state of the streamed run" if pinned else "")
                          + f", more than {planner.HOST_FRACTION:.0%} of the {avail / 2 ** 30:.1f} GiB this host has available: store the "
                          "cube in the compute dtype (it is then read in place through its memory map) and write .npy / raw "
                          "output (written in place), or use more nodes")


__all__ = ["open_cube", "CubeWriter", "LazyCube", "denoise_file", "EMD_DATA"]
