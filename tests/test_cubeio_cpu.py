"""File formats of the streaming ingest / egress layer: round trips that need no GPU."""
import numpy as np
import pytest

from cytvdn_amd import cubeio


def test_npy_memmap_roundtrip_with_dtype_conversion(tmp_path):
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 4000, (11, 3, 4, 8)).astype(np.uint16)        # detector counts as stored
    np.save(tmp_path / "scan.npy", raw)
    cube = cubeio.open_cube(str(tmp_path / "scan.npy"), np.float32)
    assert cube.shape == raw.shape and cube.dtype == np.float32 and cube.ndim == 4 and cube.nbytes == raw.size * 4
    assert isinstance(cube.src, np.memmap)                               # mapped, not loaded
    blk = cube.read_rows(3, 7)
    assert blk.dtype == np.float32 and blk.flags["C_CONTIGUOUS"] and np.array_equal(blk, raw[3:7].astype(np.float32))
    assert 1 <= cube.block_rows(1024) <= 11 and cube.block_rows(1 << 40) == 11
    w = cubeio.CubeWriter(str(tmp_path / "out.npy"), raw.shape, np.float32)
    for a in range(0, 11, 4):
        w.write_rows(a, cube.read_rows(a, min(a + 4, 11)))
    w.close()
    cube.close()
    assert np.array_equal(np.load(tmp_path / "out.npy"), raw.astype(np.float32))


def test_raw_binary_and_errors(tmp_path):
    x = np.arange(5 * 6 * 7, dtype=np.float64).reshape(5, 6, 7)
    x.tofile(tmp_path / "cube.raw")
    with pytest.raises(ValueError):
        cubeio.open_cube(str(tmp_path / "cube.raw"))
    c = cubeio.open_cube(str(tmp_path / "cube.raw"), np.float64, shape=x.shape, file_dtype=np.float64)
    assert np.array_equal(c.read_rows(0, 5), x)
    with pytest.raises(NotImplementedError):
        cubeio.open_cube(str(tmp_path / "cube.tiff"))
    with pytest.raises(NotImplementedError):
        cubeio.CubeWriter(str(tmp_path / "cube.tiff"), (2, 2, 2), np.float32)


def test_hdf5_branch_needs_h5py_and_says_so(tmp_path):
    try:
        import h5py  # noqa: F401
    except Exception:
        with pytest.raises(ImportError, match="h5py"):
            cubeio.open_cube(str(tmp_path / "scan.emd"))
        with pytest.raises(ImportError, match="h5py"):
            cubeio.CubeWriter(str(tmp_path / "out.h5"), (2, 2, 2, 2), np.float32)
        return
    w = cubeio.CubeWriter(str(tmp_path / "out.emd"), (3, 2, 4, 4), np.float32)     # the EMD v0.7 layout of mpi.py:446-491
    w.write_rows(0, np.ones((3, 2, 4, 4), np.float32))
    w.close()
    c = cubeio.open_cube(str(tmp_path / "out.emd"))
    assert c.shape == (3, 2, 4, 4) and float(c.read_rows(0, 3).sum()) == 96.0
    c.close()
