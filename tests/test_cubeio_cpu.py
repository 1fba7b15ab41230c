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


# ---- the HDF5 / EMD branch, driven through a recording stand-in for h5py (the module is absent from this image) --------
class _Attrs(dict):
    def create(self, name, value):
        self[name] = value


class _Dataset:
    def __init__(self, shape, dtype=None):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype or np.float32)
        self.ndim = len(self.shape)
        self.data = np.zeros(self.shape, self.dtype)
        self.attrs = _Attrs()

    def __getitem__(self, k):
        return self.data[k]

    def __setitem__(self, k, v):
        self.data[k] = v


class _Group:
    def __init__(self):
        self.children, self.attrs = {}, _Attrs()

    def create_group(self, name):
        assert name not in self.children
        g = self.children[name] = _Group()
        return g

    def create_dataset(self, name, shape, dtype=None):
        assert name not in self.children
        d = self.children[name] = _Dataset(shape, dtype)
        return d

    def _walk(self, path):
        node = self
        for part in path.split("/"):
            node = node.children[part]
        return node

    def __contains__(self, path):
        try:
            self._walk(path)
            return True
        except (KeyError, AttributeError):
            return False

    def __getitem__(self, path):
        return self._walk(path)

    def visititems(self, fn, prefix=""):
        for name, node in self.children.items():
            fn(prefix + name, node)
            if isinstance(node, _Group):
                node.visititems(fn, prefix + name + "/")

    def tree(self, prefix=""):
        """{path: ("group", attrs) | ("dataset", shape, attrs)} of everything below this node."""
        out = {}
        for name, node in self.children.items():
            if isinstance(node, _Group):
                out[prefix + name] = ("group", dict(node.attrs))
                out.update(node.tree(prefix + name + "/"))
            else:
                out[prefix + name] = ("dataset", node.shape, dict(node.attrs))
        return out


class _FakeH5:
    """File / Dataset of an h5py look-alike whose files live in a dict keyed by path."""
    Dataset = _Dataset

    def __init__(self):
        self.files, self.closed = {}, []

    def File(self, path, mode="r"):
        fake = self
        if mode == "w":
            self.files[path] = _Group()
        root = self.files[path]
        root.close = lambda: fake.closed.append(path)
        return root


def _emd_v07_expected(shape):
    """The tree the reference writes by hand (cyTVDN/mpi.py:446-491), as data: EMD v0.7 top-level group with its version
    attributes, the six data groups, datacube_0 with `data` and one calibrated-in-pixels axis dataset per dimension."""
    top = "4DSTEM_experiment"
    t = {top: ("group", {"emd_group_type": 2, "version_major": 0, "version_minor": 7}),
         f"{top}/metadata": ("group", {}), f"{top}/data": ("group", {})}
    for g in ("datacubes", "counted_datacubes", "diffractionslices", "realslices", "pointlists", "pointlistarrays"):
        t[f"{top}/data/{g}"] = ("group", {})
    dc = f"{top}/data/datacubes/datacube_0"
    t[dc] = ("group", {"emd_group_type": 1, "metadata": -1})
    t[f"{dc}/data"] = ("dataset", tuple(shape), {})
    names = ("R_x", "R_y", "Q_x", "Q_y") if len(shape) == 4 else ("R_x", "R_y", "E")
    for i, (n, nm) in enumerate(zip(shape, names)):
        t[f"{dc}/dim{i + 1}"] = ("dataset", (n,), {"name": np.bytes_(nm), "units": np.bytes_("[pix]")})
    return t


@pytest.mark.parametrize("shape", [(5, 4, 6, 8), (7, 3, 16)])
def test_emd_v07_layout_matches_the_reference_writer(monkeypatch, shape):
    fake = _FakeH5()
    monkeypatch.setattr(cubeio, "_h5py", lambda: fake)
    w = cubeio.CubeWriter("/virtual/out.emd", shape, np.float32)
    x = np.arange(int(np.prod(shape)), dtype=np.float32).reshape(shape)
    for a in range(0, shape[0], 2):
        w.write_rows(a, x[a:a + 2])
    w.close()
    assert fake.closed == ["/virtual/out.emd"]
    root = fake.files["/virtual/out.emd"]
    assert root.tree() == _emd_v07_expected(shape)
    dc = root["4DSTEM_experiment/data/datacubes/datacube_0"]
    assert np.array_equal(dc["data"].data, x) and dc["data"].dtype == np.float32
    for i, n in enumerate(shape):                      # uncalibrated pixel axes, mpi.py:478-489
        assert np.array_equal(dc[f"dim{i + 1}"].data, np.arange(n))
    # ... and it reads back through open_cube: EMD path by default, an explicit dataset, the first 3-D/4-D dataset otherwise
    c = cubeio.open_cube("/virtual/out.emd", np.float64)
    assert c.shape == shape and c.dtype == np.float64 and np.array_equal(c.read_rows(1, 3), x[1:3].astype(np.float64))
    c.close()
    assert fake.closed[-1] == "/virtual/out.emd"
    c = cubeio.open_cube("/virtual/out.emd", np.float32, dataset=cubeio.EMD_DATA)
    assert np.array_equal(c.read_rows(0, shape[0]), x)
    other = fake.File("/virtual/plain.h5", "w")
    other.create_group("scan").create_dataset("counts", shape, np.uint16)[...] = 3
    other.create_dataset("calibration", (4,))
    c = cubeio.open_cube("/virtual/plain.h5", np.float32)
    assert c.shape == shape and float(c.read_rows(0, 1).max()) == 3.0
    empty = fake.File("/virtual/empty.h5", "w")
    empty.create_dataset("vector", (9,))
    with pytest.raises(ValueError, match="no 3-D or 4-D dataset"):
        cubeio.open_cube("/virtual/empty.h5")
    assert fake.closed[-1] == "/virtual/empty.h5"


def test_denoise_file_refuses_to_overwrite_its_input(tmp_path):
    x = np.zeros((4, 3, 4, 8), np.float32)
    p = tmp_path / "scan.npy"
    np.save(p, x)
    link = tmp_path / "alias.npy"
    link.symlink_to(p)
    for out in (p, link):
        with pytest.raises(ValueError, match="input file itself"):
            cubeio.denoise_file(str(p), str(out), mu=[1, 1, 0.5, 0.5], iterations=2)
    assert np.array_equal(np.load(p), x)               # untouched


def test_denoise_file_counts_the_copies_it_will_hold(tmp_path, monkeypatch):
    """A stored cube of another dtype is converted into one host array, a result bound for HDF5 is held in RAM: both are added to
    what the run page-locks and refused BEFORE anything is allocated when the host cannot hold them."""
    x = (np.arange(4 * 3 * 4 * 8) % 7).astype(np.uint16).reshape(4, 3, 4, 8)
    p = tmp_path / "counts.npy"
    np.save(p, x)
    monkeypatch.setenv("TVDN_HOST_LIMIT", "1K")                         # a host with (next to) no memory
    with pytest.raises(MemoryError, match="converted to float32"):
        cubeio.denoise_file(str(p), str(tmp_path / "out.npy"), mu=[1, 1, 0.5, 0.5], iterations=2)
    assert not (tmp_path / "out.npy").exists()                          # refused before the output was created
    # same dtype in, memory-mapped out: nothing is copied, nothing to refuse (the run itself then needs the GPU)
    cubeio._check_host_holds_the_copies(x.shape, np.dtype(np.float32), 4, True, False, None, converted=False, result_in_ram=False)
    with pytest.raises(MemoryError, match="HDF5 writer"):
        cubeio._check_host_holds_the_copies(x.shape, np.dtype(np.float32), 4, True, False, None, converted=False, result_in_ram=True)
