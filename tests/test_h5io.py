"""The reference's on-disk format (fuxictr/datasets/data_utils.py:37-54: one root-level HDF5 dataset per key — `data` float64 [N, L+1]
for the encoded splits, `indices` / `values` / `lens` for retrieval_{K}_{split}.h5, fuxictr/pytorch/data_generator.py:104-113) read and
written WITHOUT h5py: rat_amd/h5io.py binds the HDF5 C library of the image with ctypes.  Checked against two independent parties: the
HDF5 distribution's own `h5dump` (structure of the files we write) and sample files written by other producers (the HDF Group's
little- / big-endian test arrays shipped with PyTables)."""
import glob
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "www24-rat_amd"))

from rat_amd import data as rd  # noqa: E402
from rat_amd import h5io  # noqa: E402

try:
    h5io.library()
    HAVE = True
except h5io.Hdf5Unavailable:
    HAVE = False
pytestmark = pytest.mark.skipif(not HAVE, reason="no libhdf5 in this environment")


def _split(n=23, L=4, K=3, seed=0):
    rs = np.random.RandomState(seed)
    data = np.concatenate([rs.randint(0, 9, size=(n, L)), rs.randint(0, 2, size=(n, 1))], axis=1).astype(np.float64)
    return data, rs.randint(-1, n, size=(n, K)).astype(np.int64), rs.rand(n, K), np.full(n, K, dtype=np.int64)


def test_round_trip_and_structure_as_h5py_writes_it(tmp_path):
    data, idx, val, lens = _split()
    path = str(tmp_path / "retrieval_3_train.h5")
    h5io.write_arrays(path, {"indices": idx, "values": val, "lens": lens})
    got = h5io.read_arrays(path, ["indices", "values", "lens"])
    assert got["indices"].dtype == np.int64 and got["values"].dtype == np.float64 and got["lens"].shape == (23,)
    assert np.array_equal(got["indices"], idx) and np.array_equal(got["values"], val) and np.array_equal(got["lens"], lens)
    with pytest.raises(KeyError):
        h5io.read_arrays(path, ["data"])
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):                               # the HDF5 distribution's own tool agrees on what the file is
        out = subprocess.run([h5dump, "-H", path], capture_output=True, text=True).stdout
        assert 'DATASET "indices"' in out and "H5T_STD_I64LE" in out and "H5T_IEEE_F64LE" in out and "( 23, 3 ) / ( 23, 3 )" in out


def test_files_written_by_other_producers_are_read(tmp_path):
    """6 x 5 arrays with value = row + column, float64 / int64 / int32, little- AND big-endian, written by the HDF Group's C test
    programs (shipped with PyTables): the library converts byte order and width, the reader only has to ask for the right class."""
    files = sorted(glob.glob("/opt/conda/lib/python*/site-packages/tables/tests/smpl_[fi]*[bl]e.h5"))
    if not files:
        pytest.skip("no third-party sample files in this environment")
    want = np.add.outer(np.arange(6), np.arange(5))
    for f in files:
        arr = h5io.read_arrays(f, ["TestArray"])["TestArray"]
        assert arr.shape == (6, 5) and np.array_equal(arr, want), f
        assert arr.dtype.kind == ("f" if "smpl_f" in f else "i")


def test_batches_from_h5_files_equal_batches_from_npz_files(tmp_path):
    """rat_amd.data.batches_from_files (what run_expid.py calls) on `train.h5` + `retrieval_3_train.h5` == on the .npz twins"""
    data, idx, val, lens = _split()
    h5io.write_arrays(str(tmp_path / "train.h5"), {"data": data})
    h5io.write_arrays(str(tmp_path / "retrieval_3_train.h5"), {"indices": idx, "values": val, "lens": lens})
    np.savez(str(tmp_path / "train.npz"), data=data)
    np.savez(str(tmp_path / "retrieval_3_train.npz"), indices=idx, values=val, lens=lens)
    a = list(rd.batches_from_files(str(tmp_path / "train.h5"), str(tmp_path / "retrieval_3_train.h5"), 8, shuffle=True, seed=3))
    b = list(rd.batches_from_files(str(tmp_path / "train.npz"), str(tmp_path / "retrieval_3_train.npz"), 8, shuffle=True, seed=3))
    assert len(a) == len(b) == 3
    for x, y in zip(a, b):
        assert all(bool((u == v).all()) for u, v in zip(x, y))


def test_unsigned_datasets_keep_their_values(tmp_path):
    """unsigned ids used to be read through a signed native type of the same width, where the library clamps at the signed
    maximum (200 -> 127 for a uint8 dataset).  They now come back unsigned, value for value."""
    from rat_amd import h5io
    path = str(tmp_path / "unsigned.h5")
    want = {"u1": np.array([0, 127, 128, 200, 255], dtype=np.uint8), "u2": np.array([1, 40000, 65535], dtype=np.uint16),
            "u4": np.array([7, 3_000_000_000, 4_294_967_295], dtype=np.uint32), "i4": np.array([-5, 5], dtype=np.int32)}
    h5io.write_arrays(path, want)
    got = h5io.read_arrays(path, list(want))
    for k, v in want.items():
        assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k
