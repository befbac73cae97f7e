"""HDF5 without h5py: the reference's on-disk format for encoded splits and retrieval files (fuxictr/datasets/data_utils.py:37-54 —
`save_hdf5` / `load_hdf5`: one root-level dataset per key, written by `h5py.File.create_dataset(key, data=array)`; the data generator
reads `data`, `indices`, `values`, `lens`, fuxictr/pytorch/data_generator.py:104-113) read and written through the HDF5 C library
itself, bound with ctypes.

h5py is not installed in this image, but `libhdf5.so` (1.10) is (`/opt/conda/lib`), and a file made of plain contiguous root-level
datasets needs a dozen C calls.  The writer issues the calls h5py issues for `create_dataset(name, data=array)` (H5Screate_simple,
H5Dcreate2 with default property lists, H5Dwrite), so the files have the structure the reference's tools produce; the reader accepts
any root-level (or path-addressed) dataset of a float / integer type, whatever its layout or byte order (the library converts).
`rat_amd.data.load_array_file` tries h5py first, then this module; without either it explains how to export to `.npz`.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_LIB = None
_CANDIDATES = ("/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/libhdf5.so",
               "/usr/local/lib/libhdf5.so")
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 0, 2, 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
hid_t, hsize_t, herr_t = ctypes.c_int64, ctypes.c_uint64, ctypes.c_int


class Hdf5Unavailable(RuntimeError):
    pass


def library():
    """the HDF5 C library (RAT_HDF5_LIBRARY, the loader's search path, then the usual locations); raises Hdf5Unavailable"""
    global _LIB
    if _LIB is not None:
        return _LIB
    names = [os.environ.get("RAT_HDF5_LIBRARY"), ctypes.util.find_library("hdf5")] + list(_CANDIDATES)
    last = None
    for name in names:
        if not name:
            continue
        try:
            lib = ctypes.CDLL(name)
        except OSError as exc:
            last = exc
            continue
        lib.H5open.restype = herr_t
        if lib.H5open() < 0:
            continue
        # hid_t is 64-bit from HDF5 1.10 on (a 32-bit int before): refuse an older library instead of passing it truncated handles
        ver = (ctypes.c_uint * 3)()
        lib.H5get_libversion.restype = herr_t
        if lib.H5get_libversion(ctypes.byref(ver, 0), ctypes.byref(ver, 4), ctypes.byref(ver, 8)) < 0 or (ver[0], ver[1]) < (1, 10):
            last = OSError("%s is HDF5 %d.%d.%d; 1.10 or newer is needed (64-bit hid_t)" % (name, ver[0], ver[1], ver[2]))
            continue
        sig = {
            "H5Fopen": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
            "H5Fcreate": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
            "H5Fclose": (herr_t, [hid_t]),
            "H5Dopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
            "H5Dcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dget_space": (hid_t, [hid_t]),
            "H5Dget_type": (hid_t, [hid_t]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
            "H5Dclose": (herr_t, [hid_t]),
            "H5Screate_simple": (hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
            "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
            "H5Sclose": (herr_t, [hid_t]),
            "H5Tget_class": (ctypes.c_int, [hid_t]),
            "H5Tget_size": (ctypes.c_size_t, [hid_t]),
            "H5Tget_sign": (ctypes.c_int, [hid_t]),
            "H5Tclose": (herr_t, [hid_t]),
            "H5Eset_auto2": (herr_t, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
        }
        for fn, (res, args) in sig.items():
            f = getattr(lib, fn)
            f.restype, f.argtypes = res, args
        lib.H5Eset_auto2(0, None, None)                     # errors come back as negative ids: no stack dumps on stderr
        _LIB = lib
        return lib
    raise Hdf5Unavailable("no usable libhdf5 found (set RAT_HDF5_LIBRARY); last error: %s" % (last,))


def _native(lib, name):
    return hid_t.in_dll(lib, name).value            # H5T_NATIVE_* are globals filled in by H5open()


def _mem_type(lib, dtype):
    dtype = np.dtype(dtype)
    table = {"f8": "H5T_NATIVE_DOUBLE_g", "f4": "H5T_NATIVE_FLOAT_g", "i8": "H5T_NATIVE_INT64_g", "i4": "H5T_NATIVE_INT32_g",
             "i2": "H5T_NATIVE_INT16_g", "i1": "H5T_NATIVE_INT8_g", "u8": "H5T_NATIVE_UINT64_g", "u4": "H5T_NATIVE_UINT32_g",
             "u2": "H5T_NATIVE_UINT16_g", "u1": "H5T_NATIVE_UINT8_g"}
    key = dtype.kind + str(dtype.itemsize)
    if key not in table:
        raise TypeError("HDF5 I/O here covers float / integer arrays, not %s" % dtype)
    return _native(lib, table[key])


def read_arrays(path, keys):
    """{key: numpy array} of the named datasets (float datasets come back as float64 / float32, integer ones as int64 / int32 / ...)"""
    lib = library()
    fid = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if fid < 0:
        raise OSError("%s is not an HDF5 file the library can open" % path)
    out = {}
    try:
        for key in keys:
            did = lib.H5Dopen2(fid, key.encode(), H5P_DEFAULT)
            if did < 0:
                raise KeyError("%s has no dataset %r" % (path, key))
            sid = tid = -1
            try:
                sid, tid = lib.H5Dget_space(did), lib.H5Dget_type(did)
                nd = lib.H5Sget_simple_extent_ndims(sid)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    lib.H5Sget_simple_extent_dims(sid, dims, None)
                cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
                if cls == H5T_FLOAT:
                    dtype = np.float64 if size >= 8 else np.float32
                elif cls == H5T_INTEGER:
                    # unsigned data keeps an unsigned type: read through a signed native type of the same width the library would
                    # CLAMP values above the signed maximum (H5T_SGN_NONE = 0)
                    unsigned = lib.H5Tget_sign(tid) == 0
                    dtype = ({1: np.uint8, 2: np.uint16, 4: np.uint32}.get(size, np.uint64) if unsigned else
                             {1: np.int8, 2: np.int16, 4: np.int32}.get(size, np.int64))
                else:
                    raise TypeError("dataset %r of %s is neither float nor integer (HDF5 type class %d)" % (key, path, cls))
                arr = np.empty(tuple(int(d) for d in dims[:nd]), dtype=dtype)
                if lib.H5Dread(did, _mem_type(lib, dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise OSError("H5Dread failed on %r of %s" % (key, path))
                out[key] = arr
            finally:
                if tid >= 0:
                    lib.H5Tclose(tid)
                if sid >= 0:
                    lib.H5Sclose(sid)
                lib.H5Dclose(did)
    finally:
        lib.H5Fclose(fid)
    return out


def write_arrays(path, arrays):
    """one contiguous root-level dataset per item — what `h5py.File(path, "w").create_dataset(key, data=array)` writes"""
    lib = library()
    fid = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if fid < 0:
        raise OSError("cannot create %s" % path)
    try:
        for key, value in arrays.items():
            arr = np.ascontiguousarray(value)
            tid = _mem_type(lib, arr.dtype)
            dims = (hsize_t * max(arr.ndim, 1))(*arr.shape)
            sid = lib.H5Screate_simple(arr.ndim, dims, None)
            did = -1
            try:
                did = lib.H5Dcreate2(fid, key.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
                if did < 0 or lib.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise OSError("writing dataset %r to %s failed" % (key, path))
            finally:
                if did >= 0:
                    lib.H5Dclose(did)
                if sid >= 0:
                    lib.H5Sclose(sid)
    finally:
        lib.H5Fclose(fid)
