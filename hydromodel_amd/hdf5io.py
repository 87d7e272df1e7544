"""HDF5 results files without h5py: a ctypes binding of the HDF5 C library (libhdf5, 1.10 API).

Replaces the two h5py uses of the reference (code/src/simulation.py:697-706 `saveResults`: one
gzip-compressed root-level dataset per key of ``Simulation.output``; :735-741 `loadResults`: every
root-level dataset back into a dict of arrays).  Files written here open with h5py / h5dump and files
written by the reference's h5py open here -- the on-disk format is the library's, not ours.

The library is looked up in this order: ``$HYDROMODEL_HDF5_LIB``, the dynamic loader
(``ctypes.util.find_library('hdf5')``), ``/opt/conda/lib/libhdf5.so`` (where this image keeps it).
"""
import ctypes as C
import ctypes.util
import os
from pathlib import Path

import numpy as np

_hid = C.c_int64          # hid_t is 64-bit since HDF5 1.10
_hsize = C.c_uint64
_GZIP_LEVEL = 4           # h5py's default for compression='gzip' (simulation.py:703 comment)
_CHUNK_BYTES = 256 * 1024

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_SGN_NONE = 0
H5T_DIR_ASCEND = 1
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5O_TYPE_DATASET = 1

_lib = None


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", _hsize), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


def _candidates():
    env = os.environ.get("HYDROMODEL_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    yield "/opt/conda/lib/libhdf5.so"
    yield "libhdf5.so"


def _load():
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for name in _candidates():
        try:
            lib = C.CDLL(name)
            break
        except OSError as exc:
            last = exc
    else:
        raise ImportError(f"libhdf5 not found ({last}); set HYDROMODEL_HDF5_LIB")
    sig = {
        "H5open": (C.c_int, []),
        "H5get_libversion": (C.c_int, [C.POINTER(C.c_uint)] * 3),
        "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
        "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fclose": (C.c_int, [_hid]),
        "H5Screate_simple": (_hid, [C.c_int, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Screate": (_hid, [C.c_int]),
        "H5Sclose": (C.c_int, [_hid]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Pcreate": (_hid, [_hid]),
        "H5Pclose": (C.c_int, [_hid]),
        "H5Pset_chunk": (C.c_int, [_hid, C.c_int, C.POINTER(_hsize)]),
        "H5Pset_deflate": (C.c_int, [_hid, C.c_uint]),
        "H5Zfilter_avail": (C.c_int, [C.c_int]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Dclose": (C.c_int, [_hid]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dget_space": (_hid, [_hid]),
        "H5Dget_type": (_hid, [_hid]),
        "H5Tget_class": (C.c_int, [_hid]),
        "H5Tget_size": (C.c_size_t, [_hid]),
        "H5Tget_sign": (C.c_int, [_hid]),
        "H5Tget_native_type": (_hid, [_hid, C.c_int]),
        "H5Tclose": (C.c_int, [_hid]),
        "H5Gget_info": (C.c_int, [_hid, C.POINTER(_GInfo)]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [_hid, C.c_char_p, C.c_int, C.c_int, _hsize, C.c_char_p, C.c_size_t,
                                             _hid]),
        "H5Oopen": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Oclose": (C.c_int, [_hid]),
        "H5Iget_type": (C.c_int, [_hid]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise ImportError("H5open failed")
    lib.H5Eset_auto2(0, None, None)      # errors are reported through return codes -> Python exceptions
    _lib = lib
    return lib


def available():
    try:
        _load()
        return True
    except ImportError:
        return False


def version():
    lib = _load()
    v = [C.c_uint() for _ in range(3)]
    lib.H5get_libversion(*[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def _glob(name):
    return _hid.in_dll(_load(), name).value


_NATIVE = {
    np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g", np.dtype(np.float32): "H5T_NATIVE_FLOAT_g",
    np.dtype(np.int8): "H5T_NATIVE_INT8_g", np.dtype(np.uint8): "H5T_NATIVE_UINT8_g",
    np.dtype(np.int16): "H5T_NATIVE_INT16_g", np.dtype(np.uint16): "H5T_NATIVE_UINT16_g",
    np.dtype(np.int32): "H5T_NATIVE_INT32_g", np.dtype(np.uint32): "H5T_NATIVE_UINT32_g",
    np.dtype(np.int64): "H5T_NATIVE_INT64_g", np.dtype(np.uint64): "H5T_NATIVE_UINT64_g",
}


def _check(rc, what):
    if rc < 0:
        raise OSError(f"HDF5: {what} failed")
    return rc


def _chunk_shape(shape, itemsize):
    """Row-blocked chunks of about 256 KiB (compression needs a chunked layout, as in h5py)."""
    if len(shape) == 0 or 0 in shape:
        return None
    chunk = list(shape)
    row_bytes = itemsize * int(np.prod(shape[1:], dtype=np.int64))
    chunk[0] = int(max(1, min(shape[0], _CHUNK_BYTES // max(1, row_bytes))))
    k = 1
    while itemsize * int(np.prod(chunk, dtype=np.int64)) > 4 * _CHUNK_BYTES and k < len(chunk):
        chunk[k] = max(1, chunk[k] // 2)       # very wide rows: split the trailing axes too
        if chunk[k] == 1:
            k += 1
    return tuple(chunk)


def write(path, arrays, compression="gzip"):
    """Create/truncate `path` with one root-level dataset per (key, ndarray) of `arrays`."""
    lib = _load()
    fid = lib.H5Fcreate(os.fsencode(str(Path(path))), H5F_ACC_TRUNC, 0, 0)
    _check(fid, f"create {path}")
    try:
        for key, val in arrays.items():
            a = np.require(np.asarray(val), requirements="C")      # (ascontiguousarray would make 0-d arrays 1-d)
            if a.dtype == np.bool_:
                a = a.astype(np.uint8)
            if a.dtype not in _NATIVE:
                raise TypeError(f"dataset {key!r}: dtype {a.dtype} is not supported")
            mem_t = _glob(_NATIVE[a.dtype])
            if a.ndim == 0:
                space = lib.H5Screate(0)        # H5S_SCALAR
            else:
                dims = (_hsize * a.ndim)(*a.shape)
                space = lib.H5Screate_simple(a.ndim, dims, None)
            _check(space, "dataspace")
            dcpl = lib.H5Pcreate(_glob("H5P_CLS_DATASET_CREATE_ID_g"))
            _check(dcpl, "property list")
            chunk = _chunk_shape(a.shape, a.dtype.itemsize)
            if compression == "gzip" and chunk is not None and lib.H5Zfilter_avail(1) > 0:
                _check(lib.H5Pset_chunk(dcpl, a.ndim, (_hsize * a.ndim)(*chunk)), "set_chunk")
                _check(lib.H5Pset_deflate(dcpl, _GZIP_LEVEL), "set_deflate")
            dset = lib.H5Dcreate2(fid, key.encode(), mem_t, space, 0, dcpl, 0)
            try:
                _check(dset, f"create dataset {key!r}")
                if a.size:
                    _check(lib.H5Dwrite(dset, mem_t, 0, 0, 0, a.ctypes.data_as(C.c_void_p)), f"write {key!r}")
            finally:
                if dset >= 0:
                    lib.H5Dclose(dset)
                lib.H5Pclose(dcpl)
                lib.H5Sclose(space)
    finally:
        _check(lib.H5Fclose(fid), f"close {path}")


def _numpy_dtype(lib, ftype, key):
    cls, size = lib.H5Tget_class(ftype), lib.H5Tget_size(ftype)
    if cls == H5T_FLOAT and size in (4, 8):
        return np.dtype(f"f{size}")
    if cls == H5T_INTEGER and size in (1, 2, 4, 8):
        return np.dtype(("u" if lib.H5Tget_sign(ftype) == H5T_SGN_NONE else "i") + str(size))
    raise TypeError(f"dataset {key!r}: HDF5 type class {cls} / {size} bytes is not supported")


def keys(path):
    lib = _load()
    fid = lib.H5Fopen(os.fsencode(str(Path(path))), H5F_ACC_RDONLY, 0)
    _check(fid, f"open {path}")
    try:
        return _keys(lib, fid)
    finally:
        lib.H5Fclose(fid)


def _keys(lib, fid):
    info = _GInfo()
    _check(lib.H5Gget_info(fid, C.byref(info)), "group info")
    names = []
    for k in range(info.nlinks):
        n = lib.H5Lget_name_by_idx(fid, b".", H5_INDEX_NAME, H5_ITER_INC, k, None, 0, 0)
        _check(n, "link name")
        buf = C.create_string_buffer(n + 1)
        lib.H5Lget_name_by_idx(fid, b".", H5_INDEX_NAME, H5_ITER_INC, k, buf, n + 1, 0)
        names.append(buf.value.decode())
    return names


def read(path):
    """Every root-level dataset of `path` as {name: ndarray} (simulation.py:735-741)."""
    lib = _load()
    fid = lib.H5Fopen(os.fsencode(str(Path(path))), H5F_ACC_RDONLY, 0)
    _check(fid, f"open {path}")
    out = {}
    try:
        for key in _keys(lib, fid):
            dset = lib.H5Dopen2(fid, key.encode(), 0)
            if dset < 0:
                continue                      # a sub-group: the reference's files have none
            try:
                space, ftype = lib.H5Dget_space(dset), lib.H5Dget_type(dset)
                try:
                    nd = _check(lib.H5Sget_simple_extent_ndims(space), "rank")
                    dims = (_hsize * max(nd, 1))()
                    if nd:
                        lib.H5Sget_simple_extent_dims(space, dims, None)
                    dt = _numpy_dtype(lib, ftype, key)
                    a = np.empty(tuple(int(d) for d in dims[:nd]), dtype=dt)
                    if a.size:
                        _check(lib.H5Dread(dset, _glob(_NATIVE[dt]), 0, 0, 0, a.ctypes.data_as(C.c_void_p)),
                               f"read {key!r}")
                    out[key] = a
                finally:
                    lib.H5Tclose(ftype)
                    lib.H5Sclose(space)
            finally:
                lib.H5Dclose(dset)
    finally:
        lib.H5Fclose(fid)
    return out
