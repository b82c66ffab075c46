"""extract_features.lua writes `/feats` and `/boxes` with torch-hdf5 (extract_features.lua:92-96); the image has no
h5py, so densecap_amd/hdf5_min.py writes the file.  Here it is read back by libhdf5 itself (ctypes, when the shared
library exists -- /opt/conda/lib in this image) and by the module's own reader."""
import ctypes as C
import glob
import os

import numpy as np
import pytest


def _libhdf5():
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            if "_hl" in p or "_cpp" in p or "fortran" in p:
                continue
            try:
                return C.CDLL(p)
            except OSError:
                pass
    return None


def _read_with_libhdf5(lib, path, name):
    hid = C.c_int64
    lib.H5open.restype = C.c_int
    lib.H5Fopen.restype = hid; lib.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    lib.H5Dopen2.restype = hid; lib.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Dget_space.restype = hid; lib.H5Dget_space.argtypes = [hid]
    lib.H5Sget_simple_extent_ndims.restype = C.c_int; lib.H5Sget_simple_extent_ndims.argtypes = [hid]
    lib.H5Sget_simple_extent_dims.restype = C.c_int
    lib.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.H5Dread.restype = C.c_int; lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    lib.H5Dclose.argtypes = [hid]; lib.H5Sclose.argtypes = [hid]; lib.H5Fclose.argtypes = [hid]
    assert lib.H5open() >= 0
    native_float = hid.in_dll(lib, "H5T_NATIVE_FLOAT_g").value
    f = lib.H5Fopen(path.encode(), 0, 0)
    assert f >= 0, "libhdf5 refused the file"
    d = lib.H5Dopen2(f, name.encode(), 0)
    assert d >= 0, "libhdf5 cannot open dataset %s" % name
    sp = lib.H5Dget_space(d)
    nd = lib.H5Sget_simple_extent_ndims(sp)
    dims = (C.c_uint64 * nd)()
    lib.H5Sget_simple_extent_dims(sp, dims, None)
    out = np.empty(tuple(int(x) for x in dims), np.float32)
    assert lib.H5Dread(d, native_float, 0, 0, 0, out.ctypes.data) >= 0
    lib.H5Sclose(sp); lib.H5Dclose(d); lib.H5Fclose(f)
    return out


def test_hdf5_roundtrip_own_reader(tmp_path):
    from densecap_amd import hdf5_min as H
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((3, 5, 4096)).astype(np.float32)
    boxes = rng.uniform(1, 700, (3, 5, 4)).astype(np.float32)
    p = str(tmp_path / "f.h5")
    H.write_hdf5(p, {"feats": feats, "boxes": boxes, "ids": np.arange(7, dtype=np.int32)})
    back = H.read_hdf5(p)
    assert set(back) == {"feats", "boxes", "ids"}
    np.testing.assert_array_equal(back["feats"], feats)
    np.testing.assert_array_equal(back["boxes"], boxes)
    np.testing.assert_array_equal(back["ids"], np.arange(7, dtype=np.int32))
    assert open(p, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    with pytest.raises(ValueError):
        H.write_hdf5(p, {"a/b": feats})


def test_hdf5_file_is_read_by_libhdf5(tmp_path):
    lib = _libhdf5()
    if lib is None:
        pytest.skip("no libhdf5 on this machine")
    from densecap_amd import hdf5_min as H
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((2, 3, 4096)).astype(np.float32)
    boxes = rng.uniform(1, 700, (2, 3, 4)).astype(np.float32)
    p = str(tmp_path / "g.h5")
    H.write_hdf5(p, {"feats": feats, "boxes": boxes})
    np.testing.assert_array_equal(_read_with_libhdf5(lib, p, "feats"), feats)     # extract_features.lua:94
    np.testing.assert_array_equal(_read_with_libhdf5(lib, p, "/boxes"), boxes)    # extract_features.lua:95
