"""Read-only access to the reference's HDF5 feature files without h5py (reference: VL-T5/src/vqa_data_memory.py:141-189).

The reference opens `{source}_obj36.h5` with `h5py.File(path, 'r')` and reads, per item, `f[f'{img_id}/features'].read_direct(feats)`,
`f[f'{img_id}/img_h'][()]`, `f[f'{img_id}/img_w'][()]`, `f[f'{img_id}/boxes'][()]`.  h5py is a wrapper around the HDF5 C library; this
image has no h5py but does carry that library (libhdf5 1.10, `/opt/conda/lib`), so this module binds the dozen C entry points that item
read needs with ctypes and offers the SAME subset of h5py's interface: `H5File(path)[name]` -> dataset with `.shape`, `.dtype`,
`.read_direct(array)`, `[()]` / `[...]`; `name in f`; `f.keys()` / `f[group].keys()`; `KeyError` for a missing object.  `write_feature_file`
creates a file in the reference's layout (what its `feature_extraction/` scripts do through h5py) -- for fixtures and for users who
re-extract features on a machine without h5py.

Host-side I/O only: nothing here touches the GPU.  The conda build of libhdf5 is not thread-safe; the reference's loaders use worker
PROCESSES (one open file per worker, `vqa_data_memory.py:160-164`), which is also the rule here.
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5O_TYPE_GROUP, H5O_TYPE_DATASET = 0, 1
_lib = None


class Hdf5Error(OSError):
    pass


class H5G_info_t(C.Structure):              # H5Gpublic.h (unchanged 1.8 ... 1.14)
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


def find_library():
    """Path of libhdf5: $VQACL_HDF5_LIB, the loader's search path, then the places this image and Debian put it."""
    cand = [os.environ.get("VQACL_HDF5_LIB"), ctypes.util.find_library("hdf5"), ctypes.util.find_library("hdf5_serial")]
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*",
                "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        cand += sorted(glob.glob(pat))
    for c in cand:
        if c:
            return c
    return None


def lib():
    """The loaded library with argument / result types set (hid_t is 64 bits since HDF5 1.10: the default int would truncate handles)."""
    global _lib
    if _lib is not None:
        return _lib
    path = find_library()
    if path is None:
        raise Hdf5Error("libhdf5 not found (set VQACL_HDF5_LIB to its path, or install h5py: H5FeatureSource uses it when present)")
    L = C.CDLL(path)
    sig = {
        "H5open": (herr_t, []), "H5get_libversion": (herr_t, [C.POINTER(C.c_uint)] * 3),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fclose": (herr_t, [hid_t]),
        "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]), "H5Iget_type": (C.c_int, [hid_t]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (herr_t, [hid_t]),
        "H5Gget_info": (herr_t, [hid_t, C.POINTER(H5G_info_t)]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (herr_t, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Screate": (hid_t, [C.c_int]), "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]), "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]), "H5Pset_obj_track_times": (herr_t, [hid_t, C.c_int]),
        "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]), "H5Pset_deflate": (herr_t, [hid_t, C.c_uint]), "H5Pset_shuffle": (herr_t, [hid_t]),
        "H5Zfilter_avail": (C.c_int, [C.c_int]),
        "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]), "H5Tclose": (herr_t, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tset_fields": (herr_t, [hid_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
        "H5Tset_size": (herr_t, [hid_t, C.c_size_t]), "H5Tset_ebias": (herr_t, [hid_t, C.c_size_t]),
    }
    # the version first, through the one entry point whose signature never changed: hid_t is a 32-bit int before 1.10 (every handle
    # below would be passed and returned with the wrong width), and H5Gget_info / H5Oopen are 1.8+
    ver = (C.c_uint(), C.c_uint(), C.c_uint())
    L.H5get_libversion.restype, L.H5get_libversion.argtypes = herr_t, [C.POINTER(C.c_uint)] * 3
    if L.H5get_libversion(*[C.byref(v) for v in ver]) < 0 or (ver[0].value, ver[1].value) < (1, 10):
        raise Hdf5Error(f"{path}: HDF5 {ver[0].value}.{ver[1].value}.{ver[2].value} -- vqacl_amd.hdf5_io needs libhdf5 >= 1.10 (64-bit hid_t); "
                        "set VQACL_HDF5_LIB to a newer library or install h5py (H5FeatureSource uses it when present)")
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    if L.H5open() < 0:
        raise Hdf5Error("H5open failed")
    L.H5Eset_auto2(0, None, None)            # errors come back as return codes -> exceptions here; no stack dumps on stderr (a missing key is normal)
    L._native = {np.dtype(k): hid_t.in_dll(L, v).value for k, v in (
        ("float32", "H5T_NATIVE_FLOAT_g"), ("float64", "H5T_NATIVE_DOUBLE_g"), ("int8", "H5T_NATIVE_INT8_g"), ("uint8", "H5T_NATIVE_UINT8_g"),
        ("int16", "H5T_NATIVE_INT16_g"), ("uint16", "H5T_NATIVE_UINT16_g"), ("int32", "H5T_NATIVE_INT32_g"), ("uint32", "H5T_NATIVE_UINT32_g"),
        ("int64", "H5T_NATIVE_INT64_g"), ("uint64", "H5T_NATIVE_UINT64_g"))}
    _lib = L
    return L


def library_version():
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    lib().H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def _native(dtype):
    try:
        return lib()._native[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"no HDF5 native type for {np.dtype(dtype)}") from None


class Dataset:
    """One open dataset (h5py.Dataset's read side)."""

    def __init__(self, did, name):
        L = lib()
        self._id, self.name = did, name
        sp = L.H5Dget_space(did)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            L.H5Sget_simple_extent_dims(sp, dims, None)
        L.H5Sclose(sp)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        tp = L.H5Dget_type(did)
        cls, size, sign = L.H5Tget_class(tp), L.H5Tget_size(tp), L.H5Tget_sign(tp)
        L.H5Tclose(tp)
        if cls == H5T_FLOAT and size in (2, 4, 8):
            # (IEEE half: what h5py writes for a float16 array.  libhdf5 < 1.14.4 has no native half type; the library converts between
            #  any two IEEE-style float layouts, so half datasets are read as float32 -- exact -- and narrowed back for `[()]`)
            self.dtype = np.dtype(f"float{8 * size}")
        elif cls == H5T_INTEGER and size in (1, 2, 4, 8):
            self.dtype = np.dtype(f"{'u' if sign == 0 else ''}int{8 * size}")
        else:
            self.dtype = None               # strings / compounds: nothing the feature files hold; reading raises

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    def read_direct(self, dest):
        """Read the whole dataset into `dest` (C-contiguous numpy array of the dataset's shape; HDF5 converts to its dtype)."""
        if self._id is None:
            raise ValueError("dataset of a closed file")
        if not isinstance(dest, np.ndarray) or not dest.flags.c_contiguous or not dest.flags.writeable:
            raise TypeError("read_direct needs a writeable C-contiguous numpy array")
        if tuple(dest.shape) != self.shape:
            raise TypeError(f"can't broadcast {self.shape} -> {tuple(dest.shape)}")
        if self.dtype is None:
            raise TypeError(f"{self.name}: unsupported HDF5 type class")
        if dest.dtype == np.float16:
            wide = np.empty(self.shape, dtype=np.float32)
            self.read_direct(wide)
            dest[...] = wide                    # (exact for a half dataset; round to nearest otherwise, as h5py's conversion)
            return
        if lib().H5Dread(self._id, _native(dest.dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, dest.ctypes.data_as(C.c_void_p)) < 0:
            raise Hdf5Error(f"H5Dread failed on {self.name}")

    def __getitem__(self, key):
        if self.dtype is None:
            raise TypeError(f"{self.name}: unsupported HDF5 type class")
        out = np.empty(self.shape, dtype=self.dtype)
        self.read_direct(out)
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            return out[()] if self.shape == () else out
        return out[key]

    def __len__(self):
        if not self.shape:
            raise TypeError("scalar dataset")
        return self.shape[0]

    def close(self):
        if self._id is not None:
            lib().H5Dclose(self._id)
            self._id = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """A group (or the file's root): name lookup relative to it."""

    def __init__(self, file, path):
        self._file, self._path = file, path.strip("/")

    def _abs(self, name):
        name = str(name)
        return name if name.startswith("/") else ("/" + "/".join(p for p in (self._path, name) if p))

    def __contains__(self, name):
        return self._file._exists(self._abs(name))

    def __getitem__(self, name):
        return self._file._open(self._abs(name))

    def keys(self):
        return self._file._children("/" + self._path)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())


class H5File(Group):
    """`h5py.File(path, 'r')` for the reads the reference's dataset class makes."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("read-only (write_feature_file() creates files)")
        self.filename = os.fspath(path)
        self._id = lib().H5Fopen(self.filename.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        if self._id < 0:
            self._id = None
            raise Hdf5Error(f"unable to open {self.filename!r} as an HDF5 file")
        Group.__init__(self, self, "")

    def _check(self):
        if self._id is None:
            raise ValueError("closed file")

    def _exists(self, path):
        """h5py semantics: every intermediate link must exist (H5Lexists fails on a missing parent instead of answering)."""
        self._check()
        cur = ""
        for part in [p for p in path.split("/") if p]:
            cur += "/" + part
            if lib().H5Lexists(self._id, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def _open(self, path):
        if not self._exists(path):
            raise KeyError(f"Unable to open object (object {path.strip('/')!r} doesn't exist)")
        L = lib()
        oid = L.H5Oopen(self._id, path.encode(), H5P_DEFAULT)
        if oid < 0:
            raise KeyError(f"Unable to open object {path!r}")
        kind = L.H5Iget_type(oid)               # H5I_GROUP = 2, H5I_DATASET = 5
        L.H5Oclose(oid)
        if kind == 2:
            return Group(self, path)
        if kind == 5:
            did = L.H5Dopen2(self._id, path.encode(), H5P_DEFAULT)
            if did < 0:
                raise Hdf5Error(f"H5Dopen2 failed on {path!r}")
            return Dataset(did, path)
        raise TypeError(f"{path!r}: neither a group nor a dataset")

    def _children(self, path):
        self._check()
        L = lib()
        oid = L.H5Oopen(self._id, (path or "/").encode(), H5P_DEFAULT)
        if oid < 0:
            raise KeyError(path)
        info = H5G_info_t()                   # (H5Gget_num_objs is a deprecated symbol: absent from builds without the 1.6 API)
        if L.H5Gget_info(oid, C.byref(info)) < 0:
            L.H5Oclose(oid)
            raise TypeError(f"{path!r} is not a group")
        out = []
        buf = C.create_string_buffer(1024)
        for i in range(info.nlinks):
            ln = L.H5Lget_name_by_idx(oid, b".", 0, 0, i, buf, 1024, H5P_DEFAULT)      # H5_INDEX_NAME, H5_ITER_INC
            if ln >= 0:
                out.append(buf.value.decode())
        L.H5Oclose(oid)
        return out

    def close(self):
        if self._id is not None:
            lib().H5Fclose(self._id)
            self._id = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _half_type():
    """IEEE binary16 as an HDF5 type, built the way h5py builds it (a copy of the 32-bit IEEE type with 5 exponent / 10 mantissa bits)."""
    L = lib()
    tid = L.H5Tcopy(hid_t.in_dll(L, "H5T_IEEE_F32LE_g").value)
    if tid < 0 or L.H5Tset_fields(tid, 15, 10, 5, 0, 10) < 0 or L.H5Tset_size(tid, 2) < 0 or L.H5Tset_ebias(tid, 15) < 0:
        raise Hdf5Error("could not build the IEEE half type")
    return tid


def write_feature_file(path, items, track_times=False, compression=None, compression_opts=4, shuffle=False, chunks=None):
    """Create `path` in the reference's layout: for every (img_id, dict) of `items` a group `{img_id}` holding one dataset per dict
    entry -- arrays as simple dataspaces of their dtype, Python / numpy scalars as scalar datasets (how h5py stores `grp['img_w'] = w`).
    The reference's files carry `features [n,2048] f32`, `boxes [n,4] f32`, `img_w`, `img_h` (+ `obj_id`, `obj_conf`, `attr_id`,
    `attr_conf` that its dataset class never reads).  track_times=False (h5py's default for datasets too): no modification times in
    the object headers, so the same items give the same bytes.  The reference's extraction scripts write `grp['features'] = array`
    (contiguous, uncompressed: the default here); `compression="gzip"` (+ `compression_opts` level, `shuffle`, `chunks` = a shape or None
    for one chunk per dataset) stores the non-scalar datasets chunked through the deflate filter, as `create_dataset(..., compression=
    "gzip")` would -- files re-packed that way read back through the same calls.  float16 arrays are stored as IEEE half."""
    if compression not in (None, "gzip"):
        raise ValueError("compression must be None or 'gzip'")
    if compression and lib().H5Zfilter_avail(1) <= 0:
        raise Hdf5Error("this libhdf5 was built without the deflate filter")
    L = lib()
    plists = []

    def plist(cls_symbol):
        if track_times:
            return H5P_DEFAULT
        pid = L.H5Pcreate(hid_t.in_dll(L, cls_symbol).value)
        if pid < 0 or L.H5Pset_obj_track_times(pid, 0) < 0:
            raise Hdf5Error("could not build the creation property list")
        plists.append(pid)
        return pid
    fcpl, gcpl, dcpl = plist("H5P_CLS_FILE_CREATE_ID_g"), plist("H5P_CLS_GROUP_CREATE_ID_g"), plist("H5P_CLS_DATASET_CREATE_ID_g")
    fid = L.H5Fcreate(os.fspath(path).encode(), H5F_ACC_TRUNC, fcpl, H5P_DEFAULT)
    if fid < 0:
        raise Hdf5Error(f"unable to create {path!r}")
    try:
        for img_id, entry in (items.items() if hasattr(items, "items") else items):
            gid = L.H5Gcreate2(fid, str(img_id).encode(), H5P_DEFAULT, gcpl, H5P_DEFAULT)
            if gid < 0:
                raise Hdf5Error(f"H5Gcreate2 failed for {img_id!r}")
            try:
                for key, val in entry.items():
                    a = np.asarray(val)
                    if a.ndim:
                        a = np.ascontiguousarray(a)         # (on a 0-d value it would return shape (1,): scalars stay scalar datasets)
                    if a.dtype == np.bool_ or a.dtype.kind not in "fiu":
                        raise TypeError(f"{img_id}/{key}: dtype {a.dtype} not supported")
                    if a.ndim == 0:
                        sp = L.H5Screate(H5S_SCALAR)
                    else:
                        sp = L.H5Screate_simple(a.ndim, (hsize_t * a.ndim)(*a.shape), None)
                    half = a.dtype == np.float16
                    if half:                        # stored as IEEE half, handed over as float32 (the library narrows: exact)
                        a = a.astype(np.float32)
                    ftp = _half_type() if half else _native(a.dtype)
                    mtp = _native(a.dtype)
                    this_dcpl = dcpl
                    if compression and a.ndim:
                        this_dcpl = L.H5Pcreate(hid_t.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value)
                        ch = tuple(chunks) if chunks is not None and len(chunks) == a.ndim else a.shape
                        ch = tuple(max(1, min(int(c), int(n))) for c, n in zip(ch, a.shape))
                        bad = this_dcpl < 0 or (not track_times and L.H5Pset_obj_track_times(this_dcpl, 0) < 0) \
                            or L.H5Pset_chunk(this_dcpl, a.ndim, (hsize_t * a.ndim)(*ch)) < 0 \
                            or (shuffle and L.H5Pset_shuffle(this_dcpl) < 0) or L.H5Pset_deflate(this_dcpl, int(compression_opts)) < 0
                        if bad:
                            raise Hdf5Error("could not build the chunked / deflate creation property list")
                    did = L.H5Dcreate2(gid, key.encode(), ftp, sp, H5P_DEFAULT, this_dcpl, H5P_DEFAULT)
                    ok = did >= 0 and L.H5Dwrite(did, mtp, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) >= 0
                    if did >= 0:
                        L.H5Dclose(did)
                    if this_dcpl != dcpl and this_dcpl >= 0:
                        L.H5Pclose(this_dcpl)
                    if half:
                        L.H5Tclose(ftp)
                    L.H5Sclose(sp)
                    if not ok:
                        raise Hdf5Error(f"writing {img_id}/{key} failed")
            finally:
                L.H5Gclose(gid)
    finally:
        L.H5Fclose(fid)
        for pid in plists:
            L.H5Pclose(pid)
