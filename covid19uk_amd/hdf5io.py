"""Minimal HDF5 reader/writer over the C library (ctypes -> libhdf5).

The reference reads its input with xarray/netCDF4 (NetCDF4 files ARE HDF5
files; covid19uk/inference/inference.py:481-485) and writes `posterior.hd5`
with h5py through gemlib's `Posterior` (inference.py:352-358,588-592).  Neither
package exists in this image, but libhdf5 does (/opt/conda/lib/libhdf5.so
1.10.6), so this module binds the dozen C calls the path needs: create / open
files, nested groups, N-d datasets of float64 / int32 / int8 / fixed strings,
hyperslab writes along the first axis (Posterior.write_samples'
`first_dim_offset`), whole-dataset reads with type conversion (which also
decodes chunked / deflated NetCDF4 variables) and string attributes -- and, for
the post-processing outputs the reference writes with `xarray.Dataset.to_netcdf
(group=...)` (posterior/predict.py:124-147, reproduction_number.py:73-88), the
netCDF-4 conventions on top of HDF5: dimension scales attached to every
variable (libhdf5_hl's H5DS calls), `_Netcdf4Dimid`, variable-length string and
CF-encoded time coordinates (`File.write_netcdf_group`), so that
`xarray.open_dataset(file, group=...)` opens them.  `bool` datasets are the
int8 enum {FALSE, TRUE} h5py / gemlib's Posterior store numpy bools as.

If no libhdf5 can be loaded, `available()` is False and the CLI falls back to
`.npz` files (inference.py of this package); nothing here touches the GPU.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import os

import numpy as np

_CANDIDATES = [
    os.environ.get("SEIR_LIBHDF5", ""),
    "/opt/conda/lib/libhdf5.so",
    "/opt/conda/lib/libhdf5.so.103",
    ctypes.util.find_library("hdf5") or "",
    ctypes.util.find_library("hdf5_serial") or "",
    "libhdf5.so", "libhdf5_serial.so",
]

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
_lib = None
_ids = {}

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL, H5S_SELECT_SET = 0, 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5D_CONTIGUOUS = 1
_hl = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    err = None
    for cand in _CANDIDATES:
        if not cand:
            continue
        try:
            lib = ctypes.CDLL(cand)
            lib.H5open()
            break
        except (OSError, AttributeError) as e:      # pragma: no cover - depends on the host
            err = e
            lib = None
    if lib is None:
        raise OSError(f"libhdf5 not found ({err}); set SEIR_LIBHDF5=/path/to/libhdf5.so")

    def fn(name, res, *args):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, list(args)
        return f
    c_char_p, c_void_p, c_int, c_uint, c_size_t = (ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_uint, ctypes.c_size_t)
    hp = ctypes.POINTER(hsize_t)
    fn("H5Fcreate", hid_t, c_char_p, c_uint, hid_t, hid_t)
    fn("H5Fopen", hid_t, c_char_p, c_uint, hid_t)
    fn("H5Fclose", c_int, hid_t)
    fn("H5Fflush", c_int, hid_t, c_int)
    fn("H5Gopen2", hid_t, hid_t, c_char_p, hid_t)
    fn("H5Gclose", c_int, hid_t)
    fn("H5Screate_simple", hid_t, c_int, hp, hp)
    fn("H5Sclose", c_int, hid_t)
    fn("H5Sselect_hyperslab", c_int, hid_t, c_int, hp, hp, hp, hp)
    fn("H5Sget_simple_extent_ndims", c_int, hid_t)
    fn("H5Sget_simple_extent_dims", c_int, hid_t, hp, hp)
    fn("H5Dcreate2", hid_t, hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
    fn("H5Dopen2", hid_t, hid_t, c_char_p, hid_t)
    fn("H5Dwrite", c_int, hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p)
    fn("H5Dread", c_int, hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p)
    fn("H5Dclose", c_int, hid_t)
    fn("H5Dget_space", hid_t, hid_t)
    fn("H5Dget_type", hid_t, hid_t)
    fn("H5Tget_class", c_int, hid_t)
    fn("H5Tget_size", c_size_t, hid_t)
    fn("H5Tcopy", hid_t, hid_t)
    fn("H5Tset_size", c_int, hid_t, c_size_t)
    fn("H5Tclose", c_int, hid_t)
    fn("H5Tis_variable_str", c_int, hid_t)
    fn("H5Pcreate", hid_t, hid_t)
    fn("H5Pset_chunk", c_int, hid_t, c_int, hp)
    fn("H5Pset_alloc_time", c_int, hid_t, c_int)
    fn("H5Pset_fill_time", c_int, hid_t, c_int)
    fn("H5Dget_offset", ctypes.c_uint64, hid_t)
    fn("H5Pset_create_intermediate_group", c_int, hid_t, c_uint)
    fn("H5Pclose", c_int, hid_t)
    fn("H5Lexists", c_int, hid_t, c_char_p, hid_t)
    fn("H5Aexists_by_name", c_int, hid_t, c_char_p, c_char_p, hid_t)
    fn("H5Aopen_by_name", hid_t, hid_t, c_char_p, c_char_p, hid_t, hid_t)
    fn("H5Aget_type", hid_t, hid_t)
    fn("H5Aread", c_int, hid_t, hid_t, c_void_p)
    fn("H5Aclose", c_int, hid_t)
    fn("H5Tenum_create", hid_t, hid_t)
    fn("H5Tenum_insert", c_int, hid_t, c_char_p, c_void_p)
    fn("H5Tget_super", hid_t, hid_t)
    fn("H5Tequal", c_int, hid_t, hid_t)
    fn("H5Acreate_by_name", hid_t, hid_t, c_char_p, c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
    fn("H5Awrite", c_int, hid_t, hid_t, c_void_p)
    fn("H5Screate", hid_t, c_int)
    fn("H5Dget_create_plist", hid_t, hid_t)
    fn("H5Pget_layout", c_int, hid_t)
    fn("H5Pget_nfilters", c_int, hid_t)
    fn("H5Pget_alloc_time", c_int, hid_t, ctypes.POINTER(c_int))
    fn("H5Fget_access_plist", hid_t, hid_t)
    fn("H5Pget_driver", hid_t, hid_t)
    fn("H5Gcreate2", hid_t, hid_t, c_char_p, hid_t, hid_t, hid_t)
    fn("H5Lget_name_by_idx", ctypes.c_ssize_t, hid_t, c_char_p, c_int, c_int, hsize_t, c_char_p, c_size_t, hid_t)
    fn("H5Gget_info_by_name", c_int, hid_t, c_char_p, c_void_p, hid_t)

    class _GInfo(ctypes.Structure):
        _fields_ = [("storage_type", c_int), ("nlinks", hsize_t), ("max_corder", ctypes.c_int64), ("mounted", c_uint)]
    lib._GInfo = _GInfo
    fn("H5Gget_info", c_int, hid_t, ctypes.POINTER(_GInfo))
    fn("H5Eset_auto2", c_int, hid_t, c_void_p, c_void_p)
    lib.H5Eset_auto2(0, None, None)                  # errors are reported through return codes
    for name in ("H5T_NATIVE_DOUBLE_g", "H5T_NATIVE_INT32_g", "H5T_NATIVE_INT8_g", "H5T_NATIVE_INT64_g",
                 "H5T_C_S1_g", "H5P_CLS_DATASET_CREATE_ID_g", "H5P_CLS_LINK_CREATE_ID_g"):
        _ids[name] = hid_t.in_dll(lib, name).value
    _lib = lib
    return lib


def _load_hl():
    """libhdf5_hl (the dimension-scale calls), next to the libhdf5 that was loaded; None if there is none."""
    global _hl
    if _hl is not None:
        return _hl or None
    lib = _load()
    base = os.path.dirname(getattr(lib, "_name", "") or "")
    hl = None
    for cand in (os.path.join(base, "libhdf5_hl.so") if base else "", "/opt/conda/lib/libhdf5_hl.so",
                 ctypes.util.find_library("hdf5_hl") or "", ctypes.util.find_library("hdf5_serial_hl") or "",
                 "libhdf5_hl.so", "libhdf5_serial_hl.so"):
        if not cand:
            continue
        try:
            hl = ctypes.CDLL(cand)
            hl.H5DSset_scale.restype, hl.H5DSset_scale.argtypes = ctypes.c_int, [hid_t, ctypes.c_char_p]
            hl.H5DSattach_scale.restype, hl.H5DSattach_scale.argtypes = ctypes.c_int, [hid_t, hid_t, ctypes.c_uint]
            break
        except (OSError, AttributeError):
            hl = None
    _hl = hl or False
    return hl


def available() -> bool:
    try:
        _load()
        return True
    except OSError:
        return False


def _check(rc, what):
    if rc < 0:
        raise OSError(f"HDF5: {what} failed")
    return rc


_NATIVE = {np.dtype("float64"): "H5T_NATIVE_DOUBLE_g", np.dtype("int32"): "H5T_NATIVE_INT32_g",
           np.dtype("int8"): "H5T_NATIVE_INT8_g", np.dtype("int64"): "H5T_NATIVE_INT64_g"}


def _dims(shape):
    return (hsize_t * len(shape))(*[int(x) for x in shape])


def _bool_enum(lib):
    """The HDF5 type h5py maps numpy.bool_ to: an enum over int8 with members FALSE = 0 and TRUE = 1."""
    tid = lib.H5Tenum_create(_ids["H5T_NATIVE_INT8_g"])
    for name, val in ((b"FALSE", 0), (b"TRUE", 1)):
        v = ctypes.c_int8(val)
        lib.H5Tenum_insert(tid, name, ctypes.byref(v))
    return tid


class File:
    """`File(path, "w" | "r" | "a")`; dataset names are '/'-separated paths."""

    def __init__(self, path, mode="r"):
        lib = _load()
        self._lib = lib
        p = os.fsencode(path)
        if mode == "w":
            self._f = lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "a":
            self._f = lib.H5Fopen(p, H5F_ACC_RDWR, H5P_DEFAULT) if os.path.exists(path) else \
                lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            self._f = lib.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT)
        _check(self._f, f"open {path!r} ({mode})")
        self._dsets = {}
        self._path = path
        self._raw = {}            # name -> (file offset, shape): datasets created with raw=True (write_rows_parallel)
        self._raw_fd = -1
        self._pool = None

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_f", -1) >= 0:
            for d in self._dsets.values():
                self._lib.H5Dclose(d)
            self._dsets = {}
            self._lib.H5Fclose(self._f)
            self._f = -1
            if self._raw_fd >= 0:
                os.close(self._raw_fd)
                self._raw_fd = -1
            if self._pool is not None:
                self._pool.shutdown()
                self._pool = None

    def flush(self):
        self._lib.H5Fflush(self._f, 1)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- structure ------------------------------------------------------------
    def exists(self, name) -> bool:
        parts = [p for p in name.split("/") if p]
        path = ""
        for part in parts:
            path += "/" + part
            if self._lib.H5Lexists(self._f, path.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def _open(self, name):
        if name not in self._dsets:
            d = self._lib.H5Dopen2(self._f, name.encode(), H5P_DEFAULT)
            _check(d, f"open dataset {name!r}")
            self._dsets[name] = d
        return self._dsets[name]

    def create_dataset(self, name, shape, dtype, chunk_rows=None, raw=False):
        """N-d dataset (intermediate groups are created).  dtype: float64, int32, int8, int64
        or 'S<n>' fixed-length strings.  chunk_rows: chunk extent of axis 0 (None = contiguous).
        raw: contiguous, its file space allocated now and never filled, so that write_rows_parallel can put row
        blocks straight into the file from several threads (the library serialises H5Dwrite and copies through
        one core: ~2 GB/s for the float64 event tensor, a quarter of what the sampler delivers for one chain)."""
        lib = self._lib
        own_type = False
        if isinstance(dtype, str) and dtype == "vlen_str":          # variable-length UTF-8 strings (netCDF NC_STRING)
            dt = np.dtype(object)
            tid = lib.H5Tcopy(_ids["H5T_C_S1_g"])
            lib.H5Tset_size(tid, H5T_VARIABLE)
            own_type = True
        elif np.dtype(dtype) == np.bool_:                           # h5py's bool: int8 enum {FALSE = 0, TRUE = 1}
            dt = np.dtype(np.int8)
            tid = _bool_enum(lib)
            own_type = True
        else:
            dt = np.dtype(dtype)
            if dt.kind == "S":
                tid = lib.H5Tcopy(_ids["H5T_C_S1_g"])
                lib.H5Tset_size(tid, dt.itemsize)
                own_type = True
            else:
                tid = _ids[_NATIVE[dt]]
        shape = tuple(int(x) for x in shape)
        space = lib.H5Screate_simple(len(shape), _dims(shape) if shape else None, None)
        lcpl = lib.H5Pcreate(_ids["H5P_CLS_LINK_CREATE_ID_g"])
        lib.H5Pset_create_intermediate_group(lcpl, 1)
        dcpl = H5P_DEFAULT
        raw = bool(raw) and dt.kind not in "SO" and not own_type and len(shape) >= 1 and all(x > 0 for x in shape)
        if raw:
            dcpl = lib.H5Pcreate(_ids["H5P_CLS_DATASET_CREATE_ID_g"])
            lib.H5Pset_alloc_time(dcpl, 1)           # H5D_ALLOC_TIME_EARLY
            lib.H5Pset_fill_time(dcpl, 1)            # H5D_FILL_TIME_NEVER
        elif chunk_rows and shape and shape[0] > 0:
            dcpl = lib.H5Pcreate(_ids["H5P_CLS_DATASET_CREATE_ID_g"])
            lib.H5Pset_chunk(dcpl, len(shape), _dims((min(int(chunk_rows), shape[0]),) + shape[1:]))
        d = lib.H5Dcreate2(self._f, name.encode(), tid, space, lcpl, dcpl, H5P_DEFAULT)
        lib.H5Sclose(space)
        lib.H5Pclose(lcpl)
        if dcpl != H5P_DEFAULT:
            lib.H5Pclose(dcpl)
        if own_type:
            lib.H5Tclose(tid)
        _check(d, f"create dataset {name!r}")
        self._dsets[name] = d
        if raw:
            # rows go straight into the file only if the library really laid the dataset out the way that needs: one
            # contiguous extent with a file address, no filter, allocated at creation, a file on the plain POSIX driver;
            # anything else and H5Dwrite serves it like any other dataset
            off = int(lib.H5Dget_offset(d))
            if off != 0xFFFFFFFFFFFFFFFF and self._raw_layout_ok(d):
                self._raw[name] = (off, shape, dt)

    def _raw_layout_ok(self, d):
        lib = self._lib
        ok = False
        dcpl = lib.H5Dget_create_plist(d)
        if dcpl >= 0:
            at = ctypes.c_int(-1)
            ok = (lib.H5Pget_layout(dcpl) == H5D_CONTIGUOUS and lib.H5Pget_nfilters(dcpl) == 0 and
                  lib.H5Pget_alloc_time(dcpl, ctypes.byref(at)) >= 0 and at.value == 1)        # H5D_ALLOC_TIME_EARLY
            lib.H5Pclose(dcpl)
        if ok:
            fapl = lib.H5Fget_access_plist(self._f)
            if fapl >= 0:
                # the id of the plain POSIX ("sec2") driver: H5FD_sec2_init() is what the H5FD_SEC2 macro expands to up to
                # HDF5 1.14; the development branch makes it an id global instead.  A library that exports neither: no raw
                # writes, H5Dwrite serves the dataset like any other.
                sec2 = -1
                try:
                    init = lib.H5FD_sec2_init
                    init.restype = hid_t
                    sec2 = init()
                except AttributeError:
                    for sym in ("H5FD_SEC2_id_g", "H5FD_SEC2_g"):
                        try:
                            sec2 = hid_t.in_dll(lib, sym).value
                            break
                        except ValueError:
                            continue
                ok = sec2 > 0 and lib.H5Pget_driver(fapl) == sec2
                lib.H5Pclose(fapl)
            else:
                ok = False
        return ok

    def write_rows_parallel(self, name, array, offset=0, threads=None):
        """ds[offset:offset+n] = array for a dataset created with raw=True: the rows are converted to the file's type
        and written at their file address by `threads` workers (numpy casts and pwrite both release the GIL).
        Falls back to write() for any other dataset."""
        if name not in self._raw:
            return self.write(name, array, offset)
        off, shape, dt = self._raw[name]
        a = np.asarray(array)
        if a.ndim == len(shape) - 1:
            a = a[None]
        if tuple(a.shape[1:]) != shape[1:] or offset + a.shape[0] > shape[0] or offset < 0:
            raise ValueError(f"write to {name!r}: block {a.shape} at {offset} does not fit {shape}")
        if self._raw_fd < 0:
            self._raw_fd = os.open(self._path, os.O_WRONLY)
        if threads is None:
            threads = max(2, min(8, (os.cpu_count() or 4) // 2))
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=max(1, int(threads)))
            self._scratch = {}
        n = a.shape[0]
        row_bytes = int(np.prod(shape[1:], dtype=np.int64)) * dt.itemsize
        nw = max(1, min(int(threads), n))
        bounds = [n * k // nw for k in range(nw + 1)]
        fd = self._raw_fd

        def work(k):
            lo, hi = bounds[k], bounds[k + 1]
            if hi <= lo:
                return
            src = a[lo:hi]
            if src.dtype != dt or not src.flags.c_contiguous:
                buf = self._scratch.get(k)
                if buf is None or buf.shape[0] < hi - lo or buf.shape[1:] != src.shape[1:]:
                    buf = self._scratch[k] = np.empty((hi - lo,) + src.shape[1:], dt)
                dst = buf[:hi - lo]
                np.copyto(dst, src, casting="unsafe")
                src = dst
            mv = memoryview(src).cast("B")
            pos, done = off + (offset + lo) * row_bytes, 0
            while done < len(mv):
                done += os.pwrite(fd, mv[done:done + (1 << 30)], pos + done)

        for f in [self._pool.submit(work, k) for k in range(nw)]:
            f.result()

    def write(self, name, array, offset=0):
        """ds[offset:offset+n] = array (the rest of the axes must match)."""
        if name in self._raw:
            # written at its file address, behind the library's back: every write of such a dataset goes that way, so
            # that no H5Dwrite (and no stale sieve buffer) ever covers the same bytes
            return self.write_rows_parallel(name, array, offset)
        lib = self._lib
        d = self._open(name)
        a = np.ascontiguousarray(array)
        ftype = lib.H5Dget_type(d)
        cls = lib.H5Tget_class(ftype)
        keep = None
        if cls == H5T_STRING and lib.H5Tis_variable_str(ftype) > 0:
            mtype, own = lib.H5Tcopy(ftype), True
            keep = [x if isinstance(x, bytes) else str(x).encode("utf-8") for x in np.asarray(array, dtype=object).reshape(-1)]
            a = np.array([ctypes.cast(ctypes.c_char_p(x), ctypes.c_void_p).value for x in keep], dtype=np.uint64).reshape(np.shape(array))
        elif cls == H5T_STRING:
            mtype, own = lib.H5Tcopy(ftype), True
            a = np.ascontiguousarray(a.astype(f"S{lib.H5Tget_size(ftype)}"))
        elif cls == H5T_ENUM:
            mtype, own = lib.H5Tcopy(ftype), True
            a = np.ascontiguousarray(a.astype(np.int8))
        else:
            if a.dtype == np.bool_:
                a = a.astype(np.int8)
            if a.dtype not in _NATIVE:
                a = a.astype(np.float64)
            mtype, own = _ids[_NATIVE[a.dtype]], False
        lib.H5Tclose(ftype)
        fspace = lib.H5Dget_space(d)
        nd = lib.H5Sget_simple_extent_ndims(fspace)
        if nd == 0:
            rc = lib.H5Dwrite(d, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p))
        else:
            fd = (hsize_t * nd)()
            lib.H5Sget_simple_extent_dims(fspace, fd, None)
            shape = tuple(a.shape) if a.ndim == nd else (1,) + tuple(a.shape)
            if tuple(fd)[1:] != shape[1:] or offset + shape[0] > fd[0]:
                lib.H5Sclose(fspace)
                raise ValueError(f"write to {name!r}: block {shape} at {offset} does not fit {tuple(fd)}")
            start = _dims((offset,) + (0,) * (nd - 1))
            lib.H5Sselect_hyperslab(fspace, H5S_SELECT_SET, start, None, _dims(shape), None)
            mspace = lib.H5Screate_simple(nd, _dims(shape), None)
            rc = lib.H5Dwrite(d, mtype, mspace, fspace, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p))
            lib.H5Sclose(mspace)
        lib.H5Sclose(fspace)
        if own:
            lib.H5Tclose(mtype)
        _check(rc, f"write {name!r}")

    def shape(self, name):
        lib = self._lib
        d = self._open(name)
        sp = lib.H5Dget_space(d)
        nd = lib.H5Sget_simple_extent_ndims(sp)
        fd = (hsize_t * max(nd, 1))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(sp, fd, None)
        lib.H5Sclose(sp)
        return tuple(int(x) for x in fd[:nd])

    def read(self, name):
        """Whole dataset as float64 / int64 / fixed bytes (by HDF5 type class)."""
        if name in self._raw:
            return self._read_raw(name, 0, self._raw[name][1][0], 1)
        lib = self._lib
        d = self._open(name)
        shape = self.shape(name)
        ftype = lib.H5Dget_type(d)
        cls = lib.H5Tget_class(ftype)
        if cls == H5T_STRING:
            if lib.H5Tis_variable_str(ftype) > 0:
                n = int(np.prod(shape)) if shape else 1
                ptrs = (ctypes.c_char_p * n)()
                rc = lib.H5Dread(d, ftype, H5S_ALL, H5S_ALL, H5P_DEFAULT, ptrs)
                out = np.array([p or b"" for p in ptrs]).reshape(shape)
            else:
                out = np.empty(shape, dtype=f"S{lib.H5Tget_size(ftype)}")
                rc = lib.H5Dread(d, ftype, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p))
        elif cls == H5T_ENUM:                        # h5py bools: the enum's own type in memory, values 0 / 1
            out8 = np.empty(shape, dtype=np.int8)
            rc = lib.H5Dread(d, ftype, H5S_ALL, H5S_ALL, H5P_DEFAULT, out8.ctypes.data_as(ctypes.c_void_p)) \
                if lib.H5Tget_size(ftype) == 1 else -1
            out = out8.astype(np.int64)
        else:
            out = np.empty(shape, dtype=np.float64 if cls == H5T_FLOAT else np.int64)
            rc = lib.H5Dread(d, _ids[_NATIVE[out.dtype]], H5S_ALL, H5S_ALL, H5P_DEFAULT,
                             out.ctypes.data_as(ctypes.c_void_p))
        lib.H5Tclose(ftype)
        _check(rc, f"read {name!r}")
        return out

    def read_rows(self, name, start, count, stride=1):
        """ds[start : start + count*stride : stride] read as ONE strided hyperslab along axis 0 (what h5py
        does for `ds[a:b:c]`): only the selected rows leave the file.  Numeric datasets only."""
        if name in self._raw:
            return self._read_raw(name, start, count, stride)
        lib = self._lib
        d = self._open(name)
        shape = self.shape(name)
        start, count, stride = int(start), int(count), int(stride)
        if not shape or stride < 1 or start < 0 or count < 0 or (count and start + (count - 1) * stride >= shape[0]):
            raise ValueError(f"read_rows({name!r}): rows {start}:+{count}:{stride} outside {shape}")
        ftype = lib.H5Dget_type(d)
        cls = lib.H5Tget_class(ftype)
        if cls == H5T_STRING:
            lib.H5Tclose(ftype)
            raise ValueError(f"read_rows({name!r}): string datasets are read whole")
        enum = cls == H5T_ENUM
        out = np.empty((count,) + shape[1:], dtype=np.int8 if enum else np.float64 if cls == H5T_FLOAT else np.int64)
        mem_t = lib.H5Tcopy(ftype) if enum else _ids[_NATIVE[out.dtype]]
        lib.H5Tclose(ftype)
        if count == 0:
            if enum:
                lib.H5Tclose(mem_t)
            return out.astype(np.int64) if enum else out
        nd = len(shape)
        fspace = lib.H5Dget_space(d)
        _check(lib.H5Sselect_hyperslab(fspace, H5S_SELECT_SET, _dims((start,) + (0,) * (nd - 1)),
                                       _dims((stride,) + (1,) * (nd - 1)), _dims((count,) + shape[1:]), None),
               f"select rows of {name!r}")
        mspace = lib.H5Screate_simple(nd, _dims(out.shape), None)
        rc = lib.H5Dread(d, mem_t, mspace, fspace, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p))
        lib.H5Sclose(mspace)
        lib.H5Sclose(fspace)
        if enum:
            lib.H5Tclose(mem_t)
        _check(rc, f"read rows of {name!r}")
        return out.astype(np.int64) if enum else out

    def _read_raw(self, name, start, count, stride):
        """Rows of a dataset this handle writes at its file address (write_rows_parallel), read the same way: the
        library never saw those bytes, and its buffers are not asked about them."""
        off, shape, dt = self._raw[name]
        start, count, stride = int(start), int(count), int(stride)
        if stride < 1 or start < 0 or count < 0 or (count and start + (count - 1) * stride >= shape[0]):
            raise ValueError(f"read_rows({name!r}): rows {start}:+{count}:{stride} outside {shape}")
        row = int(np.prod(shape[1:], dtype=np.int64)) * dt.itemsize
        out = np.empty((count,) + shape[1:], dt)
        fd = os.open(self._path, os.O_RDONLY)
        try:
            mv = memoryview(out).cast("B")
            for i in range(count):
                pos, done = off + (start + i * stride) * row, 0
                while done < row:
                    chunk = os.pread(fd, row - done, pos + done)
                    if not chunk:
                        raise OSError(f"read of {name!r}: unexpected end of file")
                    mv[i * row + done:i * row + done + len(chunk)] = chunk
                    done += len(chunk)
        finally:
            os.close(fd)
        return out.astype(np.float64 if dt.kind == "f" else np.int64, copy=False)

    # -- attributes / netCDF-4 conventions ---------------------------------------
    def write_attr(self, obj, attr, value):
        """Scalar or 1-d attribute on an object: str (fixed-length, as netCDF NC_CHAR), int32 / int64 or float64."""
        lib = self._lib
        if isinstance(value, str):
            raw = value.encode("utf-8")
            tid, own = lib.H5Tcopy(_ids["H5T_C_S1_g"]), True
            lib.H5Tset_size(tid, max(len(raw), 1))
            space = lib.H5Screate(0)
            buf = ctypes.create_string_buffer(raw, max(len(raw), 1))
        else:
            a = np.ascontiguousarray(value)
            if a.dtype not in _NATIVE:
                a = a.astype(np.float64 if a.dtype.kind == "f" else np.int32)
            tid, own = _ids[_NATIVE[a.dtype]], False
            space = lib.H5Screate(0) if a.ndim == 0 else lib.H5Screate_simple(1, _dims(a.shape), None)
            buf = a.ctypes.data_as(ctypes.c_void_p)
        at = lib.H5Acreate_by_name(self._f, obj.encode(), attr.encode(), tid, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        rc = lib.H5Awrite(at, tid, buf) if at >= 0 else -1
        if at >= 0:
            lib.H5Aclose(at)
        lib.H5Sclose(space)
        if own:
            lib.H5Tclose(tid)
        _check(rc, f"write attribute {attr!r} of {obj!r}")

    def num_root_links(self):
        info = self._lib._GInfo()
        g = self._lib.H5Gopen2(self._f, b"/", H5P_DEFAULT)
        _check(self._lib.H5Gget_info(g, ctypes.byref(info)), "group info")
        self._lib.H5Gclose(g)
        return int(info.nlinks)

    def _links(self, group):
        """Names of the links of a group, in name order."""
        lib = self._lib
        info = lib._GInfo()
        if lib.H5Gget_info_by_name(self._f, group.encode(), ctypes.byref(info), H5P_DEFAULT) < 0:
            return []
        out = []
        for i in range(int(info.nlinks)):
            n = lib.H5Lget_name_by_idx(self._f, group.encode(), 0, 0, i, None, 0, H5P_DEFAULT)     # H5_INDEX_NAME, H5_ITER_INC
            if n < 0:
                continue
            buf = ctypes.create_string_buffer(int(n) + 1)
            lib.H5Lget_name_by_idx(self._f, group.encode(), 0, 0, i, buf, int(n) + 1, H5P_DEFAULT)
            out.append(buf.value.decode())
        return out

    def _read_int_attr(self, obj, attr):
        lib = self._lib
        if lib.H5Aexists_by_name(self._f, obj.encode(), attr.encode(), H5P_DEFAULT) <= 0:
            return None
        a = lib.H5Aopen_by_name(self._f, obj.encode(), attr.encode(), H5P_DEFAULT, H5P_DEFAULT)
        if a < 0:
            return None
        v = ctypes.c_int32(-1)
        rc = lib.H5Aread(a, _ids["H5T_NATIVE_INT32_g"], ctypes.byref(v))
        lib.H5Aclose(a)
        return int(v.value) if rc >= 0 else None

    def next_netcdf_dimid(self):
        """Dimension ids are global to a netCDF-4 file and netCDF-C expects them dense: the next one is one more than the
        largest `_Netcdf4Dimid` any dataset of the root group or of its groups carries (0 in a file without dimensions)."""
        top = -1
        for name in self._links("/"):
            paths = ["/" + name] + ["/" + name + "/" + sub for sub in self._links("/" + name)]
            for path in paths:
                v = self._read_int_attr(path, "_Netcdf4Dimid")
                if v is not None:
                    top = max(top, v)
        return top + 1

    def write_netcdf_group(self, group, coords, variables):
        """One netCDF-4 group laid out as the netCDF-4 format specifies (and `xarray.Dataset.to_netcdf(group=...)` produces).

        coords: ordered {dimension name: coordinate values} -- int arrays, a list of str (variable-length strings) or
        datetime64[D] (CF: int64 days since the first date, with `units` / `calendar`); variables: {name: (dims, array)}.
        Every dimension is an HDF5 dimension scale (CLASS / NAME by H5DSset_scale) holding its coordinate, with a
        `_Netcdf4Dimid` that continues the file's numbering (ids are global to the file and contiguous: 0, 1, 2, ...
        over all groups), attached to every axis of the variables that use it; a variable of more than one dimension also
        carries `_Netcdf4Coordinates`, the ids of its dimensions in order, as netCDF-C writes it.  The layout is checked by
        reading it back with h5py (tests/test_hostio.py); no netCDF-C / xarray reader exists in this image, so that a file
        opens in them is what the format description promises, not something that was run."""
        hl = _load_hl()
        if hl is None:
            raise OSError("libhdf5_hl (dimension scales) not found: cannot write a netCDF-4 group")
        g = "/" + group.strip("/")
        base = self.next_netcdf_dimid()
        dimid = {}
        for i, (dim, values) in enumerate(coords.items()):
            path = f"{g}/{dim}"
            v = np.asarray(values)
            if v.dtype.kind == "M":
                days = v.astype("datetime64[D]")
                origin = days[0] if len(days) else np.datetime64("1970-01-01")
                self.create_dataset(path, v.shape, np.int64)
                self.write(path, (days - origin).astype(np.int64))
                self.write_attr(path, "units", f"days since {origin} 00:00:00")
                self.write_attr(path, "calendar", "proleptic_gregorian")
            elif v.dtype.kind in "OUS":
                self.create_dataset(path, v.shape, "vlen_str")
                self.write(path, [x.decode() if isinstance(x, bytes) else str(x) for x in v.reshape(-1)])
            else:
                self.create_dataset(path, v.shape, np.int64 if v.dtype.kind in "iu" else np.float64)
                self.write(path, v)
            _check(hl.H5DSset_scale(self._open(path), dim.encode()), f"dimension scale {path!r}")
            dimid[dim] = base + i
            self.write_attr(path, "_Netcdf4Dimid", np.int32(base + i))
        for name, (dims, array) in variables.items():
            a = np.asarray(array, dtype=np.float64)
            if a.ndim != len(dims) or any(a.shape[k] != len(np.asarray(coords[dm])) for k, dm in enumerate(dims)):
                raise ValueError(f"variable {name!r}: shape {a.shape} does not match dimensions {dims}")
            path = f"{g}/{name}"
            self.create_dataset(path, a.shape, np.float64)
            self.write(path, a)
            self.write_attr(path, "_FillValue", np.array([np.nan]))
            for k, dm in enumerate(dims):
                _check(hl.H5DSattach_scale(self._open(path), self._open(f"{g}/{dm}"), k), f"attach {dm!r} to {path!r}")
            if len(dims) > 1:
                self.write_attr(path, "_Netcdf4Coordinates", np.array([dimid[dm] for dm in dims], dtype=np.int32))

    def read_str_attr(self, obj, attr):
        """String attribute `attr` of object `obj`, or None."""
        lib = self._lib
        if lib.H5Aexists_by_name(self._f, obj.encode(), attr.encode(), H5P_DEFAULT) <= 0:
            return None
        a = lib.H5Aopen_by_name(self._f, obj.encode(), attr.encode(), H5P_DEFAULT, H5P_DEFAULT)
        if a < 0:
            return None
        t = lib.H5Aget_type(a)
        out = None
        if lib.H5Tget_class(t) == H5T_STRING:
            if lib.H5Tis_variable_str(t) > 0:
                p = ctypes.c_char_p()
                if lib.H5Aread(a, t, ctypes.byref(p)) >= 0 and p.value is not None:
                    out = p.value.decode(errors="replace")
            else:
                buf = ctypes.create_string_buffer(lib.H5Tget_size(t) + 1)
                if lib.H5Aread(a, t, buf) >= 0:
                    out = buf.value.decode(errors="replace")
        lib.H5Tclose(t)
        lib.H5Aclose(a)
        return out
