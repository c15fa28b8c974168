#!/opt/conda/bin/python3.9
"""Generate tests/golden/inferencedata_xarray_layout.nc: a small input file with the ON-DISK structure that
`assemble_data` produces (covid19uk/data/assemble.py:15-16: two xarray `Dataset.to_netcdf(group=...)` calls on
the variables of covid19uk/model_spec.py:88-105), so that the build's reader (covid19uk_amd/inference/
inference.py:read_inference_data, over ctypes->libhdf5) is exercised on a file it did not write itself.

xarray / netCDF4 are not installed anywhere in this image, so the netCDF-4 conventions are laid down with
h5py (present under /opt/conda): groups `constant_data` / `observations`; one HDF5 dimension scale per
dimension (CLASS=DIMENSION_SCALE, NAME, _Netcdf4Dimid) attached to every variable (DIMENSION_LIST); string
coordinates as variable-length strings; datetime coordinates as int64 with `units = "days since ..."` and
`calendar` attributes (xarray's CF encoding); float variables with a NaN `_FillValue`; the boolean weekday
stored as float64 (`.astype(DTYPE)`, model_spec.py:95); C on dims (location_dest, location_src) and W on
`date` (covid19uk/data/loaders.py:39-41,67-74); the case counts as int64 [location, time], chunked,
shuffled and deflated.  The expected arrays go to inferencedata_expected.npz.

    /opt/conda/bin/python3.9 tests/golden/make_netcdf_fixture.py
"""
import os

import h5py
import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
M, T = 4, 10
rng = np.random.default_rng(20211004)
codes = np.array(["N09000001", "N09000002", "N09000003", "N09000004"], dtype=object)
names = np.array(["Antrim", "Ards", "Armagh", "Belfast"], dtype=object)
C = np.floor(rng.uniform(0, 300, size=(M, M)))
np.fill_diagonal(C, 0.0)
N = np.floor(rng.uniform(1e4, 2e5, size=M))
W = rng.uniform(0.6, 1.2, size=T)
A = np.array([[0, 1, 0, 1], [1, 0, 1, 0], [0, 1, 0, 1], [1, 0, 1, 0]], dtype=np.float64)
days = np.arange(T, dtype=np.int64)
weekday = (((days + 6) % 7) < 5).astype(np.float64)          # 2020-03-01 was a Sunday
area = rng.uniform(1e8, 9e8, size=M)
cases = rng.poisson(3.0, size=(M, T)).astype(np.int64)
vstr = h5py.string_dtype(encoding="utf-8")


def scale(g, name, data, dimid, dtype=None):
    d = g.create_dataset(name, data=data, dtype=dtype)
    d.make_scale(name)
    d.attrs["_Netcdf4Dimid"] = np.int32(dimid)
    return d


def time_scale(g, name, dimid):
    d = scale(g, name, days, dimid)
    d.attrs["units"] = np.bytes_("days since 2020-03-01 00:00:00")            # fixed-length (NC_CHAR)
    d.attrs.create("calendar", "proleptic_gregorian", dtype=vstr)              # variable-length (NC_STRING)
    return d


def var(g, name, data, dims, **kw):
    d = g.create_dataset(name, data=data, **kw)
    for i, s in enumerate(dims):
        d.dims[i].attach_scale(s)
    if data.dtype.kind == "f":
        d.attrs["_FillValue"] = np.array([np.nan])
    return d


path = os.path.join(OUT, "inferencedata_xarray_layout.nc")
with h5py.File(path, "w") as f:
    f.attrs["_NCProperties"] = np.bytes_("version=2,netcdf=4.7.4,hdf5=1.10.6")
    g = f.create_group("constant_data")
    dest = scale(g, "location_dest", codes, 0, vstr)
    src = scale(g, "location_src", codes, 1, vstr)
    date = time_scale(g, "date", 2)
    loc = scale(g, "location", codes, 3, vstr)
    time = time_scale(g, "time", 4)
    var(g, "C", C, [dest, src], chunks=(2, M), compression="gzip", compression_opts=4, shuffle=True)
    var(g, "W", W, [date])
    var(g, "N", N, [loc])
    var(g, "adjacency", A, [loc, loc])
    var(g, "weekday", weekday, [time])
    var(g, "area", area, [loc])
    var(g, "locations", names, [loc], dtype=vstr)
    o = f.create_group("observations")
    loc2 = scale(o, "location", codes, 0, vstr)
    time2 = time_scale(o, "time", 1)
    var(o, "cases", cases, [loc2, time2], chunks=(M, 5), compression="gzip", compression_opts=4, shuffle=True)
np.savez(os.path.join(OUT, "inferencedata_expected.npz"), C=C, N=N, W=W, adjacency=A, weekday=weekday, area=area,
         cases=cases.astype(np.float64),
         time=np.array([str(np.datetime64("2020-03-01") + np.timedelta64(int(d), "D")) for d in days]))
print("wrote", path, os.path.getsize(path), "bytes")
