"""ctypes binding of oracle/libslicer_oracle.so (TEST INFRASTRUCTURE ONLY).

Parity pin status: "parity unpinned" by a reference build (see slicer_oracle.c);
pinned by SURVEY.md Appendix-B known answers + an independent numpy restatement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

__all__ = ["build", "lib", "weight", "transform", "min_guard", "select_project",
           "gridist_w", "OrcFile", "create_density_maps", "file_range", "reduce_sum"]


def build(force=False):
    so = os.path.join(_HERE, "libslicer_oracle.so")
    src = os.path.join(_HERE, "slicer_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


class OrcFile(C.Structure):
    _fields_ = [("npart", C.c_int32 * 6), ("massarr", C.c_double * 6), ("boxsize", C.c_double),
                ("pos", C.c_void_p), ("mass", C.c_void_p * 6)]


_fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_weight.restype = C.c_float
        L.orc_weight.argtypes = [C.c_float, C.c_float, C.c_double]
        L.orc_transform.restype = None
        L.orc_transform.argtypes = [_fp, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_double, C.c_double, C.c_double, C.c_float, _fp, _fp, _fp]
        L.orc_min_guard.restype = C.c_int
        L.orc_min_guard.argtypes = [_fp, _fp, _fp, C.c_int64]
        L.orc_select_project.restype = C.c_int64
        L.orc_select_project.argtypes = [_fp, _fp, _fp, C.c_void_p, C.c_float, C.c_int64, C.c_double, C.c_double,
                                         C.c_double, C.c_int, C.c_double, C.c_int, _fp, _fp, _fp, C.c_void_p]
        L.orc_gridist_w.restype = None
        L.orc_gridist_w.argtypes = [_fp, _fp, _fp, C.c_int64, C.c_int, C.c_int, _fp]
        L.orc_create_density_maps.restype = C.c_int
        L.orc_create_density_maps.argtypes = [C.POINTER(OrcFile), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_double, C.c_double, C.c_int, C.c_double,
                                              C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_double, C.c_double, C.c_double, C.c_float,
                                              _fp, _fp, np.ctypeslib.ndpointer(dtype=np.int64)]
        L.orc_create_density_maps_sn.restype = C.c_int
        L.orc_create_density_maps_sn.argtypes = [C.POINTER(OrcFile), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_double, C.c_double, C.c_int, C.c_double,
                                                 C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_double, C.c_double, C.c_double, C.c_float,
                                                 _fp, _fp, np.ctypeslib.ndpointer(dtype=np.int64)]
        L.orc_file_range.restype = None
        L.orc_file_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        L.orc_reduce_sum.restype = None
        L.orc_reduce_sum.argtypes = [_fp, _fp, C.c_size_t]
        _LIB = L
    return _LIB


def weight(ixx, ixh, dx):
    return float(lib().orc_weight(np.float32(ixx), np.float32(ixh), float(dx)))


def transform(raw, boxsize, sgn, face, center, rcase):
    """A1 (gadget2io.cpp:195-274). raw: [n,3] f32 -> x,y,z f32 arrays."""
    raw = np.ascontiguousarray(raw, dtype=np.float32).reshape(-1, 3)
    n = raw.shape[0]
    x = np.empty(n, np.float32)
    y = np.empty(n, np.float32)
    z = np.empty(n, np.float32)
    lib().orc_transform(raw.reshape(-1), n, float(boxsize), int(sgn[0]), int(sgn[1]), int(sgn[2]), int(face),
                        float(center[0]), float(center[1]), float(center[2]), np.float32(rcase), x, y, z)
    return x, y, z


def min_guard(x, y, z):
    return int(lib().orc_min_guard(x, y, z, len(x)))


def select_project(x, y, z, mass, mconst, ld, ld2, boxsize, nrep, fov, npix, want_index=False):
    """A2+A3 (densitymaps.cpp:346-401). Returns xs, ys, ms[, sel_index]."""
    n = len(x)
    rep = (2 * nrep + 1) ** 2
    xs = np.empty(n * rep, np.float32)
    ys = np.empty(n * rep, np.float32)
    ms = np.empty(n * rep, np.float32)
    idx = np.empty(n * rep, np.int64) if want_index else None
    mp = None
    if mass is not None:
        mass = np.ascontiguousarray(mass, dtype=np.float32)
        mp = mass.ctypes.data
    k = lib().orc_select_project(x, y, z, mp, np.float32(mconst), n, float(ld), float(ld2), float(boxsize),
                                 int(nrep), float(fov), int(npix), xs, ys, ms,
                                 idx.ctypes.data if want_index else None)
    if want_index:
        return xs[:k].copy(), ys[:k].copy(), ms[:k].copy(), idx[:k].copy()
    return xs[:k].copy(), ys[:k].copy(), ms[:k].copy()


def gridist_w(xs, ys, ws, nn, do_ngp):
    """A4 (utilities.cpp:36-97). Returns map [nn,nn] (row = slow axis = y)."""
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    ws = np.ascontiguousarray(ws, np.float32)
    assert len(xs) == len(ys) == len(ws)
    m = np.empty(nn * nn, np.float32)
    lib().orc_gridist_w(xs, ys, ws, len(xs), int(nn), int(bool(do_ngp)), m)
    return m.reshape(nn, nn)


def create_density_maps(files, ffmin, ffmax, npix, hydro, do_ngp, ld, ld2, nrepperp, fov,
                        sgn, face, center, rcase, snopt=0):
    """A5 (densitymaps.cpp:419-524). files: list of dicts {npart[6], massarr[6], boxsize, pos[n,3], mass{t:arr}}.
    Returns rc, tot[npix,npix], toti[6,npix,npix], nsel[6]."""
    arr = (OrcFile * len(files))()
    keep = []
    for i, f in enumerate(files):
        pos = np.ascontiguousarray(f["pos"], dtype=np.float32).reshape(-1)
        keep.append(pos)
        assert pos.size == 3 * int(sum(f["npart"]))
        for t in range(6):
            arr[i].npart[t] = int(f["npart"][t])
            arr[i].massarr[t] = float(f["massarr"][t])
            m = f.get("mass", {}).get(t)
            if m is not None:
                m = np.ascontiguousarray(m, dtype=np.float32)
                assert m.size == int(f["npart"][t])
                keep.append(m)
                arr[i].mass[t] = m.ctypes.data
            else:
                arr[i].mass[t] = None
        arr[i].boxsize = float(f["boxsize"])
        arr[i].pos = pos.ctypes.data
    tot = np.empty(npix * npix, np.float32)
    toti = np.empty(6 * npix * npix, np.float32)
    nsel = np.zeros(6, np.int64)
    rc = lib().orc_create_density_maps_sn(arr, int(ffmin), int(ffmax), int(npix), int(bool(hydro)), int(bool(do_ngp)),
                                          int(snopt), float(ld), float(ld2), int(nrepperp), float(fov),
                                       int(sgn[0]), int(sgn[1]), int(sgn[2]), int(face),
                                       float(center[0]), float(center[1]), float(center[2]), np.float32(rcase),
                                       tot, toti, nsel)
    return rc, tot.reshape(npix, npix), toti.reshape(6, npix, npix), nsel


def file_range(numfiles, numprocs, myid):
    a = C.c_uint()
    b = C.c_uint()
    lib().orc_file_range(numfiles, numprocs, myid, C.byref(a), C.byref(b))
    return a.value, b.value


def reduce_sum(dst, src):
    lib().orc_reduce_sum(dst.reshape(-1), np.ascontiguousarray(src, np.float32).reshape(-1), dst.size)
    return dst
