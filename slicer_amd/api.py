"""Python host-side mirror of the reference's interface for the mass-assignment path.

Same names and argument meaning as the reference (data.h:29-131 structs, densitymaps.h:161-165
createDensityMaps), over the C ABI in include/slicer_amd.h.  This module only marshals arguments
and files; every particle is processed by the HIP kernels in libslicer_amd.so -- there is no CPU
compute path here, and importing fails if the library is missing.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import _lib, gadget

_L = _lib.load()

MAS_TSC, MAS_NGP = 0, 1
ACC_F32, ACC_F64, ACC_FIXED64 = 0, 1, 2
ALGO_AUTO, ALGO_DIRECT, ALGO_BINNED = 0, 1, 2
ELEM_F32, ELEM_F64, ELEM_FIXED64 = 0, 1, 2
ERR_NEGATIVE_COORD = 1
ERR_UNSUPPORTED = 6


class SlicerError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"slicer_amd error {code}: {msg}")
        self.code = code


# ---- reference structs (field names as in data.h) ----------------------------------------------
@dataclass
class InputParams:  # data.h:29-49 (only the fields the path reads are used)
    npix: int = 256
    zs: float = 0.0
    Ds: float = 0.0
    fov: float = 0.0
    hydro: bool = False
    simType: str = "Gadget"
    rgrid: float = 0.0
    filredshiftlist: str = ""
    pathsnap: str = ""
    simulation: str = ""
    seedcenter: int = 0
    seedface: int = 0
    seedsign: int = 0
    partinplanes: bool = False
    directory: str = ""
    suffix: str = ""
    snopt: int = 0
    snpix: str = ""
    physical: bool = False
    w: float = -1.0


@dataclass
class Lens:  # data.h:104-117
    nplanes: int = 0
    replication: List[int] = field(default_factory=list)
    pll: List[int] = field(default_factory=list)
    fromsnap: List[str] = field(default_factory=list)
    fromsnapi: List[int] = field(default_factory=list)
    zsimlens: List[float] = field(default_factory=list)
    ld: List[float] = field(default_factory=list)
    ld2: List[float] = field(default_factory=list)
    zfromsnap: List[float] = field(default_factory=list)
    randomize: List[bool] = field(default_factory=list)
    nrepperp: List[int] = field(default_factory=list)


@dataclass
class Random:  # data.h:126-131
    x0: List[float] = field(default_factory=list)
    y0: List[float] = field(default_factory=list)
    z0: List[float] = field(default_factory=list)
    face: List[int] = field(default_factory=list)
    sgnX: List[int] = field(default_factory=list)
    sgnY: List[int] = field(default_factory=list)
    sgnZ: List[int] = field(default_factory=list)


def sconv_int(i):
    """sconv(ff, fINT): the sub-file suffix (densitymaps.cpp:436)."""
    return str(int(i))


# ---- handle wrapper ----------------------------------------------------------------------------
class Slicer:
    """One device context (one per GPU/process)."""

    def __init__(self, device=0, max_chunk=1 << 24):
        h = C.c_void_p()
        rc = _L.slicer_create(int(device), int(max_chunk), C.byref(h))
        if rc:
            raise SlicerError(rc, (_L.slicer_last_error(None) or b"").decode())
        self._h = h
        self.npix = 0
        self.n_planes = 0

    def close(self):
        if getattr(self, "_h", None):
            _L.slicer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc:
            raise SlicerError(rc, (_L.slicer_last_error(self._h) or b"").decode())

    def set_option(self, key, value):
        """Tuning / test knob of this handle (include/slicer_amd.h: slicer_set_option); returns the previous value."""
        old = self.get_option(key)
        self._chk(_L.slicer_set_option(self._h, key.encode(), int(value)))
        return old

    def get_option(self, key):
        v = C.c_int32(0)
        self._chk(_L.slicer_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def set_stream(self, stream_ptr):
        self._chk(_L.slicer_set_stream(self._h, C.c_void_p(stream_ptr)))

    def plane_begin(self, npix, fov_rad, ld, ld2, nrepperp=None, mas=MAS_TSC, accum=ACC_F32, algo=ALGO_AUTO,
                    hydro=False, snopt=0, want_type_maps=True, fixed_frac_bits=0, debug_flags=0):
        ld = list(np.atleast_1d(ld))
        ld2 = list(np.atleast_1d(ld2))
        n = len(ld)
        d = _lib.PlaneDesc()
        d.npix, d.n_planes, d.mas, d.accum, d.algo = int(npix), n, int(mas), int(accum), int(algo)
        d.hydro, d.snopt, d.want_type_maps = int(bool(hydro)), int(snopt), int(bool(want_type_maps))
        d.fov_rad = float(fov_rad)
        d.fixed_frac_bits = int(fixed_frac_bits)
        d.debug_flags = int(debug_flags)
        nrepperp = [0] * n if nrepperp is None else list(np.atleast_1d(nrepperp))
        for i in range(min(n, _lib.MAX_PLANES)):
            d.ld[i], d.ld2[i], d.nrepperp[i] = float(ld[i]), float(ld2[i]), int(nrepperp[i])
        self._chk(_L.slicer_plane_begin(self._h, C.byref(d)))
        self.npix, self.n_planes = int(npix), n

    def file_begin(self, npart, massarr, boxsize, sgn, face, center, rcase):
        f = _lib.FileDesc()
        for t in range(6):
            f.npart[t] = int(npart[t])
            f.massarr[t] = float(massarr[t])
        f.boxsize = float(boxsize)
        for a in range(3):
            f.sgn[a] = int(sgn[a])
            f.center[a] = float(center[a])
        f.face = int(face)
        f.rcase = float(np.float32(rcase))
        self._chk(_L.slicer_file_begin(self._h, C.byref(f)))

    def deposit_host(self, ptype, pos, mass=None):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        mp = None
        if mass is not None:
            mass = np.ascontiguousarray(mass, dtype=np.float32)
            assert mass.size == pos.shape[0]
            mp = mass.ctypes.data
        self._chk(_L.slicer_deposit_host(self._h, int(ptype), pos.ctypes.data, mp, pos.shape[0]))

    def deposit_device(self, ptype, d_pos, n, d_mass=None):
        self._chk(_L.slicer_deposit_device(self._h, int(ptype), C.c_void_p(d_pos),
                                           C.c_void_p(d_mass) if d_mass else None, int(n)))

    def file_end(self):
        self._chk(_L.slicer_file_end(self._h))

    def plane_finalize(self):
        self._chk(_L.slicer_plane_finalize(self._h))

    def synchronize(self):
        self._chk(_L.slicer_synchronize(self._h))

    def algo_mask(self):
        """Bit (1 << ALGO_*) of every deposit algorithm that ran since plane_begin (bit 3: shot-noise thinning)."""
        m = C.c_int32()
        self._chk(_L.slicer_plane_algo_mask(self._h, C.byref(m)))
        return m.value

    def plane_status(self):
        self._chk(_L.slicer_plane_status(self._h))

    def plane_flush(self):
        self._chk(_L.slicer_plane_flush(self._h))

    # --- cross-rank sum in the accumulator type (include/slicer_amd.h "cross-rank sum") ---
    def reduce_meta_get(self):
        m = _lib.ReduceMeta()
        self._chk(_L.slicer_reduce_meta_get(self._h, C.byref(m)))
        return [int(x) for x in m.v]

    def reduce_meta_get_async(self):
        """reduce_meta_get without the host synchronisation: the guard entry stays 0; combine the device flag of
        plane_device_guard() across ranks instead (parallel.reduce_planes does)."""
        m = _lib.ReduceMeta()
        self._chk(_L.slicer_reduce_meta_get_async(self._h, C.byref(m)))
        return [int(x) for x in m.v]

    def plane_device_guard(self):
        p = C.c_void_p()
        self._chk(_L.slicer_plane_device_guard(self._h, C.byref(p)))
        return p.value

    def rand_stream_set(self, v31):
        """Shot-noise deviates from a stream of this handle's own (31 words, oldest first; None: back to the process-global
        libc stream) -- include/slicer_amd.h: slicer_rand_stream_set."""
        if v31 is None:
            self._chk(_L.slicer_rand_stream_set(self._h, None))
        else:
            self._chk(_L.slicer_rand_stream_set(self._h, (C.c_uint32 * 31)(*[int(x) for x in v31])))

    def rand_stream_get(self):
        v = (C.c_uint32 * 31)()
        self._chk(_L.slicer_rand_stream_get(self._h, v))
        return list(v)

    def get_stream(self):
        p = C.c_void_p()
        self._chk(_L.slicer_get_stream(self._h, C.byref(p)))
        return p.value or 0

    def reduce_meta_set(self, ints):
        m = _lib.ReduceMeta()
        for i, x in enumerate(ints):
            m.v[i] = int(x)
        self._chk(_L.slicer_reduce_meta_set(self._h, C.byref(m)))

    def plane_accumulators(self, plane=0):
        """(device pointers of the 7 accumulator slots or None, ELEM_* kind); slot 6 = shared / all-types."""
        acc = (C.c_void_p * 7)()
        elem = C.c_int32()
        self._chk(_L.slicer_plane_accumulators(self._h, int(plane), acc, C.byref(elem)))
        return [acc[s] for s in range(7)], elem.value

    def plane_device_counts(self, plane=0):
        p = C.c_void_p()
        self._chk(_L.slicer_plane_device_counts(self._h, int(plane), C.byref(p)))
        return p.value

    def plane_device_maps(self, plane=0):
        tot = C.c_void_p()
        toti = (C.c_void_p * 6)()
        self._chk(_L.slicer_plane_device_maps(self._h, int(plane), C.byref(tot), toti))
        return tot.value, [toti[t] for t in range(6)]

    def plane_read(self, plane=0, want_types=True):
        n = self.npix
        tot = np.empty((n, n), np.float32)
        toti = np.empty((6, n, n), np.float32) if want_types else None
        nsel = np.zeros(6, np.int64)
        self._chk(_L.slicer_plane_read(self._h, int(plane), tot.ctypes.data,
                                       toti.ctypes.data if want_types else None, nsel.ctypes.data))
        return tot, toti, nsel

    # --- device utilities ---
    def malloc(self, nbytes):
        p = C.c_void_p()
        self._chk(_L.slicer_device_malloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def free(self, ptr):
        self._chk(_L.slicer_device_free(self._h, C.c_void_p(ptr)))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(max(arr.nbytes, 4))
        self._chk(_L.slicer_copy_to_device(self._h, C.c_void_p(p), arr.ctypes.data, arr.nbytes))
        return p

    def to_host(self, ptr, shape, dtype):
        out = np.empty(shape, dtype)
        self._chk(_L.slicer_copy_to_host(self._h, out.ctypes.data, C.c_void_p(ptr), out.nbytes))
        return out

    def synth_positions(self, d_pos, first, count, boxsize=1000.0, seed=0x51CE2, clustered=False):
        self._chk(_L.slicer_synth_positions(self._h, C.c_void_p(d_pos), int(first), int(count), float(boxsize),
                                            int(seed), int(bool(clustered))))

    def debug_project(self, ptype, d_pos, n, capacity):
        d_xs = self.malloc(4 * capacity)
        d_ys = self.malloc(4 * capacity)
        d_pl = self.malloc(4 * capacity)
        d_src = self.malloc(8 * capacity)
        cnt = C.c_uint64()
        try:
            self._chk(_L.slicer_debug_project(self._h, int(ptype), C.c_void_p(d_pos), int(n), C.c_void_p(d_xs),
                                              C.c_void_p(d_ys), C.c_void_p(d_pl), C.c_void_p(d_src), int(capacity),
                                              C.byref(cnt)))
            k = min(cnt.value, capacity)
            xs = self.to_host(d_xs, k, np.float32)
            ys = self.to_host(d_ys, k, np.float32)
            pl = self.to_host(d_pl, k, np.int32)
            src = self.to_host(d_src, k, np.uint64)
        finally:
            for p in (d_xs, d_ys, d_pl, d_src):
                self.free(p)
        return cnt.value, xs, ys, pl, src

    def debug_box_quotient(self, box):
        """(number of mismatches, up to 8 offending r) of the exhaustive f32-quotient sweep for this box size."""
        n = C.c_uint32()
        ex = (C.c_uint32 * 8)()
        self._chk(_L.slicer_debug_box_quotient(self._h, float(box), C.byref(n), ex))
        return n.value, np.array(list(ex), np.uint32).view(np.float32)[:min(n.value, 8)]

    def debug_dl_quotient(self, npix):
        """Exhaustive device check of the division-free grid quotient for a map size; -> (mismatches, example f32s)."""
        n = C.c_uint32(0)
        ex = (C.c_uint32 * 8)()
        self._chk(_L.slicer_debug_dl_quotient(self._h, int(npix), C.byref(n), ex))
        return n.value, np.array(list(ex), np.uint32).view(np.float32)[:min(n.value, 8)]

    def debug_math(self, op, a, b=None):
        """Device sqrt (op 0), quotient (1), small-angle asin (2) / atan (3) of float64 arrays; see slicer_amd.h."""
        a = np.ascontiguousarray(a, np.float64)
        n = a.size
        d_a = self.to_device(a)
        d_b = self.to_device(np.ascontiguousarray(b, np.float64)) if b is not None else None
        d_o = self.malloc(8 * max(n, 1))
        try:
            self._chk(_L.slicer_debug_math(self._h, int(op), C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_o), n))
            return self.to_host(d_o, n, np.float64)
        finally:
            for p in (d_a, d_b, d_o):
                if p is not None:
                    self.free(p)

    def profile_enable(self, on=True):
        self._chk(_L.slicer_profile_enable(self._h, int(bool(on))))

    def profile_reset(self):
        self._chk(_L.slicer_profile_reset(self._h))

    def profile_get(self):
        arr = (_lib.KernelTime * 16)()
        n = C.c_int()
        self._chk(_L.slicer_profile_get(self._h, arr, 16, C.byref(n)))
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n.value, 16))}


# ---- the reference's entry point ---------------------------------------------------------------
def createDensityMaps(p: InputParams, lens: Lens, random: Random, isnap: int, ffmin: int, ffmax: int, File: str,
                      fovradiants: float, rcase: float, myid: int = 1, *, slicer: Slicer = None, do_ngp: bool = False,
                      accum: int = ACC_F32, algo: int = ALGO_AUTO, true_counts: bool = False):
    """createDensityMaps (densitymaps.h:161-165 / densitymaps.cpp:419-524) on the GPU.

    Returns (status, mapxytot[npix,npix], mapxytoti[6,npix,npix], ntotxyi[6]).  status is 0, or 1 where
    the reference returns 1 (unreadable file, negative transformed coordinate).  The four unused GSL
    arguments of the reference signature are dropped.  ntotxyi is all zeros like the reference's
    out-parameter (shadowed at densitymaps.cpp:497) unless true_counts=True.
    """
    own = slicer is None
    s = slicer or Slicer()
    try:
        s.plane_begin(p.npix, fovradiants, [lens.ld[isnap]], [lens.ld2[isnap]], [lens.nrepperp[isnap]],
                      mas=MAS_NGP if do_ngp else MAS_TSC, accum=accum, algo=algo, hydro=p.hydro, snopt=p.snopt,
                      want_type_maps=True)
        for ff in range(ffmin, ffmax):
            file_in = File + "." + sconv_int(ff)
            try:
                path, hdr = gadget.open_snapshot(file_in)
            except FileNotFoundError:
                return 1, None, None, None
            pos = gadget.read_positions(path)
            masses = gadget.read_masses(path, hdr) if p.hydro else {}
            s.file_begin(hdr["npart"], hdr["massarr"], hdr["boxsize"],
                         (random.sgnX[isnap], random.sgnY[isnap], random.sgnZ[isnap]), random.face[isnap],
                         (random.x0[isnap], random.y0[isnap], random.z0[isnap]), rcase)
            off = 0
            for t in range(6):
                n = int(hdr["npart"][t])
                if n > 0:
                    s.deposit_host(t, pos[off:off + n], masses.get(t) if p.hydro else None)
                off += n
            s.file_end()
        try:
            tot, toti, nsel = s.plane_read(0, want_types=True)
        except SlicerError as e:
            if e.code == ERR_NEGATIVE_COORD:
                return 1, None, None, None
            raise
        if not true_counts:
            nsel = np.zeros(6, np.int64)
        return 0, tot, toti, nsel
    finally:
        if own:
            s.close()
