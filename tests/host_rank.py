"""Host stand-in of one rank's plane pass for the gloo rehearsals of the N>1 path (TEST INFRASTRUCTURE).

It holds what a slicer handle holds after the last file_end() of a pass -- per-slot accumulators in the pass's
accumulator type -- computed with the oracle (selection, projection) and the numpy restatement (bit-exact TSC
contributions), and answers the same five calls slicer_amd.parallel.reduce_planes makes on a real handle
(include/slicer_amd.h, "cross-rank sum in the accumulator type").  Slot s = particle type 0..5, slot 6 = the shared
all-types accumulator.  Accumulator semantics restated from slicer_capi.cpp / slicer_binned.hip:
  f32     per-slot f32 sums (order here: particle order; on the device: atomics)
  f64     per-slot f64 sums of the f32 contributions
  fixed64 per-slot integer sums of rint(c * 2^k), k = 40 - (ilogb(m) + 1): order-independent => bitwise reproducible
  ngp     per-file sequential f32 sums folded file by file (densitymaps.cpp:511-513): slot 6 = tot, slots t = toti[t]
"""
import math

import numpy as np

import np_restatement as npr
import oracle

I32_MIN = -(2 ** 31)
ELEM = {"f32": 0, "f64": 1, "fixed64": 2, "ngp": 0}


def fixed_exp(m, frac=40):
    return frac - (math.frexp(m)[1])  # ilogb(m) + 1 == frexp exponent for m > 0


class HostRankPass:
    def __init__(self, files, npix, fov, ld, ld2, rnd, mode="fixed64", want_type_maps=True, box=1000.0):
        self.npix, self.n_planes, self.mode, self.want_type_maps = npix, 1, mode, want_type_maps
        self.finalized = False
        n2 = npix * npix
        dt = {"f32": np.float32, "f64": np.float64, "fixed64": np.int64, "ngp": np.float32}[mode]
        self.acc = [None] * 7
        self.exp = [I32_MIN] * 7
        self.counts = np.zeros(6, np.int64)
        self.neg = 0
        if mode == "ngp":
            self.acc[6] = np.zeros(n2, np.float32)
        for f in files:
            off = 0
            file_maps = [None] * 6
            for t in range(6):
                n = int(f["npart"][t])
                if not n:
                    continue
                raw = np.asarray(f["pos"], np.float32).reshape(-1, 3)[off:off + n]
                off += n
                x, y, z = oracle.transform(raw, box, rnd["sgn"], rnd["face"], rnd["center"], rnd["rcase"])
                self.neg |= int(oracle.min_guard(x, y, z))
                m = float(f["massarr"][t])
                xs, ys, ms = oracle.select_project(x, y, z, None, m, ld, ld2, box, 0, fov, npix)
                self.counts[t] += len(xs)
                slot = t if (want_type_maps or mode == "ngp") else 6
                if mode == "ngp":
                    file_maps[t] = oracle.gridist_w(xs, ys, ms, npix, True).reshape(-1)
                    continue
                if self.acc[slot] is None:
                    self.acc[slot] = np.zeros(n2, dt)
                    if mode == "fixed64":
                        mm = m if want_type_maps else max(float(v) for v in f["massarr"])
                        self.exp[slot] = fixed_exp(mm)
                pix, val = npr.tsc_contributions(xs, ys, ms, npix)
                ok = pix >= 0
                if mode == "fixed64":
                    q = np.rint(val[ok].astype(np.float64) * 2.0 ** self.exp[slot]).astype(np.int64)
                    np.add.at(self.acc[slot], pix[ok], q)
                else:
                    np.add.at(self.acc[slot], pix[ok], val[ok].astype(dt))
            if mode == "ngp" and any(v is not None for v in file_maps):
                s = np.zeros(n2, np.float32)
                for t in range(6):  # ((((m0+m1)+m2)+m3)+m4)+m5, absent types contribute their zero map
                    v = file_maps[t] if file_maps[t] is not None else np.zeros(n2, np.float32)
                    s = v.copy() if t == 0 else (s + v).astype(np.float32)
                    if file_maps[t] is not None and want_type_maps:
                        if self.acc[t] is None:
                            self.acc[t] = np.zeros(n2, np.float32)
                        self.acc[t] = (self.acc[t] + v).astype(np.float32)
                self.acc[6] = (self.acc[6] + s).astype(np.float32)
        self.neg_remote = False

    # ---- the calls parallel.reduce_planes makes ----
    def reduce_meta_get(self):
        v = [0] * 24
        for s in range(7):
            live = self.acc[s] is not None
            v[s] = int(live)
            fx = live and self.mode == "fixed64"
            v[7 + s] = self.exp[s] if fx else I32_MIN
            v[14 + s] = -self.exp[s] if fx else I32_MIN
        v[21] = self.neg
        return v

    def reduce_meta_set(self, v):
        dt = {"f32": np.float32, "f64": np.float64, "fixed64": np.int64, "ngp": np.float32}[self.mode]
        for s in range(7):
            if not v[s]:
                continue
            if self.mode == "fixed64":
                assert v[7 + s] == -v[14 + s], f"ranks scaled accumulator {s} differently"
                assert self.acc[s] is None or self.exp[s] == v[7 + s]
                self.exp[s] = v[7 + s]
            if self.acc[s] is None:
                self.acc[s] = np.zeros(self.npix * self.npix, dt)
        self.neg_remote = bool(v[21])

    def plane_accumulators(self, plane=0):
        return list(self.acc), ELEM[self.mode]

    def plane_device_counts(self, plane=0):
        return self.counts

    def plane_finalize(self):
        self.finalized = True

    # ---- f32 maps as slicer_plane_finalize would produce them ----
    def maps(self):
        n = self.npix
        toti = np.zeros((6, n * n), np.float32)
        if self.mode == "ngp":
            for t in range(6):
                if self.acc[t] is not None:
                    toti[t] = self.acc[t]
            return self.acc[6].reshape(n, n), toti.reshape(6, n, n)
        if self.acc[6] is not None:
            a = self.acc[6]
            tot = (a.astype(np.float64) * 2.0 ** -self.exp[6]).astype(np.float32) if self.mode == "fixed64" \
                else a.astype(np.float32)
            return tot.reshape(n, n), toti.reshape(6, n, n)
        if self.mode == "f32":
            tot, first = np.zeros(n * n, np.float32), True
            for t in range(6):
                v = self.acc[t] if self.acc[t] is not None else np.zeros(n * n, np.float32)
                toti[t] = v
                tot = v.copy() if first else (tot + v).astype(np.float32)
                first = False
            return tot.reshape(n, n), toti.reshape(6, n, n)
        s = np.zeros(n * n, np.float64)
        for t in range(6):
            if self.acc[t] is None:
                continue
            v = self.acc[t].astype(np.float64) * (2.0 ** -self.exp[t] if self.mode == "fixed64" else 1.0)
            toti[t] = v.astype(np.float32)
            s += v
        return s.astype(np.float32).reshape(n, n), toti.reshape(6, n, n)
