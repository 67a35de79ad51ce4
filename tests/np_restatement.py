"""Second, independent restatement of the path's arithmetic in vectorised numpy.

TEST INFRASTRUCTURE ONLY.  Written from SURVEY.md Appendix A (A1-A5) without
looking at oracle/slicer_oracle.c's code paths, so that a transcription slip in
either one shows up as a bit mismatch (tests/test_oracle_cross.py).  numpy
float32/float64 scalar ops are IEEE round-to-nearest and never fused.

Reference lines: gadget2io.cpp:195-274 (A1), densitymaps.cpp:346-401 (A2/A3),
utilities.cpp:4-16,36-97 (A4), densitymaps.cpp:511-513 (A5).
"""
import numpy as np

F32 = np.float32
F64 = np.float64


def _wrap(v):
    v = v.astype(F32)
    hi = v.astype(F64) > 1.0
    v = np.where(hi, (v.astype(F64) - 1.0).astype(F32), v)
    lo = v.astype(F64) < 0.0
    v = np.where(lo, (1.0 + v.astype(F64)).astype(F32), v)
    return v.astype(F32)


_FACE = {1: (0, 1, 2), 2: (0, 2, 1), 3: (1, 2, 0), 4: (1, 0, 2), 5: (2, 0, 1), 6: (2, 1, 0)}


def transform(raw, boxsize, sgn, face, center, rcase):
    raw = np.asarray(raw, F32).reshape(-1, 3)
    b = []
    for a in range(3):
        q = raw[:, a].astype(F64) / F64(boxsize)
        b.append(_wrap((F64(int(sgn[a])) * q).astype(F32)))
    perm = _FACE[int(face)]
    out = []
    for a in range(3):
        v = b[perm[a]]
        v = (v.astype(F64) - F64(center[a])).astype(F32)
        out.append(_wrap(v))
    x, y, z = out
    z = (z + F32(rcase)).astype(F32)
    return x, y, z


def select_project(x, y, z, m, ld, ld2, boxsize, nrep, fov, npix):
    """m: f32 array per particle (already MAX_M-capped) -> xs, ys, ms, idx (reference order)."""
    minD = F64(ld) / F64(boxsize) * 1.e+3 / 1.0
    maxD = F64(ld2) / F64(boxsize) * 1.e+3 / 1.0
    zz = z.astype(F64)
    inslab = np.nonzero((zz >= minD) & (zz < maxD))[0]
    lim = F64(fov) * (1. + 2. / int(npix)) * 0.5
    res = []
    reps = [(ni, nj) for ni in range(-nrep, nrep + 1) for nj in range(-nrep, nrep + 1)]
    for r, (ni, nj) in enumerate(reps):
        X = (x[inslab] + F32(ni)).astype(F32).astype(F64) - 0.5
        Y = (y[inslab] + F32(nj)).astype(F32).astype(F64) - 0.5
        Z = zz[inslab]
        with np.errstate(invalid="ignore", divide="ignore"):
            d = np.sqrt(X * X + Y * Y + Z * Z)
            dec = np.arcsin(X / d)
            ra = np.arctan2(Y, Z)
            ok = (np.abs(ra) <= lim) & (np.abs(dec) <= lim)
            xs = (dec / F64(fov) + 0.5).astype(F32)
            ys = (ra / F64(fov) + 0.5).astype(F32)
        res.append((inslab[ok], np.full(ok.sum(), r), xs[ok], ys[ok]))
    idx = np.concatenate([a[0] for a in res])
    rr = np.concatenate([a[1] for a in res])
    xs = np.concatenate([a[2] for a in res])
    ys = np.concatenate([a[3] for a in res])
    order = np.lexsort((rr, idx))  # particle-major, then (ni,nj) order
    return xs[order], ys[order], m[idx[order]].astype(F32), idx[order]


def weight(ixx, ixh, dx):
    ixx = np.asarray(ixx, F32)
    ixh = np.asarray(ixh, F32)
    A = np.abs((ixx - ixh).astype(F32)).astype(F32)
    u = (A.astype(F64) / F64(dx)).astype(F32)
    w1 = (0.75 - (u * u).astype(F32).astype(F64)).astype(F32)
    t = 1.5 - u.astype(F64)
    w2 = (0.5 * (t * t)).astype(F32)
    Ad = A.astype(F64)
    return np.where(Ad <= 0.5 * F64(dx), w1, np.where(Ad <= 0.5 * 3.0 * F64(dx), w2, F32(0))).astype(F32)


def tsc_contributions(xs, ys, ws, nn):
    """Per-particle 9 (pixel, value) pairs in reference j-order; pixel = -1 if clipped."""
    dl = 1. / F64(nn)
    gx = np.floor(xs.astype(F64) / dl).astype(np.int64)
    gy = np.floor(ys.astype(F64) / dl).astype(np.int64)
    sw = np.sqrt(ws.astype(F32)).astype(F32)
    pix = np.empty((len(xs), 9), np.int64)
    val = np.empty((len(xs), 9), F32)
    for j in range(9):
        px = gx + (j % 3) - 1
        py = gy + (j // 3) - 1
        cx = ((px.astype(F64) + 0.5) * dl).astype(F32)
        cy = ((py.astype(F64) + 0.5) * dl).astype(F32)
        wfx = (sw * weight(xs, cx, dl)).astype(F32)
        wfy = (sw * weight(ys, cy, dl)).astype(F32)
        ok = (px >= 0) & (px < nn) & (py >= 0) & (py < nn)
        pix[:, j] = np.where(ok, px + nn * py, -1)
        val[:, j] = (wfx * wfy).astype(F32)
    return pix, val


def gridist_w(xs, ys, ws, nn, do_ngp):
    """Sequential f32 accumulation in particle order (slow python loop over entries; small cases only)."""
    grid = np.zeros(nn * nn, F32)
    dl = 1. / F64(nn)
    if do_ngp:
        gx = np.floor(xs.astype(F64) / dl).astype(np.int64)
        gy = np.floor(ys.astype(F64) / dl).astype(np.int64)
        for i in range(len(xs)):
            if 0 <= gx[i] < nn and 0 <= gy[i] < nn:
                p = gx[i] + nn * gy[i]
                grid[p] = F32(grid[p] + F32(ws[i]))
        return grid.reshape(nn, nn)
    pix, val = tsc_contributions(xs, ys, ws, nn)
    for i in range(len(xs)):
        for j in range(9):
            p = pix[i, j]
            if p >= 0:
                grid[p] = F32(grid[p] + val[i, j])
    return grid.reshape(nn, nn)


def tsc_exact_f64(xs, ys, ws, nn):
    """Order-free float64 sum of the bit-exact f32 contributions (for tolerance studies)."""
    pix, val = tsc_contributions(xs, ys, ws, nn)
    ok = pix >= 0
    g = np.zeros(nn * nn, F64)
    np.add.at(g, pix[ok], val[ok].astype(F64))
    k = np.zeros(nn * nn, np.int64)
    np.add.at(k, pix[ok], 1)
    return g.reshape(nn, nn), k.reshape(nn, nn)
