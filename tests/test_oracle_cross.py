"""oracle/slicer_oracle.c vs the independent numpy restatement: bit-for-bit."""
import itertools

import numpy as np
import pytest

import np_restatement as npr
import oracle
from slicer_amd import synth

BOX = 1000.0


def edge_positions():
    """Raw coordinates that exercise the wrap edges: 0, -0.0, box, tiny, just-inside values."""
    vals = np.array([0.0, -0.0, BOX, np.nextafter(np.float32(BOX), np.float32(0)), 1e-30, 1e-3, 0.5 * BOX,
                     np.nextafter(np.float32(0.5 * BOX), np.float32(BOX)), 0.3 * BOX, 0.6 * BOX, 0.1 * BOX,
                     0.999999 * BOX, 1.0000001 * BOX, -1e-4], np.float32)
    return np.array(list(itertools.product(vals, repeat=3)), np.float32)


@pytest.mark.parametrize("face", [1, 2, 3, 4, 5, 6])
def test_transform_all_faces_signs(face):
    raw = np.concatenate([synth.positions(0, 4096, BOX), edge_positions()])
    for sgn in itertools.product((-1, 1), repeat=3):
        for center, rcase in (((0.3, 0.6, 0.1), 3.0), ((0., 1., 0.5), 0.0)):
            a = oracle.transform(raw, BOX, sgn, face, center, rcase)
            b = npr.transform(raw, BOX, sgn, face, center, rcase)
            for u, v in zip(a, b):
                assert np.array_equal(u.view(np.uint32), v.view(np.uint32))


@pytest.mark.parametrize("nrep", [0, 1])
def test_select_project_bits(nrep):
    raw = synth.positions(0, 200000, BOX)
    x, y, z = oracle.transform(raw, BOX, (-1, 1, -1), 3, (0.3, 0.6, 0.1), 3.0)
    m = np.full(len(x), 0.0123, np.float32)
    for ld, ld2, fov, npix in ((3.0, 3.25, 0.25, 256), (3.25, 4.0, 0.12, 64)):
        xs, ys, ms, idx = oracle.select_project(x, y, z, None, 0.0123, ld, ld2, BOX, nrep, fov, npix, want_index=True)
        xs2, ys2, ms2, idx2 = npr.select_project(x, y, z, m, ld, ld2, BOX, nrep, fov, npix)
        assert len(xs) > 1000
        assert np.array_equal(idx, idx2)
        # numpy may route arcsin/arctan2 through its own SIMD kernels (<=1 ulp f64 from glibc): after
        # rounding to f32 a mismatch has probability ~1e-8 per value; demand none on this sample.
        assert np.array_equal(xs.view(np.uint32), xs2.view(np.uint32))
        assert np.array_equal(ys.view(np.uint32), ys2.view(np.uint32))
        assert np.array_equal(ms, ms2)


def test_slab_boundaries_inclusive_exclusive():
    # z == minDist is kept (>=), z == maxDist is dropped (<)   densitymaps.cpp:374
    z = np.array([3.0, np.nextafter(np.float32(3.25), np.float32(0)), 3.25], np.float32)
    x = np.full(3, 0.5, np.float32)
    xs, ys, ms, idx = oracle.select_project(x, x, z, None, 1.0, 3.0, 3.25, BOX, 0, 0.25, 64, want_index=True)
    assert list(idx) == [0, 1]


def test_fov_margin_inclusive():
    # |ang| <= fov*(1+2/npix)/2 keeps one pixel beyond the map edge   densitymaps.cpp:383
    npix, fov = 64, 0.25
    lim = fov * (1. + 2. / npix) * 0.5
    zz = np.float32(3.5)
    inside = np.float32(0.5 + float(zz) * np.tan(lim * 0.999))
    outside = np.float32(0.5 + float(zz) * np.tan(lim * 1.001))
    y = np.array([inside, outside], np.float32)
    x = np.full(2, 0.5, np.float32)
    z = np.full(2, zz, np.float32)
    xs, ys, ms, idx = oracle.select_project(x, y, z, None, 1.0, 3.0, 4.0, BOX, 0, fov, npix, want_index=True)
    assert list(idx) == [0]
    assert ys[0] > 1.0     # lands in the border ring beyond the last pixel


@pytest.mark.parametrize("nn", [16, 64, 24])
@pytest.mark.parametrize("ngp", [False, True])
def test_gridist_bits(nn, ngp):
    rng = np.random.default_rng(nn)
    n = 3000
    dl = 1.0 / nn
    xs = rng.uniform(-dl, 1 + dl, n).astype(np.float32)
    ys = rng.uniform(-dl, 1 + dl, n).astype(np.float32)
    # exact pixel edges / centres too
    xs[:nn + 1] = (np.arange(nn + 1) * dl).astype(np.float32)
    ys[:nn + 1] = 0.5
    ws = rng.uniform(0.001, 3.0, n).astype(np.float32)
    a = oracle.gridist_w(xs, ys, ws, nn, ngp)
    b = npr.gridist_w(xs, ys, ws, nn, ngp)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_weight_bits():
    rng = np.random.default_rng(5)
    for nn in (8, 24, 1024, 4096):
        dl = 1.0 / nn
        x = rng.uniform(0, 1, 5000).astype(np.float32)
        h = (x + rng.uniform(-2 * dl, 2 * dl, 5000)).astype(np.float32)
        b = npr.weight(x, h, dl)
        a = np.array([oracle.weight(xi, hi, dl) for xi, hi in zip(x, h)], np.float32)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_create_density_maps_multi_file_multi_type():
    """A5: per-file/per-type sequential maps, left-assoc f32 type sum, file-order accumulation."""
    rng = np.random.default_rng(11)
    files = []
    first = 0
    for ff in range(3):
        npart = [500, 3000, 0, 700, 0, 50] if ff != 1 else [0, 2500, 400, 0, 0, 0]
        n = sum(npart)
        files.append(dict(npart=npart, massarr=[0.5, 0.0123, 0.3, 0.07, 0, 1.5], boxsize=BOX,
                          pos=synth.positions(first, n, BOX)))
        first += n
    npix, fov, ld, ld2 = 32, 0.25, 3.0, 4.0
    args = ((-1, 1, -1), 3, (0.3, 0.6, 0.1), 3.0)
    for ngp in (False, True):
        rc, tot, toti, nsel = oracle.create_density_maps(files, 0, 3, npix, False, ngp, ld, ld2, 0, fov, *args)
        assert rc == 0
        tot2 = np.zeros((npix, npix), np.float32)
        toti2 = np.zeros((6, npix, npix), np.float32)
        for f in files:
            off = 0
            mapi = np.zeros((6, npix, npix), np.float32)
            for t in range(6):
                n = f["npart"][t]
                if n == 0:
                    continue
                x, y, z = npr.transform(f["pos"][off:off + n], BOX, *args)
                off += n
                m = np.full(n, np.float32(f["massarr"][t]), np.float32)
                xs, ys, ms, _ = npr.select_project(x, y, z, m, ld, ld2, BOX, 0, fov, npix)
                if len(xs):
                    mapi[t] = npr.gridist_w(xs, ys, ms, npix, ngp)
            s = mapi[0] + mapi[1]
            for t in range(2, 6):
                s = s + mapi[t]
            tot2 = tot2 + s
            toti2 = toti2 + mapi
        assert np.array_equal(tot.view(np.uint32), tot2.view(np.uint32))
        assert np.array_equal(toti.view(np.uint32), toti2.view(np.uint32))
        assert nsel.sum() > 1000


def test_guard_fires_on_out_of_box_input():
    f = dict(npart=[0, 2, 0, 0, 0, 0], massarr=[0, 1, 0, 0, 0, 0], boxsize=BOX,
             pos=np.array([[500, 500, 500], [-2600, 500, 500]], np.float32))
    rc, *_ = oracle.create_density_maps([f], 0, 1, 8, False, False, 0.0, 1.0, 0, 1.0, (1, 1, 1), 1, (0, 0, 0), 0.0)
    assert rc == 1
