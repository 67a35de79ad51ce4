"""The C++ createDensityMaps adapter (slicer_amd/csrc/densitymaps_amd.cpp) driven like slicer-v2.cpp drives the
reference: real format-2 sub-files on disk, reference structs, valarray outputs."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from slicer_amd import gadget, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "adapter_driver")
BOX = 1000.0


def make_files(tmp_path, hydro):
    base = str(tmp_path / "snap_042")
    files, first = [], 0
    rng = np.random.default_rng(9)
    for ff in range(2):
        npart = [4001 + ff, 50003, 0, 1501, 0, 0]
        n = sum(npart)
        pos = synth.positions(first, n, BOX)
        first += n
        massarr = [0.0 if hydro else 0.02, 0.0123, 0, 0.3, 0, 0]
        m0 = rng.uniform(0.01, 0.03, npart[0]).astype(np.float32) if hydro else None
        gadget.write_snapshot(f"{base}.{ff}", pos, npart, massarr, BOX, numfiles=2, mass=m0)
        files.append(dict(npart=npart, massarr=massarr, boxsize=BOX, pos=pos, mass={0: m0} if hydro else {}))
    return base, files


def run_driver(base, ffmin, ffmax, npix, fov, ld, ld2, rcase, ngp, hydro, out, env=None):
    return subprocess.run([DRIVER, base, str(ffmin), str(ffmax), str(npix), repr(fov), repr(ld), repr(ld2), repr(rcase),
                           str(int(ngp)), str(int(hydro)), out], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, **(env or {})))


def test_driver_is_built_and_fails_cleanly_without_input(tmp_path):
    assert os.path.exists(DRIVER), "run __graft_entry__.build()"
    r = run_driver(str(tmp_path / "nothing"), 0, 1, 16, 0.25, 3.0, 4.0, 3.0, 1, 0, str(tmp_path / "o.bin"))
    assert r.returncode == 1  # no device here, or no such file on a GPU box: the reference's "return 1"
    assert "slicer_amd" in r.stderr or "Error in opening the file" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("hydro", [False, True])
def test_adapter_matches_oracle(tmp_path, hydro):
    npix, fov, ld, ld2, rcase = 64, 0.25, 3.0, 4.0, 3.0
    base, files = make_files(tmp_path, hydro)
    for ngp in (True, False):
        out = str(tmp_path / f"maps_{int(ngp)}.bin")
        r = run_driver(base, 0, 2, npix, fov, ld, ld2, rcase, ngp, hydro, out)
        assert r.returncode == 0, r.stderr
        raw = np.fromfile(out, np.float32, 7 * npix * npix).reshape(7, npix, npix)
        ntot = np.fromfile(out, np.int32, 6, offset=4 * 7 * npix * npix)
        rc, tot, toti, nsel = oracle.create_density_maps(files, 0, 2, npix, hydro, ngp, ld, ld2, 0, fov,
                                                         (-1, 1, -1), 3, (0.3, 0.6, 0.1), rcase)
        assert rc == 0 and np.all(ntot == 0)  # the reference's ntotxyi stays 0 (densitymaps.cpp:497)
        if ngp and not hydro:
            assert np.array_equal(raw[0].view(np.uint32), tot.view(np.uint32))
            assert np.array_equal(raw[1:].view(np.uint32), toti.view(np.uint32))
        else:
            for got, ref in [(raw[0], tot)] + [(raw[1 + t], toti[t]) for t in range(6)]:
                d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
                assert np.all(d <= 3e-6 * ref), float(d.max())
            if ngp:  # constant-mass species stay bit-exact under NGP even in a hydro run
                assert np.array_equal(raw[2].view(np.uint32), toti[1].view(np.uint32))
    # a missing sub-file makes createDensityMaps return 1, as readHeader's failure does (densitymaps.cpp:438)
    r = run_driver(base, 0, 3, npix, fov, ld, ld2, rcase, 1, hydro, str(tmp_path / "x.bin"))
    assert r.returncode == 1 and "Error in opening the file" in r.stderr


@pytest.mark.gpu
def test_adapter_deposits_the_planes_of_one_replication_in_one_pass(tmp_path):
    """slicer-v2.cpp calls createDensityMaps once per plane; the planes of one box replication share sub-files, Random
    entry and rcase, so the adapter deposits them all during the first call and serves the others from the device.
    Every plane must come out as if it had its own pass: bit for bit (NGP), and identical bytes with the look-ahead off."""
    npix, fov, rcase = 64, 0.25, 3.0
    lds, ld2s = [3.0, 3.3, 3.7], [3.3, 3.7, 4.0]
    base, files = make_files(tmp_path, False)
    per_plane = 7 * npix * npix * 4 + 6 * 4
    outs = {}
    for look in ("1", "0"):
        out = str(tmp_path / f"planes_{look}.bin")
        env = dict(os.environ, SLICER_AMD_LOOKAHEAD=look, ADAPTER_TIMES="1")
        r = subprocess.run([DRIVER, base, "0", "2", str(npix), repr(fov), ",".join(map(repr, lds)), ",".join(map(repr, ld2s)),
                            repr(rcase), "1", "0", out], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        outs[look] = open(out, "rb").read()
        assert len(outs[look]) == 3 * per_plane
        assert r.stderr.count("createDensityMaps call") == 3
    assert outs["1"] == outs["0"]
    for p in range(3):
        raw = np.frombuffer(outs["1"], np.float32, 7 * npix * npix, offset=p * per_plane).reshape(7, npix, npix)
        rc, tot, toti, nsel = oracle.create_density_maps(files, 0, 2, npix, False, True, lds[p], ld2s[p], 0, fov,
                                                         (-1, 1, -1), 3, (0.3, 0.6, 0.1), rcase)
        assert rc == 0 and nsel.sum() > 0
        assert np.array_equal(raw[0].view(np.uint32), tot.view(np.uint32))
        assert np.array_equal(raw[1:].view(np.uint32), toti.view(np.uint32))


@pytest.mark.gpu
def test_adapter_can_skip_the_per_type_maps_the_caller_discards(tmp_path):
    """Without partinplanes the reference's caller never reads mapxytoti (writeMaps, densitymaps.cpp:537-584).  With the
    opt-in switch (slicer_amd_adapter_skip_type_maps / SLICER_AMD_SKIP_TYPE_MAPS=1) the adapter neither builds nor copies
    them: mapxytot is unchanged (bitwise under NGP, inside the TSC bar otherwise), mapxytoti reads zero.  With
    partinplanes the switch does nothing."""
    npix, fov, ld, ld2, rcase = 64, 0.25, 3.0, 4.0, 3.0
    base, files = make_files(tmp_path, False)
    for ngp in (True, False):
        rc, tot, toti, nsel = oracle.create_density_maps(files, 0, 2, npix, False, ngp, ld, ld2, 0, fov,
                                                         (-1, 1, -1), 3, (0.3, 0.6, 0.1), rcase)
        assert rc == 0
        for pip, expect_types in (("0", False), ("1", True)):
            out = str(tmp_path / f"skip_{int(ngp)}_{pip}.bin")
            r = run_driver(base, 0, 2, npix, fov, ld, ld2, rcase, ngp, False, out,
                           env={"SLICER_AMD_SKIP_TYPE_MAPS": "1", "ADAPTER_PARTINPLANES": pip})
            assert r.returncode == 0, r.stderr
            raw = np.fromfile(out, np.float32, 7 * npix * npix).reshape(7, npix, npix)
            if ngp:
                assert np.array_equal(raw[0].view(np.uint32), tot.view(np.uint32))
            else:
                d = np.abs(raw[0].astype(np.float64) - tot.astype(np.float64))
                assert np.all(d <= 3e-6 * tot), float(d.max())
            if expect_types:
                assert float(raw[2].sum()) > 0
                if ngp:
                    assert np.array_equal(raw[1:].view(np.uint32), toti.view(np.uint32))
            else:
                assert not raw[1:].any()


@pytest.mark.gpu
def test_adapter_streams_black_hole_masses_from_bhma(tmp_path):
    """densitymaps.cpp:358-372: per-particle masses stream from the MASS block in type order, except type 5, which skips its
    entries there and streams from BHMA.  Through the C++ reader and adapter: gas and black holes with per-particle
    masses next to two constant-mass species, every per-type map against the oracle."""
    npix, fov, ld, ld2, rcase = 64, 0.25, 3.0, 4.0, 3.0
    base = str(tmp_path / "snap_077")
    rng = np.random.default_rng(21)
    files, first = [], 0
    for ff in range(2):
        npart = [3001 + ff, 40003, 0, 1201, 0, 907]
        n = sum(npart)
        pos = synth.positions(first, n, BOX)
        first += n
        massarr = [0.0, 0.0123, 0, 0.3, 0, 0.0]
        m0 = rng.uniform(0.01, 0.03, npart[0]).astype(np.float32)
        bh = rng.uniform(0.5, 5.0, npart[5]).astype(np.float32)
        # MASS holds an entry for every particle of a massarr == 0 species, type 5 included (its entries are skipped)
        mass_block = np.concatenate([m0, np.full(npart[5], 123.0, np.float32)])
        gadget.write_snapshot(f"{base}.{ff}", pos, npart, massarr, BOX, numfiles=2, mass=mass_block, bhmass=bh)
        files.append(dict(npart=npart, massarr=massarr, boxsize=BOX, pos=pos, mass={0: m0, 5: bh}))
    out = str(tmp_path / "maps.bin")
    r = run_driver(base, 0, 2, npix, fov, ld, ld2, rcase, True, True, out)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, np.float32, 7 * npix * npix).reshape(7, npix, npix)
    rc, tot, toti, nsel = oracle.create_density_maps(files, 0, 2, npix, True, True, ld, ld2, 0, fov, (-1, 1, -1), 3,
                                                     (0.3, 0.6, 0.1), rcase)
    assert rc == 0 and nsel[5] > 100 and nsel[0] > 100
    for t in (1, 3):  # constant-mass species: bit for bit under NGP
        assert np.array_equal(raw[1 + t].view(np.uint32), toti[t].view(np.uint32)), t
    for t in (0, 5):
        d = np.abs(raw[1 + t].astype(np.float64) - toti[t])
        assert np.all(d <= 3e-6 * toti[t]), (t, float(d.max()))
    d = np.abs(raw[0].astype(np.float64) - tot)
    assert np.all(d <= 3e-6 * tot)


@pytest.mark.gpu
def test_adapter_shot_noise_follows_the_stream_randomizebox_left(tmp_path):
    """snopt > 0 through the C++ adapter in a fresh process: the adapter reads libc's rand() stream at its first call --
    before the HIP runtime starts, whose threads draw from that stream now and then -- and thins from its own copy, plane
    after plane.  NGP maps bit for bit like the oracle started from the same srand + five draws."""
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    npix, fov, rcase = 64, 0.25, 3.0
    base, files = make_files(tmp_path, False)
    lds, ld2s = [3.0, 3.3, 3.6], [3.3, 3.6, 4.0]
    out = str(tmp_path / "maps.bin")
    env = dict(os.environ, ADAPTER_SNOPT="2", ADAPTER_SRAND="777")
    r = subprocess.run([DRIVER, base, "0", "2", str(npix), repr(fov), ",".join(map(repr, lds)), ",".join(map(repr, ld2s)),
                        repr(rcase), "1", "0", out], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    per_plane = 7 * npix * npix * 4 + 6 * 4
    blob = open(out, "rb").read()
    assert len(blob) == 3 * per_plane
    libc.srand(777)
    for _ in range(5):
        libc.rand()
    for p in range(3):
        rc, ref_tot, ref_toti, nsel = oracle.create_density_maps(files, 0, 2, npix, False, True, lds[p], ld2s[p], 0, fov,
                                                                 (-1, 1, -1), 3, (0.3, 0.6, 0.1), rcase, snopt=2)
        assert rc == 0 and nsel.sum() > 0
        raw = np.frombuffer(blob, np.float32, 7 * npix * npix, offset=p * per_plane).reshape(7, npix, npix)
        assert np.array_equal(raw[0].view(np.uint32), ref_tot.view(np.uint32)), p
        assert np.array_equal(raw[1:].view(np.uint32), ref_toti.view(np.uint32)), p
