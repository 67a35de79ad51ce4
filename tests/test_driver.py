"""SLICER_amd driver (slicer_amd/csrc/slicer_main.cpp + planner.cpp): planning on CPU, full run on the GPU.

The planner restates readInput / readRedList / buildPlanes / randomizeBox / testFov (densitymaps.cpp:9-283,
data.cpp:8-87, gadget2io.cpp:613-661).  GSL is absent, so spline-interpolated values are unpinned against GSL; what
is checked here: the plane grid (multiples of box/4), the snapshot choice, the randomisation plan against an
independent evaluation through libc's srand/rand, the comoving distance against numerical quadrature, and -- on the
GPU -- every written FITS plane against the oracle fed with the dumped plan."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import oracle
from slicer_amd import gadget, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "slicer_amd", "SLICER_amd")
BOX = 100000.0  # kpc/h


def make_cone(tmp_path, npix=32, partinplanes=0, zs=0.2, snopt=0, hydro=False, fov=2.0):
    snaps = [("snapdir_003/snap_003", 0.0), ("snapdir_002/snap_002", 0.1), ("snapdir_001/snap_001", 0.25)]
    files = {}
    first = 0
    rng = np.random.default_rng(11)
    for name, z in snaps:
        os.makedirs(tmp_path / os.path.dirname(name), exist_ok=True)
        fl = []
        for ff in range(2):
            # hydro: gas (type 0) with massarr[0] = 0 and a MASS block -- per-particle masses, some above MAX_M
            npart = [2000 + 3 * ff if hydro else 0, 3000 + 7 * ff, 501, 0, 0, 0]
            n = sum(npart)
            pos = synth.positions(first, n, BOX)
            first += n
            m0 = None
            if hydro:
                m0 = rng.uniform(0.001, 0.05, npart[0]).astype(np.float32)
                m0[::211] = 2000.0
            gadget.write_snapshot(str(tmp_path / f"{name}.{ff}"), pos, npart, [0, 0.0123, 0.3, 0, 0, 0], BOX, numfiles=2,
                                  redshift=z, om0=0.3, oml=0.7, h=0.7, **({"mass": m0} if hydro else {}))
            fl.append(dict(npart=npart, massarr=[0, 0.0123, 0.3, 0, 0, 0], boxsize=BOX, pos=pos,
                           **({"mass": {0: m0}} if hydro else {})))
        files[name] = fl
    (tmp_path / "snapshot_list.txt").write_text("\n".join(n for n, _ in snaps))
    out = tmp_path / "out"
    out.mkdir()
    vals = [npix, zs, fov, str(tmp_path / "snapshot_list.txt"), str(tmp_path) + "/", "gadget", -229, -230, -231,
            partinplanes, str(out) + "/cone_", "t0", snopt, -1.0]
    ini = tmp_path / "InputParams.ini"
    ini.write_text("".join(f"##### {i + 1}. #####\n{v}\n" for i, v in enumerate(vals)))
    return str(ini), files, str(out)


def run(args, **kw):
    return subprocess.run([EXE] + args, capture_output=True, text=True, timeout=600, **kw)


def test_plan_only_matches_independent_checks(tmp_path):
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    ini, files, out = make_cone(tmp_path)
    plan_path = str(tmp_path / "plan.json")
    r = run([ini, "--plan-only", "--dump-plan", plan_path])
    assert r.returncode == 0, r.stderr
    plan = json.load(open(plan_path))
    planes = plan["planes"]
    n = plan["nplanes"]
    # comoving distance to zs = 0.2 in flat LCDM (h = 1 units): c/H0 * int dz / E(z)
    z = np.linspace(0, 0.2, 20001)
    dc = 2997.92458 * np.trapezoid(1 / np.sqrt(0.3 * (1 + z) ** 3 + 0.7), z)
    # the reference's integrator (w0waCDM.cpp:51-54) steps "for (zi = lastZ; zi < z; zi += dz)" with dz = (z-lastZ)/100:
    # rounding makes some segments take a 101st trapezoid, a +0.06 % bias that the restatement keeps on purpose
    assert 0 <= plan["Ds"] / dc - 1 < 2e-3
    step = BOX / 1e3 / 4
    assert n == len(planes) == int(np.ceil(plan["Ds"] / step)) and n >= 20
    for i, pl in enumerate(planes):
        assert pl["ld"] == pytest.approx(i * step, abs=1e-9) and pl["ld2"] == pytest.approx((i + 1) * step, abs=1e-9)
        assert pl["randomize"] == (1 if i % 4 == 0 else 0)
        assert pl["nrepperp"] == 0
    # the snapshot whose distance is closest to the plane centre is used: z=0 first, later planes move to higher z
    seq = [pl["fromsnapi"] for pl in planes]
    assert seq[0] == 0 and seq == sorted(seq) and seq[-1] >= 1
    # randomisation plan re-derived through libc (densitymaps.cpp:187-217)
    libc = C.CDLL("libc.so.6")
    RAND_MAX = np.float32(2147483647)

    def frand():
        return float(np.float32(libc.rand()) / RAND_MAX)
    for i, pl in enumerate(planes):
        if not pl["randomize"]:
            for k in ("x0", "y0", "z0", "face", "sgn"):
                assert pl[k] == planes[i - 1][k]
            continue
        g = i // 4
        libc.srand(C.c_uint((-229 + g * 13) & 0xFFFFFFFF))
        exp = [frand(), frand(), frand()]
        assert [pl["x0"], pl["y0"], pl["z0"]] == exp
        libc.srand(C.c_uint((-230 + g * 5) & 0xFFFFFFFF))
        face = 7
        while face > 6 or face < 1:
            face = int(1 + float(np.float32(libc.rand()) / RAND_MAX) * 5. + 0.5)
        assert pl["face"] == face
        libc.srand(C.c_uint((-231 + g * 8) & 0xFFFFFFFF))
        sg = []
        for _ in range(3):
            v = 2
            while v > 1 or v < 0:
                v = int(float(np.float32(libc.rand()) / RAND_MAX) + 0.5)
            sg.append(v if v else -1)
        assert pl["sgn"] == sg
    lines = open(os.path.join(out, "cone_planes_list_t0.txt")).read().strip().split("\n")
    assert len(lines) == n and lines[0].split()[0] == "0" and lines[0].split()[5] == "snapdir_003/snap_003"


def test_bad_inputs_fail_like_the_reference(tmp_path):
    assert run([]).returncode == 2
    assert run([str(tmp_path / "missing.ini"), "--plan-only"]).returncode == 1
    ini, _, _ = make_cone(tmp_path)
    txt = open(ini).read().replace("\n2.0\n", "\n40.0\n")  # field of view wider than the box at the last plane
    open(ini, "w").write(txt)
    r = run([ini, "--plan-only"])
    assert r.returncode == 1 and "Field view too large" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("partinplanes", [0, 1])
def test_full_run_writes_planes_that_match_the_oracle(tmp_path, partinplanes):
    ini, files, out = make_cone(tmp_path, partinplanes=partinplanes)
    plan_path = str(tmp_path / "plan.json")
    r = run([ini, "--ngp", "--dump-plan", plan_path])
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.load(open(plan_path))
    rcase = 0.0
    checked = 0
    for i, pl in enumerate(plan["planes"]):
        if pl["randomize"]:
            rcase = float(np.float32(pl["ld"] / pl["snapbox"] * 1e3))
        fl = files[pl["fromsnap"]]
        rc, tot, toti, nsel = oracle.create_density_maps(fl, 0, 2, 32, False, True, pl["ld"], pl["ld2"], 0,
                                                         plan["fovradiants"], pl["sgn"], pl["face"],
                                                         (pl["x0"], pl["y0"], pl["z0"]), rcase)
        assert rc == 0
        label = "%03d" % i
        if not partinplanes:
            raw = open(os.path.join(out, f"cone_gadget.{label}.plane_32_t0.fits"), "rb").read()
            data = np.frombuffer(raw[2880:2880 + 4 * 1024], ">f4").reshape(32, 32).astype(np.float32)
            assert np.array_equal(data.view(np.uint32), tot.view(np.uint32))
            assert (b"HIERARCH NPARTTYPE1 = %8d" % nsel[1]) in raw[:2880]
        else:
            for t in (1, 2):
                path = os.path.join(out, f"cone_gadget.{label}.ptype{t}_plane_32_t0.fits")
                if nsel[t] == 0:
                    assert not os.path.exists(path)
                    continue
                raw = open(path, "rb").read()
                data = np.frombuffer(raw[2880:2880 + 4 * 1024], ">f4").reshape(32, 32).astype(np.float32)
                assert np.array_equal(data.view(np.uint32), toti[t].view(np.uint32))
        checked += 1
    assert checked == plan["nplanes"]
    if not partinplanes:  # resume: a second run finds every plane and does nothing
        r2 = run([ini, "--ngp"])
        assert r2.returncode == 0 and r2.stdout.count("Already exists") == plan["nplanes"]
        # and the single-plane (reference-shaped) passes give identical files
        for f in os.listdir(out):
            if f.endswith(".fits"):
                os.rename(os.path.join(out, f), os.path.join(out, f + ".multi"))
        assert run([ini, "--ngp", "--single-plane"]).returncode == 0
        for f in os.listdir(out):
            if f.endswith(".fits"):
                assert open(os.path.join(out, f), "rb").read() == open(os.path.join(out, f + ".multi"), "rb").read()


@pytest.mark.gpu
def test_reference_counts_switch_reproduces_the_reference_quirks(tmp_path):
    """The reference's ntotxyi stays 0 (an inner array shadows the out-parameter, densitymaps.cpp:497): its FITS headers
    carry NPARTTYPE* = 0 and, with partinplanes, `if (ntotxyi[i] > 0)` (densitymaps.cpp:593) is never true, so no
    per-type file is ever written.  --reference-counts keeps both quirks; the default writes the real counts."""
    ini, files, out = make_cone(tmp_path)
    assert run([ini, "--ngp", "--reference-counts"]).returncode == 0
    names = sorted(f for f in os.listdir(out) if f.endswith(".fits"))
    assert len(names) >= 20
    for f in names:
        hdr = open(os.path.join(out, f), "rb").read(2880)
        for t in range(6):
            assert (b"HIERARCH NPARTTYPE%d = %8d" % (t, 0)) in hdr, (f, t)
    ini2, _, out2 = make_cone(tmp_path / "pip", partinplanes=1)
    assert run([ini2, "--ngp", "--reference-counts"]).returncode == 0
    assert not [f for f in os.listdir(out2) if f.endswith(".fits")]


@pytest.mark.gpu
def test_full_run_with_lateral_replication(tmp_path):
    """--replication (the reference's -DReplicationOnPerpendicularPlane): a 60-degree field is wider than the box from the
    second box replication on, so the planner gives the far planes lateral copies of the box (computeReplications) and the
    deposit emits every copy inside the field (densitymaps.cpp:377-401).  Without the flag the same field is refused."""
    ini, files, out = make_cone(tmp_path, fov=60.0)
    r0 = run([ini, "--plan-only"])
    assert r0.returncode == 1 and "Field view too large" in r0.stderr
    plan_path = str(tmp_path / "plan.json")
    r = run([ini, "--ngp", "--replication", "--dump-plan", plan_path])
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.load(open(plan_path))
    reps = [pl["nrepperp"] for pl in plan["planes"]]
    assert reps[0] == 0 and max(reps) >= 3 and reps == sorted(reps)
    rcase = 0.0
    for i, pl in enumerate(plan["planes"]):
        if pl["randomize"]:
            rcase = float(np.float32(pl["ld"] / pl["snapbox"] * 1e3))
        rc, tot, toti, nsel = oracle.create_density_maps(files[pl["fromsnap"]], 0, 2, 32, False, True, pl["ld"], pl["ld2"],
                                                         pl["nrepperp"], plan["fovradiants"], pl["sgn"], pl["face"],
                                                         (pl["x0"], pl["y0"], pl["z0"]), rcase)
        assert rc == 0
        raw = open(os.path.join(out, "cone_gadget.%03d.plane_32_t0.fits" % i), "rb").read()
        data = np.frombuffer(raw[2880:2880 + 4 * 1024], ">f4").reshape(32, 32).astype(np.float32)
        assert np.array_equal(data.view(np.uint32), tot.view(np.uint32)), (i, pl["nrepperp"])
        assert (b"HIERARCH NPARTTYPE1 = %8d" % nsel[1]) in raw[:2880]
        if pl["nrepperp"] >= 3:
            assert nsel[1] > 6000   # more entries than the snapshot has particles of the species


@pytest.mark.gpu
def test_full_run_with_a_physical_pixel_size(tmp_path):
    """npix < 0 in the parameter file asks for pixels of -npix kpc/h: every plane gets its own map size
    (slicer-v2.cpp:142-143: int(mean distance * fov / rgrid * 1e3) + 1) and the files carry "<n>_kpc" in their names.
    Each plane bit for bit (NGP) like the oracle at that plane's size."""
    ini, files, out = make_cone(tmp_path, npix=-150)
    plan_path = str(tmp_path / "plan.json")
    r = run([ini, "--ngp", "--dump-plan", plan_path])
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.load(open(plan_path))
    rcase, sizes = 0.0, set()
    for i, pl in enumerate(plan["planes"]):
        if pl["randomize"]:
            rcase = float(np.float32(pl["ld"] / pl["snapbox"] * 1e3))
        npix = int((pl["ld2"] + pl["ld"]) / 2 * plan["fovradiants"] / 150 * 1e3 / 1.0) + 1
        sizes.add(npix)
        rc, tot, toti, nsel = oracle.create_density_maps(files[pl["fromsnap"]], 0, 2, npix, False, True, pl["ld"], pl["ld2"], 0,
                                                         plan["fovradiants"], pl["sgn"], pl["face"],
                                                         (pl["x0"], pl["y0"], pl["z0"]), rcase)
        assert rc == 0
        raw = open(os.path.join(out, "cone_gadget.%03d.plane_150_kpc_t0.fits" % i), "rb").read()
        assert (b"NAXIS1  = %20d" % npix) in raw[:2880] and (b"NAXIS2  = %20d" % npix) in raw[:2880]
        data = np.frombuffer(raw[2880:2880 + 4 * npix * npix], ">f4").reshape(npix, npix).astype(np.float32)
        assert np.array_equal(data.view(np.uint32), tot.view(np.uint32)), i
    assert len(sizes) > 5 and max(sizes) > 3 * min(sizes)


@pytest.mark.gpu
def test_full_run_with_per_particle_masses(tmp_path):
    """A hydro snapshot through the whole driver (testHydro: a species with npart > 0 and massarr = 0 carries a MASS
    block; densitymaps.cpp:358-372 reads one mass per particle, MAX_M zeroes the outliers), with per-type files
    (partinplanes): the constant-mass species bit for bit (NGP), the gas map and the total inside the f32 bar."""
    ini, files, out = make_cone(tmp_path, partinplanes=1, hydro=True)
    plan_path = str(tmp_path / "plan.json")
    r = run([ini, "--ngp", "--dump-plan", plan_path])
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.load(open(plan_path))
    assert plan["hydro"] == 1
    rcase, gas = 0.0, 0
    for i, pl in enumerate(plan["planes"]):
        if pl["randomize"]:
            rcase = float(np.float32(pl["ld"] / pl["snapbox"] * 1e3))
        rc, tot, toti, nsel = oracle.create_density_maps(files[pl["fromsnap"]], 0, 2, 32, True, True, pl["ld"], pl["ld2"], 0,
                                                         plan["fovradiants"], pl["sgn"], pl["face"],
                                                         (pl["x0"], pl["y0"], pl["z0"]), rcase)
        assert rc == 0
        for t in (0, 1, 2):
            path = os.path.join(out, "cone_gadget.%03d.ptype%d_plane_32_t0.fits" % (i, t))
            if nsel[t] == 0:
                assert not os.path.exists(path)
                continue
            raw = open(path, "rb").read()
            data = np.frombuffer(raw[2880:2880 + 4 * 1024], ">f4").reshape(32, 32).astype(np.float32)
            if t == 0:
                gas += int(nsel[0])
                assert np.all(np.abs(data.astype(np.float64) - toti[0]) <= 3e-6 * toti[0]), (i, t)
            else:
                assert np.array_equal(data.view(np.uint32), toti[t].view(np.uint32)), (i, t)
    assert gas > 100


@pytest.mark.gpu
@pytest.mark.parametrize("accum", ["f32", "f64", "fixed64"])
def test_full_run_with_tsc_stays_inside_the_tolerance(tmp_path, accum):
    """The default mass assignment (TSC, densitymaps.h:22 DO_NGP false) through the whole driver, every accumulator
    type: each FITS plane within the T-TSC bar of the oracle's map, zero pixels where the oracle has zeros."""
    ini, files, out = make_cone(tmp_path)
    plan_path = str(tmp_path / "plan.json")
    r = run([ini, "--accum", accum, "--dump-plan", plan_path])
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.load(open(plan_path))
    rcase = 0.0
    for i, pl in enumerate(plan["planes"]):
        if pl["randomize"]:
            rcase = float(np.float32(pl["ld"] / pl["snapbox"] * 1e3))
        rc, tot, toti, nsel = oracle.create_density_maps(files[pl["fromsnap"]], 0, 2, 32, False, False, pl["ld"], pl["ld2"], 0,
                                                         plan["fovradiants"], pl["sgn"], pl["face"],
                                                         (pl["x0"], pl["y0"], pl["z0"]), rcase)
        assert rc == 0
        raw = open(os.path.join(out, "cone_gadget.%03d.plane_32_t0.fits" % i), "rb").read()
        data = np.frombuffer(raw[2880:2880 + 4 * 1024], ">f4").reshape(32, 32).astype(np.float64)
        assert np.all(data[tot == 0] == 0), i
        if accum != "fixed64":  # (a contribution below the fixed-point quantum rounds to zero there)
            assert np.array_equal(data == 0, tot == 0), i
        # FIXED64 resolves 2^-40 of the mass scale absolutely: a pixel of 1e-9 carries that as a relative 1e-4
        slack = 1e-11 if accum == "fixed64" else 0.0
        assert np.all(np.abs(data - tot) <= 3e-6 * tot + slack), i


@pytest.mark.gpu
@pytest.mark.parametrize("partinplanes", [0, 1])
def test_two_rank_driver_equals_one_rank_bitwise_with_fixed64(tmp_path, partinplanes):
    """slicer-v2.cpp:162-175 + 214-217 in the driver: `--devices 0,0` runs two ranks (two handles, one host thread each;
    here on the same GPU, so the sum goes through host memory with --reduce host -- the RCCL clique needs distinct
    GPUs), each depositing its contiguous range of sub-files; the partial FIXED64 accumulators are summed as integers
    onto rank 0, which writes the planes.  Every FITS file must be byte-identical to the one-rank run.  With
    partinplanes the second sub-file range holds species the first lacks in no case here, but the per-type maps go
    through the same rank-invariant protocol."""
    ini, files, out = make_cone(tmp_path, partinplanes=partinplanes)
    assert run([ini, "--accum", "fixed64"]).returncode == 0
    one = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out)) if f.endswith(".fits")}
    assert len(one) >= 20
    for f in one:
        os.remove(os.path.join(out, f))
    r = run([ini, "--accum", "fixed64", "--devices", "0,0", "--reduce", "host"])
    assert r.returncode == 0, r.stderr[-2000:]
    two = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out)) if f.endswith(".fits")}
    assert two.keys() == one.keys()
    for f in one:
        assert one[f] == two[f], f
    # f32 accumulators: the two-rank sum differs from the one-rank one only by f32 reordering
    for f in two:
        os.remove(os.path.join(out, f))
    assert run([ini, "--devices", "0,0", "--reduce", "host"]).returncode == 0
    for f in one:
        a = np.frombuffer(open(os.path.join(out, f), "rb").read()[2880:2880 + 4 * 1024], ">f4").astype(np.float64)
        b = np.frombuffer(one[f][2880:2880 + 4 * 1024], ">f4").astype(np.float64)
        assert np.allclose(a, b, rtol=3e-6, atol=2.0 ** -29)


@pytest.mark.gpu
def test_a_rank_that_fails_locally_stops_every_rank_before_the_collective(tmp_path):
    """a rank whose deposit phase fails (here: a sub-file of its range is missing) must not leave the
    others alone in the rank sum.  The rank threads agree on the outcome before the collective (Rendezvous,
    slicer_main.cpp) -- none enters it, the run returns 1 at once (the reference: MPI_Abort, slicer-v2.cpp:204-207)."""
    ini, files, out = make_cone(tmp_path)
    os.remove(str(tmp_path / "snapdir_003/snap_003.1"))   # rank 1's only sub-file of the first snapshot
    r = run([ini, "--devices", "0,0", "--reduce", "host"])
    assert r.returncode == 1
    assert "Error in opening the file" in r.stderr and "another rank failed" in r.stderr
    assert not [f for f in os.listdir(out) if f.endswith(".fits")]


@pytest.mark.gpu
@pytest.mark.parametrize("single_plane", [False, True])
def test_shot_noise_thinning_through_the_driver_matches_the_oracle(tmp_path, single_plane):
    """snopt > 0 end to end: the plan carries libc's rand() state as randomizeBox left it; the oracle, started from that
    state and run plane after plane, must give every FITS plane bit for bit (NGP) -- with the planes of a box replication
    in ONE pass (the library replays the pass plane-major) and with one pass per plane."""
    libc = C.CDLL("libc.so.6")
    from slicer_amd import _lib
    L = _lib.load()
    ini, files, out = make_cone(tmp_path, snopt=2)
    plan_path = str(tmp_path / "plan.json")
    r = run([ini, "--ngp", "--dump-plan", plan_path] + (["--single-plane"] if single_plane else []))
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.load(open(plan_path))
    assert plan["libc_rand_state"] is not None and len(plan["libc_rand_state"]) == 31
    assert L.slicer_libc_rand_state_set((C.c_uint32 * 31)(*plan["libc_rand_state"])) == 0
    rcase, thinned = 0.0, 0
    for i, pl in enumerate(plan["planes"]):
        if pl["randomize"]:
            rcase = float(np.float32(pl["ld"] / pl["snapbox"] * 1e3))
        rc, tot, toti, nsel = oracle.create_density_maps(files[pl["fromsnap"]], 0, 2, 32, False, True, pl["ld"], pl["ld2"], 0,
                                                         plan["fovradiants"], pl["sgn"], pl["face"],
                                                         (pl["x0"], pl["y0"], pl["z0"]), rcase, snopt=2)
        assert rc == 0
        raw = open(os.path.join(out, "cone_gadget.%03d.plane_32_t0.fits" % i), "rb").read()
        data = np.frombuffer(raw[2880:2880 + 4 * 1024], ">f4").reshape(32, 32).astype(np.float32)
        assert np.array_equal(data.view(np.uint32), tot.view(np.uint32)), i
        assert (b"HIERARCH NPARTTYPE1 = %8d" % nsel[1]) in raw[:2880]
        thinned += int(nsel.sum())
    assert thinned > 300  # (a 2-degree cone: few entries per plane, every one of them drawn for)


@pytest.mark.gpu
def test_shot_noise_thinning_on_several_devices_is_reproducible(tmp_path):
    """snopt > 0 draws from libc's rand() stream (densitymaps.cpp:387-397).  The reference's MPI ranks each own an
    identically seeded copy of it; the rank threads of this process each get one as well (slicer_rand_stream_set, started
    from the process state randomizeBox left), instead of interleaving their draws on the process-global stream (ADVICE
    r2; the combination used to be refused): two runs write identical files, and they differ from the one-rank run only
    where the thinning differs -- the selected counts in the headers agree."""
    ini, _, out = make_cone(tmp_path, snopt=2)
    runs = []
    for k in range(2):
        r = run([ini, "--ngp", "--devices", "0,0", "--reduce", "host"])
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append({f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out)) if f.endswith(".fits")})
        for f in runs[-1]:
            os.remove(os.path.join(out, f))
    assert len(runs[0]) >= 10 and runs[0].keys() == runs[1].keys()
    for f in runs[0]:
        assert runs[0][f] == runs[1][f], f
    assert run([ini, "--ngp"]).returncode == 0
    for f in runs[0]:
        one = open(os.path.join(out, f), "rb").read()
        assert one[:2880] == runs[0][f][:2880], f   # same header: same selected counts (NPARTTYPE*), same plane
    assert run([ini, "--plan-only", "--devices", "0,0"]).returncode == 0


def test_driver_device_lists(tmp_path):
    ini, _, _ = make_cone(tmp_path)
    assert run([ini, "--plan-only", "--devices", "0-1"]).returncode == 0   # planning needs no device
    assert run([ini, "--devices", "0-1", "--reduce", "bogus"]).returncode == 2
    assert run([ini, "--devices", "0-1", "--reduce-algo", "bogus"]).returncode == 2
    r = run([ini, "--devices", "0,63"])   # no such device (or no device at all here): fails loudly, nothing written
    assert r.returncode == 1 and "slicer_amd" in r.stderr
