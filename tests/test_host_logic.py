"""Host-side logic that needs no GPU: format-2 files, synthetic generator, file partition."""
import numpy as np
import pytest

import oracle
from slicer_amd import gadget, synth

BOX = 1000.0


def test_gadget_roundtrip_and_block_scan(tmp_path):
    npart = [3, 10, 0, 2, 0, 1]
    n = sum(npart)
    pos = synth.positions(7, n, BOX)
    mass = np.arange(3 + 1, dtype=np.float32)  # types 0 and 5 have massarr == 0
    bh = np.array([42.0], np.float32)
    path = str(tmp_path / "snap.0")
    gadget.write_snapshot(path, pos, npart, [0, 0.0123, 0, 0.5, 0, 0], BOX, numfiles=3, mass=mass, bhmass=bh)
    p2, hdr = gadget.open_snapshot(path)
    assert p2 == path and list(hdr["npart"]) == npart and hdr["numfiles"] == 3 and hdr["boxsize"] == BOX
    assert hdr["massarr"][1] == 0.0123
    assert np.array_equal(gadget.read_positions(path), pos)
    m = gadget.read_masses(path, hdr)
    assert np.array_equal(m[0], mass[:3]) and np.array_equal(m[5], bh)  # type 5 streams from BHMA
    # on-disk framing the reference reader walks (gadget2io.cpp:24-26, 133-165)
    raw = open(path, "rb").read()
    assert raw[4:8] == b"HEAD" and int.from_bytes(raw[16:20], "little") == 256
    assert raw[20 + 256 + 8:20 + 256 + 12] == b"POS "


def test_open_snapshot_falls_back_to_name_without_suffix(tmp_path):
    # gadget2io.cpp:14-17: if "<name>.0" cannot be opened, "<name>" is tried
    pos = synth.positions(0, 4, BOX)
    gadget.write_snapshot(str(tmp_path / "single"), pos, [0, 4, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0], BOX)
    p, hdr = gadget.open_snapshot(str(tmp_path / "single.0"))
    assert p.endswith("single") and hdr["npart"][1] == 4
    with pytest.raises(FileNotFoundError):
        gadget.open_snapshot(str(tmp_path / "missing.0"))


def test_synth_is_chunking_invariant_and_in_box():
    for clustered in (False, True):
        a = synth.positions(0, 5000, BOX, clustered=clustered)
        b = np.concatenate([synth.positions(0, 1234, BOX, clustered=clustered),
                            synth.positions(1234, 5000 - 1234, BOX, clustered=clustered)])
        assert np.array_equal(a, b)
        assert a.min() >= 0 and a.max() <= BOX
    u = synth.positions(0, 200000, BOX)
    assert abs(u.mean() / BOX - 0.5) < 5e-3
    c = synth.positions(0, 200000, BOX, clustered=True)
    h, _ = np.histogramdd(c, bins=16, range=[(0, BOX)] * 3)
    hu, _ = np.histogramdd(u, bins=16, range=[(0, BOX)] * 3)
    assert h.std() > 3 * hu.std()  # clustered boxes really are clustered


def test_oracle_rank_partition_and_reduce_is_linear():
    """a7 (slicer-v2.cpp:162-175, 214-217): per-rank file ranges + rank sum == single-rank result (NGP exact
    when every pixel's partial sums are exactly representable: one species, power-of-two mass)."""
    files, first = [], 0
    for ff in range(5):
        n = 3000 + 17 * ff
        files.append(dict(npart=[0, n, 0, 0, 0, 0], massarr=[0, 0.25, 0, 0, 0, 0], boxsize=BOX,
                          pos=synth.positions(first, n, BOX)))
        first += n
    args = (32, False, True, 3.0, 4.0, 0, 0.25, (-1, 1, -1), 3, (0.3, 0.6, 0.1), 3.0)
    rc, tot, toti, nsel = oracle.create_density_maps(files, 0, 5, *args)
    acc = np.zeros_like(tot)
    seen = []
    for r in range(3):
        a, b = oracle.file_range(5, 3, r)
        seen += list(range(a, b))
        rc, t, _, _ = oracle.create_density_maps(files, a, b, *args)
        oracle.reduce_sum(acc, t)
    assert seen == [0, 1, 2, 3, 4]
    assert np.array_equal(acc, tot)
