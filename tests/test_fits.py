"""The plane FITS writer against libcfitsio itself (the library under the reference's CCfits calls,
densitymaps.cpp:549-583): whole files must be byte-identical.  Runs without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from slicer_amd import fits
from slicer_amd.api import InputParams, Lens

CFITSIO = "/opt/conda/lib/libcfitsio.so"
pytestmark = pytest.mark.skipif(not os.path.exists(CFITSIO), reason="libcfitsio not present")


def cfitsio_write(path, img, keys):
    """The call sequence CCfits issues for FITS(name, FLOAT_IMG, 2, naxes); pHDU().write(...); addKey(...)*"""
    L = C.CDLL(CFITSIO)
    f, st = C.c_void_p(), C.c_int(0)
    L.ffinit(C.byref(f), path.encode(), C.byref(st))
    naxes = (C.c_long * 2)(img.shape[1], img.shape[0])
    L.ffcrim(f, -32, 2, naxes, C.byref(st))
    flat = np.ascontiguousarray(img, np.float32).reshape(-1)
    L.ffppr(f, 42, C.c_longlong(1), C.c_longlong(flat.size), flat.ctypes.data_as(C.c_void_p), C.byref(st))
    for name, val, comm in keys:
        if isinstance(val, int):
            v = C.c_int(val)
            L.ffuky(f, 31, name.encode(), C.byref(v), comm.encode(), C.byref(st))
        else:
            v = C.c_double(val)
            L.ffuky(f, 82, name.encode(), C.byref(v), comm.encode(), C.byref(st))
    L.ffclos(f, C.byref(st))
    assert st.value == 0
    return open(path, "rb").read()


def params(tmp_path, partinplanes=False):
    return InputParams(npix=16, fov=2.0, simType="Gadget", partinplanes=partinplanes, directory=str(tmp_path) + "/",
                       simulation="sim", suffix="test", snpix="16")


@pytest.mark.parametrize("npix", [8, 27, 64])
def test_all_types_file_is_byte_identical_to_cfitsio(tmp_path, npix):
    rng = np.random.default_rng(npix)
    img = rng.uniform(0, 3, (npix, npix)).astype(np.float32)
    img[0, 1] = 0.0
    p = params(tmp_path)
    p.npix = npix
    lens = Lens(nplanes=1, ld=[123.456789], ld2=[223.4])
    hdr = dict(h=0.6774, om0=0.3089, oml=0.6911, massarr=[0.0, 0.0123456789, 0, 1e-30, 2.5e10, 0.5])
    ntot = [0, 123456, 0, 7, 0, 2147483647]
    out = fits.writeMaps(p, hdr, lens, 0, 0.512345678901234567, "007", "16", img, None, ntot, 0)
    assert out == [fits.fileOutput(p, "007")] and out[0].endswith("sim.007.plane_16_test.fits")
    keys = fits.plane_keys(p, 0.6774, 0.3089, 0.6911, hdr["massarr"], 123.456789, 223.4, 0.512345678901234567, ntot)
    ref = cfitsio_write(str(tmp_path / "ref.fits"), img, keys)
    got = open(out[0], "rb").read()
    assert len(got) % 2880 == 0 and got == ref


def test_double_formatting_edge_cases_match_cfitsio(tmp_path):
    vals = [0.0, 1.0, -1.0, 0.1, 1e-30, -2.5e300, 123456789012345678.0, 1e15, 1e16, 123456.5, 1 / 3, 2 / 3 * 1e-7,
            9.99999999999999e22, 5e-324]
    keys = [("K%d" % i, v, "c") for i, v in enumerate(vals)] + [("LONGKEYNAME%d" % i, v, " ") for i, v in enumerate(vals)]
    img = np.zeros((4, 4), np.float32)
    ref = cfitsio_write(str(tmp_path / "ref.fits"), img, keys)
    fits.write_image(str(tmp_path / "mine.fits"), img, keys)
    assert open(str(tmp_path / "mine.fits"), "rb").read() == ref


def test_partinplanes_writes_only_types_with_particles_and_refuses_overwrite(tmp_path):
    p = params(tmp_path, partinplanes=True)
    lens = Lens(nplanes=1, ld=[3.0], ld2=[4.0])
    hdr = dict(h=0.7, om0=0.3, oml=0.7, massarr=[0.5, 0.0123, 0, 0, 0, 0])
    maps = np.arange(6 * 16 * 16, dtype=np.float32).reshape(6, 16, 16)
    out = fits.writeMaps(p, hdr, lens, 0, 0.3, "012", "16", None, maps, [10, 20, 0, 0, 0, 0], 0)
    assert [os.path.basename(o) for o in out] == ["sim.012.ptype0_plane_16_test.fits", "sim.012.ptype1_plane_16_test.fits"]
    raw = open(out[1], "rb").read()
    assert b"HIERARCH NPARTTYPE0 =       20" in raw and b"M1      =               0.0123" in raw and b"M0 " not in raw
    data = np.frombuffer(raw[2880:2880 + 4 * 256], ">f4").reshape(16, 16)
    assert np.array_equal(data, maps[1])           # row-major, NAXIS1 = fast axis = x = dec
    with pytest.raises(FileExistsError):           # CCfits: FITS::CantCreate -> the reference aborts
        fits.writeMaps(p, hdr, lens, 0, 0.3, "012", "16", None, maps, [10, 20, 0, 0, 0, 0], 0)
    assert fits.writeMaps(p, hdr, lens, 0, 0.3, "013", "16", None, maps, [1] * 6, 1) == []  # only rank 0 writes
    # with the reference's own (always zero) counts no per-type file is ever written (densitymaps.cpp:593)
    assert fits.writeMaps(p, hdr, lens, 0, 0.3, "014", "16", None, maps, [0] * 6, 0) == []


def test_cpp_writeMaps_twin_is_byte_identical_to_python(tmp_path):
    """slicer_amd/csrc/fits_writer.cpp (the reference-signature writeMaps used behind slicer-v2.cpp) vs fits.py."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "tests", "cpp", "fits_driver")
    assert os.path.exists(drv), "run __graft_entry__.build()"
    hdr = dict(h=0.6774, om0=0.3089, oml=0.6911, massarr=[0.0, 0.0123456789, 0, 1e-30, 2.5e10, 0.5])
    ntot = [0, 123456, 0, 7, 0, 2147483647]
    lens = Lens(nplanes=1, ld=[123.456789], ld2=[223.4])
    tot = (np.arange(256, dtype=np.float32) * np.float32(0.37)).reshape(16, 16)
    toti = np.stack([(t * 1000 + np.arange(256)).astype(np.float32).reshape(16, 16) for t in range(6)])
    for pip in (0, 1):
        dc, dp = tmp_path / f"cpp{pip}", tmp_path / f"py{pip}"
        dc.mkdir()
        dp.mkdir()
        assert subprocess.run([drv, str(dc) + "/", str(pip)], capture_output=True).returncode == 0
        p = params(dp, partinplanes=bool(pip))
        out = fits.writeMaps(p, hdr, lens, 0, 0.512345678901234567, "007", "16", tot, toti, ntot, 0)
        assert len(out) == (3 if pip else 1)
        for path in out:
            twin = os.path.join(str(dc), os.path.basename(path))
            assert open(twin, "rb").read() == open(path, "rb").read()
        # second call: file exists -> CantCreate -> non-zero exit, as the reference aborts
        assert subprocess.run([drv, str(dc) + "/", str(pip)], capture_output=True).returncode == 1
