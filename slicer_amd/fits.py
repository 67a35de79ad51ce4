"""FITS plane writer with the byte layout of the reference's CCfits/cfitsio output.

Reference: writeMaps / fileOutput, densitymaps.cpp:530-649: one FLOAT_IMG primary HDU of npix x npix (NAXIS1 = x =
dec, the fast axis of the map), then the keys REDSHIFT, PHYSICALSIZE, PIXELUNIT, DlLOW, DlUP, nparttype0..5 (or only
nparttype0 for per-type files), HUBBLE, OMEGAMATTER, OMEGALAMBDA, m0..m5, through CCfits' addKey -> cfitsio
fits_update_key.  The card formatting below reproduces cfitsio 3.47 (the library under CCfits): upper-cased names,
HIERARCH convention for names longer than 8 characters, "%.15G" doubles with a forced decimal point, values
right-justified to column 30; tests/test_fits.py compares whole files byte for byte with files written by
libcfitsio itself (CCfits is not installed, so CCfits-specific behaviour beyond these calls is unpinned).
"""
import os

import numpy as np

_COMMENTS = ("COMMENT   FITS (Flexible Image Transport System) format is defined in 'Astronomy",
             "COMMENT   and Astrophysics', volume 376, page 359; bibcode: 2001A&A...376..359H")


def _fmt_double(v):
    s = "%.15G" % float(v)
    if "." not in s and "N" not in s and "I" not in s:
        s = s.replace("E", ".0E") if "E" in s else s + "."
    return s


def card(name, value, comment=" "):
    name = name.upper()
    val = str(int(value)) if isinstance(value, (int, np.integer)) and not isinstance(value, bool) else _fmt_double(value)
    head = (name.ljust(8) + "= ") if len(name) <= 8 else ("HIERARCH " + name + " = ")
    pad = max(0, 30 - len(head) - len(val))
    c = head + " " * pad + val + " / " + comment
    return c[:80].ljust(80)


def header_bytes(npix, keys):
    cards = ["SIMPLE  =                    T / file does conform to FITS standard",
             "BITPIX  =                  -32 / number of bits per data pixel",
             "NAXIS   =                    2 / number of data axes",
             "NAXIS1  = %20d / length of data axis 1" % npix,
             "NAXIS2  = %20d / length of data axis 2" % npix,
             "EXTEND  =                    T / FITS dataset may contain extensions"]
    cards += list(_COMMENTS)
    cards += [card(*k) for k in keys]
    cards.append("END")
    raw = "".join(c.ljust(80) for c in cards).encode("ascii")
    return raw + b" " * (-len(raw) % 2880)


def write_image(path, image, keys):
    """Refuses to overwrite, like CCfits' FITS(name, FLOAT_IMG, ...) without a leading '!' (FITS::CantCreate)."""
    image = np.asarray(image, dtype=np.float32)
    assert image.ndim == 2 and image.shape[0] == image.shape[1]
    if os.path.exists(path):
        raise FileExistsError(f"It was not possible to create the map: {path}")
    data = image.astype(">f4").tobytes()
    with open(path, "wb") as f:
        f.write(header_bytes(image.shape[0], keys))
        f.write(data + b"\0" * (-len(data) % 2880))


def fileOutput(p, snappl, label=0):
    """densitymaps.cpp:636-649 (Gadget branches)."""
    if p.simType == "Gadget" and not p.partinplanes:
        return f"{p.directory}{p.simulation}.{snappl}.plane_{p.snpix}_{p.suffix}.fits"
    if p.simType == "Gadget" and p.partinplanes:
        return f"{p.directory}{p.simulation}.{snappl}.ptype{int(label)}_plane_{p.snpix}_{p.suffix}.fits"
    raise ValueError("Output name format not recognized")


def plane_keys(p, h, om0, oml, massarr, ld, ld2, zsim, ntotxyi, ptype=None):
    keys = [("REDSHIFT", float(zsim), " "), ("PHYSICALSIZE", float(p.fov), " "),
            ("PIXELUNIT", 1.e+10 / h, "Mass unit in M_Sun"), ("DlLOW", ld / h, "comoving distance in Mpc"),
            ("DlUP", ld2 / h, "comoving distance in Mpc")]
    if ptype is None:
        keys += [(f"nparttype{i}", int(ntotxyi[i]), " ") for i in range(6)]
    else:
        keys += [("nparttype0", int(ntotxyi[ptype]), " ")]
    keys += [("HUBBLE", float(h), " "), ("OMEGAMATTER", float(om0), " "), ("OMEGALAMBDA", float(oml), " ")]
    if ptype is None:
        keys += [(f"m{i}", float(massarr[i]), " ") for i in range(6)]
    else:
        keys += [(f"m{ptype}", float(massarr[ptype]), " ")]
    return keys


def writeMaps(p, data, lens, isnap, zsim, snappl, snpix, mapxytotrecv, mapxytotirecv, ntotxyi, myid):
    """writeMaps (densitymaps.cpp:530-630).  data: header record with h, om0, oml, massarr.  Only rank 0 writes.
    With partinplanes, a type's file is written when ntotxyi[i] > 0 -- pass the true counts
    (createDensityMaps(..., true_counts=True)): with the reference's always-zero counts no file would appear."""
    if myid != 0:
        return []
    h, om0, oml, massarr = float(data["h"]), float(data["om0"]), float(data["oml"]), data["massarr"]
    ld, ld2 = lens.ld[isnap], lens.ld2[isnap]
    written = []
    if not p.partinplanes:
        path = fileOutput(p, snappl)
        write_image(path, mapxytotrecv, plane_keys(p, h, om0, oml, massarr, ld, ld2, zsim, ntotxyi))
        written.append(path)
    else:
        for i in range(6):
            if ntotxyi[i] > 0:
                path = fileOutput(p, snappl, i)
                write_image(path, mapxytotirecv[i], plane_keys(p, h, om0, oml, massarr, ld, ld2, zsim, ntotxyi, i))
                written.append(path)
    return written
