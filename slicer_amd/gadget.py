"""GADGET-2 *format 2* snapshot sub-files: writer for synthetic boxes and a bulk block reader.

Byte layout (what the reference's reader expects: gadget2io.cpp:24-26 readHeader, :133-165
fastforwardToBlock, data.h:59-95 Header/Block):

    [i32 8]["HEAD"][i32 264][i32 8]   [i32 256]{Header, 256 B}[i32 256]
    [i32 8]["POS "][i32 12n+8][i32 8] [i32 12n]{f32 xyz AoS, types 0..5 concatenated}[i32 12n]
    [i32 8]["MASS"][i32 4k+8][i32 8]  [i32 4k]{f32, only types with massarr == 0}[i32 4k]   (optional)

The reference reads the POS block with three 4-byte stream reads per particle
(gadget2io.cpp:200-202); here a block is one numpy.fromfile call.
"""
import os
import struct

import numpy as np

HEADER_DTYPE = np.dtype([
    ("npart", "<i4", 6), ("massarr", "<f8", 6), ("time", "<f8"), ("redshift", "<f8"),
    ("flag_sfr", "<i4"), ("flag_feedback", "<i4"), ("npartTotal", "<u4", 6), ("flag_cooling", "<i4"),
    ("numfiles", "<i4"), ("boxsize", "<f8"), ("om0", "<f8"), ("oml", "<f8"), ("h", "<f8"),
    ("flag_sage", "<i4"), ("flag_metals", "<i4"), ("nTotalHW", "<i4", 6), ("flag_entropy", "<i4"),
    ("la", "<i4", 14),
])  # data.h:59-79; 252 bytes of fields + 4 bytes tail padding = sizeof(Header) = 256
assert HEADER_DTYPE.itemsize == 252


def _block_head(f, name, nbytes):
    f.write(struct.pack("<i4sii", 8, name.encode("ascii"), nbytes + 8, 8))
    f.write(struct.pack("<i", nbytes))


def write_snapshot(path, pos, npart, massarr, boxsize, numfiles=1, mass=None, bhmass=None, redshift=0.0,
                   om0=0.3, oml=0.7, h=0.7, npart_total=None):
    """Write one sub-file.  pos: [sum(npart),3] f32.  mass: f32 array for the MASS block (types with
    massarr==0 in type order, type 5 included) or None.  bhmass: BHMA block for type 5 or None."""
    pos = np.ascontiguousarray(pos, dtype="<f4").reshape(-1, 3)
    assert pos.shape[0] == int(np.sum(npart))
    hdr = np.zeros(1, HEADER_DTYPE)
    hdr["npart"] = npart
    hdr["massarr"] = massarr
    hdr["time"] = 1.0 / (1.0 + redshift)
    hdr["redshift"] = redshift
    hdr["npartTotal"] = npart if npart_total is None else npart_total
    hdr["numfiles"] = numfiles
    hdr["boxsize"] = boxsize
    hdr["om0"], hdr["oml"], hdr["h"] = om0, oml, h
    with open(path, "wb") as f:
        _block_head(f, "HEAD", 256)
        f.write(hdr.tobytes() + b"\0" * 4)
        f.write(struct.pack("<i", 256))
        _block_head(f, "POS ", pos.nbytes)
        pos.tofile(f)
        f.write(struct.pack("<i", pos.nbytes))
        if mass is not None:
            m = np.ascontiguousarray(mass, dtype="<f4")
            _block_head(f, "MASS", m.nbytes)
            m.tofile(f)
            f.write(struct.pack("<i", m.nbytes))
        if bhmass is not None:
            m = np.ascontiguousarray(bhmass, dtype="<f4")
            _block_head(f, "BHMA", m.nbytes)
            m.tofile(f)
            f.write(struct.pack("<i", m.nbytes))


def open_snapshot(file_in):
    """readHeader semantics (gadget2io.cpp:8-31): try `file_in`, then the name without its last two
    characters (".0"); returns (path, header record)."""
    path = file_in
    if not os.path.exists(path):
        path = file_in[:-2]
    if not os.path.exists(path):
        raise FileNotFoundError(f"Error in opening the file: {file_in}!")
    with open(path, "rb") as f:
        f.seek(20)  # int32 blockheader[5]
        raw = f.read(256)
    hdr = np.frombuffer(raw[:252], HEADER_DTYPE, 1)[0]
    return path, hdr


def find_block(f, name, start=20 + 256):
    """fastforwardToBlock (gadget2io.cpp:133-165): scan 24-byte Block records from `start`
    (the trailing size marker of the previous block) until `name`; returns (data offset, nbytes)."""
    f.seek(start)
    while True:
        rec = f.read(24)
        if len(rec) < 24:
            raise EOFError(f"block {name!r} not found")
        _, _, nm, _, _, size2 = struct.unpack("<ii4siii", rec)
        if nm.decode("ascii", "replace") == name:
            return f.tell(), size2
        f.seek(size2, os.SEEK_CUR)


def read_block(path, name, dtype="<f4"):
    with open(path, "rb") as f:
        off, nbytes = find_block(f, name)
        f.seek(off)
        return np.fromfile(f, dtype=dtype, count=nbytes // np.dtype(dtype).itemsize)


def read_positions(path):
    """Whole POS block as [n,3] f32 (raw file units)."""
    return read_block(path, "POS ").reshape(-1, 3)


def read_masses(path, hdr):
    """Per-type mass arrays for types with massarr == 0 (densitymaps.cpp:358-372): types 0..4 stream from
    MASS in type order; type 5 skips its npart[5] MASS entries and streams from BHMA instead."""
    out = {}
    need = [t for t in range(6) if hdr["npart"][t] > 0 and hdr["massarr"][t] == 0]
    if not need:
        return out
    m = read_block(path, "MASS")
    off = 0
    for t in need:
        n = int(hdr["npart"][t])
        if t == 5:
            out[5] = read_block(path, "BHMA")[:n]
        else:
            out[t] = m[off:off + n]
        off += n
    return out
