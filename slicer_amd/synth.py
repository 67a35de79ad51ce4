"""Synthetic GADGET-2 particle boxes: counter-based, libm-free.

Every coordinate is a pure function of (seed, global particle index, axis), so a
box can be generated in any chunking, on the host (numpy, here) or on the device
(slicer_synth_positions in csrc/slicer_kernels.hip) with bit-identical results:
only integer ops and exactly-rounded double multiply/add are used.

uniform  : u = (splitmix64(seed, 3*i+axis) >> 40) * box / 2^24
clustered: half the particles as above; the other half sit in one of 4096 blobs,
           centre from the same generator (stream seed ^ BLOB_SALT), offset =
           sigma*box*sqrt(3)*(Irwin-Hall(4) - 2) built from four 16-bit fields.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)
BLOB_SALT = 0xB10B5EEDC0FFEE11
SEL_SALT = 0x5E1EC7A11CE5A17D
OFF_SALT = 0x0FF5E7DEADBEEF01
BASE_SEED = 0x51CE2
N_BLOBS = 4096
SIGMA = 0.004
SQRT3 = 1.7320508075688772


def splitmix64(seed, counter):
    """Stateless splitmix64: the (counter+1)-th output of the stream started at `seed`."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(counter, dtype=np.uint64) + np.uint64(1)) * GOLDEN
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        z = z ^ (z >> np.uint64(31))
    return z


def positions(first, count, box=1000.0, seed=BASE_SEED, clustered=False):
    """Raw POS-block rows [count,3] f32 for global particle indices first..first+count-1."""
    i = np.arange(first, first + count, dtype=np.uint64)
    out = np.empty((count, 3), np.float32)
    scale = np.float64(box)
    for a in range(3):
        v = (splitmix64(seed, i * np.uint64(3) + np.uint64(a)) >> np.uint64(40)).astype(np.float64)
        u = v * scale / 16777216.0
        if clustered:
            sel = splitmix64(np.uint64(seed) ^ np.uint64(SEL_SALT), i)
            in_blob = (sel >> np.uint64(63)) == np.uint64(1)
            b = (sel >> np.uint64(20)) & np.uint64(N_BLOBS - 1)
            c = (splitmix64(np.uint64(seed) ^ np.uint64(BLOB_SALT), b * np.uint64(3) + np.uint64(a))
                 >> np.uint64(40)).astype(np.float64) * scale / 16777216.0
            h = splitmix64(np.uint64(seed) ^ np.uint64(OFF_SALT), i * np.uint64(3) + np.uint64(a))
            s = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
                 + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + ((h >> np.uint64(48)) & np.uint64(0xFFFF)))
            off = (s.astype(np.float64) - 131070.0) * (SIGMA * SQRT3 / 65536.0) * scale
            p = c + off
            p = np.where(p < 0.0, p + scale, p)
            p = np.where(p >= scale, p - scale, p)
            u = np.where(in_blob, p, u)
        out[:, a] = u.astype(np.float32)
    return out
