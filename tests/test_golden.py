"""The oracle reproduces the committed golden vectors bit for bit (guards the checker against drift: compiler,
flags, libm).  The vectors are oracle-generated -- see tests/golden/make_golden.py for the provenance statement."""
import numpy as np
import pytest

import golden_util
import oracle


@pytest.mark.parametrize("name", golden_util.names())
def test_oracle_matches_golden(name):
    files, c, tot, toti, nsel = golden_util.load(name)
    rc, t2, ti2, n2 = oracle.create_density_maps(files, 0, len(files), c["npix"], c["hydro"], c["ngp"], c["ld"], c["ld2"],
                                                 c["nrep"], c["fov"], c["rnd"]["sgn"], c["rnd"]["face"],
                                                 c["rnd"]["center"], c["rnd"]["rcase"])
    assert rc == 0 and np.array_equal(n2, nsel)
    assert np.array_equal(t2.view(np.uint32), tot.view(np.uint32))
    assert np.array_equal(ti2.view(np.uint32), toti.view(np.uint32))


def test_golden_set_is_complete():
    assert len(golden_util.names()) >= 6
