"""The C-ABI library loads, exports every symbol include/slicer_amd.h declares, and its PODs have the
layout the ctypes mirror assumes.  No compute calls: runs without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

from slicer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "slicer_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(slicer_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in slicer_amd.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype"
    assert sorted(_lib.SYMBOLS) == names


def test_rccl_library_exports_its_header():
    """include/slicer_amd_rccl.h <-> libslicer_amd_rccl.so (load + symbols only; no communicator is created)."""
    from slicer_amd import rccl
    src = open(os.path.join(ROOT, "include", "slicer_amd_rccl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(slicer_rccl_[a-z_0-9]+)\s*\(", src)))
    lib = rccl.load()
    assert names == sorted(rccl.SYMBOLS)
    for n in names:
        assert hasattr(lib, n)


def test_version_and_null_handle_errors():
    lib = _lib.load()
    assert lib.slicer_version() == 200
    assert lib.slicer_plane_begin(None, None) == 2  # SLICER_ERR_ARG, no crash
    assert lib.slicer_file_end(None) == 2
    assert b"null" in lib.slicer_last_error(None)


def test_struct_layout_matches_header(tmp_path):
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "slicer_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(slicer_plane_desc), offsetof(slicer_plane_desc, fov_rad),
         offsetof(slicer_plane_desc, ld2), offsetof(slicer_plane_desc, nrepperp), offsetof(slicer_plane_desc, fixed_frac_bits),
         offsetof(slicer_plane_desc, want_type_maps));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(slicer_file_desc), offsetof(slicer_file_desc, massarr),
         offsetof(slicer_file_desc, boxsize), offsetof(slicer_file_desc, sgn), offsetof(slicer_file_desc, center),
         offsetof(slicer_file_desc, rcase));
  printf("%zu %zu\n", sizeof(slicer_kernel_time), offsetof(slicer_kernel_time, total_ms));
  return 0; }''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    P, F, K = _lib.PlaneDesc, _lib.FileDesc, _lib.KernelTime
    assert [int(x) for x in out[0].split()] == [C.sizeof(P), P.fov_rad.offset, P.ld2.offset, P.nrepperp.offset,
                                                P.fixed_frac_bits.offset, P.want_type_maps.offset]
    assert [int(x) for x in out[1].split()] == [C.sizeof(F), F.massarr.offset, F.boxsize.offset, F.sgn.offset,
                                                F.center.offset, F.rcase.offset]
    assert [int(x) for x in out[2].split()] == [C.sizeof(K), K.total_ms.offset]


def test_no_gpu_fails_loudly_not_silently():
    """Without a device the product refuses to run (no CPU fallback)."""
    import slicer_amd
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.slicer_create(0, 1 << 20, C.byref(h))
    if rc == 0:  # we are on a GPU box
        lib.slicer_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == 7 and b"HIP device" in lib.slicer_last_error(None)
    with pytest.raises(slicer_amd.SlicerError):
        slicer_amd.Slicer(0)


def test_product_does_not_touch_the_oracle():
    """The product package must never import / link / load anything under oracle/."""
    pkg = os.path.join(ROOT, "slicer_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "slicer_oracle" not in txt, f
    out = subprocess.check_output(["ldd", _lib.LIB_PATH]).decode()
    assert "oracle" not in out
