"""ctypes binding of libslicer_amd_rccl.so (include/slicer_amd_rccl.h): the per-plane rank sum over RCCL
for C/C++ hosts that do not use torch.distributed (slicer-v2.cpp:214-217 replacement)."""
import ctypes as C
import os

from . import _lib

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libslicer_amd_rccl.so")
ID_BYTES = 128
_H = C.c_void_p
SYMBOLS = {
    "slicer_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "slicer_rccl_comm_init_rank": (C.c_int, [C.POINTER(_H), C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "slicer_rccl_comm_init_all": (C.c_int, [C.POINTER(_H), C.c_int, C.POINTER(C.c_int)]),
    "slicer_rccl_comm_destroy": (C.c_int, [_H]),
    "slicer_rccl_last_error": (C.c_char_p, []),
    "slicer_rccl_plane_reduce": (C.c_int, [_H, _H, C.c_int, C.c_int]),
    "slicer_rccl_plane_reduce_ex": (C.c_int, [_H, _H, C.c_int, C.c_int]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
    _lib.load()  # libslicer_amd.so first (rpath $ORIGIN also finds it)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib
