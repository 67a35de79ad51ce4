"""ctypes loader of the product library (HIP kernels + C ABI).  No fallback: if the library
is missing or a symbol is absent, importing slicer_amd fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslicer_amd.so")

MAX_PLANES = 8


class PlaneDesc(C.Structure):
    _fields_ = [("npix", C.c_int32), ("n_planes", C.c_int32), ("mas", C.c_int32), ("accum", C.c_int32),
                ("algo", C.c_int32), ("hydro", C.c_int32), ("snopt", C.c_int32), ("want_type_maps", C.c_int32),
                ("fov_rad", C.c_double), ("ld", C.c_double * MAX_PLANES), ("ld2", C.c_double * MAX_PLANES),
                ("nrepperp", C.c_int32 * MAX_PLANES), ("fixed_frac_bits", C.c_int32), ("debug_flags", C.c_int32)]


class FileDesc(C.Structure):
    _fields_ = [("npart", C.c_int32 * 6), ("massarr", C.c_double * 6), ("boxsize", C.c_double),
                ("sgn", C.c_int32 * 3), ("face", C.c_int32), ("center", C.c_double * 3), ("rcase", C.c_float),
                ("reserved", C.c_int32)]


REDUCE_META_INTS = 24


class ReduceMeta(C.Structure):
    _fields_ = [("v", C.c_int32 * REDUCE_META_INTS)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("total_ms", C.c_double)]


# every symbol include/slicer_amd.h declares: (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "slicer_version": (C.c_int, []),
    "slicer_create": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(_H)]),
    "slicer_destroy": (C.c_int, [_H]),
    "slicer_last_error": (C.c_char_p, [_H]),
    "slicer_set_option": (C.c_int, [_H, C.c_char_p, C.c_int32]),
    "slicer_get_option": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_int32)]),
    "slicer_libc_rand_supported": (C.c_int, []),
    "slicer_libc_rand_state_get": (C.c_int, [C.POINTER(C.c_uint32)]),
    "slicer_libc_rand_state_set": (C.c_int, [C.POINTER(C.c_uint32)]),
    "slicer_rand_stream_set": (C.c_int, [_H, C.POINTER(C.c_uint32)]),
    "slicer_rand_stream_get": (C.c_int, [_H, C.POINTER(C.c_uint32)]),
    "slicer_set_stream": (C.c_int, [_H, C.c_void_p]),
    "slicer_plane_begin": (C.c_int, [_H, C.POINTER(PlaneDesc)]),
    "slicer_file_begin": (C.c_int, [_H, C.POINTER(FileDesc)]),
    "slicer_deposit_host": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "slicer_deposit_stream": (C.c_int, [_H, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]),
    "slicer_deposit_device": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "slicer_file_end": (C.c_int, [_H]),
    "slicer_plane_finalize": (C.c_int, [_H]),
    "slicer_plane_device_maps": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "slicer_plane_read": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "slicer_synchronize": (C.c_int, [_H]),
    "slicer_get_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "slicer_plane_info": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "slicer_plane_device_counts": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p)]),
    "slicer_plane_algo_mask": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "slicer_plane_status": (C.c_int, [_H]),
    "slicer_plane_flush": (C.c_int, [_H]),
    "slicer_reduce_meta_get": (C.c_int, [_H, C.POINTER(ReduceMeta)]),
    "slicer_reduce_meta_set": (C.c_int, [_H, C.POINTER(ReduceMeta)]),
    "slicer_reduce_meta_get_async": (C.c_int, [_H, C.POINTER(ReduceMeta)]),
    "slicer_plane_device_guard": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "slicer_plane_accumulators": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "slicer_device_malloc": (C.c_int, [_H, C.c_size_t, C.POINTER(C.c_void_p)]),
    "slicer_device_free": (C.c_int, [_H, C.c_void_p]),
    "slicer_copy_to_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t]),
    "slicer_copy_to_host": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_size_t]),
    "slicer_synth_positions": (C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_uint64, C.c_int]),
    "slicer_debug_project": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "slicer_debug_box_quotient": (C.c_int, [_H, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "slicer_debug_dl_quotient": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "slicer_debug_math": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "slicer_profile_enable": (C.c_int, [_H, C.c_int]),
    "slicer_profile_reset": (C.c_int, [_H]),
    "slicer_profile_get": (C.c_int, [_H, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C slicer_amd/csrc`). slicer_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    return lib
