"""slicer_amd -- MI355X-native particle->grid mass assignment (the SLICER densitymaps hot path).

Importing loads slicer_amd/libslicer_amd.so (HIP kernels + C ABI, include/slicer_amd.h) and fails
loudly if it is missing: there is no CPU fallback in this package.
"""
from . import _lib, gadget, synth  # noqa: F401
from .api import (ACC_F32, ACC_F64, ACC_FIXED64, ALGO_AUTO, ALGO_BINNED, ALGO_DIRECT, ELEM_F32, ELEM_F64,  # noqa: F401
                  ELEM_FIXED64, MAS_NGP, MAS_TSC, InputParams, Lens, Random, Slicer, SlicerError, createDensityMaps)

__version__ = "0.2.0"
