// densitymaps_amd.hpp -- the reference's createDensityMaps entry point, backed by the MI355X path.
//
// Signature, argument meaning and return convention are those of SLICER/densitymaps.h:161-165
// (body densitymaps.cpp:419-524; caller slicer-v2.cpp:204-206).  Linking this translation unit
// instead of the reference's definition makes slicer-v2.cpp's plane loop run on the GPU unchanged.
#pragma once
#include <string>
#include <valarray>

#include "slicer_types.hpp"

// Run-time replacements of the reference's compile-time switches and of choices the reference does
// not have.  mas: 0 TSC / 1 NGP (DO_NGP, densitymaps.h:22).  accum / algo: SLICER_ACC_* / SLICER_ALGO_*.
// true_counts: 0 keeps ntotxyi at 0 exactly like the reference (its inner array shadows the
// out-parameter, densitymaps.cpp:497); 1 returns the real selected counts.  device < 0: myid % #GPUs.
extern "C" void slicer_amd_adapter_config(int mas, int accum, int algo, int true_counts, int device);
// on != 0: when InputParams.partinplanes is false, do not build or copy back the six per-type maps -- the reference's
// caller discards them then (writeMaps, densitymaps.cpp:537-584, writes mapxytot only); mapxytoti comes back
// zero-filled.  Halves the device-to-host traffic of a call.  Default off: all seven maps, as the reference fills them.
extern "C" void slicer_amd_adapter_skip_type_maps(int on);
extern "C" void slicer_amd_adapter_shutdown(void);

int createDensityMaps(InputParams &p, Lens &lens, Random &random, int isnap, unsigned int ffmin, unsigned int ffmax,
                      std::string File, double fovradiants, double rcase, gsl_spline *GetDl,
                      gsl_interp_accel *accGetDl, gsl_spline *GetZl, gsl_interp_accel *accGetZl,
                      std::valarray<float> &mapxytot, std::valarray<float> (&mapxytoti)[6], int (&ntotxyi)[6],
                      int myid);
