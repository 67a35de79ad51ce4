// slicer_device.hpp -- device-side arithmetic of the mass-assignment path, gfx950 only.
//
// Every function below reproduces the *type flow* of the reference expression it cites
// (which operations happen in binary32, which in binary64, where values are rounded),
// because the parity bar is bit-exact NGP binning and bit-exact per-particle TSC
// contributions.  This translation unit is compiled with -ffp-contract=off and the
// pragma below: the reference is built for baseline x86-64 (no FMA; CMakeLists.txt:8-14),
// so no product-sum here may be fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace slicer {

constexpr int kMaxPlanes = 8;

// Uniform parameters of one deposit call (one particle type of one sub-file, 1..8 planes).
struct PassParams {
    // --- file / Random entry (gadget2io.cpp:204-270) ---
    double box;      // Header.boxsize
    double inv_box;  // RN(1/box): fast path of the quotient, see div_by_box()
    double c0[3];    // Random.x0,y0,z0
    float sgn[3];    // Random.sgnX,Y,Z as +-1.0f
    int perm[3];     // face permutation: out[a] = wrapped[perm[a]]
    float rcase;
    // --- planes (densitymaps.cpp:346-347,374) ---
    int n_planes;
    float zlo[kMaxPlanes];  // smallest f32 >= minDist  => (double)z >= minDist  <=>  z >= zlo
    float zhi[kMaxPlanes];  // smallest f32 >= maxDist  => (double)z <  maxDist  <=>  z <  zhi
    int nrep[kMaxPlanes];
    // --- projection (densitymaps.cpp:382-386) ---
    double fov;
    double lim;  // fov * (1. + 2. / npix) * 0.5
    // --- grid (utilities.cpp:50,69-70) ---
    int nn;
    int pow2;        // nn is a power of two: x / dl == x * nn exactly
    double dl;       // 1. / double(nn)
    double nn_d;     // (double)nn
    double half_dl;  // 0.5 * dl
    double onehalf_dl;  // 0.5 * 3.0 * dl
    // --- mass (densitymaps.cpp:358-372) ---
    float mconst;    // (float)massarr[t]
    float sm_const;  // sqrtf(mconst), IEEE correctly rounded
    // --- fixed-point accumulation ---
    double fixed_scale;  // 2^k
};

// (float)(sgn * ((double)r / box))      gadget2io.cpp:204-206
// The f64 quotient is only consumed through a rounding to f32, so RN(r * RN(1/box)) (<= 2.5 ulp
// from the correctly rounded quotient) gives the same f32 unless it sits within a few ulp of an
// f32 rounding tie (bit 28 of the f64 mantissa set, bits 27..0 clear) or in the f32 subnormal
// range; those rare cases (p ~ 2^-24) take the exact division.
__device__ __forceinline__ float div_by_box(float r, const PassParams &P)
{
    double q = (double)r * P.inv_box;
    unsigned long long b = (unsigned long long)__double_as_longlong(q);
    unsigned lo = (unsigned)b & 0x1FFFFFFFu;
    unsigned ex = (unsigned)(b >> 52) & 0x7FFu;
    bool risky = ((lo - 0x0FFFFFF0u) <= 0x20u) | (ex < 1023u - 125u);
    if (risky)
        q = (double)r / P.box;
    return (float)q;
}

// gadget2io.cpp:209-220 / 258-269.  The reference evaluates "v - 1." and "1. + v" in double and
// stores to float; for an f32 v both are exactly what the f32 operation returns (v-1 is exact for
// v > 1; 1+v is exact in f64 whenever |v| >= 2^-29 and rounds to 1.0f either way below that).
__device__ __forceinline__ float wrap01(float v)
{
    if (v > 1.0f)
        v = v - 1.0f;
    if (v < 0.0f)
        v = 1.0f + v;
    return v;
}

// A1: raw POS triple -> (x, y, z) in box units, z piled by rcase.   gadget2io.cpp:204-273
__device__ __forceinline__ void transform(float rx, float ry, float rz, const PassParams &P, float &x, float &y,
                                          float &z)
{
    float b[3];
    b[0] = wrap01(P.sgn[0] * div_by_box(rx, P));  // sign flip is exact in any precision
    b[1] = wrap01(P.sgn[1] * div_by_box(ry, P));
    b[2] = wrap01(P.sgn[2] * div_by_box(rz, P));
    // perm is wave-uniform: three selects each
    float v0 = P.perm[0] == 0 ? b[0] : (P.perm[0] == 1 ? b[1] : b[2]);
    float v1 = P.perm[1] == 0 ? b[0] : (P.perm[1] == 1 ? b[1] : b[2]);
    float v2 = P.perm[2] == 0 ? b[0] : (P.perm[2] == 1 ? b[1] : b[2]);
    x = wrap01((float)((double)v0 - P.c0[0]));
    y = wrap01((float)((double)v1 - P.c0[1]));
    z = wrap01((float)((double)v2 - P.c0[2]));
    z = z + P.rcase;
}

// A3: getPolar(radec) + FOV test + map coordinates.   densitymaps.cpp:382-386, utilities.cpp:23-25
// sqrt and the two divisions are IEEE correctly rounded; asin/atan2 are OCML's (<= 1-2 ulp, like
// glibc's): after the rounding to f32 the outputs agree with the CPU except when the f64 value lies
// within ~2 ulp(f64) of an f32 tie (p ~ 1e-8 per coordinate; see DESIGN.md "libm").
__device__ __forceinline__ bool project(float x, float y, float z, int ni, int nj, const PassParams &P, float &xs,
                                        float &ys)
{
    float xf = x + (float)ni;
    float yf = y + (float)nj;
    double X = (double)xf - 0.5;
    double Y = (double)yf - 0.5;
    double Z = (double)z;
    double d = sqrt(X * X + Y * Y + Z * Z);
    double dec = asin(X / d);
    double ra = atan2(Y, Z);
    if (!(fabs(ra) <= P.lim && fabs(dec) <= P.lim))
        return false;  // NaN (d == 0) is rejected, as in the reference
    xs = (float)(dec / P.fov + 0.5);
    ys = (float)(ra / P.fov + 0.5);
    return true;
}

// floor(x / dl) as int.   utilities.cpp:69-70
template <bool POW2>
__device__ __forceinline__ int grid_index(float v, const PassParams &P)
{
    double q = POW2 ? (double)v * P.nn_d : (double)v / P.dl;
    return (int)floor(q);
}

// TSC weights of the three cells g-1, g, g+1 along one axis.   utilities.cpp:4-16, 82-88
template <bool POW2>
__device__ __forceinline__ void tsc_axis(float v, int g, const PassParams &P, float w[3])
{
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int p = g + a - 1;
        float c = (float)(((double)p + 0.5) * P.dl);
        float D = v - c;
        float A = fabsf(D);
        double Ad = (double)A;
        float u = (float)(POW2 ? Ad * P.nn_d : Ad / P.dl);
        float W;
        if (Ad <= P.half_dl) {
            float uu = u * u;
            W = (float)(0.75 - (double)uu);
        } else if (Ad <= P.onehalf_dl) {
            double t = 1.5 - (double)u;
            W = (float)(0.5 * (t * t));
        } else {
            W = 0.0f;
        }
        w[a] = W;
    }
}

// densitymaps.cpp:367-369: masses above MAX_M (1e3) are zeroed
__device__ __forceinline__ float cap_mass(float m) { return m > 1000.0f ? 0.0f : m; }

// ---- synthetic boxes: must match slicer_amd/synth.py bit for bit ----
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long seed, unsigned long long counter)
{
    unsigned long long z = seed + (counter + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace slicer
