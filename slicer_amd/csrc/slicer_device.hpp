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

// rare exact paths: out of line by default (inlined, the compiler if-converts them into the hot stream)
#ifndef SLICER_SLOWPATH
#define SLICER_SLOWPATH __attribute__((noinline))
#endif

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
    unsigned pm[3][3];  // the same as bit masks: pm[a][c] = all ones iff perm[a] == c
    float rcase;
    // --- planes (densitymaps.cpp:346-347,374) ---
    int n_planes;
    float zlo[kMaxPlanes];  // smallest f32 >= minDist  => (double)z >= minDist  <=>  z >= zlo
    float zhi[kMaxPlanes];  // smallest f32 >= maxDist  => (double)z <  maxDist  <=>  z <  zhi
    int nrep[kMaxPlanes];
    // window of lateral replicas (ni in [rep_i0, rep_i1], nj in [rep_j0, rep_j1]) this launch of the binned project
    // kernel emits; the host walks the (2n+1)^2 replicas of densitymaps.cpp:377-381 in windows of at most 7 x 7
    int rep_i0, rep_i1, rep_j0, rep_j1;
    // --- projection (densitymaps.cpp:382-386) ---
    double fov;
    double inv_fov;  // RN(1/fov): fast path of the map coordinate, see map_coord_risky()
    double lim;      // fov * (1. + 2. / npix) * 0.5
    float tan_lim_hi, sin2_lim_hi;  // tan(lim) and sin^2(lim), inflated by 1e-5: surely_outside_fov()
    int force_libm;  // debug: bit 0 always use OCML asin/atan2, bit 1 always the 15-term series
    double series_max;  // largest |sin dec|, |tan ra| the series path takes: kSeriesMax9 or kSeriesMax15
    // --- grid (utilities.cpp:50,69-70) ---
    int nn;
    int pow2;        // nn is a power of two: x / dl == x * nn exactly
    double dl;       // 1. / double(nn)
    double nn_d;     // (double)nn
    double half_dl;  // 0.5 * dl
    double onehalf_dl;  // 0.5 * 3.0 * dl
    float dl_f, nn_f, half_dl_f, onehalf_dl_f;  // the same as f32: exact when nn is a power of two (pow2 paths only)
    float half_dl_lo, onehalf_dl_lo;  // largest f32 <= half_dl / onehalf_dl: (double)A <= half_dl  <=>  A <= half_dl_lo
    double inv_dl;   // RN64(1 / dl)
    int dl_quot_ok;  // quot_dl()'s three-operation quotient was proven against the IEEE division for this dl (device
                     // sweep over every f32 operand below 2, k_check_dl_quotient); 0: divide
    // --- mass (densitymaps.cpp:358-372) ---
    float mconst;    // (float)massarr[t]
    float sm_const;  // sqrtf(mconst), IEEE correctly rounded
    // --- fixed-point accumulation ---
    double fixed_scale;  // 2^k
    // --- integer tile cells of the F32 / F64 modes (k_tile_deposit) ---
    double tile_scale;      // 2^(49 - le), le = ilogb(m) + 1: a contribution c adds rint(c * tile_scale) to its cell
    double tile_inv_scale;  // 2^(le - 49)
    float tile_cmin;        // 2^(le - 25): every contribution >= this is an exact multiple of 2^(le - 49)
};

// (float)(sgn * ((double)r / box))      gadget2io.cpp:204-206
// The f64 quotient is only consumed through a rounding to f32, so RN(r * RN(1/box)) (<= 2.5 ulp
// from the correctly rounded quotient) gives the same f32 unless it sits within a few ulp of an
// f32 rounding tie (bit 28 of the f64 mantissa set, bits 27..0 clear) or in the f32 subnormal
// range; those rare cases (p ~ 2^-24) take the exact division.
__device__ __forceinline__ bool box_quotient_risky(double q)
{
    // near an f32 rounding tie (bits 28..0 of the mantissa within 2^4 of 0x10000000) or so small that the f32
    // result is subnormal or zero-ish (|q| < 2^-100, tested on the rounded value): 4 VALU ops
    const unsigned lo = (unsigned)__double_as_longlong(q) & 0x1FFFFFFFu;
    return ((lo - 0x0FFFFFF0u) <= 0x20u) | (fabsf((float)q) < 0x1p-100f);
}

// exact quotient for the rare lanes that need it; out of line (and by value: no address-taken locals) so
// that the IEEE division is really branched around -- inlined, the compiler if-converts it into every
// particle's instruction stream
__device__ SLICER_SLOWPATH double div_by_box_exact(float r, double box, double q)
{
    return box_quotient_risky(q) ? (double)r / box : q;
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
    double qx = (double)rx * P.inv_box, qy = (double)ry * P.inv_box, qz = (double)rz * P.inv_box;
    const int risky = (int)box_quotient_risky(qx) | (int)box_quotient_risky(qy) | (int)box_quotient_risky(qz);
    if (__ballot(risky != 0) != 0ull) {
        qx = div_by_box_exact(rx, P.box, qx);  // wave-uniform branch, p ~ 1e-5 per wave
        qy = div_by_box_exact(ry, P.box, qy);
        qz = div_by_box_exact(rz, P.box, qz);
    }
    unsigned b[3];
    b[0] = __float_as_uint(wrap01(P.sgn[0] * (float)qx));  // sign flip is exact in any precision
    b[1] = __float_as_uint(wrap01(P.sgn[1] * (float)qy));
    b[2] = __float_as_uint(wrap01(P.sgn[2] * (float)qz));
    // face permutation as bit selects with scalar masks: out[a] = b[perm[a]]
    const float v0 = __uint_as_float((b[0] & P.pm[0][0]) | (b[1] & P.pm[0][1]) | (b[2] & P.pm[0][2]));
    const float v1 = __uint_as_float((b[0] & P.pm[1][0]) | (b[1] & P.pm[1][1]) | (b[2] & P.pm[1][2]));
    const float v2 = __uint_as_float((b[0] & P.pm[2][0]) | (b[1] & P.pm[2][1]) | (b[2] & P.pm[2][2]));
    x = wrap01((float)((double)v0 - P.c0[0]));
    y = wrap01((float)((double)v1 - P.c0[1]));
    z = wrap01((float)((double)v2 - P.c0[2]));
    z = z + P.rcase;
}

// ---- small-angle asin / atan ----------------------------------------------------------------
// For |x| <= 0.3125 (fields of view up to ~36 deg) asin and atan are evaluated by their Taylor series
// in the form x + x*z*P(z), z = x^2, 15 terms: truncation < 0.03 ulp, and because the correction term is
// < 3.5 % of the result the Horner rounding stays below 0.02 ulp, so the value is within ~0.52 ulp (asin)
// and ~1.02 ulp (atan, including the rounding of the quotient y/z) of the exact one -- the accuracy class
// of glibc's and OCML's routines, which are used outside that range.  A1..A15 = (2k)!/(4^k k!^2 (2k+1)).
// Both polynomials are evaluated as two interleaved Horner chains in w = z^2 (even and odd coefficients), which
// halves the dependent-FMA depth; the series value is p = pe(w) + z*po(w).
// r * w + c with the coefficient in a scalar register pair: the coefficients of both series would otherwise sit in 52
// VGPRs (and cost a v_mov_b64 per term, v_fmac_f64 being destructive); s_mov of a literal is off the VALU port.
// The literal is materialised by two s_mov inside volatile asm, i.e. next to its use: left to the compiler, the 18
// coefficients of the two series are hoisted out of the particle loop, where they pin 36 SGPRs and push the loop's own
// uniform operands into spill lanes (a v_readlane per use: VALU slots, the kernel's bound).  s_mov runs on the scalar
// port.
constexpr unsigned long long dbits(double d) { return __builtin_bit_cast(unsigned long long, d); }

template <unsigned long long C>
__device__ __forceinline__ double fma_sc(double r, double w)
{
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((unsigned)(C & 0xFFFFFFFFull)));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((unsigned)(C >> 32)));
    const double c = __hiloint2double((int)hi, (int)lo);
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(r), "v"(w), "s"(c));
    return o;
}
#define FMA_SC(r, w, c) fma_sc<dbits(c)>(r, w)

// N = 15 terms serve |x| <= 0.3125; N = 9 terms serve |x| <= 0.155 (fields of view up to ~17 deg) with the same
// truncation bound (A10 z^10 and z^10 / 21 are below 0.03 ulp there) and six fewer dependent fp64 FMAs per series.
constexpr double kSeriesMax15 = 0.3125, kSeriesMax9 = 0.155;

template <int N>
__device__ __forceinline__ double asin_small(double x)
{
    static_assert(N == 9 || N == 15, "term counts with a proven range");
    const double z = x * x, w = z * z;
    // A1..AN split: even-index chain (A1, A3, ..., AN) and odd-index chain (A2, A4, ..., A(N-1))
    double pe, po;
    if (N == 15) {
        pe = FMA_SC(0x1.31683bdef7bdfp-8, w, 0x1.782dda12f684cp-8);  // A15, A13
        po = FMA_SC(0x1.51ba308d3dcb1p-8, w, 0x1.a6863d70a3d71p-8);  // A14, A12
        pe = FMA_SC(pe, w, 0x1.df3bd37a6f4dfp-8);     // A11
        po = FMA_SC(po, w, 0x1.12ef3cf3cf3cfp-7);     // A10
        pe = FMA_SC(pe, w, 0x1.3fde50d79435ep-7);     // A9
        po = FMA_SC(po, w, 0x1.7a87878787878p-7);     // A8
        pe = FMA_SC(pe, w, 0x1.c99999999999ap-7);     // A7
        po = FMA_SC(po, w, 0x1.1c4ec4ec4ec4fp-6);     // A6
    } else {
        pe = FMA_SC(0x1.3fde50d79435ep-7, w, 0x1.c99999999999ap-7);  // A9, A7
        po = FMA_SC(0x1.7a87878787878p-7, w, 0x1.1c4ec4ec4ec4fp-6);  // A8, A6
    }
    pe = FMA_SC(pe, w, 0x1.6e8ba2e8ba2e9p-6);         // A5
    po = FMA_SC(po, w, 0x1.f1c71c71c71c7p-6);         // A4
    pe = FMA_SC(pe, w, 0x1.6db6db6db6db7p-5);         // A3
    po = FMA_SC(po, w, 0x1.3333333333333p-4);         // A2
    pe = FMA_SC(pe, w, 0x1.5555555555555p-3);         // A1
    const double p = fma(po, z, pe);                  // A1 + A2 z + A3 z^2 + ...
    return fma(x * z, p, x);
}

template <int N>
__device__ __forceinline__ double atan_small(double t)
{
    static_assert(N == 9 || N == 15, "term counts with a proven range");
    const double z = t * t, w = z * z;
    double pe, po;
    if (N == 15) {
        pe = FMA_SC(-0x1.0842108421084p-5, w, -0x1.2f684bda12f68p-5);  // -1/31, -1/27
        po = FMA_SC(0x1.1a7b9611a7b96p-5, w, 0x1.47ae147ae147bp-5);    // +1/29, +1/25
        pe = FMA_SC(pe, w, -0x1.642c8590b2164p-5);    // -1/23
        po = FMA_SC(po, w, 0x1.8618618618618p-5);     // +1/21
        pe = FMA_SC(pe, w, -0x1.af286bca1af28p-5);    // -1/19
        po = FMA_SC(po, w, 0x1.e1e1e1e1e1e1ep-5);     // +1/17
        pe = FMA_SC(pe, w, -0x1.1111111111111p-4);    // -1/15
        po = FMA_SC(po, w, 0x1.3b13b13b13b14p-4);     // +1/13
    } else {
        pe = FMA_SC(-0x1.af286bca1af28p-5, w, -0x1.1111111111111p-4);  // -1/19, -1/15
        po = FMA_SC(0x1.e1e1e1e1e1e1ep-5, w, 0x1.3b13b13b13b14p-4);    // +1/17, +1/13
    }
    pe = FMA_SC(pe, w, -0x1.745d1745d1746p-4);        // -1/11
    po = FMA_SC(po, w, 0x1.c71c71c71c71cp-4);         // +1/9
    pe = FMA_SC(pe, w, -0x1.2492492492492p-3);        // -1/7
    po = FMA_SC(po, w, 0x1.999999999999ap-3);         // +1/5
    pe = FMA_SC(pe, w, -0x1.5555555555555p-2);        // -1/3
    const double p = fma(po, z, pe);
    return fma(t * z, p, t);
}

// Two arctangents at once through ONE pass over the coefficients: every coefficient is materialised once (two s_mov)
// and feeds two independent v_fma_f64 -- half the scalar moves and wait states of two separate series, and two
// independent dependency chains per Horner step.  Used by the fast project+bin kernel, which takes BOTH angles as
// arctangents: ra = atan(Y / Z) and dec = atan(X / sqrt(Y^2 + Z^2)) (= asin(X / d), utilities.cpp:23-25, in exact
// arithmetic; its results are only trusted inside the error window of project_emit either way).
template <unsigned long long C>
__device__ __forceinline__ void fma_sc2(double ra, double wa, double rb, double wb, double &oa, double &ob)
{
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((unsigned)(C & 0xFFFFFFFFull)));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((unsigned)(C >> 32)));
    const double c = __hiloint2double((int)hi, (int)lo);
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(oa) : "v"(ra), "v"(wa), "s"(c));
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ob) : "v"(rb), "v"(wb), "s"(c));
}
#define FMA_SC2(ra, wa, rb, wb, c) fma_sc2<dbits(c)>(ra, wa, rb, wb, ra, rb)

template <int N>
__device__ __forceinline__ void atan_small_pair(double ta, double tb, double &outa, double &outb)
{
    static_assert(N == 9 || N == 15, "term counts with a proven range");
    const double za = ta * ta, wa = za * za, zb = tb * tb, wb = zb * zb;
    double pea, poa, peb, pob;
    if (N == 15) {
        pea = peb = -0x1.0842108421084p-5;  // -1/31
        poa = pob = 0x1.1a7b9611a7b96p-5;   // +1/29
        FMA_SC2(pea, wa, peb, wb, -0x1.2f684bda12f68p-5);  // -1/27
        FMA_SC2(poa, wa, pob, wb, 0x1.47ae147ae147bp-5);   // +1/25
        FMA_SC2(pea, wa, peb, wb, -0x1.642c8590b2164p-5);  // -1/23
        FMA_SC2(poa, wa, pob, wb, 0x1.8618618618618p-5);   // +1/21
        FMA_SC2(pea, wa, peb, wb, -0x1.af286bca1af28p-5);  // -1/19
        FMA_SC2(poa, wa, pob, wb, 0x1.e1e1e1e1e1e1ep-5);   // +1/17
        FMA_SC2(pea, wa, peb, wb, -0x1.1111111111111p-4);  // -1/15
        FMA_SC2(poa, wa, pob, wb, 0x1.3b13b13b13b14p-4);   // +1/13
    } else {
        pea = peb = -0x1.af286bca1af28p-5;  // -1/19
        poa = pob = 0x1.e1e1e1e1e1e1ep-5;   // +1/17
        FMA_SC2(pea, wa, peb, wb, -0x1.1111111111111p-4);  // -1/15
        FMA_SC2(poa, wa, pob, wb, 0x1.3b13b13b13b14p-4);   // +1/13
    }
    FMA_SC2(pea, wa, peb, wb, -0x1.745d1745d1746p-4);  // -1/11
    FMA_SC2(poa, wa, pob, wb, 0x1.c71c71c71c71cp-4);   // +1/9
    FMA_SC2(pea, wa, peb, wb, -0x1.2492492492492p-3);  // -1/7
    FMA_SC2(poa, wa, pob, wb, 0x1.999999999999ap-3);   // +1/5
    FMA_SC2(pea, wa, peb, wb, -0x1.5555555555555p-2);  // -1/3
    const double pa = fma(poa, za, pea), pb = fma(pob, zb, peb);
    outa = fma(ta * za, pa, ta);
    outb = fma(tb * zb, pb, tb);
}

// The same for four arguments (two particles' two angles): one pass over the coefficients feeds four independent
// chains -- a quarter of the scalar moves per arctangent, and enough independent fp64 work in ONE wave to cover the
// latency of a dependent v_fma_f64 without help from other waves.
template <unsigned long long C>
__device__ __forceinline__ void fma_sc4(double (&r)[4], const double (&w)[4])
{
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((unsigned)(C & 0xFFFFFFFFull)));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((unsigned)(C >> 32)));
    const double c = __hiloint2double((int)hi, (int)lo);
#pragma unroll
    for (int i = 0; i < 4; i++)
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(w[i]), "s"(c));
}
#define FMA_SC4(r, w, c) fma_sc4<dbits(c)>(r, w)

template <int N>
__device__ __forceinline__ void atan_small_quad(const double (&t)[4], double (&out)[4])
{
    static_assert(N == 9 || N == 15, "term counts with a proven range");
    double z[4], w[4], pe[4], po[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        z[i] = t[i] * t[i];
        w[i] = z[i] * z[i];
    }
    if (N == 15) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            pe[i] = -0x1.0842108421084p-5;  // -1/31
            po[i] = 0x1.1a7b9611a7b96p-5;   // +1/29
        }
        FMA_SC4(pe, w, -0x1.2f684bda12f68p-5);  // -1/27
        FMA_SC4(po, w, 0x1.47ae147ae147bp-5);   // +1/25
        FMA_SC4(pe, w, -0x1.642c8590b2164p-5);  // -1/23
        FMA_SC4(po, w, 0x1.8618618618618p-5);   // +1/21
        FMA_SC4(pe, w, -0x1.af286bca1af28p-5);  // -1/19
        FMA_SC4(po, w, 0x1.e1e1e1e1e1e1ep-5);   // +1/17
        FMA_SC4(pe, w, -0x1.1111111111111p-4);  // -1/15
        FMA_SC4(po, w, 0x1.3b13b13b13b14p-4);   // +1/13
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            pe[i] = -0x1.af286bca1af28p-5;  // -1/19
            po[i] = 0x1.e1e1e1e1e1e1ep-5;   // +1/17
        }
        FMA_SC4(pe, w, -0x1.1111111111111p-4);  // -1/15
        FMA_SC4(po, w, 0x1.3b13b13b13b14p-4);   // +1/13
    }
    FMA_SC4(pe, w, -0x1.745d1745d1746p-4);  // -1/11
    FMA_SC4(po, w, 0x1.c71c71c71c71cp-4);   // +1/9
    FMA_SC4(pe, w, -0x1.2492492492492p-3);  // -1/7
    FMA_SC4(po, w, 0x1.999999999999ap-3);   // +1/5
    FMA_SC4(pe, w, -0x1.5555555555555p-2);  // -1/3
#pragma unroll
    for (int i = 0; i < 4; i++)
        out[i] = fma(t[i] * z[i], fma(po[i], z[i], pe[i]), t[i]);
}

// ---- correctly rounded sqrt and quotient without the range scaling ---------------------------------------------
// The compiler's IEEE sequences (v_rsq_f64 / v_rcp_f64 + FMA iterations) wrap the same iterations in exponent
// scaling (v_div_scale, v_div_fmas, v_div_fixup, v_ldexp + class tests) that only matters for operands near the
// ends of the f64 range.  project() feeds them S in [2^-298, ~1e2] and quotients of f32-derived values, all far
// from those ends, so the unscaled iterations return the same correctly rounded results; operands outside the
// stated ranges take polar_exact().
__device__ __forceinline__ double sqrt_midrange(double S)  // 2^-500 <= S <= 2^500
{
    const double y = __builtin_amdgcn_rsq(S);
    double g = S * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double e = fma(-g, g, S);
    g = fma(e, h, g);
    e = fma(-g, g, S);
    return fma(e, h, g);
}

__device__ __forceinline__ double div_midrange(double n, double d)  // 2^-500 <= |d| <= 2^500, n = 0 or 2^-500 <= |n| <= 2^500
{
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    const double q0 = n * y;
    const double r = fma(-d, q0, n);
    return fma(r, y, q0);
}

// ---- fast (not correctly rounded) reciprocal square root and reciprocal -------------------------------------------
// One cubically convergent step on the hardware estimates (v_rsq_f64 / v_rcp_f64 are good to ~2^-20..2^-23; measured
// by tests/test_gpu_parity.py::test_fast_projection_error_budget through slicer_debug_math ops 6-9):
//   rsqrt: e = 1 - S y^2,  y <- y (1 + e/2 + 3 e^2/8)      error ~ e^3      5 instructions
//   rcp:   e = 1 - d y,    y <- y (1 + e + e^2)            error ~ e^3      3 instructions
// Used by k_project_bin_fast, whose results are only trusted where a 2^-41 window around them decides the rounding.
__device__ __forceinline__ double rsqrt_fast(double S)
{
    const double y = __builtin_amdgcn_rsq(S);
    const double e = fma(-(S * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

__device__ __forceinline__ double rcp_fast(double d)
{
    const double y = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, y, 1.0);
    return fma(y, fma(e, e, e), y);
}

// (float)(ang / fov + 0.5)      densitymaps.cpp:385-386
// s = RN64(ang * RN64(1/fov)) + 0.5 is within 2^-51 (absolute, |ang/fov| < 1) of the reference's f64 value, so its
// rounding to f32 is the reference's unless s sits within 2^-50 of an f32 rounding tie.  The test is made cheap rather
// than tight: for 2^-13 <= s < 2 one ulp(f64) of s is >= 2^-65, so "bits 28..0 of the mantissa within 2^15 of the tie
// pattern" covers 2^-50 everywhere in that range (p = 1.2e-4 per coordinate); values outside the range (the leftmost
// 0.01 % of the map, negative values) are treated as risky too.  Risky lanes take the exact division, out of line and
// behind one wave-uniform branch for both coordinates.
__device__ __forceinline__ bool map_coord_risky(double s)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(s);
    const unsigned hi = (unsigned)(b >> 32), lo = (unsigned)b & 0x1FFFFFFFu;
    const bool near_tie = __usad(lo, 0x10000000u, 0u) <= (1u << 15);           // v_sad_u32
    const bool in_range = (hi - 0x3F200000u) <= (0x3FFFFFFFu - 0x3F200000u);   // 2^-13 <= s < 2
    return near_tie || !in_range;
}

__device__ SLICER_SLOWPATH double map_coord_exact(double ang, double fov, double s)
{
    return map_coord_risky(s) ? ang / fov + 0.5 : s;
}

// A3: getPolar(radec) + FOV test + map coordinates.   densitymaps.cpp:382-386, utilities.cpp:23-25
// sqrt and the divisions are IEEE correctly rounded; asin/atan2 are the small-angle series above or OCML's
// (<= 1-2 ulp, like glibc's): after the rounding to f32 the outputs agree with the CPU except when the f64
// value lies within ~2 ulp(f64) of an f32 tie (p ~ 1e-8 per coordinate; see DESIGN.md "Arithmetic fidelity").
// out-of-range arguments (wide fields of view, z <= 0, NaN): OCML's asin / atan2.  Kept out of line so that
// the register budget of the callers is set by the series path.
struct Polar {
    double dec, ra;
};
__device__ SLICER_SLOWPATH Polar polar_exact(double X, double Y, double Z)
{
    Polar r;
    r.dec = asin(X / sqrt(X * X + Y * Y + Z * Z));
    r.ra = atan2(Y, Z);
    return r;
}

// SERIES: terms of the small-angle series compiled in (9 or 15); 0 = chosen at run time from P.series_max (costs
// registers: both variants are live code).  The hot kernel is instantiated for 9 and 15 and picked at launch.
template <int SERIES = 0>
__device__ __forceinline__ bool project(float x, float y, float z, int ni, int nj, const PassParams &P, float &xs,
                                        float &ys)
{
    float xf = x + (float)ni;
    float yf = y + (float)nj;
    double X = (double)xf - 0.5;
    double Y = (double)yf - 0.5;
    double Z = (double)z;
    // X, Y are 0 or >= 2^-25 in magnitude (differences of an f32 and 0.5); Z > 2^-150 below
    const double S = X * X + Y * Y + Z * Z;
    const bool mid = z > 0.0f && S < 0x1p100;  // false for NaN
    const double d = sqrt_midrange(S);
    const double q = div_midrange(X, d);
    double dec, ra;
    // P.series_max: 0.155 with 9-term series when the field of view (plus the pre-test's margin) stays below it,
    // else 0.3125 with 15 terms; wave-uniform choice
    if (mid && fabs(q) <= P.series_max && fabs(Y) <= P.series_max * Z && !(P.force_libm & 1)) {
        const double t = div_midrange(Y, Z);
        if (SERIES == 9 || (SERIES == 0 && P.series_max < 0.2)) {
            dec = asin_small<9>(q);
            ra = atan_small<9>(t);
        } else {
            dec = asin_small<15>(q);
            ra = atan_small<15>(t);
        }
    } else {
        const Polar pl = polar_exact(X, Y, Z);
        dec = pl.dec;
        ra = pl.ra;
    }
    if (!(fabs(ra) <= P.lim && fabs(dec) <= P.lim))
        return false;  // NaN (d == 0) is rejected, as in the reference
    double sx = dec * P.inv_fov + 0.5, sy = ra * P.inv_fov + 0.5;
    if (__ballot(map_coord_risky(sx) || map_coord_risky(sy)) != 0ull) {
        sx = map_coord_exact(dec, P.fov, sx);
        sy = map_coord_exact(ra, P.fov, sy);
    }
    xs = (float)sx;
    ys = (float)sy;
    return true;
}

// Conservative f32 pre-test of the FOV cut for the unreplicated case: true only if the entry certainly
// fails |ra| <= lim or |dec| <= lim (margins 1e-5 relative + 1e-6 absolute dwarf every f32 rounding here),
// so skipping it cannot change the result.  |ra| > lim <=> |Y| > Z tan(lim);  |dec| > lim <=> X^2 > sin^2(lim) d^2.
__device__ __forceinline__ bool surely_outside_fov(float x, float y, float z, const PassParams &P)
{
    const float X = x - 0.5f, Y = y - 0.5f;
    const float s = X * X + Y * Y + z * z;
    const bool out_ra = fabsf(Y) > z * P.tan_lim_hi + 1e-6f;
    const bool out_dec = X * X > P.sin2_lim_hi * s + 1e-6f;
    return (out_ra | out_dec) & (z > 0.0f) & (P.lim < 1.5);
}

// floor(x / dl) as int.   utilities.cpp:69-70
// POW2 (dl = 2^-k): x / dl = x * 2^k, an exact scaling in f32 as well as in f64, so the f32 product and floorf()
// give the reference's double-precision result without any fp64 instruction.
// RN64(n / dl) for the f32 operands of the grid arithmetic (0 <= n < 2).  With y = RN64(1/dl): q0 = RN(n y), the exact
// residual r = n - dl q0 by FMA, q = RN(q0 + r y) -- Markstein's correction step, which returns the correctly rounded
// quotient for all but special divisors; whether dl is one is not argued but measured: the host enables this path
// for a map size only after k_check_dl_quotient compared it with the IEEE division on every f32 operand below 2.
__device__ __forceinline__ double quot_dl3(double n, double dl, double inv_dl)
{
    const double q0 = n * inv_dl;
    const double r = fma(-dl, q0, n);
    return fma(r, inv_dl, q0);
}

// Any other nn (dl = RN64(1/nn) is not a power of two): the reference divides by dl in f64.  The product t = v * nn is
// exact in f64 (24 + 17 bits), and the correctly rounded quotient RN64(v / dl) lies within one f64 ulp of it (dl is within
// 2^-53 of 1/nn), so the two have the same floor unless t itself is an integer -- then the division decides.
// grid_tie() tells the kernels that keep such entries out of their loops (k_project_bin_fast).
__device__ __forceinline__ bool grid_tie(float v, const PassParams &P)
{
    const double t = (double)v * P.nn_d;
    return t == floor(t);
}

template <bool POW2>
__device__ __forceinline__ int grid_index(float v, const PassParams &P)
{
    if (POW2)
        return (int)floorf(v * P.nn_f);
    if (P.dl_quot_ok)  // (holds for |v| < 2: anything further out is far off the map either way -- but keep it exact)
        return (int)floor(fabsf(v) < 2.0f ? quot_dl3((double)v, P.dl, P.inv_dl) : (double)v / P.dl);
    const double t = (double)v * P.nn_d, f = floor(t);
    if (t == f)  // rare (v = k / nn exactly)
        return (int)floor((double)v / P.dl);
    return (int)f;
}

// TSC weights of the three cells g-1, g, g+1 along one axis.   utilities.cpp:4-16, 82-88
// General case: the reference's mixed f32/f64 sequence, operation by operation.
// POW2 (dl = 2^-k): every fp64 step of that sequence is exact or rounds once to f32, so the same values come out of
// f32 instructions alone (fp64 VALU ops issue at half the f32 rate and the conversions are extra instructions):
//   c  = (float)((p + 0.5) * dl)     (p + 0.5) * 2^-k is exact                 = ((float)p + 0.5f) * dl, c(g +- 1) = c(g) +- dl
//   u  = (float)((double)A / dl)     exact scaling                             = A * nn
//   A <= 0.5 dl, A <= 1.5 dl         both thresholds are f32 values            = f32 compares (decided statically, below)
//   (float)(0.75 - (double)(u*u))    the f64 difference is exact (u*u < 2^-2 has 24 bits; below 2^-50 both give 0.75f)
//                                                                              = 0.75f - u*u
//   t = 1.5 - (double)u              exact, and a multiple of 2^-24 <= 1: an f32  = 1.5f - u
//   (float)(0.5 * (t * t))           t*t exact in f64, one rounding to f32; halving commutes with it  = 0.5f * (t * t)
template <bool POW2>
__device__ __forceinline__ void tsc_axis(float v, int g, const PassParams &P, float w[3])
{
    if (POW2) {
        // g = floor(v * nn) for this same v, so |v - c| <= 0.5 dl at the centre cell and 0.5 dl <= |v - c| <= 1.5 dl at
        // its neighbours (rounding is monotonic and the bounds are f32 values): the reference's branch is known, and
        // at |v - c| = 0.5 dl, where it would take the other one, both formulas give exactly 0.5.
        const float cg = ((float)g + 0.5f) * P.dl_f;
        const float u0 = fabsf(v - (cg - P.dl_f)) * P.nn_f;
        const float u1 = fabsf(v - cg) * P.nn_f;
        const float u2 = fabsf(v - (cg + P.dl_f)) * P.nn_f;
        const float t0 = 1.5f - u0, t2 = 1.5f - u2;
        w[0] = 0.5f * (t0 * t0);
        w[1] = 0.75f - u1 * u1;
        w[2] = 0.5f * (t2 * t2);
        return;
    }
    // Any other nn: the reference's sequence with every step that is provably exact moved to f32 --
    //   (double)p + 0.5 = (double)g + (a - 0.5), exact;  c = (float)(that * dl) as written
    //   (double)A <= half_dl  <=>  A <= RD32(half_dl): A is an f32 value (thresholds prepared by the host)
    //   u = (float)((double)A / dl): exact product with nn unless on an f32 midpoint (see below)
    //   (float)(0.75 - (double)(u*u)), 1.5 - (double)u, (float)(0.5 * (t*t)): as in the power-of-two case above -- those
    //   arguments use only that u*u and (for u >= 0.5) 1.5 - u are f32 values, not the map size
    const double gd = (double)g;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float c = (float)((gd + ((double)a - 0.5)) * P.dl);
        const float A = fabsf(v - c);
        // u = (float)(Ad / dl): the exact product q = A * nn (41 bits) and RN64(Ad / dl), one f64 ulp apart at most, round
        // to the same f32 unless q sits exactly on the midpoint of two f32 values (its 29 low mantissa bits are 100..0):
        // only then the division is carried out
        float u;
        if (P.dl_quot_ok && A < 2.0f) {  // wave-uniform in practice: three operations instead of the division
            u = (float)quot_dl3((double)A, P.dl, P.inv_dl);
        } else {
            const double q = (double)A * P.nn_d;
            u = (float)q;
            if (((unsigned)__double_as_longlong(q) & 0x1FFFFFFFu) == 0x10000000u)
                u = (float)((double)A / P.dl);
        }
        const float t = 1.5f - u;
        const float w_in = 0.75f - u * u, w_out = 0.5f * (t * t);
        w[a] = A <= P.half_dl_lo ? w_in : (A <= P.onehalf_dl_lo ? w_out : 0.0f);
    }
}

// rint(w * scale) for 0 <= w * scale < 2^51 (scale a power of two) as an integer, with one fp64 FMA: the low 52 bits
// of w * scale + 2^52 (the product is exact, so the FMA rounds exactly once, to nearest even -- what __double2ll_rn
// of the product returns, without its 64-bit conversion sequence)
__device__ __forceinline__ unsigned long long rn_scaled_u64(float w, double scale)
{
    return (unsigned long long)__double_as_longlong(fma((double)w, scale, 0x1p52)) & 0xFFFFFFFFFFFFFull;
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
