// slicer_kernels.hip -- gfx950 (CDNA4) kernels of the mass-assignment path.
//
// Reference loops replaced (there are no reference kernels; these are CPU loops):
//   gadget2io.cpp:195-274   readPos transform          -> transform()          [slicer_device.hpp]
//   densitymaps.cpp:355-401 slab select + projection   -> project()
//   utilities.cpp:66-95     TSC / NGP scatter          -> deposit_*()
//   densitymaps.cpp:511-513 per-file map sums          -> k_fold_ngp / k_finalize_tsc
#include "slicer_kernels.hpp"

#pragma clang fp contract(off)

namespace slicer {

// ---------------------------------------------------------------------------------------------
// deposit of one selected entry into a global accumulator map
// ---------------------------------------------------------------------------------------------
template <int ACC>
__device__ __forceinline__ void add_global(void *map, size_t idx, float c, const PassParams &P)
{
    if (ACC == kF32) {
        atomicAdd(reinterpret_cast<float *>(map) + idx, c);  // global_atomic_add_f32, no CAS loop
    } else if (ACC == kF64) {
        atomicAdd(reinterpret_cast<double *>(map) + idx, (double)c);
    } else if (ACC == kFixed64) {
        long long v = __double2ll_rn((double)c * P.fixed_scale);
        atomicAdd(reinterpret_cast<unsigned long long *>(map) + idx, (unsigned long long)v);
    }
}

template <int MAS, int ACC, bool POW2>
__device__ __forceinline__ void deposit_global(void *map, float xs, float ys, float m, float sm, const PassParams &P)
{
    const int nn = P.nn;
    int gx = grid_index<POW2>(xs, P);
    int gy = grid_index<POW2>(ys, P);
    if (MAS == kNGP) {
        if (gx >= 0 && gx < nn && gy >= 0 && gy < nn) {
            size_t idx = (size_t)gx + (size_t)nn * (size_t)gy;
            if (ACC == kCountU32)
                atomicAdd(reinterpret_cast<unsigned *>(map) + idx, 1u);
            else
                atomicAdd(reinterpret_cast<float *>(map) + idx, m);
        }
    } else {
        float wx[3], wy[3];
        tsc_axis<POW2>(xs, gx, P, wx);
        tsc_axis<POW2>(ys, gy, P, wy);
#pragma unroll
        for (int a = 0; a < 3; a++) {
            wx[a] = sm * wx[a];  // wfx = sqrt(w) * weight(...)   utilities.cpp:88
            wy[a] = sm * wy[a];
        }
#pragma unroll
        for (int b = 0; b < 3; b++) {
            int py = gy + b - 1;
            if (py < 0 || py >= nn)
                continue;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                int px = gx + a - 1;
                if (px < 0 || px >= nn)
                    continue;
                add_global<ACC>(map, (size_t)px + (size_t)nn * (size_t)py, wx[a] * wy[b], P);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SLICER_ALGO_DIRECT: one thread per particle, everything fused, global atomics
// ---------------------------------------------------------------------------------------------
template <int MAS, int ACC, bool POW2, bool HAS_MASS>
__global__ __launch_bounds__(256) void k_direct(const float *__restrict__ pos, const float *__restrict__ mass,
                                                uint64_t n, PassParams P, Targets T)
{
    __shared__ unsigned s_cnt[kMaxPlanes];
    __shared__ int s_neg;
    if (threadIdx.x < kMaxPlanes)
        s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        s_neg = 0;
    __syncthreads();

    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool neg = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float rx = pos[3 * i + 0], ry = pos[3 * i + 1], rz = pos[3 * i + 2];
        float x, y, z;
        transform(rx, ry, rz, P, x, y, z);
        neg |= (x < 0.0f) | (y < 0.0f) | (z < 0.0f);  // densitymaps.cpp:334
        float m = P.mconst, sm = P.sm_const;
        if (HAS_MASS) {
            m = cap_mass(mass[i]);
            sm = __fsqrt_rn(m);
        }
        for (int p = 0; p < P.n_planes; p++) {
            if (!(z >= P.zlo[p] && z < P.zhi[p]))
                continue;
            const int nr = P.nrep[p];
            for (int ni = -nr; ni <= nr; ni++)
                for (int nj = -nr; nj <= nr; nj++) {
                    float xs, ys;
                    if (!project(x, y, z, ni, nj, P, xs, ys))
                        continue;
                    atomicAdd(&s_cnt[p], 1u);
                    deposit_global<MAS, ACC, POW2>(T.acc[p], xs, ys, m, sm, P);
                }
        }
    }
    if (neg)
        s_neg = 1;
    __syncthreads();
    if (threadIdx.x < P.n_planes && s_cnt[threadIdx.x])
        atomicAdd(T.nsel[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    if (threadIdx.x == 0 && s_neg)
        atomicOr(T.neg_flag, 1);
}

// ---------------------------------------------------------------------------------------------
// Shot-noise thinning (InputParams.snopt > 0, densitymaps.cpp:387-397): every selected entry, in the reference's
// order (particle-major, then the (ni, nj) replicas), consumes one libc rand(); it keeps mass 2^snopt * m when
// rand()/float(RAND_MAX) < 1/2^snopt and gets mass 0 otherwise.  The deviates come from the process-global stream the
// reference uses -- continued on the device (slicer_rand.hip) or drawn on the host by the C ABI -- as an array in
// selection order, so the device needs each entry's ordinal:
//   k_thin_count  : selected entries per 64-particle chunk (lane l of a chunk = particle 64*c + l)
//   k_thin_scan   : exclusive prefix over chunks (single workgroup)
//   k_thin_deposit: recomputes the selection, rank = base[chunk] + lanes before + replica index, then deposits
// Two full projections per particle and global atomics for the kept entries.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_prefix(unsigned v, unsigned &total)
{
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = (unsigned)__shfl_up((int)x, d);
        if ((int)(threadIdx.x & 63) >= d)
            x += y;
    }
    total = (unsigned)__shfl((int)x, 63);
    return x - v;
}

__global__ __launch_bounds__(256) void k_thin_count(const float *__restrict__ pos, uint64_t n, PassParams P,
                                                    unsigned *__restrict__ counts, int *neg_flag)
{
    const uint64_t nchunks = (n + 63) / 64;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    for (uint64_t c = wave0; c < nchunks; c += nwaves) {
        const uint64_t i = c * 64 + lane;
        unsigned v = 0;
        if (i < n) {
            float x, y, z;
            transform(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], P, x, y, z);
            if ((x < 0.0f) | (y < 0.0f) | (z < 0.0f))
                atomicOr(neg_flag, 1);
            if (z >= P.zlo[0] && z < P.zhi[0]) {
                const int nr = P.nrep[0];
                for (int ni = -nr; ni <= nr; ni++)
                    for (int nj = -nr; nj <= nr; nj++) {
                        float xs, ys;
                        v += project(x, y, z, ni, nj, P, xs, ys) ? 1u : 0u;
                    }
            }
        }
        unsigned tot;
        (void)lane_prefix(v, tot);
        if (lane == 0)
            counts[c] = tot;
    }
}

__global__ __launch_bounds__(1024) void k_thin_scan(const unsigned *__restrict__ counts, unsigned long long *__restrict__ base,
                                                    uint64_t nchunks)
{
    __shared__ unsigned long long s_part[1024];
    const int tid = threadIdx.x;
    const uint64_t per = (nchunks + 1023) / 1024;
    const uint64_t lo = (uint64_t)tid * per, hi = lo + per < nchunks ? lo + per : nchunks;
    unsigned long long sum = 0;
    for (uint64_t i = lo; i < hi; i++)
        sum += counts[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned long long v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = tid ? s_part[tid - 1] : 0;
    for (uint64_t i = lo; i < hi; i++) {
        base[i] = run;
        run += counts[i];
    }
    if (tid == 1023)
        base[nchunks] = s_part[1023];
}

template <int MAS, int ACC, bool POW2, bool HAS_MASS>
__global__ __launch_bounds__(256) void k_thin_deposit(const float *__restrict__ pos, const float *__restrict__ mass,
                                                      uint64_t n, PassParams P, Targets T,
                                                      const unsigned long long *__restrict__ base,
                                                      const float *__restrict__ urand, double thr, double mfac)
{
    const uint64_t nchunks = (n + 63) / 64;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    unsigned long long nsel = 0;
    for (uint64_t c = wave0; c < nchunks; c += nwaves) {
        const uint64_t i = c * 64 + lane;
        float x = 0, y = 0, z = 0;
        bool inslab = false;
        unsigned v = 0;
        if (i < n) {
            transform(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], P, x, y, z);
            inslab = z >= P.zlo[0] && z < P.zhi[0];
        }
        const int nr = P.nrep[0];
        if (inslab)
            for (int ni = -nr; ni <= nr; ni++)
                for (int nj = -nr; nj <= nr; nj++) {
                    float xs, ys;
                    v += project(x, y, z, ni, nj, P, xs, ys) ? 1u : 0u;
                }
        unsigned tot;
        unsigned long long rank = base[c] + lane_prefix(v, tot);
        nsel += v;
        if (!inslab || v == 0)
            continue;
        float m0 = P.mconst;
        if (HAS_MASS)
            m0 = cap_mass(mass[i]);
        for (int ni = -nr; ni <= nr; ni++)
            for (int nj = -nr; nj <= nr; nj++) {
                float xs, ys;
                if (!project(x, y, z, ni, nj, P, xs, ys))
                    continue;
                const float u = urand[rank++];
                if (!((double)u < thr))
                    continue;  // ms = 0: contributes nothing (adds of +0.0f in the reference)
                const float m = (float)(mfac * (double)m0);  // ms.push_back(pow(2, snopt) * num_float1)
                deposit_global<MAS, ACC, POW2>(T.acc[0], xs, ys, m, __fsqrt_rn(m), P);
            }
    }
    // selected-entry counter (all selected entries, kept or not, as totPartxyi counts them)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        nsel += (unsigned long long)__shfl_down((long long)nsel, d);
    if (lane == 0 && nsel)
        atomicAdd(T.nsel[0], nsel);
}

__global__ __launch_bounds__(256) void k_zero_many(ZeroList Z)
{
    const unsigned long long total = Z.quad0[Z.n];
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        int b = 0;
        while (b + 1 < Z.n && q >= Z.quad0[b + 1])
            b++;
        const unsigned long long w = (q - Z.quad0[b]) * 4ull;  // first word of this quad inside buffer b
        unsigned *dst = reinterpret_cast<unsigned *>(Z.p[b]) + w;
        if (w + 4ull <= Z.words[b]) {
            *reinterpret_cast<uint4 *>(dst) = make_uint4(0u, 0u, 0u, 0u);
        } else {
            for (unsigned long long k = w; k < Z.words[b]; k++)
                dst[k - w] = 0u;
        }
    }
}

hipError_t launch_zero_many(const ZeroList &Z, hipStream_t s)
{
    if (Z.n <= 0 || Z.quad0[Z.n] == 0)
        return hipSuccess;
    const unsigned long long total = Z.quad0[Z.n];
    const unsigned long long want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < 256ull * 32ull ? want : 256ull * 32ull);  // grid-stride beyond 32 blocks per CU
    k_zero_many<<<grid, 256, 0, s>>>(Z);
    return hipGetLastError();
}

static inline int grid_for(uint64_t n, int block, int per_thread = 1)
{
    uint64_t g = (n + (uint64_t)block * per_thread - 1) / ((uint64_t)block * per_thread);
    const uint64_t cap = 256ull * 16ull;  // 256 CUs x 16 resident blocks: grid-stride the rest
    if (g > cap)
        g = cap;
    if (g < 1)
        g = 1;
    return (int)g;
}

template <int MAS, int ACC>
static hipError_t launch_direct_2(bool pow2, bool has_mass, const float *pos, const float *mass, uint64_t n,
                                  const PassParams &P, const Targets &T, hipStream_t s)
{
    dim3 grid(grid_for(n, 256)), block(256);
    if (pow2) {
        if (has_mass)
            k_direct<MAS, ACC, true, true><<<grid, block, 0, s>>>(pos, mass, n, P, T);
        else
            k_direct<MAS, ACC, true, false><<<grid, block, 0, s>>>(pos, mass, n, P, T);
    } else {
        if (has_mass)
            k_direct<MAS, ACC, false, true><<<grid, block, 0, s>>>(pos, mass, n, P, T);
        else
            k_direct<MAS, ACC, false, false><<<grid, block, 0, s>>>(pos, mass, n, P, T);
    }
    return hipGetLastError();
}

hipError_t launch_direct(const LaunchCfg &cfg, const float *d_pos, const float *d_mass, uint64_t n,
                         const PassParams &P, const Targets &T, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    const bool pow2 = P.pow2 != 0;
    if (cfg.mas == kNGP) {
        if (cfg.acc == kCountU32)
            return launch_direct_2<kNGP, kCountU32>(pow2, false, d_pos, d_mass, n, P, T, s);
        return launch_direct_2<kNGP, kF32>(pow2, cfg.has_mass, d_pos, d_mass, n, P, T, s);
    }
    switch (cfg.acc) {
    case kF32: return launch_direct_2<kTSC, kF32>(pow2, cfg.has_mass, d_pos, d_mass, n, P, T, s);
    case kF64: return launch_direct_2<kTSC, kF64>(pow2, cfg.has_mass, d_pos, d_mass, n, P, T, s);
    case kFixed64: return launch_direct_2<kTSC, kFixed64>(pow2, cfg.has_mass, d_pos, d_mass, n, P, T, s);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// finalize (TSC): accumulators -> f32 per-type maps and the all-types map
// ---------------------------------------------------------------------------------------------
template <int ACC>
__device__ __forceinline__ double acc_value(const void *a, uint64_t i, double inv_scale)
{
    if (ACC == kF64)
        return reinterpret_cast<const double *>(a)[i];
    return (double)reinterpret_cast<const long long *>(a)[i] * inv_scale;
}

template <int ACC>
__global__ __launch_bounds__(256) void k_finalize_tsc(FinalizeArgs A)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.npix2; i += stride) {
        if (ACC == kF32) {
            if (A.acc_shared) {
                A.tot[i] = reinterpret_cast<const float *>(A.acc_shared)[i];
            } else {
                // mapxytot = m0 + m1 + ... + m5, left-associated f32 (densitymaps.cpp:511)
                float s = 0.0f;
                bool first = true;
#pragma unroll
                for (int t = 0; t < 6; t++) {
                    float v = A.acc[t] ? reinterpret_cast<const float *>(A.acc[t])[i] : 0.0f;
                    s = first ? v : s + v;
                    first = false;
                }
                A.tot[i] = s;
            }
        } else {
            if (A.acc_shared) {
                A.tot[i] = (float)acc_value<ACC>(A.acc_shared, i, A.inv_scale_shared);
            } else {
                double s = 0.0;
#pragma unroll
                for (int t = 0; t < 6; t++) {
                    if (!A.acc[t])
                        continue;
                    double v = acc_value<ACC>(A.acc[t], i, A.inv_scale[t]);
                    if (A.toti[t])
                        A.toti[t][i] = (float)v;
                    s += v;
                }
                A.tot[i] = (float)s;
            }
        }
    }
}

hipError_t launch_finalize_tsc(int acc, const FinalizeArgs &A, hipStream_t s)
{
    dim3 grid(grid_for(A.npix2, 256, 4)), block(256);
    switch (acc) {
    case kF32: k_finalize_tsc<kF32><<<grid, block, 0, s>>>(A); break;
    case kF64: k_finalize_tsc<kF64><<<grid, block, 0, s>>>(A); break;
    case kFixed64: k_finalize_tsc<kFixed64><<<grid, block, 0, s>>>(A); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// NGP per-file fold.  With one constant mass m the reference pixel of a (file, type) map is the
// k-fold sequential f32 sum s <- fl(s + m) (utilities.cpp:75), a function of the count k alone.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float seq_sum(unsigned k, float m)
{
    float s = 0.0f;
    for (unsigned j = 0; j < k; j++)
        s = s + m;
    return s;
}

__global__ __launch_bounds__(256) void k_fold_ngp(FoldArgs A)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.npix2; i += stride) {
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < 6; t++) {
            float v = 0.0f;  // an absent type contributes its zero map
            if (A.mode[t] == 1) {
                unsigned *c = reinterpret_cast<unsigned *>(A.scratch[t]);
                unsigned k = c[i];
                if (k) {
                    v = seq_sum(k, A.mconst[t]);
                    c[i] = 0;
                }
            } else if (A.mode[t] == 2) {
                float *c = reinterpret_cast<float *>(A.scratch[t]);
                v = c[i];
                if (v != 0.0f)
                    c[i] = 0.0f;
            }
            sum = (t == 0) ? v : sum + v;  // ((((m0+m1)+m2)+m3)+m4)+m5   densitymaps.cpp:511
            if (A.mode[t] && A.toti[t])
                A.toti[t][i] = A.toti[t][i] + v;  // densitymaps.cpp:513
        }
        A.tot[i] = A.tot[i] + sum;
    }
}

hipError_t launch_fold_ngp(const FoldArgs &A, hipStream_t s)
{
    dim3 grid(grid_for(A.npix2, 256, 4)), block(256);
    k_fold_ngp<<<grid, block, 0, s>>>(A);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// synthetic boxes (bench / tests): bit-identical to slicer_amd/synth.py
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_synth(float *pos, uint64_t first, uint64_t count, double box,
                                               unsigned long long seed, int clustered)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t total = count * 3;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        uint64_t i = first + e / 3;
        unsigned a = (unsigned)(e % 3);
        double v = (double)(splitmix64(seed, i * 3ull + a) >> 40);
        double u = v * box / 16777216.0;
        if (clustered) {
            unsigned long long sel = splitmix64(seed ^ 0x5E1EC7A11CE5A17Dull, i);
            if (sel >> 63) {
                unsigned long long b = (sel >> 20) & 4095ull;
                double c = (double)(splitmix64(seed ^ 0xB10B5EEDC0FFEE11ull, b * 3ull + a) >> 40) * box / 16777216.0;
                unsigned long long h = splitmix64(seed ^ 0x0FF5E7DEADBEEF01ull, i * 3ull + a);
                unsigned long long sft = (h & 0xFFFFull) + ((h >> 16) & 0xFFFFull) + ((h >> 32) & 0xFFFFull) +
                                         ((h >> 48) & 0xFFFFull);
                double off = ((double)sft - 131070.0) * (0.004 * 1.7320508075688772 / 65536.0) * box;
                double p = c + off;
                if (p < 0.0)
                    p = p + box;
                if (p >= box)
                    p = p - box;
                u = p;
            }
        }
        pos[e] = (float)u;
    }
}

hipError_t launch_synth(float *d_pos, uint64_t first, uint64_t count, double box, uint64_t seed, int clustered,
                        hipStream_t s)
{
    if (count == 0)
        return hipSuccess;
    dim3 grid(grid_for(count * 3, 256)), block(256);
    k_synth<<<grid, block, 0, s>>>(d_pos, first, count, box, seed, clustered);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// debug: A1-A3 only, outputs (xs, ys, plane, source index) of every selected entry
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_debug_project(const float *__restrict__ pos, uint64_t n, PassParams P,
                                                       float *xs_out, float *ys_out, int32_t *plane_out,
                                                       uint64_t *src_out, uint64_t capacity,
                                                       unsigned long long *count, int *neg_flag)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float x, y, z;
        transform(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], P, x, y, z);
        if ((x < 0.0f) | (y < 0.0f) | (z < 0.0f))
            atomicOr(neg_flag, 1);
        for (int p = 0; p < P.n_planes; p++) {
            if (!(z >= P.zlo[p] && z < P.zhi[p]))
                continue;
            const int nr = P.nrep[p];
            int r = 0;
            for (int ni = -nr; ni <= nr; ni++)
                for (int nj = -nr; nj <= nr; nj++, r++) {
                    float xs, ys;
                    if (!project(x, y, z, ni, nj, P, xs, ys))
                        continue;
                    unsigned long long k = atomicAdd(count, 1ull);
                    if (k < capacity) {
                        xs_out[k] = xs;
                        ys_out[k] = ys;
                        plane_out[k] = p | (r << 8);
                        src_out[k] = i;
                    }
                }
        }
    }
}

hipError_t launch_debug_project(const float *d_pos, uint64_t n, const PassParams &P, float *d_xs, float *d_ys,
                                int32_t *d_plane, uint64_t *d_src, uint64_t capacity, unsigned long long *d_count,
                                int *neg_flag, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    dim3 grid(grid_for(n, 256)), block(256);
    k_debug_project<<<grid, block, 0, s>>>(d_pos, n, P, d_xs, d_ys, d_plane, d_src, capacity, d_count, neg_flag);
    return hipGetLastError();
}

// Exhaustive check of quot_dl3 against the IEEE division for one dl: every non-negative f32 below 2 (2^30 operands; the
// negative ones follow by symmetry of every operation involved).  out[0] = mismatches, out[1..8] = examples (f32 bits).
__global__ __launch_bounds__(256) void k_check_dl_quotient(double dl, double inv_dl, unsigned *out)
{
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < 0x40000000ull; u += stride) {
        const double n = (double)__uint_as_float((unsigned)u);
        const double ref = n / dl;
        const double got = quot_dl3(n, dl, inv_dl);
        if (__double_as_longlong(ref) != __double_as_longlong(got)) {
            const unsigned k = atomicAdd(out, 1u);
            if (k < 8)
                out[1 + k] = (unsigned)u;
        }
    }
}

__global__ __launch_bounds__(256) void k_debug_math(int op, const double *__restrict__ a, const double *__restrict__ b,
                                                    double *__restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double r;
    switch (op) {
    case 0: r = sqrt_midrange(a[i]); break;
    case 1: r = div_midrange(a[i], b[i]); break;
    case 2: r = asin_small<15>(a[i]); break;
    case 3: r = atan_small<15>(a[i]); break;
    case 4: r = asin_small<9>(a[i]); break;
    case 5: r = atan_small<9>(a[i]); break;
    case 6: r = __builtin_amdgcn_rsq(a[i]); break;  // the raw hardware estimates ...
    case 7: r = __builtin_amdgcn_rcp(a[i]); break;
    case 8: r = rsqrt_fast(a[i]); break;            // ... and their one-step refinements (k_project_bin_fast)
    case 9: r = rcp_fast(a[i]); break;
    // 10 / 11: the grid arithmetic of a map that is not a power of two wide (npix in the low 20 bits of b): 10 = cell
    // index of the f32 coordinate a; 11 = TSC weight number (int)b >> 20 (0..2) of that coordinate
    default: {
        PassParams P{};
        const int nn = (int)b[i] & 0xFFFFF;
        P.nn = nn;
        P.pow2 = 0;
        P.dl = 1. / double(nn);
        P.nn_d = (double)nn;
        P.half_dl = 0.5 * P.dl;
        P.onehalf_dl = 0.5 * 3.0 * P.dl;
        float lo = (float)P.half_dl;
        P.half_dl_lo = (double)lo > P.half_dl ? __uint_as_float(__float_as_uint(lo) - 1u) : lo;
        lo = (float)P.onehalf_dl;
        P.onehalf_dl_lo = (double)lo > P.onehalf_dl ? __uint_as_float(__float_as_uint(lo) - 1u) : lo;
        P.inv_dl = 1.0 / P.dl;
        P.dl_quot_ok = ((int)b[i] >> 22) & 1;  // bit 22 of b: the three-operation quotient (as after a clean sweep)
        const float v = (float)a[i];
        const int g = grid_index<false>(v, P);
        if (op == 10) {
            r = (double)g;
        } else {
            float w[3];
            tsc_axis<false>(v, g, P, w);
            r = (double)w[(((int)b[i] >> 20) & 3) % 3];
        }
    } break;
    }
    out[i] = r;
}

hipError_t launch_check_dl_quotient(double dl, unsigned *d_out9, hipStream_t s)
{
    k_check_dl_quotient<<<256 * 16, 256, 0, s>>>(dl, 1.0 / dl, d_out9);
    return hipGetLastError();
}

hipError_t launch_debug_math(int op, const double *d_a, const double *d_b, double *d_out, uint64_t n, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    k_debug_math<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(op, d_a, d_b, d_out, n);
    return hipGetLastError();
}

hipError_t launch_thin_count(const float *d_pos, uint64_t n, const PassParams &P, unsigned *counts,
                             unsigned long long *base, int *neg_flag, hipStream_t s)
{
    const uint64_t nchunks = (n + 63) / 64;
    k_thin_count<<<grid_for(n, 256), 256, 0, s>>>(d_pos, n, P, counts, neg_flag);
    k_thin_scan<<<1, 1024, 0, s>>>(counts, base, nchunks);
    return hipGetLastError();
}

template <int MAS, int ACC>
static hipError_t launch_thin_2(bool pow2, bool has_mass, const float *pos, const float *mass, uint64_t n,
                                const PassParams &P, const Targets &T, const unsigned long long *base,
                                const float *urand, double thr, double mfac, hipStream_t s)
{
    dim3 grid(grid_for(n, 256)), block(256);
#define THIN(P2_, HM_) k_thin_deposit<MAS, ACC, P2_, HM_><<<grid, block, 0, s>>>(pos, mass, n, P, T, base, urand, thr, mfac)
    if (pow2) {
        if (has_mass) THIN(true, true); else THIN(true, false);
    } else {
        if (has_mass) THIN(false, true); else THIN(false, false);
    }
#undef THIN
    return hipGetLastError();
}

hipError_t launch_thin_deposit(const LaunchCfg &cfg, const float *d_pos, const float *d_mass, uint64_t n,
                               const PassParams &P, const Targets &T, const unsigned long long *base,
                               const float *urand, double thr, double mfac, hipStream_t s)
{
    const bool pow2 = P.pow2 != 0;
    if (cfg.mas == kNGP) {
        if (cfg.acc == kCountU32)
            return launch_thin_2<kNGP, kCountU32>(pow2, false, d_pos, d_mass, n, P, T, base, urand, thr, mfac, s);
        return launch_thin_2<kNGP, kF32>(pow2, cfg.has_mass, d_pos, d_mass, n, P, T, base, urand, thr, mfac, s);
    }
    switch (cfg.acc) {
    case kF32: return launch_thin_2<kTSC, kF32>(pow2, cfg.has_mass, d_pos, d_mass, n, P, T, base, urand, thr, mfac, s);
    case kF64: return launch_thin_2<kTSC, kF64>(pow2, cfg.has_mass, d_pos, d_mass, n, P, T, base, urand, thr, mfac, s);
    case kFixed64:
        return launch_thin_2<kTSC, kFixed64>(pow2, cfg.has_mass, d_pos, d_mass, n, P, T, base, urand, thr, mfac, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace slicer
