// slicer_rand.hip -- glibc's rand() stream continued on the device (shot-noise thinning, InputParams.snopt > 0).
//
// densitymaps.cpp:393 draws one libc rand() per selected entry, in selection order, from the process-global stream
// (seeded by randomizeBox, densitymaps.cpp:187-217).  Drawing ~10^7 deviates per sub-file on the host costs ~100 ms
// (a lock and a function call each) against ~0.3 ms of kernels, so the stream moves to the GPU:
//
//   * glibc's default generator (random_r.c, TYPE_3) is the additive feedback recurrence over Z / 2^32
//         x[n] = x[n - 31] + x[n - 3],   rand() = x[n] >> 1,
//     i.e. the 31-word state advances by a fixed linear map A.  A^k is a 31 x 31 matrix of 32-bit words, so any
//     position of the stream can be reached directly: A^(2^k) (k < 48) serve the jump over a count known only on the
//     device, A^(1024 l) (l < 64) give lane l of a wave its start state from the wave's.
//   * the process-global state is read and written back through initstate() / setstate(), which hand out the state
//     array (type and rear pointer encoded in its first word, random_r.c:__initstate_r / __setstate_r).  The layout
//     knowledge is checked once per process on a private state array (never on the process's own): a failed check, or
//     a generator type other than TYPE_3, leaves thinning on the host loop.
//
//   k_rand_wave_states : one wave; start states of all generating waves (a chain of A^65536 products) and the state
//                        after all `nsel` draws (binary decomposition of nsel) -- nsel is read from device memory
//   k_rand_generate    : lane l of wave w produces deviates [w * 65536 + l * 1024, + 1024) as rand() / float(RAND_MAX)
#include <hip/hip_runtime.h>

#include <array>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "slicer_kernels.hpp"

namespace slicer {

namespace {

constexpr int kDeg = 31;             // DEG_3
constexpr int kSep = 3;              // SEP_3
constexpr int kLaneBlock = 1024;     // deviates per lane
constexpr int kLaneLog2 = 10;
constexpr int kWaveLog2 = kLaneLog2 + 6;  // deviates per wave: 65536

using Mat = std::array<uint32_t, kDeg * kDeg>;

Mat mat_mul(const Mat &a, const Mat &b)
{
    Mat c;
    for (int i = 0; i < kDeg; i++)
        for (int j = 0; j < kDeg; j++) {
            uint32_t s = 0;
            for (int k = 0; k < kDeg; k++)
                s += a[i * kDeg + k] * b[k * kDeg + j];
            c[i * kDeg + j] = s;
        }
    return c;
}

// state vector v[j] = x[n - 31 + j] (oldest first); one draw: v' = (v[1], ..., v[30], v[0] + v[28])
Mat step_matrix()
{
    Mat a{};
    for (int i = 0; i + 1 < kDeg; i++)
        a[i * kDeg + i + 1] = 1;
    a[(kDeg - 1) * kDeg + 0] = 1;
    a[(kDeg - 1) * kDeg + (kDeg - kSep)] = 1;
    return a;
}

struct Tables {
    std::vector<uint32_t> pow2;  // [kRandPow2][31][31]: A^(2^k)
    std::vector<uint32_t> lane;  // [31][31][64]: (A^(1024 l))[r][j] at ((r * 31 + j) * 64 + l)
};

const Tables &tables()
{
    static Tables T;
    static std::once_flag once;
    std::call_once(once, [] {
        T.pow2.resize((size_t)kRandPow2 * kDeg * kDeg);
        Mat p = step_matrix();
        Mat lane_step{};
        for (int k = 0; k < kRandPow2; k++) {
            memcpy(&T.pow2[(size_t)k * kDeg * kDeg], p.data(), sizeof(uint32_t) * kDeg * kDeg);
            if (k == kLaneLog2)
                lane_step = p;
            p = mat_mul(p, p);
        }
        T.lane.resize((size_t)kDeg * kDeg * 64);
        Mat cur{};
        for (int i = 0; i < kDeg; i++)
            cur[i * kDeg + i] = 1;
        for (int l = 0; l < 64; l++) {
            for (int e = 0; e < kDeg * kDeg; e++)
                T.lane[(size_t)e * 64 + l] = cur[e];
            cur = mat_mul(lane_step, cur);
        }
    });
    return T;
}

// ---- the process-global generator state, through initstate / setstate ----
alignas(8) char g_scratch[128];
std::mutex g_libc_mutex;

// reads the state behind `arr` (what initstate / setstate returned) into v[0..30], oldest first
bool decode_state(const int32_t *arr, uint32_t *v)
{
    const int type = arr[0] % 5, rear = arr[0] / 5;
    if (type != 3 || rear < 0 || rear >= kDeg)
        return false;
    const int front = (rear + kSep) % kDeg;  // the slot the next draw overwrites: x[n - 31]
    for (int j = 0; j < kDeg; j++)
        v[j] = (uint32_t)arr[1 + (front + j) % kDeg];
    return true;
}

void encode_state(int32_t *arr, const uint32_t *v)
{
    arr[0] = 5 * 0 + 3;  // rear = 0 => front = 3
    for (int j = 0; j < kDeg; j++)
        arr[1 + (kSep + j) % kDeg] = (int32_t)v[j];
}

uint32_t model_draw(uint32_t *v)
{
    const uint32_t x = v[0] + v[kDeg - kSep];
    for (int j = 0; j + 1 < kDeg; j++)
        v[j] = v[j + 1];
    v[kDeg - 1] = x;
    return x >> 1;
}

// Checks the layout knowledge on a private state array: decode, predict 100 draws, rewind by encode, draw again.
bool self_test_locked()
{
    alignas(8) static char priv[128];
    char *procs = initstate(20240917u, priv, sizeof priv);  // rand() now runs on `priv`; `procs` = the process's own
    bool ok = procs != nullptr;
    if (ok) {
        char *mine = initstate(1u, g_scratch, sizeof g_scratch);  // hands back `priv` with its rear pointer encoded
        ok = mine == priv;
        uint32_t v0[kDeg], v[kDeg];
        ok = ok && decode_state((const int32_t *)priv, v0);
        if (ok) {
            setstate(priv);
            memcpy(v, v0, sizeof v);
            int first[100];
            for (int i = 0; i < 100 && ok; i++) {
                first[i] = rand();
                ok = (uint32_t)first[i] == model_draw(v);
            }
            // rewind through the encoder and draw again
            initstate(1u, g_scratch, sizeof g_scratch);
            encode_state((int32_t *)priv, v0);
            setstate(priv);
            for (int i = 0; i < 100 && ok; i++)
                ok = rand() == first[i];
        }
        setstate(procs);  // the process's own state: untouched by all of the above
    }
    return ok;
}

bool layout_known_locked()
{
    static int known = -1;
    if (known < 0)
        known = self_test_locked() ? 1 : 0;
    return known == 1;
}

}  // namespace

bool libc_rand_grab(uint32_t *v31)
{
    std::lock_guard<std::mutex> g(g_libc_mutex);
    if (!layout_known_locked())
        return false;
    char *procs = initstate(1u, g_scratch, sizeof g_scratch);
    if (!procs)
        return false;
    const bool ok = decode_state((const int32_t *)procs, v31);
    setstate(procs);
    return ok;
}

bool libc_rand_put(const uint32_t *v31)
{
    std::lock_guard<std::mutex> g(g_libc_mutex);
    if (!layout_known_locked())
        return false;
    char *procs = initstate(1u, g_scratch, sizeof g_scratch);
    if (!procs)
        return false;
    uint32_t probe[kDeg];
    const bool ok = decode_state((const int32_t *)procs, probe);  // still a TYPE_3 array
    if (ok)
        encode_state((int32_t *)procs, v31);
    setstate(procs);
    return ok;
}

// n draws of the model recurrence on the host (a handle-private stream without the device path): out[k] as
// rand() / float(RAND_MAX); v31 is advanced
void libc_rand_model_fill(uint32_t *v31, float *out, unsigned long long n)
{
    for (unsigned long long k = 0; k < n; k++)
        out[k] = (float)(int)model_draw(v31) / float(RAND_MAX);
}

size_t rand_tables_bytes() { return ((size_t)kRandPow2 * kDeg * kDeg + (size_t)kDeg * kDeg * 64) * sizeof(uint32_t); }

hipError_t rand_tables_upload(void *d_tables, hipStream_t s)
{
    const Tables &T = tables();
    hipError_t e = hipMemcpyAsync(d_tables, T.pow2.data(), T.pow2.size() * 4, hipMemcpyHostToDevice, s);
    if (e != hipSuccess)
        return e;
    return hipMemcpyAsync((uint32_t *)d_tables + T.pow2.size(), T.lane.data(), T.lane.size() * 4, hipMemcpyHostToDevice, s);
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// y = M x for a 31 x 31 matrix in global memory (row-major), x in LDS; lane r < 31 returns row r
__device__ __forceinline__ uint32_t mat_row_times(const uint32_t *__restrict__ M, const uint32_t *x, unsigned lane)
{
    uint32_t s = 0;
    if (lane < (unsigned)kDeg) {
#pragma unroll
        for (int j = 0; j < kDeg; j++)
            s += M[lane * kDeg + j] * x[j];
    }
    return s;
}

__global__ __launch_bounds__(64) void k_rand_wave_states(const unsigned long long *__restrict__ nsel_ptr,
                                                         uint32_t *__restrict__ state, uint32_t *__restrict__ wave_states,
                                                         const uint32_t *__restrict__ pow2, unsigned long long max_draws)
{
    __shared__ uint32_t s_x[32];
    const unsigned lane = threadIdx.x;
    unsigned long long nsel = *nsel_ptr;
    if (nsel > max_draws)  // the host sized the deviate buffer for max_draws (never exceeded: one draw per entry)
        nsel = max_draws;
    const unsigned long long nw = (nsel + ((1ull << kWaveLog2) - 1)) >> kWaveLog2;
    const uint32_t s0 = lane < (unsigned)kDeg ? state[lane] : 0u;
    // chain of wave start states: S_0 = state, S_(w+1) = A^65536 S_w; the matrix row stays in registers
    uint32_t row[kDeg];
    const uint32_t *Mw = pow2 + (size_t)kWaveLog2 * kDeg * kDeg;
#pragma unroll
    for (int j = 0; j < kDeg; j++)
        row[j] = lane < (unsigned)kDeg ? Mw[lane * kDeg + j] : 0u;
    uint32_t cur = s0;
    for (unsigned long long w = 0; w < nw; w++) {
        if (lane < (unsigned)kDeg) {
            wave_states[w * 32 + lane] = cur;
            s_x[lane] = cur;
        }
        __syncthreads();
        uint32_t nx = 0;
#pragma unroll
        for (int j = 0; j < kDeg; j++)
            nx += row[j] * s_x[j];
        __syncthreads();
        cur = nx;
    }
    // the state after nsel draws: product of the A^(2^k) of nsel's set bits, applied to the start state
    cur = s0;
    for (int k = 0; k < kRandPow2; k++) {
        if (!((nsel >> k) & 1ull))
            continue;  // (uniform)
        if (lane < (unsigned)kDeg)
            s_x[lane] = cur;
        __syncthreads();
        cur = mat_row_times(pow2 + (size_t)k * kDeg * kDeg, s_x, lane);
        __syncthreads();
    }
    if (lane < (unsigned)kDeg)
        state[lane] = cur;
}

__global__ __launch_bounds__(256) void k_rand_generate(const unsigned long long *__restrict__ nsel_ptr,
                                                       const uint32_t *__restrict__ wave_states,
                                                       const uint32_t *__restrict__ lane_tab, float *__restrict__ urand,
                                                       unsigned long long max_draws)
{
    unsigned long long nsel = *nsel_ptr;
    if (nsel > max_draws)
        nsel = max_draws;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long w = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long first = (w << kWaveLog2) + ((unsigned long long)lane << kLaneLog2);
    if ((w << kWaveLog2) >= nsel)
        return;  // (whole wave)
    // start state of this lane: A^(1024 lane) S_w
    uint32_t S[kDeg], v[kDeg];
#pragma unroll
    for (int j = 0; j < kDeg; j++)
        S[j] = wave_states[w * 32 + j];
#pragma unroll
    for (int r = 0; r < kDeg; r++) {
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < kDeg; j++)
            s += lane_tab[(size_t)(r * kDeg + j) * 64 + lane] * S[j];
        v[r] = s;
    }
    if (first >= nsel)
        return;
    const unsigned long long end = first + kLaneBlock < nsel ? first + kLaneBlock : nsel;
    float *out = urand + first;
    const unsigned count = (unsigned)(end - first);
    // in place: before step j slot j holds x[n - 31 + j] (the oldest) and slot (j + 28) % 31 holds x[n + j - 3]
    for (unsigned o = 0; o < count; o += kDeg) {
#pragma unroll
        for (int j = 0; j < kDeg; j++) {
            v[j] += v[(j + kDeg - kSep) % kDeg];
            if (o + j < count)  // rand() / float(RAND_MAX): the int rounds to f32, RAND_MAX rounds to 2^31
                out[o + j] = __uint2float_rn(v[j] >> 1) * 4.656612873077392578125e-10f;
        }
    }
}

hipError_t launch_rand_deviates(const unsigned long long *d_nsel, uint32_t *d_state, uint32_t *d_wave_states,
                                const void *d_tables, float *d_urand, unsigned long long max_draws, hipStream_t s)
{
    const uint32_t *pow2 = (const uint32_t *)d_tables;
    const uint32_t *lane_tab = pow2 + (size_t)kRandPow2 * kDeg * kDeg;
    k_rand_wave_states<<<1, 64, 0, s>>>(d_nsel, d_state, d_wave_states, pow2, max_draws);
    const unsigned long long nw = (max_draws + ((1ull << kWaveLog2) - 1)) >> kWaveLog2;
    if (nw) {
        const unsigned blocks = (unsigned)((nw + 3) / 4);
        k_rand_generate<<<blocks, 256, 0, s>>>(d_nsel, d_wave_states, lane_tab, d_urand, max_draws);
    }
    return hipGetLastError();
}

size_t rand_wave_states_bytes(unsigned long long max_draws)
{
    return (size_t)(((max_draws + ((1ull << kWaveLog2) - 1)) >> kWaveLog2) + 1) * 32 * sizeof(uint32_t);
}

}  // namespace slicer
