// Micro-benchmark: how ds_add_u64 (the tile kernel's LDS atomic) reacts to the way a wave's 64 cells fall on the LDS
// banks.  Tile 66 x 130 cells of 8 bytes as in k_tile_deposit, 1024-thread workgroups, 9 atomics (3 x 3 neighbourhood)
// per lane and iteration.  Patterns: random base cell; lane-private (conflict-free); random but with the base cells
// of every group of G consecutive lanes distinct modulo G (G = 16, 32, 64) -- what an in-wave arrangement of the
// records by bank class could reach at best.   Build: hipcc --offload-arch=gfx950 -O3 tools/lds_bank_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int W = 66, H = 130, CELLS = W * H;

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters)
{
    extern __shared__ unsigned long long tile[];
    for (int i = threadIdx.x; i < CELLS; i += 1024) tile[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const unsigned lane = threadIdx.x & 63u;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        unsigned cx = 1 + ((s >> 8) & 63), cy = 1 + ((s >> 20) & 127);
        unsigned cell = cy * W + cx;
        if (MODE == 1) cell = W + 1 + ((threadIdx.x * 3 + it) % (CELLS - 3 * W));
        if (MODE >= 16) {  // make (cell mod G) == (lane mod G) by moving the column (stays inside the row's 64 + slack)
            constexpr unsigned G = MODE;
            const unsigned d = (cell + G - (lane % G)) % G;
            cell = cell >= d + W + 1 ? cell - d : cell + (G - d);
        }
#pragma unroll
        for (int b = -1; b <= 1; b++)
#pragma unroll
            for (int a = -1; a <= 1; a++) {
                const int idx = (int)cell + b * W + a;
                atomicAdd(&tile[idx < 0 ? 0 : (idx >= CELLS ? CELLS - 1 : idx)], 1ull);
            }
    }
    __syncthreads();
    unsigned long long acc = 0;
    for (int i = threadIdx.x; i < CELLS; i += 1024) acc += tile[i];
    if (acc == 123456789ull) out[blockIdx.x] = acc;
}

template <int MODE>
void run(const char *name)
{
    unsigned long long *d;
    hipMalloc(&d, 8192 * 8);
    const int iters = 200, grid = 2048;
    size_t lds = CELLS * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, 1024, lds>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 1024, lds>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)grid * 1024 * iters * 9;
    printf("%-40s %8.3f ms  %8.2f G lane-ops/s  (%.2f LDS cycles per wave instruction and CU at 2.4 GHz)\n", name, ms,
           ops / ms / 1e6, 64.0 * 2.4e9 * 256 / (ops / (ms * 1e-3)));
    hipFree(d);
}

int main()
{
    run<0>("ds_add_u64 random cells");
    run<1>("ds_add_u64 lane-private cells");
    run<16>("ds_add_u64 distinct mod 16 per 16 lanes");
    run<32>("ds_add_u64 distinct mod 32 per 32 lanes");
    run<64>("ds_add_u64 distinct mod 64 per 64 lanes");
    return 0;
}
