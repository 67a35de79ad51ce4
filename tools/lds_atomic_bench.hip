// Micro-benchmark: LDS atomic-add throughput on gfx950 for the access shape of the tile-deposit kernel
// (random cells of a 130x130 tile, 512-thread workgroups, 2 per CU).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int MODE>
__global__ __launch_bounds__(512) void k(T *out, int iters)
{
    extern __shared__ unsigned char raw[];
    T *tile = reinterpret_cast<T *>(raw);
    const int cells = 130 * 130;
    for (int i = threadIdx.x; i < cells; i += 512) tile[i] = (T)0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        unsigned cx = 1 + ((s >> 8) & 127), cy = 1 + ((s >> 20) & 127);
        if (MODE == 0) {  // 9 atomics, 3x3 neighbourhood
#pragma unroll
            for (int b = -1; b <= 1; b++)
#pragma unroll
                for (int a = -1; a <= 1; a++) atomicAdd(&tile[(cy + b) * 130 + cx + a], (T)1);
        } else if (MODE == 1) {  // 9 plain stores (race; bandwidth reference)
#pragma unroll
            for (int b = -1; b <= 1; b++)
#pragma unroll
                for (int a = -1; a <= 1; a++) tile[(cy + b) * 130 + cx + a] = (T)it;
        } else if (MODE == 2) {  // 9 atomics to lane-private (conflict-free) cells
#pragma unroll
            for (int j = 0; j < 9; j++) atomicAdd(&tile[(threadIdx.x + 512 * j) % cells], (T)1);
        } else {  // MODE 3 / 4: nine lanes per record (7 records per wave), one atomic per lane and record; row pitch
                  // 130 (MODE 3) or 131 (MODE 4: the three rows of a record fall on disjoint banks)
            const int pitch = MODE == 3 ? 130 : 131;
            const unsigned lane = threadIdx.x & 63u, rec = lane / 9u, cell = lane % 9u;
#pragma unroll
            for (int j = 0; j < 9; j++) {  // same number of lane-ops per iteration as MODE 0 (lane 63 idles)
                unsigned r = (threadIdx.x & ~63u) * 2654435761u + rec * 97u + blockIdx.x * 40503u + it * 9u + j;
                r = r * 1664525u + 1013904223u;
                r ^= r >> 15;
                r *= 2246822519u;
                const unsigned cx = 1 + ((r >> 8) & 127), cy = 1 + ((r >> 20) & 63);
                if (lane < 63)
                    atomicAdd(&tile[(cy + cell / 3 - 1) * pitch + cx + cell % 3 - 1], (T)1);
            }
        }
    }
    __syncthreads();
    T acc = 0;
    for (int i = threadIdx.x; i < cells; i += 512) acc += tile[i];
    if (acc == (T)123456789) out[blockIdx.x] = acc;
}

template <typename T, int MODE>
void run(const char *name)
{
    T *d;
    hipMalloc(&d, 4096 * sizeof(T));
    const int iters = 400, grid = 4096;
    size_t lds = 130 * 130 * sizeof(T);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<T, MODE><<<grid, 512, lds>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T, MODE><<<grid, 512, lds>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)grid * 512 * iters * 9;
    printf("%-28s %8.3f ms  %8.2f G lane-ops/s  (%.2f cyc/lane-op/CU at 2.4 GHz)\n", name, ms, ops / ms / 1e6,
           2.4e9 * 256 / (ops / (ms * 1e-3)));
    hipFree(d);
}

int main()
{
    run<float, 0>("ds_add_f32 3x3 random");
    run<unsigned, 0>("ds_add_u32 3x3 random");
    run<unsigned long long, 0>("ds_add_u64 3x3 random");
    run<double, 0>("ds_add_f64 3x3 random");
    run<float, 1>("ds_write_b32 3x3 random");
    run<float, 2>("ds_add_f32 conflict-free");
    run<unsigned, 2>("ds_add_u32 conflict-free");
    run<unsigned long long, 2>("ds_add_u64 conflict-free");
    run<double, 2>("ds_add_f64 conflict-free");
    run<double, 3>("ds_add_f64 9 lanes/rec p130");
    run<double, 4>("ds_add_f64 9 lanes/rec p131");
    run<unsigned long long, 4>("ds_add_u64 9 lanes/rec p131");
    return 0;
}
