// Micro-benchmark: VALU issue cost per wave64 instruction on gfx950 at the occupancy of k_project_bin
// (512-thread workgroups, 2 per CU => 4 waves per SIMD).  Each kernel runs `iters` x 64 independent ops per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(512, 4) void k(float *out, int iters, float a, double da)
{
    float x[8];
    double d[8];
    int n[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        x[j] = threadIdx.x * 1e-3f + j;
        d[j] = x[j];
        n[j] = threadIdx.x + j;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (OP == 0) x[j] = x[j] * a + 1.0f;                // v_fma_f32 (contracted)
                if (OP == 1) d[j] = fma(d[j], da, 1.0);             // v_fma_f64
                if (OP == 2) d[j] = d[j] * da;                      // v_mul_f64
                if (OP == 3) d[j] = d[j] + da;                      // v_add_f64
                if (OP == 4) {                                       // cvt pair
                    x[j] = (float)((double)x[j]);
                    asm volatile("" : "+v"(x[j]));
                }
                if (OP == 5) n[j] = (n[j] & 0x1fffffff) + 0x10;     // int and + add
                if (OP == 6) x[j] = x[j] > a ? x[j] - 1.0f : x[j];  // cmp + sub + cndmask
                if (OP == 7) d[j] = sqrt(d[j]);                     // f64 sqrt (IEEE)
                if (OP == 8) d[j] = da / d[j];                      // f64 div (IEEE)
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
        s += x[j] + (float)d[j] + n[j];
    if (s == 12345.678f)
        out[0] = s;
}

template <int OP>
void run(const char *name)
{
    float *d;
    (void)hipMalloc(&d, 4);
    const int iters = (OP >= 7) ? 40 : 400, grid = 512;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<OP><<<grid, 512>>>(d, 4, 1.0001f, 1.0000001);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<grid, 512>>>(d, iters, 1.0001f, 1.0000001);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // source ops per SIMD: grid * 8 waves / 1024 SIMDs * iters * 64
    const double wi = (double)grid * 8 / 1024.0 * iters * 64;
    printf("%-28s %8.3f ms   %.2f cycles per source op per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / wi);
    (void)hipFree(d);
}

int main()
{
    run<0>("v_fma_f32");
    run<1>("v_fma_f64");
    run<2>("v_mul_f64");
    run<3>("v_add_f64");
    run<4>("cvt f32->f64->f32 (2 ops)");
    run<5>("v_and + v_add_u32 (2 ops)");
    run<6>("cmp+sub+cndmask (3 ops)");
    run<7>("f64 sqrt");
    run<8>("f64 div");
    return 0;
}
