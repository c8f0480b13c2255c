// ubench_valu.hip -- issue-rate microbenchmark for the VALU instructions the Hausdorff
// kernels are built from (gfx950).  Prints instructions/clk/CU-equivalent rates so the
// roofline in DESIGN.md is priced against measured, not assumed, throughput.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float a[8];
    v2f p[8];
    double d[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 1.f}; d[i] = a[i]; u[i] = threadIdx.x + i;
    }
    const float c = 1.0001f, e = 1e-4f;
    const v2f pc = {c, c}, pe = {e, e};
    const double dc = 1.0001, de = 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pe));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pe));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (MODE == 4) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (MODE == 5) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (MODE == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(de));
                if (MODE == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
                if (MODE == 8) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (MODE == 9) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dc), "v"(de));
                if (MODE == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
                if (MODE == 11) asm volatile("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i] + (float)u[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
int run(const char* name, float* dout, double flop_per_lane_instr)
{
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks/CU = wps (256 threads = 1 wave/SIMD)
        int grid = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, dout, 64);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(t0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, dout, iters);
        CHECK(hipEventRecord(t1));
        CHECK(hipEventSynchronize(t1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, t0, t1));
        double instr = (double)grid * 4 /*waves*/ * iters * 32.0;  // wave-instructions
        double per_s = instr / (ms * 1e-3);
        // wave-instructions per cycle per SIMD at 2.4 GHz nominal
        double per_clk_simd = per_s / (256.0 * 4.0) / 2.4e9;
        printf("%-16s waves/SIMD=%d  %8.3f ms  %7.2f Gwave-instr/s  %.3f instr/clk/SIMD(@2.4GHz)  %7.1f TFLOP/s-equiv\n",
               name, wps, ms, per_s * 1e-9, per_clk_simd, per_s * 64.0 * flop_per_lane_instr * 1e-12);
    }
    return 0;
}

int main()
{
    float* dout;
    CHECK(hipMalloc(&dout, 1024));
    run<0>("v_fma_f32", dout, 2);
    run<10>("v_add_f32", dout, 1);
    run<1>("v_pk_fma_f32", dout, 4);
    run<2>("v_pk_add_f32", dout, 2);
    run<3>("v_pk_mul_f32", dout, 2);
    run<4>("v_min3_u32", dout, 2);
    run<5>("v_min_u32", dout, 1);
    run<11>("v_min_u32_dpp", dout, 1);
    run<6>("v_add_f64", dout, 1);
    run<7>("v_mul_f64", dout, 1);
    run<8>("v_min_f64", dout, 1);
    run<9>("v_fma_f64", dout, 2);
    CHECK(hipFree(dout));
    return 0;
}
