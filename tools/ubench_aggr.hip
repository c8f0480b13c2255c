// ubench_aggr.hip -- runs an MFMA aggressor (mode 1: compiler MFMA chain, 3: asm MFMA + v_min3 in k_screen_mx's shape) for a
// few seconds, as a separate process beside a victim under test.  hipcc --offload-arch=gfx950 -O3 tools/ubench_aggr.hip -o tools/bin/ubench_aggr
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16f __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256, 2) aggressor(float* out, int iters)
{
    h8v a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(1.0f + (threadIdx.x & 3)); b[k] = (_Float16)0.5f; }
    f16f acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
    const f16f z = acc0;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, z, 0, 0, 0);
        asm volatile("" : "+v"(acc0), "+v"(acc1));
    }
    if (acc0[0] + acc1[3] == 123.0f) out[0] = acc0[0];
}
__global__ void __launch_bounds__(256, 2) aggressor_valu(float* out, int iters)
{
    float x = (float)threadIdx.x, y = x + 1.0f, z = x + 2.0f, w = x + 3.0f;
    for (int i = 0; i < iters * 16; ++i) {
        x = __builtin_fmaf(x, 0.5f, 1.0f); y = __builtin_fmaf(y, 0.5f, 1.0f); z = __builtin_fmaf(z, 0.5f, 1.0f); w = __builtin_fmaf(w, 0.5f, 1.0f);
        asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    }
    if (x + y + z + w == 123.0f) out[0] = x;
}
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256, 2) aggressor_pk(float* out, int iters)     // mode 3: packed FMAs at full rate, no MFMA
{
    v2f x[8];
    for (int k = 0; k < 8; ++k) x[k] = (v2f){(float)threadIdx.x + k, 1.0f + k};
    const v2f a = {0.5f, 0.25f}, b = {1.0f, 2.0f};
    for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = __builtin_elementwise_fma(x[k], a, b);
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    float t = 0; for (int k = 0; k < 8; ++k) t += x[k].x + x[k].y;
    if (t == 123.0f) out[0] = t;
}
int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 20.0;
    float* out; hipMalloc(&out, 4);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        if (argc > 2 && atoi(argv[2]) == 3) hipLaunchKernelGGL(aggressor_pk, dim3(256 * 8), dim3(256), 0, 0, out, 20000);
        else if (argc > 2 && atoi(argv[2]) == 2) hipLaunchKernelGGL(aggressor_valu, dim3(256 * 2), dim3(256), 0, 0, out, 20000);
        else hipLaunchKernelGGL(aggressor, dim3(256 * 2), dim3(256), 0, 0, out, 20000);
        hipDeviceSynchronize(); ++n;
    }
    printf("aggressor: %ld launches\n", n);
    return 0;
}
