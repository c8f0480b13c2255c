// ubench_mfma16b.hip -- second look at "f16 matrix pipe beside the minima" with shader-cycle counts (s_memtime) and a
// hand-ordered software pipeline: [MFMA of tile t+1] then [the 16 v_min3_i32 of tile t].  MI355X_MICROARCH.md: a
// 32x32x16 MFMA occupies its pipe for 32 cycles and holds the SIMD's vector issue for 8 of them; fillers whose issue
// costs fit the rest are nearly free, past that each adds its cost.  So a tile should cost ~max(32, 8 + 16 * 4) = 72.
// Build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form=1] tools/ubench_mfma16b.hip -o tools/bin/ubench_mfma16b
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int min3i(int a, int b, int c) { int m = a < b ? a : b; return m < c ? m : c; }

// MODE 0: MFMA only, two accumulation chains   MODE 1: 16 min3 per "tile" only
// MODE 2: pipeline, MFMA first then the minima of the previous tile   MODE 3: the same with 8 minima per tile
template <int MODE>
__global__ void __launch_bounds__(256) k(int* out, long long* cyc, int iters)
{
    __shared__ h8 s_b[64 * 34];
    for (int i = threadIdx.x; i < 64 * 34; i += 256) {
        h8 t;
        for (int e = 0; e < 8; ++e) t[e] = (_Float16)(0.002f * ((i * 7 + e) % 97));
        s_b[i] = t;
    }
    __syncthreads();
    const int l = threadIdx.x & 63;
    h8 a;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.001f * (l + e));
    f16v d0 = {0}, d1 = {0};
    const f16v zero = {0};
    int cm = 0x7f800000, rm[16], u[16];
    for (int v = 0; v < 16; ++v) { rm[v] = 0x7f800000; u[v] = threadIdx.x * 131 + v; }
    if (MODE >= 2) d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, s_b[l], zero, 0, 0, 0);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int t = (it & 15) * 2;
        if (MODE == 0) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, s_b[t * 64 + l], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, s_b[(t + 1) * 64 + l], d1, 0, 0, 0);
        } else if (MODE == 1) {
            for (int q = 0; q < 2; ++q)
                for (int v = 0; v < 16; ++v)
                    asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(u[v]) : "v"(u[(v + 1) & 15]), "v"(u[(v + 2) & 15]));
        } else {
            const h8 b1 = s_b[(t + 1) * 64 + l], b2 = s_b[(t + 2) * 64 + l];
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, zero, 0, 0, 0);
            for (int v = 0; v < 16; v += 2) cm = min3i(cm, __float_as_int(d0[v]), __float_as_int(d0[v + 1]));
            if (MODE == 2) for (int v = 0; v < 8; ++v) rm[v] = min3i(rm[v], __float_as_int(d0[v]), __float_as_int(d0[v + 8]));
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // the MFMA first
            __builtin_amdgcn_sched_group_barrier(0x002, MODE == 2 ? 16 : 8, 0);   // then the minima of the previous tile
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, zero, 0, 0, 0);
            for (int v = 0; v < 16; v += 2) cm = min3i(cm, __float_as_int(d1[v]), __float_as_int(d1[v + 1]));
            if (MODE == 2) for (int v = 0; v < 8; ++v) rm[v + 8] = min3i(rm[v + 8], __float_as_int(d1[v]), __float_as_int(d1[v + 8]));
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, MODE == 2 ? 16 : 8, 0);
        }
    }
    const long long t1 = clock64();
    unsigned s = (unsigned)cm;
    for (int v = 0; v < 16; ++v) s ^= (unsigned)rm[v] * (2u * v + 3u) ^ (unsigned)u[v] ^ __float_as_uint(d0[v]) ^ __float_as_uint(d1[v]);
    if (s == 123456789u) out[0] = (int)s;
    if (l == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
int run(const char* name, int* dout, long long* dcyc)
{
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    const int iters = 16384;
    for (int wps : {1, 2, 3, 4}) {
        const int grid = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, dout, dcyc, 2048);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(t0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, dout, dcyc, iters);
        CHECK(hipEventRecord(t1));
        CHECK(hipEventSynchronize(t1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, t0, t1));
        std::vector<long long> h(grid * 4);
        CHECK(hipMemcpy(h.data(), dcyc, sizeof(long long) * grid * 4, hipMemcpyDeviceToHost));
        double avg = 0;
        for (long long v : h) avg += (double)v;
        avg /= (double)h.size();
        const double tiles_per_wave = (double)iters * 2.0, tiles = tiles_per_wave * wps;
        printf("%-40s waves/SIMD=%d  %8.3f ms  %6.2f ns/tile/SIMD  %6.1f s_memtime ticks/tile/SIMD  (ticks/ns %.3f)  -> %6.1f Tdist/s per chip\n",
               name, wps, ms, ms * 1e6 / tiles, avg / tiles, avg / (ms * 1e6), tiles * 1024.0 * 1024.0 / (ms * 1e-3) * 1e-12);
    }
    return 0;
}

int main()
{
    int* dout; long long* dcyc;
    CHECK(hipMalloc(&dout, 1024));
    CHECK(hipMalloc(&dcyc, sizeof(long long) * 8192));
    run<0>("mfma 32x32x16 f16 only (2 chains)", dout, dcyc);
    run<1>("16 min3 per tile only", dout, dcyc);
    run<2>("pipeline: mfma, then 16 min3 of tile t-1", dout, dcyc);
    run<3>("pipeline: mfma, then 8 min3 of tile t-1", dout, dcyc);
    CHECK(hipFree(dout)); CHECK(hipFree(dcyc));
    return 0;
}
