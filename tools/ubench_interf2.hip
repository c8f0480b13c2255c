// ubench_interf2.hip -- what does a k_screen_mx launch disturb in OTHER kernels on the chip?  (Round 4: the packed-FMA
// screens returned different values beside it.)  A host thread calls mm_best_rotation (MM_PRECISION_F32_MATRIX, or FAST as
// the control) in a loop through libmm_hausdorff.so; meanwhile this program launches victims that hold a known pattern in
// LDS / in VGPRs / stream it through global memory for a few hundred microseconds and count what changed.
// Build: hipcc --offload-arch=gfx950 -O3 -Iinclude tools/ubench_interf2.hip -o tools/bin/ubench_interf2 -ldl -pthread
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <thread>
#include <vector>
#include "mm_hausdorff.h"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256, 3) victim_lds(unsigned long long* bad, int spins, int lds_words)
{
    extern __shared__ unsigned int s[];
    const int tid = threadIdx.x;
    for (int i = tid; i < lds_words; i += 256) s[i] = 0xA5000000u ^ (unsigned)i ^ (blockIdx.x << 12);
    __syncthreads();
    unsigned long long nb = 0;
    for (int k = 0; k < spins; ++k) {
        for (int i = tid; i < lds_words; i += 256) if (s[i] != (0xA5000000u ^ (unsigned)i ^ (blockIdx.x << 12))) ++nb;
        __syncthreads();
    }
    if (nb) atomicAdd(bad, nb);
}

__global__ void __launch_bounds__(256, 3) victim_vgpr(unsigned long long* bad, int spins)
{
    unsigned v[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) { v[i] = 0x5A000000u ^ (threadIdx.x * 97u + i); asm volatile("" : "+v"(v[i])); }
    unsigned long long nb = 0;
    for (int k = 0; k < spins; ++k) {
#pragma unroll
        for (int i = 0; i < 96; ++i) { asm volatile("" : "+v"(v[i])); if (v[i] != (0x5A000000u ^ (threadIdx.x * 97u + i))) ++nb; }
        __builtin_amdgcn_s_sleep(2);
    }
    if (nb) atomicAdd(bad, nb);
}

__global__ void __launch_bounds__(256, 3) victim_global(unsigned long long* bad, unsigned* buf, int n, int spins)
{
    const int gid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int i = gid; i < n; i += stride) buf[i] = 0xC3000000u ^ (unsigned)i;
    __syncthreads();
    unsigned long long nb = 0;
    for (int k = 0; k < spins; ++k)
        for (int i = gid; i < n; i += stride) if (__builtin_nontemporal_load(buf + i) != (0xC3000000u ^ (unsigned)i)) ++nb;
    if (nb) atomicAdd(bad, nb);
}

int main(int argc, char** argv)
{
    void* lib = dlopen(argc > 1 ? argv[1] : "multimoda-rs_amd/lib/libmm_hausdorff.so", RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    auto create = (int (*)(int, void*, mm_engine**))dlsym(lib, "mm_engine_create");
    auto destroy = (void (*)(mm_engine*))dlsym(lib, "mm_engine_destroy");
    auto best = (int (*)(mm_engine*, const double*, const double*, int, const double*, const double*, int, double, double, const double*, int,
                         int, int, double*, double*, int*, double*))dlsym(lib, "mm_best_rotation");
    if (!create || !best) { printf("symbols missing\n"); return 1; }
    const int N = 512, NA = 2881;
    std::vector<double> rx(N), ry(N), tx(N), ty(N), ang(NA);
    for (int i = 0; i < N; ++i) {
        const double t = 6.283185307179586 * i / N;
        rx[i] = 2.5 * cos(t) * (1 + 0.1 * cos(3 * t)); ry[i] = 1.8 * sin(t);
        tx[i] = 2.4 * cos(t + 0.3) * (1 + 0.1 * cos(3 * t)); ty[i] = 1.9 * sin(t + 0.3);
    }
    for (int i = 0; i < NA; ++i) ang[i] = -3.14159 + i * (6.28318 / (NA - 1));
    unsigned long long* bad; unsigned* buf;
    CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&buf, 64 << 20));
    hipStream_t sv;
    CHECK(hipStreamCreate(&sv));
    for (int prec : {MM_PRECISION_F32_FAST, MM_PRECISION_F32_MATRIX}) {
        std::atomic<bool> stop{false};
        std::atomic<long> calls{0};
        std::thread th([&] {
            mm_engine* e = nullptr;
            if (create(0, nullptr, &e)) { printf("engine\n"); return; }
            double ba, bc; int bi;
            while (!stop) { best(e, rx.data(), ry.data(), N, tx.data(), ty.data(), N, 0.0, 0.0, ang.data(), NA, 1, prec, &ba, &bc, &bi, nullptr); ++calls; }
            destroy(e);
        });
        const char* pn = prec == MM_PRECISION_F32_MATRIX ? "k_screen_mx" : "k_screen_fast";
        for (int which = 0; which < 3; ++which) {
            unsigned long long h = 0;
            CHECK(hipMemsetAsync(bad, 0, 8, sv));
            for (int k = 0; k < 300; ++k) {
                if (which == 0) hipLaunchKernelGGL(victim_lds, dim3(768), dim3(256), 16384, sv, bad, 400, 4096);
                if (which == 1) hipLaunchKernelGGL(victim_vgpr, dim3(768), dim3(256), 0, sv, bad, 3000);
                if (which == 2) hipLaunchKernelGGL(victim_global, dim3(768), dim3(256), 0, sv, bad, buf, 4 << 20, 6);
            }
            CHECK(hipStreamSynchronize(sv));
            CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
            printf("%-16s pattern beside %-14s (%ld calls so far): %llu words changed\n", which == 0 ? "LDS (16 KB)" : which == 1 ? "96 VGPRs" : "global (16 MB)", pn, calls.load(), h);
        }
        stop = true; th.join();
    }
    return 0;
}
