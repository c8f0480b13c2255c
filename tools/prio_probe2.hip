// prio_probe2.hip -- which operation of a side stream stalls behind another stream's long launch?
// Long launch (186k workgroups of 256 threads, ~14 KB LDS, 3 per CU) on stream L; on stream S, one after the
// other and timed on the host: H2D 25 MB (pinned), small kernel, D2H 32 KB (pinned), H2D 64 KB, 512-thread kernel.
// Build: hipcc --offload-arch=gfx950 -O2 tools/prio_probe2.hip -o tools/bin/prio_probe2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NT>
__global__ void __launch_bounds__(NT, NT == 256 ? 3 : 1) k_busy(float* out, int iters)
{
    __shared__ float s[3584];
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    s[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, b, s[(threadIdx.x + i) & 255]);
    if (a == 12345.678f) out[blockIdx.x] = a;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run(const char* name, hipStream_t L, hipStream_t S, float* d, char* dbuf, char* hbuf)
{
    hipLaunchKernelGGL(k_busy<256>, dim3(256), dim3(256), 0, L, d, 10);
    hipLaunchKernelGGL(k_busy<256>, dim3(92), dim3(256), 0, S, d, 10);
    hipLaunchKernelGGL(k_busy<512>, dim3(8), dim3(512), 0, S, d, 10);
    CK(hipMemcpyAsync(dbuf, hbuf, 1 << 20, hipMemcpyHostToDevice, S));
    CK(hipMemcpyAsync(hbuf, dbuf, 1 << 15, hipMemcpyDeviceToHost, S));
    CK(hipDeviceSynchronize());
    const double t0 = now_ms();
    hipLaunchKernelGGL(k_busy<256>, dim3(186000), dim3(256), 0, L, d, 6000);
    while (now_ms() - t0 < 3.0) {}
    double t[8]; int k = 0;
    t[k++] = now_ms();
    CK(hipMemcpyAsync(dbuf, hbuf, 25 << 20, hipMemcpyHostToDevice, S)); CK(hipStreamSynchronize(S)); t[k++] = now_ms();
    hipLaunchKernelGGL(k_busy<256>, dim3(2048), dim3(256), 0, S, d, 100); CK(hipStreamSynchronize(S)); t[k++] = now_ms();
    CK(hipMemcpyAsync(hbuf, dbuf, 1 << 15, hipMemcpyDeviceToHost, S)); CK(hipStreamSynchronize(S)); t[k++] = now_ms();
    CK(hipMemcpyAsync(dbuf, hbuf, 1 << 16, hipMemcpyHostToDevice, S)); CK(hipStreamSynchronize(S)); t[k++] = now_ms();
    CK(hipMemsetAsync(dbuf, 0, 32, S)); CK(hipStreamSynchronize(S)); t[k++] = now_ms();
    hipLaunchKernelGGL(k_busy<512>, dim3(8), dim3(512), 0, S, d, 100); CK(hipStreamSynchronize(S)); t[k++] = now_ms();
    CK(hipStreamSynchronize(L));
    const double te = now_ms();
    printf("%-34s H2D25M %6.2f | k256 %6.2f | D2H32K %6.2f | H2D64K %6.2f | memset %6.2f | k512 %6.2f | long total %6.2f ms\n", name,
           t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], te - t0);
    return 0;
}

int main()
{
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    float* d; char *dbuf, *hbuf;
    CK(hipMalloc(&d, 1 << 22));
    CK(hipMalloc(&dbuf, 32 << 20));
    CK(hipHostMalloc(&hbuf, 32 << 20, hipHostMallocDefault));
    hipStream_t a, b, lo, hi, lo2, hi2;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least));
    CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&lo2, hipStreamNonBlocking, least));
    CK(hipStreamCreateWithPriority(&hi2, hipStreamNonBlocking, greatest));
    for (int rep = 0; rep < 2; ++rep) {
        if (run("default L / default S", a, b, d, dbuf, hbuf)) return 1;
        if (run("lowest L / highest S", lo, hi, d, dbuf, hbuf)) return 1;
        if (run("lowest L(2) / highest S(1)", lo2, hi, d, dbuf, hbuf)) return 1;
        if (run("lowest L / highest S(2)", lo, hi2, d, dbuf, hbuf)) return 1;
        if (run("default L / highest S", a, hi, d, dbuf, hbuf)) return 1;
    }
    return 0;
}
