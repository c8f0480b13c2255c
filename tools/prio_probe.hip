// prio_probe.hip -- does a short kernel on a second stream get CUs while a long launch of many short
// workgroups fills the device?  Measures the latency of the short kernel (launch -> done) for:
//   (a) both streams default priority, (b) long = lowest / short = highest priority,
//   (c) long stream restricted by a CU mask (all but the first `r` CUs of every XCD... see below), short unrestricted,
//   (d) the long launch split into chunks.
// Build: hipcc --offload-arch=gfx950 -O2 tools/prio_probe.hip -o tools/bin/prio_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256, 3) k_busy(float* out, int iters)
{
    __shared__ float s[3584];   // ~14 KB like k_screen_fast
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    s[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, b, s[(threadIdx.x + i) & 255]);
    if (a == 12345.678f) out[blockIdx.x] = a;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run(const char* name, hipStream_t s_long, hipStream_t s_short, float* d, int chunks)
{
    const int total_wg = 186000, it_long = 6000, short_wg = 92, it_short = 6000;
    // warm
    hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s_long, d, 10);
    hipLaunchKernelGGL(k_busy, dim3(92), dim3(256), 0, s_short, d, 10);
    CK(hipDeviceSynchronize());
    const double t0 = now_ms();
    for (int c = 0; c < chunks; ++c) hipLaunchKernelGGL(k_busy, dim3(total_wg / chunks), dim3(256), 0, s_long, d, it_long);
    // give the long launch time to fill the device
    while (now_ms() - t0 < 3.0) {}
    const double t1 = now_ms();
    hipLaunchKernelGGL(k_busy, dim3(short_wg), dim3(256), 0, s_short, d, it_short);
    CK(hipStreamSynchronize(s_short));
    const double t2 = now_ms();
    CK(hipStreamSynchronize(s_long));
    const double t3 = now_ms();
    printf("%-46s short kernel done after %7.3f ms; long launch total %7.3f ms\n", name, t2 - t1, t3 - t0);
    return 0;
}

int main()
{
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("stream priority range: least %d greatest %d\n", least, greatest);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d\n", prop.multiProcessorCount);
    float* d;
    CK(hipMalloc(&d, 1 << 22));
    hipStream_t a, b, lo, hi;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least));
    CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
    {   // the short kernel alone
        const double t1 = now_ms();
        hipLaunchKernelGGL(k_busy, dim3(92), dim3(256), 0, b, d, 6000);
        CK(hipStreamSynchronize(b));
        printf("%-46s %7.3f ms\n", "short kernel alone (cold)", now_ms() - t1);
        const double t2 = now_ms();
        hipLaunchKernelGGL(k_busy, dim3(92), dim3(256), 0, b, d, 6000);
        CK(hipStreamSynchronize(b));
        printf("%-46s %7.3f ms\n", "short kernel alone", now_ms() - t2);
    }
    if (run("(a) default / default", a, b, d, 1)) return 1;
    if (run("(b) lowest / highest priority", lo, hi, d, 1)) return 1;
    if (run("(b') default / highest priority", a, hi, d, 1)) return 1;
    for (int reserve : {1, 2, 4, 8}) {
        // CU mask: bit i = CU i enabled.  Reserve `reserve` CUs (the first ones) for the short stream only.
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((ncu + 31) / 32, 0xffffffffu);
        if (ncu % 32) mask.back() = (1u << (ncu % 32)) - 1;
        for (int i = 0; i < reserve; ++i) mask[i / 32] &= ~(1u << (i % 32));
        hipStream_t m;
        CK(hipExtStreamCreateWithCUMask(&m, (uint32_t)mask.size(), mask.data()));
        char nm[96];
        snprintf(nm, sizeof nm, "(c) long stream masked off %d CU(s)", reserve);
        if (run(nm, m, b, d, 1)) return 1;
        CK(hipStreamDestroy(m));
    }
    if (run("(d) default / default, long in 8 chunks", a, b, d, 8)) return 1;
    if (run("(d') lowest / highest, long in 8 chunks", lo, hi, d, 8)) return 1;
    if (run("(d'') default / default, long in 32 chunks", a, b, d, 32)) return 1;
    return 0;
}
