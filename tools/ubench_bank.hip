// ubench_bank.hip -- does a three-operand vector instruction pay for reading its operands from the same VGPR bank?
// One wave on an idle chip, s_memtime around 1024 x 64 v_min3_i32 on fixed registers; the patterns differ only in the
// register numbers (bank = register number mod 4).  k_screen_mx folds MFMA results elementwise -- result buffers at
// multiples of 16 put all three operands of those minima in one bank.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_bank.hip -o /tmp/ubench_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41","v42","v43"

// R(i): one instruction of the pattern for element i
#define REP16(R) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
#define STR2(x) #x
#define STR(x) STR2(x)

template <int MODE>
__global__ void __launch_bounds__(64) k(unsigned long long* out, int iters)
{
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
#define A(i) "v_min3_i32 v[20+" #i "], v[20+" #i "], v[100+" #i "], v[116+" #i "]\n"
#define B(i) "v_min3_i32 v[20+" #i "], v[20+" #i "], v[101+" #i "], v[118+" #i "]\n"
#define C(i) "v_min3_i32 v[20+" #i "], v[20+" #i "], v[100+" #i "], v[118+" #i "]\n"
#define D(i) "v_min3_i32 v[20+" #i "], v[20+" #i "], v[101+" #i "], v[117+" #i "]\n"
#define E(i) "v_min3_i32 v[40+(" #i "&3)], v[40+(" #i "&3)], v[100+" #i "], v[117+" #i "]\n"
#define F(i) "v_min_i32 v[20+" #i "], v[20+" #i "], v[100+" #i "]\n"
#define G(i) "v_min_i32 v[20+" #i "], v[20+" #i "], v[101+" #i "]\n"
#define H(i) "v_min3_i32 v[20+" #i "], v[21+" #i "], v[102+" #i "], v[119+" #i "]\n"
        if (MODE == 0) asm volatile(REP16(A) REP16(A) REP16(A) REP16(A) ::: CLOB);
        if (MODE == 1) asm volatile(REP16(B) REP16(B) REP16(B) REP16(B) ::: CLOB);
        if (MODE == 2) asm volatile(REP16(C) REP16(C) REP16(C) REP16(C) ::: CLOB);
        if (MODE == 3) asm volatile(REP16(D) REP16(D) REP16(D) REP16(D) ::: CLOB);
        if (MODE == 4) asm volatile(REP16(E) REP16(E) REP16(E) REP16(E) ::: CLOB);
        if (MODE == 5) asm volatile(REP16(F) REP16(F) REP16(F) REP16(F) ::: CLOB);
        if (MODE == 6) asm volatile(REP16(G) REP16(G) REP16(G) REP16(G) ::: CLOB);
        if (MODE == 7) asm volatile(REP16(H) REP16(H) REP16(H) REP16(H) ::: CLOB);
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name, unsigned long long* dout)
{
    const int iters = 1024;
    for (int waves : {1, 2}) {      // waves on the one SIMD that is used: 1 = alone, 2 = two workgroups of one wave... (one CU)
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, dout, 16);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, dout, iters);
        CHECK(hipDeviceSynchronize());
        unsigned long long h = 0;
        CHECK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
        printf("%-58s %8.3f cycles per instruction (one wave, %d x 64 instructions)\n", name, (double)h / (iters * 64.0), iters);
        break;
    }
    return 0;
}

int main()
{
    unsigned long long* d;
    CHECK(hipMalloc(&d, 4096));
    run<0>("v_min3 d=s0 bank b, s1 bank b, s2 bank b   (row minima)", d);
    run<1>("v_min3 d=s0 bank b, s1 b+1, s2 b+2         (all distinct)", d);
    run<2>("v_min3 d=s0 bank b, s1 b, s2 b+2           (two of a kind)", d);
    run<3>("v_min3 d=s0 bank b, s1 b+1, s2 b+1         (s1 = s2)", d);
    run<4>("v_min3 4 accumulators, s1 b, s2 b+1         (column fold)", d);
    run<5>("v_min  d=s0 bank b, s1 bank b", d);
    run<6>("v_min  d=s0 bank b, s1 bank b+1", d);
    run<7>("v_min3 d b, s0 b+1, s1 b+2, s2 b+3          (nothing shared)", d);
    return 0;
}
