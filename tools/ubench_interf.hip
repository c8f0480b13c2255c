// ubench_interf.hip -- does a wave running v_mfma_f32_32x32x16_f16 disturb the vector arithmetic of OTHER waves on the chip?
// Found at full size in round 4: the packed-FMA screens (k_screen_fast, k_search<float>, k_screen_lb) returned different
// values when a k_screen_mx launch of another engine ran beside them.  Victim kernels iterate an exactly representable
// recurrence with one kind of instruction and compare with the closed form; the aggressor loops MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_interf.hip -o tools/bin/ubench_interf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16f __attribute__((ext_vector_type(16)));

// MODE 0: v_pk_fma_f32   1: v_fma_f32   2: v_pk_add_f32 + v_pk_mul_f32   3: DPP row minima   4: LDS float4 read + pk_fma
template <int MODE>
__global__ void __launch_bounds__(256, 3) victim(unsigned long long* bad, int iters)
{
    __shared__ float4 s_p[256];
    const int tid = threadIdx.x;
    s_p[tid] = make_float4(1.0f, 0.0f, 1.0f, 0.0f);
    __syncthreads();
    unsigned long long nbad = 0;
    for (int rep = 0; rep < 8; ++rep) {
        if (MODE == 0) {
            v2f x = {(float)(tid & 15), (float)(tid & 7)}, one = {1.0f, 1.0f};
            for (int i = 0; i < iters; ++i) { x = __builtin_elementwise_fma(x, one, one); asm volatile("" : "+v"(x)); }
            if (x.x != (float)((tid & 15) + iters) || x.y != (float)((tid & 7) + iters)) ++nbad;
        } else if (MODE == 1) {
            float x = (float)(tid & 15);
            for (int i = 0; i < iters; ++i) { x = __builtin_fmaf(x, 1.0f, 1.0f); asm volatile("" : "+v"(x)); }
            if (x != (float)((tid & 15) + iters)) ++nbad;
        } else if (MODE == 2) {
            v2f x = {(float)(tid & 15), (float)(tid & 7)}, one = {1.0f, 1.0f};
            for (int i = 0; i < iters; ++i) { x = (x + one) * one; asm volatile("" : "+v"(x)); }
            if (x.x != (float)((tid & 15) + iters) || x.y != (float)((tid & 7) + iters)) ++nbad;
        } else if (MODE == 3) {
            int v = tid & 15;
            for (int i = 0; i < iters; ++i) {
                int w = v + i;
                int o = __builtin_amdgcn_update_dpp(0x7fffffff, w, 0xB1, 0xF, 0xF, false); w = o < w ? o : w;
                o = __builtin_amdgcn_update_dpp(0x7fffffff, w, 0x4E, 0xF, 0xF, false); w = o < w ? o : w;
                o = __builtin_amdgcn_update_dpp(0x7fffffff, w, 0x141, 0xF, 0xF, false); w = o < w ? o : w;
                o = __builtin_amdgcn_update_dpp(0x7fffffff, w, 0x140, 0xF, 0xF, false); w = o < w ? o : w;
                if (w != i) ++nbad;      // the minimum over a row of 16 lanes holding 0..15 + i
            }
        } else if (MODE == 5) {           // ds_bpermute_b32 (__shfl from a run-time lane), as k_screen_lb fetches a candidate's cos / sin
            const int lane = tid & 63;
            for (int i = 0; i < iters; ++i) {
                const int src = (i * 7 + rep) & 63;
                const int got = __shfl(lane * 3 + i, src, 64);
                if (got != src * 3 + i) ++nbad;
            }
        } else if (MODE == 6) {           // v_min3_f32 behind v_pk_fma_f32 (inline asm, as lb_pass)
            v2f q = {(float)(tid & 15), (float)(tid & 7)}, rm = {1e30f, 1e30f};
            for (int i = 0; i < iters; ++i) {
                const v2f e0 = __builtin_elementwise_fma(q, (v2f)(1.0f), (v2f)((float)(i & 1023)));
                const v2f e1 = __builtin_elementwise_fma(q, (v2f)(1.0f), (v2f)((float)((i & 1023) + 1)));
                float r0, r1;
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(rm.x), "v"(e0.x), "v"(e1.x));
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(rm.y), "v"(e0.y), "v"(e1.y));
                rm.x = r0; rm.y = r1;
            }
            if (rm.x != (float)(tid & 15) || rm.y != (float)(tid & 7)) ++nbad;
        } else if (MODE == 7) {           // __shfl_xor butterflies (wave maximum)
            for (int i = 0; i < iters; ++i) {
                int m = ((tid & 63) * 5 + i) & 1023;
                int want = 0;
                for (int sh = 1; sh < 64; sh <<= 1) { const int o = __shfl_xor(m, sh, 64); m = o > m ? o : m; }
                for (int l = 0; l < 64; ++l) { const int v = (l * 5 + i) & 1023; want = v > want ? v : want; }
                if (m != want) ++nbad;
            }
        } else {
            v2f x = {(float)(tid & 15), (float)(tid & 7)};
            for (int i = 0; i < iters; ++i) {
                const float4 p = s_p[(tid + i) & 255];
                x = __builtin_elementwise_fma(x, (v2f)(p.x), (v2f)(p.z));
                asm volatile("" : "+v"(x));
            }
            if (x.x != (float)((tid & 15) + iters) || x.y != (float)((tid & 7) + iters)) ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ void __launch_bounds__(256, 2) aggressor(float* out, int iters)
{
    h8v a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(1.0f + (threadIdx.x & 3)); b[k] = (_Float16)0.5f; }
    f16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if ((i & 255) == 255) { for (int k = 0; k < 16; ++k) acc[k] *= 0.0f; }
    }
    if (acc[0] == 123.0f) out[0] = acc[0];
}

__global__ void __launch_bounds__(256, 2) aggressor_valu(float* out, int iters)     // control: no MFMA
{
    float x = (float)threadIdx.x;
    for (int i = 0; i < iters * 8; ++i) { x = __builtin_fmaf(x, 0.5f, 1.0f); asm volatile("" : "+v"(x)); }
    if (x == 123.0f) out[0] = x;
}

// the shape of k_screen_mx's main phase: independent MFMAs with C = 0 into alternating high VGPR ranges, folded by v_min3_i32
__global__ void __launch_bounds__(256, 2) aggressor_asm(float* out, int iters)
{
    int r = 0;
    const int x = 0x3c003c00 + (threadIdx.x & 7), y = 0x38003800 + (threadIdx.x & 3);
    asm volatile("s_mov_b32 s20, %1\n"
                 "v_mov_b32 v80, %2\n v_mov_b32 v81, %2\n v_mov_b32 v82, %2\n v_mov_b32 v83, %2\n"
                 "v_mov_b32 v84, %3\n v_mov_b32 v85, %3\n v_mov_b32 v86, %3\n v_mov_b32 v87, %3\n"
                 "v_mov_b32 v60, 0x7f800000\n"
                 "v_mfma_f32_32x32x16_f16 v[100:115], v[80:83], v[84:87], 0\n"
                 "v_mfma_f32_32x32x16_f16 v[164:179], v[80:83], v[84:87], 0\n s_nop 15\n s_nop 15\n"
                 "1:\n"
                 "v_mfma_f32_32x32x16_f16 v[116:131], v[80:83], v[84:87], 0\n"
                 "v_min3_i32 v60, v60, v100, v101\n v_min3_i32 v60, v60, v102, v103\n v_min3_i32 v60, v60, v104, v105\n v_min3_i32 v60, v60, v106, v107\n"
                 "v_min3_i32 v60, v60, v108, v109\n v_min3_i32 v60, v60, v110, v111\n v_min3_i32 v60, v60, v112, v113\n v_min3_i32 v60, v60, v114, v115\n"
                 "v_mfma_f32_32x32x16_f16 v[180:195], v[80:83], v[84:87], 0\n"
                 "v_min3_i32 v60, v60, v164, v165\n v_min3_i32 v60, v60, v166, v167\n v_min3_i32 v60, v60, v168, v169\n v_min3_i32 v60, v60, v170, v171\n"
                 "v_min3_i32 v60, v60, v172, v173\n v_min3_i32 v60, v60, v174, v175\n v_min3_i32 v60, v60, v176, v177\n v_min3_i32 v60, v60, v178, v179\n"
                 "v_mfma_f32_32x32x16_f16 v[100:115], v[80:83], v[84:87], 0\n"
                 "v_min3_i32 v60, v60, v116, v117\n v_min3_i32 v60, v60, v118, v119\n v_min3_i32 v60, v60, v120, v121\n v_min3_i32 v60, v60, v122, v123\n"
                 "v_min3_i32 v60, v60, v124, v125\n v_min3_i32 v60, v60, v126, v127\n v_min3_i32 v60, v60, v128, v129\n v_min3_i32 v60, v60, v130, v131\n"
                 "v_mfma_f32_32x32x16_f16 v[164:179], v[80:83], v[84:87], 0\n"
                 "v_min3_i32 v60, v60, v180, v181\n v_min3_i32 v60, v60, v182, v183\n v_min3_i32 v60, v60, v184, v185\n v_min3_i32 v60, v60, v186, v187\n"
                 "v_min3_i32 v60, v60, v188, v189\n v_min3_i32 v60, v60, v190, v191\n v_min3_i32 v60, v60, v192, v193\n v_min3_i32 v60, v60, v194, v195\n"
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
                 "s_nop 15\n s_nop 15\n v_mov_b32 %0, v60\n"
                 : "=v"(r) : "s"(iters), "v"(x), "v"(y)
                 : "v60", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87",
                   "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115",
                   "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131",
                   "v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179",
                   "v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195", "s20", "scc");
    if (r == 123456789) out[0] = (float)r;
}

template <int MODE>
static int run(const char* name, hipStream_t sv, hipStream_t sa, unsigned long long* bad, float* out, int agg)
{
    unsigned long long h = 0;
    CHECK(hipMemset(bad, 0, 8));
    CHECK(hipDeviceSynchronize());
    if (agg == 1) hipLaunchKernelGGL(aggressor, dim3(256 * 2), dim3(256), 0, sa, out, 4000000);
    if (agg == 2) hipLaunchKernelGGL(aggressor_valu, dim3(256 * 2), dim3(256), 0, sa, out, 4000000);
    if (agg == 3) hipLaunchKernelGGL(aggressor_asm, dim3(256 * 2), dim3(256), 0, sa, out, 1000000);
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(victim<MODE>, dim3(256 * 3), dim3(256), 0, sv, bad, MODE == 7 ? 300 : 20000);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("%-34s beside %-18s wrong results: %llu\n", name, agg == 1 ? "an MFMA loop" : (agg == 2 ? "a plain FMA loop" : (agg == 3 ? "asm MFMA + min3" : "nothing")), h);
    return 0;
}

int main()
{
    hipStream_t sv, sa;
    CHECK(hipStreamCreate(&sv)); CHECK(hipStreamCreate(&sa));
    unsigned long long* bad; float* out;
    CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&out, 4));
    for (int agg = 0; agg < 2; ++agg) {
        run<0>("v_pk_fma_f32", sv, sa, bad, out, agg);
        run<1>("v_fma_f32", sv, sa, bad, out, agg);
        run<2>("v_pk_add_f32 + v_pk_mul_f32", sv, sa, bad, out, agg);
        run<3>("v_min_i32 row DPP", sv, sa, bad, out, agg);
        run<4>("ds_read_b128 + v_pk_fma_f32", sv, sa, bad, out, agg);
        run<5>("ds_bpermute_b32 (__shfl)", sv, sa, bad, out, agg);
        run<6>("v_pk_fma_f32 -> v_min3_f32", sv, sa, bad, out, agg);
        run<7>("__shfl_xor wave maximum", sv, sa, bad, out, agg);
    }
    return 0;
}
