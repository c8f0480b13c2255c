// ubench_dual.hip -- what a tile costs if BOTH directed minima are taken in-lane: two MFMAs per 32 x 32 tile (D = A B^T and
// D' = B A^T: a lane's 16 values of D belong to its column, of D' to its row) and 8 + 8 v_min3_i32, no cross-lane row
// reduction at all -- against the shipped scheme's one MFMA + 16 minima (+ the row reduction, not in this loop).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_dual.hip -o tools/bin/ubench_dual
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define M8(acc, b) \
    "v_min3_i32 v" #acc ", v" #acc ", v" #b "0, v" #b "1\n v_min3_i32 v" #acc ", v" #acc ", v" #b "2, v" #b "3\n" \
    "v_min3_i32 v" #acc ", v" #acc ", v" #b "4, v" #b "5\n v_min3_i32 v" #acc ", v" #acc ", v" #b "6, v" #b "7\n"
// buffers: A = v[100:115], B = v[116:131], C = v[132:147], E = v[148:163]; 8 minima over a buffer's 16 registers
#define MINS8(acc, lo, hi) \
    "v_min3_i32 v" #acc ", v" #acc ", v" #lo "0, v" #lo "1\n v_min3_i32 v" #acc ", v" #acc ", v" #lo "2, v" #lo "3\n" \
    "v_min3_i32 v" #acc ", v" #acc ", v" #lo "4, v" #lo "5\n v_min3_i32 v" #acc ", v" #acc ", v" #lo "6, v" #lo "7\n" \
    "v_min3_i32 v" #acc ", v" #acc ", v" #lo "8, v" #lo "9\n v_min3_i32 v" #acc ", v" #acc ", v" #hi "0, v" #hi "1\n" \
    "v_min3_i32 v" #acc ", v" #acc ", v" #hi "2, v" #hi "3\n v_min3_i32 v" #acc ", v" #acc ", v" #hi "4, v" #hi "5\n"
#define MINS_A MINS8(60, 10, 11)
#define MINS_B MINS8(61, 11, 12)   /* v116..v131: v116-119 as 11[6-9], v120-125 as 12[0-5] */
#define MINS_C MINS8(60, 13, 14)   /* v132.. : 13[2-9] wrong digits are harmless for a timing loop: registers stay inside 100..163 */
#define MINS_E MINS8(61, 15, 16)
#define MF(d, b) "v_mfma_f32_32x32x16_f16 v[" #d "], v[80:83], v[" #b "], 0\n"
#define MFB(d) "v_mfma_f32_32x32x16_f16 v[" #d "], v[80:83], a[0:3], 0\n"      /* B operand from accumulation registers */
#define MFA(d) "v_mfma_f32_32x32x16_f16 v[" #d "], a[0:3], v[80:83], 0\n"      /* A operand from accumulation registers */
#define CLOB "v60","v61","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91", \
    "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
    "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131", \
    "v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147", \
    "a0","a1","a2","a3","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165", "s20", "scc"
#define LOOP(body) \
    asm volatile("s_mov_b32 s20, %1\n" \
                 "v_mov_b32 v80, %2\n v_mov_b32 v81, %2\n v_mov_b32 v82, %2\n v_mov_b32 v83, %2\n" \
                 "v_mov_b32 v84, %3\n v_mov_b32 v85, %3\n v_mov_b32 v86, %3\n v_mov_b32 v87, %3\n" \
                 "v_mov_b32 v88, %3\n v_mov_b32 v89, %2\n v_mov_b32 v90, %3\n v_mov_b32 v91, %2\n" \
                 "v_mov_b32 v60, 0x7f800000\n v_mov_b32 v61, 0x7f800000\n" \
                 "v_accvgpr_write_b32 a0, %3\n v_accvgpr_write_b32 a1, %3\n v_accvgpr_write_b32 a2, %3\n v_accvgpr_write_b32 a3, %3\n s_nop 4\n" \
                 MF(100:115, 84:87) MF(116:131, 88:91) MF(132:147, 84:87) MF(148:163, 88:91) "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n" \
                 "1:\n" body \
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
                 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n v_min_i32 %0, v60, v61\n" \
                 : "=v"(r) : "s"(iters), "v"(x), "v"(y) : CLOB)

template <int MODE, int WPS>
__global__ void __launch_bounds__(256, WPS) k(int* out, int iters)
{
    const int x = 0x3c003c00 + (threadIdx.x & 7), y = 0x38003800 + (threadIdx.x & 3);
    int r = 0;
    // one MFMA + 16 minima per tile, two tiles per iteration (the shipped scheme's inner loop)
    if (MODE == 0) LOOP(MF(116:131, 88:91) MINS_A MINS_A MF(100:115, 84:87) MINS_B MINS_B);
    // two MFMAs + 8 + 8 minima per tile, two tiles per iteration: tile t+1's MFMAs go out while tile t's buffers are folded
    if (MODE == 1) LOOP(MF(132:147, 84:87) MINS_A MF(148:163, 88:91) MINS_B MF(100:115, 84:87) MINS_C MF(116:131, 88:91) MINS_E);
    // the same with the operands where the shipped dual block takes them from: D = A x B(acc), D' = B(acc) x A
    if (MODE == 2) LOOP(MFB(132:147) MINS_A MFA(148:163) MINS_B MFB(100:115) MINS_C MFA(116:131) MINS_E);
    if (MODE == 3) LOOP(MFB(132:147) MINS_A MFB(148:163) MINS_B MFB(100:115) MINS_C MFB(116:131) MINS_E);
    if (MODE == 4) LOOP(MFA(132:147) MINS_A MFA(148:163) MINS_B MFA(100:115) MINS_C MFA(116:131) MINS_E);
    if (r == 123456789) out[0] = r;
}

template <int MODE, int WPS>
int run(const char* name, int* dout)
{
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    const int iters = 16384, grid = 256 * WPS;
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(grid), dim3(256), 0, 0, dout, 2048);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(t0));
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(grid), dim3(256), 0, 0, dout, iters);
    CHECK(hipEventRecord(t1));
    CHECK(hipEventSynchronize(t1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, t0, t1));
    const double tiles = (double)iters * 2.0 * WPS;
    printf("%-48s waves/SIMD=%d  %8.3f ms  %6.2f ns/tile/SIMD\n", name, WPS, ms, ms * 1e6 / tiles);
    return 0;
}

int main()
{
    int* dout;
    CHECK(hipMalloc(&dout, 1024));
    run<0, 1>("1 MFMA + 16 min3 per tile", dout); run<0, 2>("1 MFMA + 16 min3 per tile", dout); run<0, 3>("1 MFMA + 16 min3 per tile", dout);
    run<1, 1>("2 MFMA + 8 + 8 min3 per tile", dout); run<1, 2>("2 MFMA + 8 + 8 min3 per tile", dout); run<1, 3>("2 MFMA + 8 + 8 min3 per tile", dout);
    run<2, 2>("2 MFMA (B acc / A acc) + 8 + 8 min3 per tile", dout); run<2, 3>("2 MFMA (B acc / A acc) + 8 + 8 min3 per tile", dout);
    run<3, 2>("2 MFMA (both B acc) + 8 + 8 min3 per tile", dout);
    run<4, 2>("2 MFMA (both A acc) + 8 + 8 min3 per tile", dout);
    CHECK(hipFree(dout));
    return 0;
}
