// ubench_mfma16c.hip -- the hand-ordered version: the whole loop is one asm block on fixed registers, so the order is
// exactly [MFMA -> buffer 1][NMIN minima of buffer 0][MFMA -> buffer 0][NMIN minima of buffer 1], nothing the compiler
// can re-order.  s_memtime before and after.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma16c.hip -o tools/bin/ubench_mfma16c
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define MIN3_ROW(base) \
    "v_min3_i32 v60, v60, v" #base "0, v" #base "1\n"

// minima on buffer at v[100:115] (A) or v[116:131] (B): 8 column folds + 8 row folds
#define MINS16_A \
    "v_min3_i32 v60, v60, v100, v101\n v_min3_i32 v60, v60, v102, v103\n v_min3_i32 v60, v60, v104, v105\n v_min3_i32 v60, v60, v106, v107\n" \
    "v_min3_i32 v60, v60, v108, v109\n v_min3_i32 v60, v60, v110, v111\n v_min3_i32 v60, v60, v112, v113\n v_min3_i32 v60, v60, v114, v115\n" \
    "v_min3_i32 v61, v61, v100, v108\n v_min3_i32 v62, v62, v101, v109\n v_min3_i32 v63, v63, v102, v110\n v_min3_i32 v64, v64, v103, v111\n" \
    "v_min3_i32 v65, v65, v104, v112\n v_min3_i32 v66, v66, v105, v113\n v_min3_i32 v67, v67, v106, v114\n v_min3_i32 v68, v68, v107, v115\n"
#define MINS16_B \
    "v_min3_i32 v60, v60, v116, v117\n v_min3_i32 v60, v60, v118, v119\n v_min3_i32 v60, v60, v120, v121\n v_min3_i32 v60, v60, v122, v123\n" \
    "v_min3_i32 v60, v60, v124, v125\n v_min3_i32 v60, v60, v126, v127\n v_min3_i32 v60, v60, v128, v129\n v_min3_i32 v60, v60, v130, v131\n" \
    "v_min3_i32 v69, v69, v116, v124\n v_min3_i32 v70, v70, v117, v125\n v_min3_i32 v71, v71, v118, v126\n v_min3_i32 v72, v72, v119, v127\n" \
    "v_min3_i32 v73, v73, v120, v128\n v_min3_i32 v74, v74, v121, v129\n v_min3_i32 v75, v75, v122, v130\n v_min3_i32 v76, v76, v123, v131\n"
#define MINS8_A \
    "v_min3_i32 v60, v60, v100, v101\n v_min3_i32 v60, v60, v102, v103\n v_min3_i32 v60, v60, v104, v105\n v_min3_i32 v60, v60, v106, v107\n" \
    "v_min3_i32 v60, v60, v108, v109\n v_min3_i32 v60, v60, v110, v111\n v_min3_i32 v60, v60, v112, v113\n v_min3_i32 v60, v60, v114, v115\n"
#define MINS8_B \
    "v_min3_i32 v60, v60, v116, v117\n v_min3_i32 v60, v60, v118, v119\n v_min3_i32 v60, v60, v120, v121\n v_min3_i32 v60, v60, v122, v123\n" \
    "v_min3_i32 v60, v60, v124, v125\n v_min3_i32 v60, v60, v126, v127\n v_min3_i32 v60, v60, v128, v129\n v_min3_i32 v60, v60, v130, v131\n"
#define MFMA_A "v_mfma_f32_32x32x16_f16 v[100:115], v[80:83], v[84:87], 0\n"
#define MFMA_B "v_mfma_f32_32x32x16_f16 v[116:131], v[80:83], v[88:91], 0\n"
#define CLOBBERS "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76", \
    "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91", \
    "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
    "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131", "s20", "scc"
#define LOOP(body) \
    asm volatile("s_mov_b32 s20, %1\n" \
                 "v_mov_b32 v80, %2\n v_mov_b32 v81, %2\n v_mov_b32 v82, %2\n v_mov_b32 v83, %2\n" \
                 "v_mov_b32 v84, %3\n v_mov_b32 v85, %3\n v_mov_b32 v86, %3\n v_mov_b32 v87, %3\n" \
                 "v_mov_b32 v88, %3\n v_mov_b32 v89, %2\n v_mov_b32 v90, %3\n v_mov_b32 v91, %2\n" \
                 "v_mov_b32 v60, 0x7f800000\n" \
                 MFMA_A MFMA_B "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n" \
                 "1:\n" body \
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
                 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n v_mov_b32 %0, v60\n" \
                 : "=v"(r) : "s"(iters), "v"(x), "v"(y) : CLOBBERS)

template <int MODE>
__global__ void __launch_bounds__(256) k(int* out, long long* cyc, int iters)
{
    const int x = 0x3c003c00 + (threadIdx.x & 7), y = 0x38003800 + (threadIdx.x & 3);   // pairs of f16 near 1.0 / 0.5
    int r = 0;
    const long long t0 = clock64();
    if (MODE == 0) LOOP(MFMA_B MFMA_A);
    if (MODE == 1) LOOP(MINS16_A MINS16_B);
    if (MODE == 2) LOOP(MFMA_B MINS16_A MFMA_A MINS16_B);
    if (MODE == 3) LOOP(MFMA_B MINS8_A MFMA_A MINS8_B);
    if (MODE == 4) LOOP(MFMA_B "s_nop 1\n" MINS16_A MFMA_A "s_nop 1\n" MINS16_B);
    const long long t1 = clock64();
    if (r == 123456789) out[0] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
int run(const char* name, int* dout, long long* dcyc)
{
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    const int iters = 16384;
    for (int wps : {1, 2, 3}) {
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
        const double tiles = (double)iters * 2.0 * wps;
        printf("%-44s waves/SIMD=%d  %8.3f ms  %6.2f ns/tile/SIMD  %6.1f s_memtime ticks/tile/SIMD  (ticks/ns %.3f)  -> %6.1f Tdist/s per chip\n",
               name, wps, ms, ms * 1e6 / tiles, avg / tiles, avg / (ms * 1e6), tiles * 1024.0 * 1024.0 / (ms * 1e-3) * 1e-12);
    }
    return 0;
}

int main()
{
    int* dout; long long* dcyc;
    CHECK(hipMalloc(&dout, 1024));
    CHECK(hipMalloc(&dcyc, sizeof(long long) * 8192));
    run<0>("mfma 32x32x16 f16 only", dout, dcyc);
    run<1>("16 min3 per tile only", dout, dcyc);
    run<2>("mfma, then 16 min3 of the other buffer", dout, dcyc);
    run<4>("mfma, s_nop 1, 16 min3 of the other buffer", dout, dcyc);
    run<3>("mfma, then 8 min3 of the other buffer", dout, dcyc);
    CHECK(hipFree(dout)); CHECK(hipFree(dcyc));
    return 0;
}
