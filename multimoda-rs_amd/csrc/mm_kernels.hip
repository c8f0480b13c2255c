// mm_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the Hausdorff pose search.
//
// What one candidate ("pose-eval") is, in the reference (yungselm/multimoda-rs):
//   process_utils.rs:78-121  hausdorff_distance = max(directed(A,B), directed(B,A)),
//                            directed = sqrt(max_a min_b ((ax-bx)^2 + (ay-by)^2))
//   contour_point.rs:38-52   B = rotate(target, angle, centre)
// The reference evaluates the Na x Nb squared-distance matrix twice (once per direction).
// (ax-bx)^2 and (bx-ax)^2 are the same bits, so one pass over the matrix with a running
// row-min (per reference point) and column-min (per target point) yields both directed
// distances exactly.
//
// Mapping to the hardware (vector pipe; the one kernel that puts the distance arithmetic on the f16 matrix pipe and
// keeps only the minima here is k_screen_mx, further down):
//   * one workgroup = (pair, slice of its candidate angles); threads form a 16-wide
//     lane grid: lj = tid & 15 picks columns (target points), li = tid >> 4 picks rows
//     (reference points).  A DPP row (16 lanes) shares li, so row-min reduction is
//     in-row cross-lane; the column-min reduction goes through LDS ds_min atomics.
//   * reference points live in VGPRs for the whole angle slice (R rows per lane);
//     the rotated target is staged in LDS once per angle (float2/double2, 16 distinct
//     consecutive addresses per wave-instruction -> conflict-free broadcast reads).
//   * f32 screening uses 2x2 micro-tiles: packed v_pk_add/mul/fma_f32 for the distance
//     and three-operand integer min (v_min3_u32 on the bit patterns, valid because
//     squared distances are >= +0) for the row/column minima.
//   * the f64 kernel reproduces the reference's operation order with contraction off,
//     so its squared distances are bit-identical to the Rust code's.
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include <cstdio>

#include "mm_device.h"
#include "../../include/mm_hausdorff.h"

namespace mm {

typedef float v2f __attribute__((ext_vector_type(2)));

template <typename T> struct Tr;
template <> struct Tr<float> {
    using U = unsigned int;
    using T2 = float2;
    static __device__ __forceinline__ U bits(float v) { return __float_as_uint(v); }
    static __device__ __forceinline__ float from(U u) { return __uint_as_float(u); }
    static __device__ __forceinline__ float inf() { return __uint_as_float(0x7f800000u); }
    static constexpr U INF_BITS = 0x7f800000u;
};
template <> struct Tr<double> {
    using U = unsigned long long;
    using T2 = double2;
    static __device__ __forceinline__ U bits(double v) { return (U)__double_as_longlong(v); }
    static __device__ __forceinline__ double from(U u) { return __longlong_as_double((long long)u); }
    static __device__ __forceinline__ double inf() { return __longlong_as_double(0x7ff0000000000000ll); }
    static constexpr U INF_BITS = 0x7ff0000000000000ull;
};

// min of non-negative floats through their bit patterns (v_min_u32 / v_min3_u32).
static __device__ __forceinline__ float umin3f(float a, float b, float c)
{
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b), uc = __float_as_uint(c);
    unsigned m = ua < ub ? ua : ub;
    m = m < uc ? m : uc;
    return __uint_as_float(m);
}
static __device__ __forceinline__ float umin2f(float a, float b)
{
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    return __uint_as_float(ua < ub ? ua : ub);
}
// squared distances are never NaN (finite inputs), so IEEE minNum == the reference's
// `if d2 < min_sq { min_sq = d2 }` (process_utils.rs:108-110); one v_min_f64.  Written as the
// instruction itself: __builtin_fmin makes the compiler canonicalise (v_max_f64 x, x) every running
// minimum that is carried around the loop, one extra fp64 operation per row and step (+14 %).
static __device__ __forceinline__ double dmin2(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// All-reduce min over the 16 lanes of a DPP row (lanes sharing li).  DPP keeps this on
// the VALU (v_min_u32_dpp) instead of a ds_bpermute round trip through the LDS crossbar:
// quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror.
template <int CTRL>
static __device__ __forceinline__ unsigned dpp_min_u32(unsigned v)
{
    // old = UINT_MAX is the identity of min, all rows/banks enabled -> v_min_u32_dpp
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, CTRL, 0xF, 0xF, false);
    return o < v ? o : v;
}
static __device__ __forceinline__ float lane_min16(float f)
{
    unsigned v = __float_as_uint(f);
    v = dpp_min_u32<0xB1>(v);
    v = dpp_min_u32<0x4E>(v);
    v = dpp_min_u32<0x141>(v);
    v = dpp_min_u32<0x140>(v);
    return __uint_as_float(v);
}
static __device__ __forceinline__ double dpp_f64(double d, const int ctrl_sel)
{
    const long long b = __double_as_longlong(d);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);  hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false);  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break;
    }
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
static __device__ __forceinline__ double lane_min16(double v)
{
    double o;
    o = dpp_f64(v, 0); v = o < v ? o : v;
    o = dpp_f64(v, 1); v = o < v ? o : v;
    o = dpp_f64(v, 2); v = o < v ? o : v;
    o = dpp_f64(v, 3); v = o < v ? o : v;
    return v;
}

template <typename T>
static __device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        T o = __shfl_xor(v, m, 64);
        v = (o > v) ? o : v;
    }
    return v;
}

// -------------------------------------------------------------------------------------
// The search kernel.
//   T      float (screening) or double (exact)
//   R      reference points (rows) per lane held in registers
//   NLI    row lanes per workgroup (threads = 16 * NLI)
//   EXACT  reference operation order, absolute coordinates, angle==0 shortcut
// XCD-aware work order.  Workgroups are dealt round-robin over the 8 XCDs (observed, not contractual:
// b and b+8 share an XCD and its private 4 MiB L2), while the work list is pair-major (all candidate
// blocks of a pair are adjacent).  With the identity mapping every pair's point sets and tables are
// pulled into all eight L2s; this bijective remap hands each XCD one contiguous eighth of the list, so
// a pair is fetched from HBM by one XCD (or two, at a boundary).  Speed/traffic only, never correctness.
static __device__ __forceinline__ int xcd_work_index(int b, int n)
{
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// Work items come either from a host-built table (n_work_dev == nullptr) or from the
// device shortlist queue (count read from *n_work_dev); workgroups stride over them, so
// every wave terminates whatever the queue length is.
// -------------------------------------------------------------------------------------
//   MINB   workgroups per CU the register allocation must leave room for (launch bounds)
template <typename T, int R, int NLI, bool EXACT, bool MULTI_RB, int MINB = 1>
__global__ void __launch_bounds__(NLI * 16, MINB)
k_search(const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ work,
         int n_work_host, const int* __restrict__ n_work_dev,
         const T* __restrict__ ptx, const T* __restrict__ pty,
         const T* __restrict__ cosv, const T* __restrict__ sinv,
         T* __restrict__ out_sq)
{
    using TT = Tr<T>;
    using U = typename TT::U;
    using T2 = typename TT::T2;
    constexpr int NT = NLI * 16;
    constexpr int RP = R / 2;          // row pairs
    constexpr bool ODD = (R & 1) != 0; // one extra single row
    constexpr int ROWS_PER_BLOCK = NLI * R;

    extern __shared__ __align__(16) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lj = tid & 15;
    const int li = tid >> 4;
    const int n_work = n_work_dev ? *n_work_dev : n_work_host;

    // one workgroup per item (host-built table): XCD-aware order; otherwise stride over the queue
    for (int wi = (int)gridDim.x == n_work ? xcd_work_index(blockIdx.x, n_work) : (int)blockIdx.x; wi < n_work;
         wi += gridDim.x) {
        const WorkItem w = work[wi];
        const PairDesc pd = pairs[w.pair];
        const int na = pd.n_ref, nb = pd.n_tgt;
        const int nbp = (nb + 15) & ~15;
        const int ncg = nbp >> 4;  // 16-column groups
        const int nrb = MULTI_RB ? (na + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK : 1;

        T2* s_tgt = reinterpret_cast<T2*>(smem);
        T2* s_b = s_tgt + nbp;
        U* s_colmin = reinterpret_cast<U*>(s_b + nbp);
        U* s_red = s_colmin + nbp;

        __syncthreads();  // previous work item is done with LDS
        for (int j = tid; j < nbp; j += NT) {
            // padding columns duplicate the last target point: a duplicate cannot change a row
            // minimum, and its own column minimum is never read
            const int jc = j < nb ? j : nb - 1;
            T2 t;
            t.x = ptx[pd.tgt_off + jc]; t.y = pty[pd.tgt_off + jc];
            s_tgt[j] = t;
        }

        // reference rows -> registers (row = rb*ROWS_PER_BLOCK + r*NLI + li); padding rows
        // duplicate the last reference point: they repeat its row minimum (the max is unchanged)
        // and cannot lower a column minimum -- no far-away sentinels, no validity tests.
        T ax[R], ay[R];
        auto load_rows = [&](int rb) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = rb * ROWS_PER_BLOCK + r * NLI + li;
                const int rc = row < na ? row : na - 1;
                ax[r] = ptx[pd.ref_off + rc];
                ay[r] = pty[pd.ref_off + rc];
            }
        };
        if constexpr (!MULTI_RB) load_rows(0);

        const T cx = (T)pd.cx, cy = (T)pd.cy;
        const bool skip_zero = (pd.flags & MM_SEARCH_SKIP_ZERO) != 0;

        for (int a = w.a0; a < w.a0 + w.cnt; ++a) {
            const T c = cosv[pd.tab_off + a];
            const T s = sinv[pd.tab_off + a];

            __syncthreads();  // S0: readers of s_b / s_colmin / s_red from the previous angle are done
            for (int j = tid; j < nbp; j += NT) {
                const T2 t = s_tgt[j];
                T2 b;
                if constexpr (EXACT) {
                    // contour_point.rs:39-52 / align_between.rs:194-206
                    if (skip_zero && s == (T)0) {  // angle == 0.0  <=>  sin(angle) == 0 for f64
                        b = t;
                    } else {
                        const T x = t.x - cx;
                        const T y = t.y - cy;
                        b.x = (x * c - y * s) + cx;
                        b.y = (x * s + y * c) + cy;
                    }
                } else {
                    b.x = __builtin_fmaf(t.x, c, -(t.y * s));
                    b.y = __builtin_fmaf(t.x, s, t.y * c);
                }
                s_b[j] = b;
                s_colmin[j] = TT::INF_BITS;
            }
            if (tid == 0) s_red[0] = 0;
            __syncthreads();  // S1

            T rowmax = (T)0;
            for (int rb = 0; rb < nrb; ++rb) {
                if constexpr (MULTI_RB) load_rows(rb);
                T rmin[R];
#pragma unroll
                for (int r = 0; r < R; ++r) rmin[r] = TT::inf();

                int k = 0;
                for (; k + 1 < ncg; k += 2) {
                    const T2 b0 = s_b[k * 16 + lj];
                    const T2 b1 = s_b[(k + 1) * 16 + lj];
                    T cm0 = TT::inf(), cm1 = TT::inf();
                    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                        for (int q = 0; q < RP; ++q) {
                            const v2f axp = {ax[2 * q], ax[2 * q + 1]};
                            const v2f ayp = {ay[2 * q], ay[2 * q + 1]};
                            const v2f dx0 = axp - (v2f)(b0.x);
                            const v2f dy0 = ayp - (v2f)(b0.y);
                            const v2f dx1 = axp - (v2f)(b1.x);
                            const v2f dy1 = ayp - (v2f)(b1.y);
                            const v2f d0 = __builtin_elementwise_fma(dy0, dy0, dx0 * dx0);
                            const v2f d1 = __builtin_elementwise_fma(dy1, dy1, dx1 * dx1);
                            rmin[2 * q]     = umin3f(rmin[2 * q], d0.x, d1.x);
                            rmin[2 * q + 1] = umin3f(rmin[2 * q + 1], d0.y, d1.y);
                            cm0 = umin3f(cm0, d0.x, d0.y);
                            cm1 = umin3f(cm1, d1.x, d1.y);
                        }
                        if constexpr (ODD) {
                            const float dx0 = ax[R - 1] - b0.x, dy0 = ay[R - 1] - b0.y;
                            const float dx1 = ax[R - 1] - b1.x, dy1 = ay[R - 1] - b1.y;
                            const float d0 = __builtin_fmaf(dy0, dy0, dx0 * dx0);
                            const float d1 = __builtin_fmaf(dy1, dy1, dx1 * dx1);
                            rmin[R - 1] = umin3f(rmin[R - 1], d0, d1);
                            cm0 = umin2f(cm0, d0);
                            cm1 = umin2f(cm1, d1);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            // process_utils.rs:105-110 (no fma: -ffp-contract=off)
                            const T dx0 = ax[r] - b0.x, dy0 = ay[r] - b0.y;
                            const T dx1 = ax[r] - b1.x, dy1 = ay[r] - b1.y;
                            const T d0 = dx0 * dx0 + dy0 * dy0;
                            const T d1 = dx1 * dx1 + dy1 * dy1;
                            rmin[r] = dmin2(dmin2(rmin[r], d0), d1);
                            cm0 = dmin2(cm0, d0);
                            cm1 = dmin2(cm1, d1);
                        }
                    }
                    atomicMin(&s_colmin[k * 16 + lj], TT::bits(cm0));
                    atomicMin(&s_colmin[(k + 1) * 16 + lj], TT::bits(cm1));
                }
                if (k < ncg) {  // odd tail: one column group
                    const T2 b0 = s_b[k * 16 + lj];
                    T cm0 = TT::inf();
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const T dx0 = ax[r] - b0.x, dy0 = ay[r] - b0.y;
                        T d0;
                        if constexpr (std::is_same<T, float>::value) {
                            d0 = __builtin_fmaf(dy0, dy0, dx0 * dx0);
                            rmin[r] = umin2f(rmin[r], d0);
                            cm0 = umin2f(cm0, d0);
                        } else {
                            d0 = dx0 * dx0 + dy0 * dy0;
                            rmin[r] = dmin2(rmin[r], d0);
                            cm0 = dmin2(cm0, d0);
                        }
                    }
                    atomicMin(&s_colmin[k * 16 + lj], TT::bits(cm0));
                }

                // directed(A,B): min over all columns (16 lanes of the row), max over valid rows
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T v = lane_min16(rmin[r]);
                    // process_utils.rs:112-114: `if min_sq.is_finite() && min_sq > local_max_sq` -- a point whose
                    // distances are all inf / NaN (non-finite coordinates) does not take part in the maximum.
                    // Only where the value IS the reference's (EXACT): in a screen an inf is an f32 overflow of a
                    // finite distance and must stay inf -- dropping it would understate the candidate's cost and
                    // could shortlist past the true minimum.
                    if constexpr (EXACT) rowmax = (v > rowmax && v < TT::inf()) ? v : rowmax;
                    else rowmax = v > rowmax ? v : rowmax;
                }
            }
            __syncthreads();  // S2: all column minima are in LDS

            // directed(B,A): max over valid columns of the (finite) column minima
            T m = rowmax;
            for (int j = tid; j < nb; j += NT) {
                const T v = TT::from(s_colmin[j]);
                if constexpr (EXACT) m = (v > m && v < TT::inf()) ? v : m;
                else m = v > m ? v : m;
            }
            m = wave_max(m);
            if ((tid & 63) == 0) atomicMax(&s_red[0], TT::bits(m));
            __syncthreads();  // S3
            if (tid == 0) out_sq[pd.out_off + a] = TT::from(s_red[0]);
        }
    }
}

// -------------------------------------------------------------------------------------
// Fast f32 screening kernel (MM_PRECISION_F32_FAST): same lane grid and data flow as
// k_search<float>, but the squared distance is evaluated in the expanded form
//     d^2 = |a|^2 + (|b|^2 - 2 a.b) = A2 + fma(-2ax, bx, fma(-2ay, by, B2))
// i.e. 2 v_pk_fma_f32 + 1 v_pk_add_f32 per two pair-distances instead of
// 2 v_pk_add + 1 v_pk_mul + 1 v_pk_fma: 6 packed + 4 min3 per 2x2 micro-tile instead of 8 + 4.
// The price is cancellation: |d2_f32 - d2| <= 5 u (rho_a + rho_b)^2 (u = 2^-24) instead of a
// relative 2u, so the host widens the shortlist interval accordingly (PairDesc::e2) and the
// exact f64 re-score still decides every winner.  Rounded results can be a few ulp below
// zero; minima therefore use SIGNED integer order on the bit patterns (v_min3_i32): exact for
// non-negative values, and any negative value is within the error bound of zero.
// Single row block only (Na <= 16 R); larger reference sets use k_search.
// -------------------------------------------------------------------------------------
static __device__ __forceinline__ float imin3f(float a, float b, float c)
{
    int ia = __float_as_int(a), ib = __float_as_int(b), ic = __float_as_int(c);
    int m = ia < ib ? ia : ib;
    m = m < ic ? m : ic;
    return __int_as_float(m);
}
static __device__ __forceinline__ float imin2f(float a, float b)
{
    int ia = __float_as_int(a), ib = __float_as_int(b);
    return __int_as_float(ia < ib ? ia : ib);
}
template <int CTRL>
static __device__ __forceinline__ int dpp_min_i32(int v)
{
    const int o = __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, 0xF, 0xF, false);
    return o < v ? o : v;
}

template <int R, bool EMIT>
__global__ void __launch_bounds__(256, EMIT ? 2 : 3)
k_screen_fast(const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ work,
              int n_work_host, const int* __restrict__ n_work_dev,
              const float* __restrict__ ptx, const float* __restrict__ pty,
              const float* __restrict__ cosv, const float* __restrict__ sinv, float* __restrict__ out_sq,
              float* __restrict__ emit, int emit_rows, int emit_cols)
{
    // EMIT: per pair emit_rows + emit_cols floats; the item's LAST candidate leaves its row minima (one
    // per reference point) and column minima (one per target point) there -- used on single-candidate
    // items, to find the points that decide the pair's Hausdorff distance.  A separate instantiation:
    // the stores cost the main screen 5 % when they were a run-time branch.
    constexpr int NT = 256, NLI = 16;
    constexpr int RP = R / 2;
    constexpr bool ODD = (R & 1) != 0;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lj = tid & 15, li = tid >> 4;
    const int n_work = n_work_dev ? *n_work_dev : n_work_host;   // device queue: workgroups stride over it

    for (int wi = (int)gridDim.x == n_work ? xcd_work_index(blockIdx.x, n_work) : (int)blockIdx.x; wi < n_work;
         wi += gridDim.x) {
        const WorkItem w = work[wi];
        const PairDesc pd = pairs[w.pair];
        const int na = pd.n_ref, nb = pd.n_tgt;
        const int nbp = (nb + 15) & ~15;
        const int ncg = nbp >> 4;

        float4* s_b = reinterpret_cast<float4*>(smem);            // (bx, by, |b|^2, 0)
        float2* s_tgt = reinterpret_cast<float2*>(s_b + nbp);
        int* s_colmin = reinterpret_cast<int*>(s_tgt + nbp);
        int* s_red = s_colmin + nbp;

        __syncthreads();
        for (int j = tid; j < nbp; j += NT) {
            float2 t;
            const int jc = j < nb ? j : nb - 1;   // padding columns duplicate the last point: they
            t.x = ptx[pd.tgt_off + jc];            // cannot change a row minimum, and their own
            t.y = pty[pd.tgt_off + jc];            // column minimum is never read
            s_tgt[j] = t;
        }
        // rows: (-2ax, -2ay) and |a|^2; padding rows duplicate the last reference point, so they
        // repeat its row minimum (max unchanged) and cannot lower any column minimum
        v2f m2x[RP + 1], m2y[RP + 1], a2[RP + 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = r * NLI + li;
            const int rc = row < na ? row : na - 1;
            const float vx = ptx[pd.ref_off + rc], vy = pty[pd.ref_off + rc];
            const float n2 = __builtin_fmaf(vx, vx, vy * vy);
            if (r & 1) { m2x[r >> 1].y = -2.0f * vx; m2y[r >> 1].y = -2.0f * vy; a2[r >> 1].y = n2; }
            else       { m2x[r >> 1].x = -2.0f * vx; m2y[r >> 1].x = -2.0f * vy; a2[r >> 1].x = n2; }
        }

        for (int a = w.a0; a < w.a0 + w.cnt; ++a) {
            const float c = cosv[pd.tab_off + a], s = sinv[pd.tab_off + a];
            __syncthreads();  // S0
            for (int j = tid; j < nbp; j += NT) {
                const float2 t = s_tgt[j];
                float4 b;
                b.x = __builtin_fmaf(t.x, c, -(t.y * s));
                b.y = __builtin_fmaf(t.x, s, t.y * c);
                b.z = __builtin_fmaf(b.x, b.x, b.y * b.y);
                b.w = 0.0f;
                s_b[j] = b;
                s_colmin[j] = 0x7f800000;
            }
            if (tid == 0) s_red[0] = 0;
            __syncthreads();  // S1

            float rmin[R];
#pragma unroll
            for (int r = 0; r < R; ++r) rmin[r] = __int_as_float(0x7f800000);

            auto tile2 = [&](const float4 b0, const float4 b1, float& cm0, float& cm1) {
#pragma unroll
                for (int q = 0; q < RP; ++q) {
                    const v2f t0 = __builtin_elementwise_fma(m2y[q], (v2f)(b0.y), (v2f)(b0.z));
                    const v2f t1 = __builtin_elementwise_fma(m2y[q], (v2f)(b1.y), (v2f)(b1.z));
                    const v2f e0 = __builtin_elementwise_fma(m2x[q], (v2f)(b0.x), t0);
                    const v2f e1 = __builtin_elementwise_fma(m2x[q], (v2f)(b1.x), t1);
                    const v2f d0 = e0 + a2[q];
                    const v2f d1 = e1 + a2[q];
                    rmin[2 * q]     = imin3f(rmin[2 * q], d0.x, d1.x);
                    rmin[2 * q + 1] = imin3f(rmin[2 * q + 1], d0.y, d1.y);
                    cm0 = imin3f(cm0, d0.x, d0.y);
                    cm1 = imin3f(cm1, d1.x, d1.y);
                }
                if constexpr (ODD) {
                    const float d0 = __builtin_fmaf(m2x[RP].x, b0.x, __builtin_fmaf(m2y[RP].x, b0.y, b0.z)) + a2[RP].x;
                    const float d1 = __builtin_fmaf(m2x[RP].x, b1.x, __builtin_fmaf(m2y[RP].x, b1.y, b1.z)) + a2[RP].x;
                    rmin[R - 1] = imin3f(rmin[R - 1], d0, d1);
                    cm0 = imin2f(cm0, d0);
                    cm1 = imin2f(cm1, d1);
                }
            };

            int k = 0;
            for (; k + 1 < ncg; k += 2) {
                const float4 b0 = s_b[k * 16 + lj];
                const float4 b1 = s_b[(k + 1) * 16 + lj];
                float cm0 = __int_as_float(0x7f800000), cm1 = cm0;
                tile2(b0, b1, cm0, cm1);
                atomicMin(&s_colmin[k * 16 + lj], __float_as_int(cm0));
                atomicMin(&s_colmin[(k + 1) * 16 + lj], __float_as_int(cm1));
            }
            if (k < ncg) {  // odd tail: the column group paired with itself
                const float4 b0 = s_b[k * 16 + lj];
                float cm0 = __int_as_float(0x7f800000), cmd = cm0;
                tile2(b0, b0, cm0, cmd);
                atomicMin(&s_colmin[k * 16 + lj], __float_as_int(cm0));
            }

            // rows: min over the 16 column lanes, then max over valid rows (floor 0)
            int rowmax = 0;
            float* const em = EMIT ? emit + (size_t)w.pair * (size_t)(emit_rows + emit_cols) : nullptr;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int v = __float_as_int(rmin[r]);
                v = dpp_min_i32<0xB1>(v); v = dpp_min_i32<0x4E>(v); v = dpp_min_i32<0x141>(v); v = dpp_min_i32<0x140>(v);
                rowmax = v > rowmax ? v : rowmax;
                if (EMIT && lj == 0 && r * NLI + li < na) em[r * NLI + li] = __int_as_float(v);
            }
            __syncthreads();  // S2
            int m = rowmax;
            for (int j = tid; j < nb; j += NT) {
                const int v = s_colmin[j];
                m = v > m ? v : m;
                if (EMIT) em[emit_rows + j] = __int_as_float(v);
            }
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) { const int o = __shfl_xor(m, sh, 64); m = o > m ? o : m; }
            if ((tid & 63) == 0) atomicMax(&s_red[0], m);
            __syncthreads();  // S3
            if (tid == 0) out_sq[pd.out_off + a] = __int_as_float(s_red[0]);
        }
    }
}


// -------------------------------------------------------------------------------------
// Matrix-pipe screening kernel (MM_PRECISION_F32_MATRIX).  Same contract and work decomposition as
// k_screen_fast -- one workgroup = (pair, <= 8 consecutive candidates), one screened squared Hausdorff
// value per candidate, the exact f64 re-score decides every winner -- but the squared distances come from
// the f16 matrix pipe:
//     d^2 = |a|^2 + |b|^2 - 2 a.b   as ONE v_mfma_f32_32x32x16_f16 per 32 x 32 tile (1024 distances),
// every coordinate split into f16 hi + lo pieces (22 significant bits), fp32 accumulation.  K slots of a
// fragment (lane l: row or column l & 31, slots 8 * (l >> 5) .. + 7):
//     rows    A:  [-2x1, -2y1, -2x1, -2y1, -2x2, -2y2, -2x2, -2y2 | n2h, n2l, 256, 1,   0, 0,   0,   0]
//     columns B:  [  x1,   y1,   x2,   y2,   x1,   y1,   x2,   y2 | 256,   1, n2h, n2l, 256, 1, n2h, n2l]
// (a column fragment's slots 4..7 repeat its slots 0..3, so it is stored as 8 bytes and read twice; its coordinate half
// is the packed hi pieces followed by the packed lo pieces of (x, y): two conversions, no shuffle)
// with x = S * (coordinate), S = 2^e chosen per pair so that the larger set radius lands in [256, 512) (all pieces,
// their doubles and n2 / 256 stay inside f16's range), n2 = 256 * n2h + n2l the squared norm -- rows: of the split
// point; columns: of the scaled point before the rotation (once per work item).
// The vector pipe is left with the minima: per tile 8 v_min3_i32 fold its 16 values into the column minimum and
// 8 more take the elementwise row minima of two tiles at once.  tools/ubench_mfma16c.hip: hand-ordered, the MFMA
// runs entirely beside the 16 minima (34 ns per tile per SIMD against 82 for the packed-FMA form); the compiler's
// own schedule does not (46 ns), so the main phase is one generated asm block (tools/gen_screen_mx.py).
// The workgroup shares the pair's row fragments; below that every WAVE takes candidates of its own (a0 + wave, + 4, ...)
// and computes all 17 x 17 tiles of each: column fragments written to a wave-private LDS copy and read from there ONCE, into
// accumulation registers (every MFMA of a column tile takes its B operand from a[4t : 4t + 3]), column minima in registers
// for the whole candidate, row minima across the 32 column lanes through a wave-private LDS transpose (or, for small sets,
// in-lane from a second, transposed MFMA per tile: the DUAL form) -- no barrier and no shared accumulator in the candidate loop.  Error of the screened value: PairDesc::e2 = mx_e2(rho_a, rho_b) (mm_engine.cpp).
// Set sizes: the column-tile count NCT (target set, <= 17 per block) is a template parameter -- the column minima live in
// one register per column tile -- and the row-tile count (reference set) is a run-time operand of the asm block; a target
// set of more than 544 points is cut into equal blocks of NCT tiles (MULTI: the row minima are carried from block to block
// through a wave-private row store in LDS).  Sets of 64 .. 2048 points take this kernel (mx_min_points .. mx_max_points);
// the engine groups the work items by variant, one launch per group.
// -------------------------------------------------------------------------------------
#include "mm_screen_mx_asm.inc"

typedef _Float16 h8v __attribute__((ext_vector_type(8)));

static constexpr int MX_TMAX = MM_SCREEN_MX_NCT_MAX;          // column tiles of one block
static constexpr int MX_ROW_TILES_MAX = 64;                     // row tiles (LDS: 1 KB of fragments per row tile)
static constexpr int MX_RED = 32 * (MM_SCREEN_MX_RED_STRIDE / 4);   // ints of one wave's reduction scratch
static constexpr int MX_ITEM = 64;                              // candidates of one work item, at most (the engine's apb)

static __device__ __forceinline__ void mx_split(float X, _Float16& h, _Float16& l)
{
    // X and h are made opaque to the optimiser.  Without that the compiler fuses the producer of X (the rotation's fma)
    // with the conversion (v_fma_mixlo_f16: ONE rounding of the exact fma to f16) where it computes the residual, and
    // converts the f32 value (v_cvt_pk_f16_f32) where it packs the fragment: at f16 ties the two disagree by one ulp of h
    // and the stored pair (h, l) no longer sums to X -- one column in a few thousand came out a quarter unit off.
    asm volatile("" : "+v"(X));
    unsigned hb = __builtin_bit_cast(unsigned short, (_Float16)X);
    asm volatile("" : "+v"(hb));
    h = __builtin_bit_cast(_Float16, (unsigned short)hb);
    l = (_Float16)(X - (float)h);
}

// fragment of one point for lane half `hi` (0: coordinate slots, 1: norm slots); ROWS: the A (row) form
template <bool ROWS>
static __device__ __forceinline__ h8v mx_fragment(float X, float Y, int hi)
{
    _Float16 x1, x2, y1, y2;
    mx_split(X, x1, x2);
    mx_split(Y, y1, y2);
    h8v f;
    if (hi == 0) {
        if (ROWS) {
            const _Float16 m = (_Float16)-2.0f;
            f = h8v{m * x1, m * y1, m * x1, m * y1, m * x2, m * y2, m * x2, m * y2};
        } else {
            f = h8v{x1, y1, x2, y2, x1, y1, x2, y2};
        }
    } else {
        const float xs = (float)x1 + (float)x2, ys = (float)y1 + (float)y2;          // the split point, exact in f32
        const float n2 = __builtin_fmaf(xs, xs, ys * ys);
        const _Float16 nh = (_Float16)(n2 * 0.00390625f);
        const _Float16 nl = (_Float16)__builtin_fmaf(-256.0f, (float)nh, n2);
        const _Float16 c256 = (_Float16)256.0f, one = (_Float16)1.0f, z = (_Float16)0.0f;
        if (ROWS) f = h8v{nh, nl, c256, one, z, z, z, z};
        else      f = h8v{c256, one, nh, nl, c256, one, nh, nl};
    }
    return f;
}

typedef _Float16 h4v __attribute__((ext_vector_type(4)));

// the n points at (px, py) as ntiles x 64 row fragments in LDS (padding rows duplicate the last point), by 256 threads.  Four
// slots per thread and pass: the loads of a pass are issued together -- fragment by fragment the loop exposed one global-memory
// round trip per slot (8.5 of them for a 544-point set: ~25 us of a bound kernel's work item, more than its MFMAs)
template <int NT = 256>
static __device__ __forceinline__ void mx_stage_rows(h8v* __restrict__ s_dst, int ntiles, int n, const float* __restrict__ px,
                                                     const float* __restrict__ py, float S, int tid)
{
    constexpr int U = NT >= 256 ? 4 : 9;        // slots per thread and pass (one wave: 17 row tiles in two passes)
    const int total = ntiles * 64;
    for (int s0 = tid; s0 < total; s0 += NT * U) {
        float x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int slot = s0 + NT * u < total ? s0 + NT * u : total - 1;
            const int row = (slot >> 6) * 32 + (slot & 31), rc = row < n ? row : n - 1;
            x[u] = px[rc]; y[u] = py[rc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int slot = s0 + NT * u;
            if (slot < total) s_dst[slot] = mx_fragment<true>(S * x[u], S * y[u], (slot & 63) >> 5);
        }
    }
}


typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

// coordinate half (8 bytes) of one column's fragment from the rotated, scaled point: (x1, y1, x2, y2) -- the hi pieces of
// both coordinates in one packed conversion, their residuals in another.  X, Y and the hi pieces are opaque to the
// optimiser for the reason given at mx_split.
static __device__ __forceinline__ h4v mx_col_coords(float X, float Y)
{
    asm volatile("" : "+v"(X), "+v"(Y));
    h2v h = __builtin_convertvector(f2v{X, Y}, h2v);
    unsigned hb = __builtin_bit_cast(unsigned, h);
    asm volatile("" : "+v"(hb));
    h = __builtin_bit_cast(h2v, hb);
    const h2v l = __builtin_convertvector(f2v{X - (float)h.x, Y - (float)h.y}, h2v);
    return h4v{h.x, h.y, l.x, l.y};
}

// norm half (8 bytes) of a column's fragment: (256, 1, n2h, n2l), n2 = |point|^2 of the scaled, UNROTATED point -- a
// rotation keeps it (to the few units of rounding that PairDesc::e2 budgets for), so it is built once per work item
static __device__ __forceinline__ h4v mx_col_norm(float X, float Y)
{
    const float n2 = __builtin_fmaf(X, X, Y * Y);
    const _Float16 nh = (_Float16)(n2 * 0.00390625f);
    const _Float16 nl = (_Float16)__builtin_fmaf(-256.0f, (float)nh, n2);
    return h4v{(_Float16)256.0f, (_Float16)1.0f, nh, nl};
}

template <int CTRL>
static __device__ __forceinline__ int dpp_max_i32(int v)
{
    const int o = __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, 0xF, 0xF, false);
    return o > v ? o : v;
}
// max over the wave without an LDS round trip: four DPP steps inside each row of 16 lanes, then the four rows as scalars
static __device__ __forceinline__ int wave_max_i32_dpp(int v)
{
    v = dpp_max_i32<0xB1>(v); v = dpp_max_i32<0x4E>(v); v = dpp_max_i32<0x141>(v); v = dpp_max_i32<0x140>(v);
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int a = r0 > r1 ? r0 : r1, c = r2 > r3 ? r2 : r3;
    return a > c ? a : c;
}

// The generated main phase in its two forms (tools/gen_screen_mx.py): SINGLE -- one MFMA per tile, the row minima cross the
// lanes through an LDS transpose once per row tile -- and DUAL -- two MFMAs per tile (D and its transpose), both directed
// minima in-lane, no row reduction.  Column-tile counts up to MM_MX_DUAL_MAX take the dual form: it wins where a row tile has
// too few column tiles to amortise the single form's reduction (ns per tile per SIMD, dual / single: 64 points 80 / 127,
// 96: 65 / 82, 128: 57 / 68, 160: 55 / 57, 192: 54 / 54, 208: 53.5 / 51, 521: 51 / 43 -- tools/exp_mx4.sh); from six column
// tiles on the second MFMA per tile costs more than the reduction it saves (two waves' four MFMAs per tile pair keep the
// matrix pipe busy 128 of 160 cycles, and a wave that finds it busy stalls its minima too).
#ifndef MM_MX_DUAL_MAX
#define MM_MX_DUAL_MAX 5
#endif
template <int NCT, bool CARRY>
static __device__ __forceinline__ int mx_main(unsigned vB, unsigned vA, unsigned vRW, unsigned vRR, int nloop, int tail, unsigned vRS)
{
    if constexpr (NCT <= MM_MX_DUAL_MAX) return MxMainD<NCT, CARRY>::run(vB, vA, nloop, tail, vRS);
    else return MxMain<NCT, CARRY>::run(vB, vA, vRW, vRR, 0u, nloop, tail, vRS);
}
template <int NCT>
static __device__ __forceinline__ int mx_emit(unsigned vB, unsigned vA, unsigned vRW, unsigned vRR, int nloop, int tail, unsigned vRS,
                                              unsigned vCS)
{
    if constexpr (NCT <= MM_MX_DUAL_MAX) return MxEmitD<NCT>::run(vB, vA, nloop, tail, vRS, vCS);
    else return MxEmit<NCT>::run(vB, vA, vRW, vRR, 0u, nloop, tail, vRS, vCS);
}

// unsigned minimum over the wave (order-reversing map onto the signed maximum above)
static __device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v)
{
    return ~(unsigned)wave_max_i32_dpp((int)~(v ^ 0x80000000u)) ^ 0x80000000u;
}

// NCT: column tiles of one block (the whole target set when !MULTI); a_cap: row tiles the launch's LDS layout provides for.
// WAVES: waves of a workgroup, one candidate each at a time.  4 for the host-built work lists (8 consecutive candidates of
// a pair share the staged rows); 1 for the device queues of the bounded search, whose items are single candidates (the
// picks) or short runs: with 4, three waves of every workgroup sat idle and the launch took four rounds of workgroups.
template <int NCT, bool MULTI, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES, 2)
k_screen_mx(const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ work, int n_work_host,
            const int* __restrict__ n_work_dev, const int32_t* __restrict__ sel, int a_cap, const float* __restrict__ ptx,
            const float* __restrict__ pty, const float* __restrict__ cosv, const float* __restrict__ sinv, float* __restrict__ out_sq)
{
    constexpr int NB = NCT * 32;                                      // padded columns of one block
    constexpr int NQ = (NCT + 1) / 2;                                 // columns per lane
    extern __shared__ __align__(16) unsigned char smem[];
    h8v* s_a = reinterpret_cast<h8v*>(smem);                          // [a_cap][64] row fragments of the pair (all waves)
    h4v* s_bw = reinterpret_cast<h4v*>(s_a + a_cap * 64);            // [WAVES][NCT][64] column fragments, 8 bytes each: a wave's
                                                                      //            own copy, for the candidate it is on
    int* s_redx = reinterpret_cast<int*>(s_bw + WAVES * NCT * 64);   // [WAVES][MX_RED] row-reduction scratch, one per wave
    int* s_rsx = s_redx + WAVES * MX_RED;                             // MULTI: [WAVES][a_cap * 32] row store, one per wave
    float* s_cs = reinterpret_cast<float*>(s_rsx + (MULTI ? WAVES * a_cap * 32 : 0));   // [MX_ITEM][2] cos, sin of the work item's candidates
    int* s_ci = reinterpret_cast<int*>(s_cs + 2 * MX_ITEM);           // [MX_ITEM] their indices in the pair's list

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, hi = lane >> 5;
    h4v* s_b = s_bw + wave * NCT * 64;
    int* s_rs = s_rsx + wave * a_cap * 32;
    // LDS byte addresses of this lane's slots (generator docstring); a generic pointer's low 32 bits are its LDS offset
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned vB = lds0 + (unsigned)((size_t)s_b - (size_t)smem) + lane * 8;
    const unsigned vA = lds0 + (unsigned)((size_t)s_a - (size_t)smem) + lane * 16;
    const unsigned red_w = lds0 + (unsigned)((size_t)s_redx - (size_t)smem) + wave * MX_RED * 4;
    const unsigned vRW = red_w + (hi * 16) * MM_SCREEN_MX_RED_STRIDE + l32 * 4;
    const int rh = (l32 >> 2) & 1, rv = (l32 & 3) + 4 * (l32 >> 3);        // row l32 of a tile lives at (half rh, element rv)
    const unsigned vRR = red_w + (rh * 16 + rv) * MM_SCREEN_MX_RED_STRIDE + hi * 64;
    const unsigned vRS = lds0 + (unsigned)((size_t)s_rs - (size_t)smem) + l32 * 4;   // MULTI: both halves hold the same value
    const int n_work = n_work_dev ? *n_work_dev : n_work_host;   // device queue: a bounded grid strides over it

    for (int wi = (int)gridDim.x == n_work ? xcd_work_index(blockIdx.x, n_work) : (int)blockIdx.x; wi < n_work;
         wi += gridDim.x) {
        const WorkItem w = work[wi];
        const PairDesc pd = pairs[w.pair];
        const int na = pd.n_ref, nb = pd.n_tgt;
        const int nrt = (na + 31) >> 5;                                 // row tiles (<= a_cap: the engine sized the launch)
        const int nloop = (nrt - 1) >> 1, tail = (nrt - 1) & 1;         // the asm block: row tile 0, nloop x 2, (tail)
        const float S = __builtin_ldexpf(1.0f, pd.pad0), inv_s2 = __builtin_ldexpf(1.0f, -2 * pd.pad0);

        __syncthreads();
        // the candidates' cos / sin once per work item (a global load at the top of every candidate would be exposed).
        // sel (the survivors of a bounded search): the item is entries a0 .. a0 + cnt of the pair's list of candidates,
        // not a run of consecutive ones
        for (int t = tid; t < 2 * w.cnt; t += 64 * WAVES) {
            const int a = sel ? sel[pd.out_off + w.a0 + (t >> 1)] : w.a0 + (t >> 1);
            s_cs[t] = (t & 1) ? sinv[pd.tab_off + a] : cosv[pd.tab_off + a];
            if (!(t & 1)) s_ci[t >> 1] = a;
        }
        mx_stage_rows<64 * WAVES>(s_a, nrt, na, ptx + pd.ref_off, pty + pd.ref_off, S, tid);      // padding rows duplicate the last reference point
        if constexpr (!MULTI) {
            // this lane's columns -- every wave holds all of them: lane + 64 q -- unrotated, scaled, in registers for all the
            // wave's candidates
            float tx[NQ], ty[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int j = lane + 64 * q;
                const int jc = j < nb ? j : nb - 1;          // padding columns duplicate the last point (see k_screen_fast)
                tx[q] = S * ptx[pd.tgt_off + jc]; ty[q] = S * pty[pd.tgt_off + jc];
                // the norm half of the column's fragment does not depend on the candidate: written here, read by all of them
                if (j < NB) s_b[(j >> 5) * 64 + (j & 31) + 32] = mx_col_norm(tx[q], ty[q]);
            }
            __syncthreads();

            // one wave, one candidate: no barrier and nothing shared below this line
            for (int k = wave; k < w.cnt; k += WAVES) {
                const float c = s_cs[2 * k], s = s_cs[2 * k + 1];
                const int a = s_ci[k];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int j = lane + 64 * q;
                    if (j < NB) {
                        const float bx = __builtin_fmaf(tx[q], c, -(ty[q] * s));     // k_screen_fast's rotation, on scaled coordinates
                        const float by = __builtin_fmaf(tx[q], s, ty[q] * c);
                        s_b[(j >> 5) * 64 + (j & 31)] = mx_col_coords(bx, by);
                    }
                }
                // (LDS operations of one wave execute in order: the block's reads below see these writes)
                int m = mx_main<NCT, false>(vB, vA, vRW, vRR, nloop, tail, vRS);
                m = wave_max_i32_dpp(m);
                if (lane == 0) out_sq[pd.out_off + a] = __int_as_float(m) * inv_s2;
            }
        } else {
            __syncthreads();
            const int nblk = (((nb + 31) >> 5) + NCT - 1) / NCT;         // equal blocks of NCT column tiles (the engine's choice)
            for (int k = wave; k < w.cnt; k += WAVES) {
                const float c = s_cs[2 * k], s = s_cs[2 * k + 1];
                const int a = s_ci[k];
                for (int i = lane; i < nrt * 32; i += 64) s_rs[i] = 0x7f800000;      // row store: +inf
                int m = 0;
                for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int j = lane + 64 * q;
                        if (j < NB) {
                            const int jg = blk * NB + j;
                            const int jc = jg < nb ? jg : nb - 1;
                            const float x = S * ptx[pd.tgt_off + jc], y = S * pty[pd.tgt_off + jc];
                            const float bx = __builtin_fmaf(x, c, -(y * s));
                            const float by = __builtin_fmaf(x, s, y * c);
                            s_b[(j >> 5) * 64 + (j & 31) + 32] = mx_col_norm(x, y);
                            s_b[(j >> 5) * 64 + (j & 31)] = mx_col_coords(bx, by);
                        }
                    }
                    const int mb = mx_main<NCT, true>(vB, vA, vRW, vRR, nloop, tail, vRS);   // max of the block's column minima
                    m = mb > m ? mb : m;
                }
                for (int i = lane; i < nrt * 32; i += 64) { const int r = s_rs[i]; m = r > m ? r : m; }
                m = wave_max_i32_dpp(m);
                if (lane == 0) out_sq[pd.out_off + a] = __int_as_float(m) * inv_s2;
            }
        }
    }
}

// The first pick of a bounded search: one wave, one candidate, and besides its screened value the candidate's ROW minima
// (per reference point) and COLUMN minima (per target point) leave for k_lb_topk -- the generated block in its `emit` form
// (row store + column store in LDS).  Same values as k_screen_mx<NCT, false> (same operands, same tiles).
template <int NCT>
__global__ void __launch_bounds__(64, 2)
k_screen_mx_emit(const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ work, const int* __restrict__ n_work_dev, int a_cap,
                 const float* __restrict__ ptx, const float* __restrict__ pty, const float* __restrict__ cosv,
                 const float* __restrict__ sinv, float* __restrict__ out_sq, float* __restrict__ emit, int emit_rows, int emit_cols)
{
    constexpr int NB = NCT * 32, NQ = (NCT + 1) / 2;
    extern __shared__ __align__(16) unsigned char smem[];
    h8v* s_a = reinterpret_cast<h8v*>(smem);                          // [a_cap][64] row fragments
    h4v* s_b = reinterpret_cast<h4v*>(s_a + a_cap * 64);             // [NCT][64] column fragments
    int* s_red = reinterpret_cast<int*>(s_b + NCT * 64);             // [MX_RED] row-reduction scratch
    int* s_rs = s_red + MX_RED;                                       // [a_cap * 32] row store
    int* s_col = s_rs + a_cap * 32;                                   // [NQ * 64] column store
    const int lane = threadIdx.x, l32 = lane & 31, hi = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned vB = lds0 + (unsigned)((size_t)s_b - (size_t)smem) + lane * 8;
    const unsigned vA = lds0 + lane * 16;
    const unsigned red_w = lds0 + (unsigned)((size_t)s_red - (size_t)smem);
    const unsigned vRW = red_w + (hi * 16) * MM_SCREEN_MX_RED_STRIDE + l32 * 4;
    const int rh = (l32 >> 2) & 1, rv = (l32 & 3) + 4 * (l32 >> 3);
    const unsigned vRR = red_w + (rh * 16 + rv) * MM_SCREEN_MX_RED_STRIDE + hi * 64;
    const unsigned vRS = lds0 + (unsigned)((size_t)s_rs - (size_t)smem) + l32 * 4;
    const unsigned vCS = lds0 + (unsigned)((size_t)s_col - (size_t)smem) + lane * 4;
    const int n_work = *n_work_dev;

    for (int wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const WorkItem w = work[wi];
        const PairDesc pd = pairs[w.pair];
        const int na = pd.n_ref, nb = pd.n_tgt, a = w.a0;
        const int nrt = (na + 31) >> 5, nloop = (nrt - 1) >> 1, tail = (nrt - 1) & 1;
        const float S = __builtin_ldexpf(1.0f, pd.pad0), inv_s2 = __builtin_ldexpf(1.0f, -2 * pd.pad0);
        const float c = cosv[pd.tab_off + a], s = sinv[pd.tab_off + a];
        __syncthreads();
        mx_stage_rows<64>(s_a, nrt, na, ptx + pd.ref_off, pty + pd.ref_off, S, lane);
        for (int i = lane; i < nrt * 32; i += 64) s_rs[i] = 0x7f800000;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = lane + 64 * q;
            if (j < NB) {
                const int jc = j < nb ? j : nb - 1;
                const float x = S * ptx[pd.tgt_off + jc], y = S * pty[pd.tgt_off + jc];
                s_b[(j >> 5) * 64 + (j & 31) + 32] = mx_col_norm(x, y);
                s_b[(j >> 5) * 64 + (j & 31)] = mx_col_coords(__builtin_fmaf(x, c, -(y * s)), __builtin_fmaf(x, s, y * c));
            }
        }
        __syncthreads();
        int m = mx_emit<NCT>(vB, vA, vRW, vRR, nloop, tail, vRS, vCS);     // max of the column minima
        __syncthreads();
        float* const em = emit + (size_t)w.pair * (size_t)(emit_rows + emit_cols);
        for (int i = lane; i < nrt * 32; i += 64) {
            const int r = s_rs[i];
            m = r > m ? r : m;
            if (i < na) em[i] = __int_as_float(r) * inv_s2;
        }
        for (int j = lane; j < nb; j += 64) em[emit_rows + j] = __int_as_float(s_col[j]) * inv_s2;
        m = wave_max_i32_dpp(m);
        if (lane == 0) out_sq[pd.out_off + a] = __int_as_float(m) * inv_s2;
    }
}

template <int NCT>
static hipError_t launch_mx_emit_t(const BatchDev& b, const WorkItem* work, const int* n_dev, int cap, int a_cap, hipStream_t s)
{
    const size_t lds = (size_t)a_cap * 64 * 16 + (size_t)NCT * 64 * 8 + (size_t)MX_RED * 4 + (size_t)a_cap * 32 * 4 +
                       (size_t)((NCT + 1) / 2) * 64 * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_screen_mx_emit<NCT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_screen_mx_emit<NCT>), dim3(std::min(cap, 256 * 8 * 2)), dim3(64), lds, s, b.pairs, work, n_dev, a_cap, b.p32x,
                       b.p32y, b.cos32, b.sin32, b.sq32, b.emit, b.emit_rows, b.emit_cols);
    return hipGetLastError();
}

static hipError_t launch_mx_emit(const BatchDev& b, const WorkItem* work, const int* n_dev, int cap, int nct, int a_cap, hipStream_t s)
{
    if (a_cap < 1 || a_cap > 17) return hipErrorInvalidValue;
#define MM_MXE(N) case N: return launch_mx_emit_t<N>(b, work, n_dev, cap, a_cap, s);
    switch (nct) {
        MM_MXE(2) MM_MXE(3) MM_MXE(4) MM_MXE(5) MM_MXE(6) MM_MXE(7) MM_MXE(8) MM_MXE(9) MM_MXE(10) MM_MXE(11) MM_MXE(12) MM_MXE(13)
        MM_MXE(14) MM_MXE(15) MM_MXE(16) MM_MXE(17)
        default: return hipErrorInvalidValue;
    }
#undef MM_MXE
}

size_t lds_bytes_mx(int nct, bool multi, int a_cap, int waves)
{
    return (size_t)a_cap * 64 * 16 + (size_t)waves * nct * 64 * 8 + (size_t)waves * MX_RED * 4 +
           (multi ? (size_t)waves * a_cap * 32 * 4 : 0) + (size_t)MX_ITEM * 12;
}

// n_dev == nullptr: one workgroup per item of the host-built list; else a device queue of at most `cap` items whose length is
// *n_dev (a bounded grid strides over it)
// waves: 4, or 1 for a queue of single candidates (the picks of a bounded search).  sel: see the kernel.
template <int NCT, bool MULTI>
static hipError_t launch_mx_t(const BatchDev& b, const WorkItem* work, int n_work, const int* n_dev, int cap, int a_cap, int waves,
                              const int32_t* sel, hipStream_t s)
{
    if constexpr (!MULTI) {
        if (waves == 1) {
            const size_t lds = lds_bytes_mx(NCT, false, a_cap, 1);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_screen_mx<NCT, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            const int grid = n_dev ? std::min(cap, 256 * 8 * 2) : n_work;
            hipLaunchKernelGGL((k_screen_mx<NCT, false, 1>), dim3(grid), dim3(64), lds, s, b.pairs, work, n_work, n_dev, sel, a_cap,
                               b.p32x, b.p32y, b.cos32, b.sin32, b.sq32);
            return hipGetLastError();
        }
    }
    const size_t lds = lds_bytes_mx(NCT, MULTI, a_cap, 4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_screen_mx<NCT, MULTI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int grid = n_dev ? std::min(cap, 256 * 8) : n_work;
    hipLaunchKernelGGL((k_screen_mx<NCT, MULTI>), dim3(grid), dim3(256), lds, s, b.pairs, work, n_work, n_dev, sel, a_cap, b.p32x, b.p32y,
                       b.cos32, b.sin32, b.sq32);
    return hipGetLastError();
}

// the variant for a target set of nb points: column tiles per block, and whether the set takes several blocks
void mx_variant(int nb, int* nct, int* multi)
{
    const int nbt = (nb + 31) / 32;
    if (nbt <= MX_TMAX) { *nct = nbt < MM_SCREEN_MX_NCT_MIN ? MM_SCREEN_MX_NCT_MIN : nbt; *multi = 0; return; }
    const int nblk = (nbt + MX_TMAX - 1) / MX_TMAX;
    *nct = (nbt + nblk - 1) / nblk; *multi = 1;
}

// work[0 .. n_work) of b.work + work_begin: items of pairs that all take the variant (nct, multi); a_cap >= their row tiles
static hipError_t launch_mx_any(const BatchDev& b, const WorkItem* w, int n_work, const int* n_dev, int cap, int nct, int multi, int a_cap,
                                hipStream_t s, int waves = 4, const int32_t* sel = nullptr)
{
    if (a_cap < 1 || a_cap > MX_ROW_TILES_MAX || lds_bytes_mx(nct, multi != 0, a_cap, 4) > 160 * 1024) return hipErrorInvalidValue;
#define MM_MX(N) case N: return multi ? launch_mx_t<(N < 9 ? 9 : N), true>(b, w, n_work, n_dev, cap, a_cap, 4, sel, s) \
                                      : launch_mx_t<N, false>(b, w, n_work, n_dev, cap, a_cap, waves, sel, s);
    if (multi && nct < 9) return hipErrorInvalidValue;       // (several blocks: at least 18 column tiles, 9 per block)
    switch (nct) {
        MM_MX(2) MM_MX(3) MM_MX(4) MM_MX(5) MM_MX(6) MM_MX(7) MM_MX(8) MM_MX(9) MM_MX(10) MM_MX(11) MM_MX(12) MM_MX(13)
        MM_MX(14) MM_MX(15) MM_MX(16) MM_MX(17)
        default: return hipErrorInvalidValue;
    }
#undef MM_MX
}

hipError_t launch_screen_mx(const BatchDev& b, int work_begin, int n_work, int nct, int multi, int a_cap, hipStream_t s)
{
    if (n_work <= 0) return hipSuccess;
    return launch_mx_any(b, b.work + work_begin, n_work, nullptr, 0, nct, multi, a_cap, s);
}
int mx_min_points() { return 64; }
int mx_max_points() { return MX_ROW_TILES_MAX * 32; }

// -------------------------------------------------------------------------------------
// Bounded screen.  For subsets A' of the reference set and B' of the target set,
//     L = max( max_{a in A'} min_{b in B} d(a,b),  max_{b in B'} min_{a in A} d(a,b) )  <=  H(A,B):
// both terms take the outer max of a directed distance over fewer points and the inner min over
// all of them, so L never exceeds the Hausdorff distance, for any subsets.  With every k-th point
// (k = stride) it costs 2/k of the full distance matrix and, the contours being smooth, lies
// within a few percent of H -- close enough that one full evaluation per pair (at the candidate
// with the smallest L) gives an upper bound that rules out ~97-99 % of the candidates.  The
// survivors go through k_screen_fast and the exact f64 re-score as before, so winners are
// unchanged; L only decides what is NOT evaluated, under the same error bounds (PairDesc::delta,
// e2) that relate every f32 squared distance to its f64 value.
//
// One wave scores one candidate at a time (no workgroup barrier in the candidate loop) with the same
// primitive run twice -- a small query set in registers (16 column lanes x 4 row groups x 2 RP rows) against
// a large set that stays in LDS for the whole work item:
//     e(q,p) = |p|^2 - 2 q.p   (2 v_pk_fma_f32 + 1 v_min3_f32 per two distances; + |q|^2 after the min)
// Only the <= 8 RP queries are rotated per candidate, never the large set: d(a, R b) = d(R^-1 a, b), so
// pass 1 takes the queries R^-1 A' against the unrotated target (whose |b|^2 does not depend on the
// angle) and pass 2 the queries R B' against the reference.  Rotating the reference instead of the target
// swaps the roles of rho_r and rho_t in the rounding analysis of section 5 of DESIGN.md; PairDesc::delta
// (24 u (rho_r + rho_t)) covers both.  (Tried: two candidates per wave sharing every LDS read -- 20 % fewer
// instructions, 202 instead of 146 VGPRs, one wave per SIMD less: no gain.)
// -------------------------------------------------------------------------------------
static constexpr int kLbRP = 5;   // row PAIRS per row group: query sets up to 8 * kLbRP points

static __device__ __forceinline__ float fmin3_f32(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // finite inputs only
    return r;
}

// max over this wave's queries of (min over the n_groups*16 points of e) + |q|^2, as an int bit
// pattern floored at +0 (same convention as k_screen_fast); identical in all 64 lanes
template <int RP>
static __device__ __forceinline__ int lb_pass(const v2f (&qx)[RP], const v2f (&qy)[RP], const v2f (&q2)[RP],
                                              const float4* __restrict__ s_p, int n_groups, int lj)
{
    v2f rm[RP];
#pragma unroll
    for (int q = 0; q < RP; ++q) rm[q] = (v2f)(__int_as_float(0x7f800000));
    float4 p0 = s_p[lj], p1 = s_p[16 + lj];
    for (int k = 0; k < n_groups; k += 2) {   // n_groups is even (sets are padded to 32 points)
        // the next two column groups are fetched while this pair is evaluated (the last round re-reads itself)
        const int kn = k + 2 < n_groups ? k + 2 : k;
        const float4 n0 = s_p[kn * 16 + lj];
        const float4 n1 = s_p[(kn + 1) * 16 + lj];
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            const v2f t0 = __builtin_elementwise_fma(qy[q], (v2f)(p0.y), (v2f)(p0.z));
            const v2f t1 = __builtin_elementwise_fma(qy[q], (v2f)(p1.y), (v2f)(p1.z));
            const v2f e0 = __builtin_elementwise_fma(qx[q], (v2f)(p0.x), t0);
            const v2f e1 = __builtin_elementwise_fma(qx[q], (v2f)(p1.x), t1);
            rm[q].x = fmin3_f32(rm[q].x, e0.x, e1.x);
            rm[q].y = fmin3_f32(rm[q].y, e0.y, e1.y);
        }
        p0 = n0; p1 = n1;
    }
    int m = 0;
#pragma unroll
    for (int q = 0; q < RP; ++q) {
        const v2f d = rm[q] + q2[q];
        int v = __float_as_int(d.x);
        v = dpp_min_i32<0xB1>(v); v = dpp_min_i32<0x4E>(v); v = dpp_min_i32<0x141>(v); v = dpp_min_i32<0x140>(v);
        m = v > m ? v : m;
        v = __float_as_int(d.y);
        v = dpp_min_i32<0xB1>(v); v = dpp_min_i32<0x4E>(v); v = dpp_min_i32<0x141>(v); v = dpp_min_i32<0x140>(v);
        m = v > m ? v : m;
    }
    int o = __shfl_xor(m, 16, 64); m = o > m ? o : m;
    o = __shfl_xor(m, 32, 64); m = o > m ? o : m;
    return m;
}

// LIST: the 8 RP queries of either side are the point indices qlist[pair * 16 RP + (0..8RP-1 reference,
// 8RP.. target)] instead of every stride-th point, and the result is merged into out_lb by maximum.
template <int RP, bool LIST>
__global__ void __launch_bounds__(256, 3)
k_screen_lb(const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ work,
            int n_work_host, const int* __restrict__ n_work_dev, int stride, const int32_t* __restrict__ qlist,
            const float* __restrict__ ptx, const float* __restrict__ pty,
            const float* __restrict__ cosv, const float* __restrict__ sinv, float* __restrict__ out_lb)
{
    constexpr int NT = 256;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, li = lane >> 4;
    const int n_work = n_work_dev ? *n_work_dev : n_work_host;   // device queue: workgroups stride over it

    for (int wi = (int)gridDim.x == n_work ? xcd_work_index(blockIdx.x, n_work) : (int)blockIdx.x; wi < n_work;
         wi += gridDim.x) {
        const WorkItem w = work[wi];
        const PairDesc pd = pairs[w.pair];
        const int na = pd.n_ref, nb = pd.n_tgt;
        const int nap = (na + 31) & ~31, nbp = (nb + 31) & ~31;
        const int qa = LIST ? 8 * RP : (na + stride - 1) / stride;   // <= 8 RP (host)
        const int qb = LIST ? 8 * RP : (nb + stride - 1) / stride;
        const int32_t* ql = LIST ? qlist + (size_t)w.pair * (16 * RP) : nullptr;

        float4* s_a = reinterpret_cast<float4*>(smem);   // (ax, ay, |a|^2, 0)
        float4* s_t = s_a + nap;                          // (tx, ty, |t|^2, 0): the target as staged, never rotated

        __syncthreads();   // the previous item's readers are done
        // padding entries duplicate the last point: no effect on a minimum over the set
        for (int i = tid; i < nap; i += NT) {
            const int ic = i < na ? i : na - 1;
            const float x = ptx[pd.ref_off + ic], y = pty[pd.ref_off + ic];
            s_a[i] = make_float4(x, y, __builtin_fmaf(x, x, y * y), 0.0f);
        }
        for (int j = tid; j < nbp; j += NT) {
            const int jc = j < nb ? j : nb - 1;
            const float x = ptx[pd.tgt_off + jc], y = pty[pd.tgt_off + jc];
            s_t[j] = make_float4(x, y, __builtin_fmaf(x, x, y * y), 0.0f);
        }
        // the queries as staged: every stride-th point of either set (or the listed ones); rows past the subset
        // repeat its last point (no effect on the maximum over the subset)
        v2f ax[RP], ay[RP], a2[RP], bx[RP], by[RP], b2[RP];
#pragma unroll
        for (int q = 0; q < RP; ++q) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = li * (2 * RP) + 2 * q + h;
                const int ia = LIST ? ql[row] : (row < qa ? row : qa - 1) * stride;
                const int ib = LIST ? ql[8 * RP + row] : (row < qb ? row : qb - 1) * stride;
                const float xa = ptx[pd.ref_off + ia], ya = pty[pd.ref_off + ia];
                const float xb = ptx[pd.tgt_off + ib], yb = pty[pd.tgt_off + ib];
                const float na2 = __builtin_fmaf(xa, xa, ya * ya), nb2 = __builtin_fmaf(xb, xb, yb * yb);
                if (h) { ax[q].y = xa; ay[q].y = ya; a2[q].y = na2; bx[q].y = xb; by[q].y = yb; b2[q].y = nb2; }
                else   { ax[q].x = xa; ay[q].x = ya; a2[q].x = na2; bx[q].x = xb; by[q].x = yb; b2[q].x = nb2; }
            }
        }
        // the item's k-th candidate is a0 + k * step, clipped to the pair's last candidate (the sparse first
        // round scores every step-th candidate and the last one); this wave takes k = wave + 4 i.  Lane i
        // fetches the i-th one's cos/sin now, so no candidate starts with a dependent global load
        const int step = w.pad > 0 ? w.pad : 1;
        float tab_c = 1.0f, tab_s = 0.0f;
        if (wave + 4 * lane < w.cnt) {
            const int al = min(w.a0 + (wave + 4 * lane) * step, pd.n_ang - 1);
            tab_c = cosv[pd.tab_off + al];
            tab_s = sinv[pd.tab_off + al];
        }
        __syncthreads();

        for (int i = 0, k = wave; k < w.cnt; ++i, k += 4) {
            const int a = min(w.a0 + k * step, pd.n_ang - 1);
            const float c = __shfl(tab_c, i, 64), sn = __shfl(tab_s, i, 64);
            int m1, m2;
            {   // pass 1: R^-1 A' against the target as staged.  x' = x c + y s, y' = y c - x s
                v2f qx[RP], qy[RP];
#pragma unroll
                for (int q = 0; q < RP; ++q) {
                    qx[q] = -2.0f * __builtin_elementwise_fma(ax[q], (v2f)(c), ay[q] * sn);
                    qy[q] = -2.0f * __builtin_elementwise_fma(ay[q], (v2f)(c), -(ax[q] * sn));
                }
                m1 = lb_pass<RP>(qx, qy, a2, s_t, nbp >> 4, lj);
            }
            {   // pass 2: R B' (the full screen's rotation, bit for bit) against the reference
                v2f qx[RP], qy[RP];
#pragma unroll
                for (int q = 0; q < RP; ++q) {
                    qx[q] = -2.0f * __builtin_elementwise_fma(bx[q], (v2f)(c), -(by[q] * sn));
                    qy[q] = -2.0f * __builtin_elementwise_fma(bx[q], (v2f)(sn), by[q] * c);
                }
                m2 = lb_pass<RP>(qx, qy, b2, s_a, nap >> 4, lj);
            }
            if (lane == 0) {
                float v = __int_as_float(m1 > m2 ? m1 : m2);
                if (LIST) { const float o = out_lb[pd.out_off + a]; v = o > v ? o : v; }   // +inf (ruled out) stays
                out_lb[pd.out_off + a] = v;
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// The same bound on the matrix pipe (BatchDev::lb_mx): the queries are the COLUMNS of one or two 32-wide column tiles
// (QT), built in registers -- a B operand's lanes 0..31 carry a query's rotated, split coordinates, lanes 32..63 its norm
// pieces, which do not depend on the candidate -- and the large set is the ROWS, staged once per work item in k_screen_mx's
// A-form fragments (1 KB per 32 points) and never rotated: pass 1 takes R^-1 A' against the target, pass 2 R B' against the
// reference, as above.  Per 32 x 32 tile one v_mfma_f32_32x32x16_f16 and 8 v_min3_i32 (only the column minima are wanted:
// a lane's 16 values all belong to ITS query), against 2.5 vector instructions per distance pair above: a candidate's
// 2 x 17 tiles cost ~340 issue slots instead of ~1300.  No LDS traffic per candidate except the row fragments; four
// workgroups per CU hide the MFMA latency of each other's waves (no hand-written schedule needed here: the minima of a tile
// are 8 instructions, the dependence is on the tile before).  Values carry k_screen_mx's error bound (the engine sets
// PairDesc::e2 to the larger of the two directions').
// -------------------------------------------------------------------------------------
// min over the nt row tiles at LDS address `rows` of the squared distances to this lane's query (column lane & 31) in each of
// the NB column fragments, both halves of the wave combined; int bit patterns (signed order: rounded values can fall a few
// units below zero).  The pass itself is a generated asm block (tools/gen_screen_mx.py, generate_bound): written in C the
// compiler puts the minima of a tile right behind that tile's MFMA without a wait state, and the hardware does not interlock.
template <int NB> struct MxBound;
template <> struct MxBound<1> {
    static __device__ __forceinline__ void run(unsigned rows, int nt, const h8v* bq, int* out)
    {
        int counter;
        asm volatile(MM_BOUND_MX_ASM_1 : "=&v"(out[0]), "=&s"(counter) : "v"(rows), "s"(nt - 1), "v"(bq[0]) : MM_BOUND_MX_CLOBBERS_1);
    }
};
template <> struct MxBound<2> {
    static __device__ __forceinline__ void run(unsigned rows, int nt, const h8v* bq, int* out)
    {
        int counter;
        asm volatile(MM_BOUND_MX_ASM_2 : "=&v"(out[0]), "=&v"(out[1]), "=&s"(counter) : "v"(rows), "s"(nt - 1), "v"(bq[0]), "v"(bq[1])
                     : MM_BOUND_MX_CLOBBERS_2);
    }
};
template <> struct MxBound<4> {
    static __device__ __forceinline__ void run(unsigned rows, int nt, const h8v* bq, int* out)
    {
        int counter;
        asm volatile(MM_BOUND_MX_ASM_4 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&s"(counter)
                     : "v"(rows), "s"(nt - 1), "v"(bq[0]), "v"(bq[1]), "v"(bq[2]), "v"(bq[3]) : MM_BOUND_MX_CLOBBERS_4);
    }
};
template <int NB>
static __device__ __forceinline__ void mx_bound_pass(unsigned rows, int nt, const h8v (&bq)[NB], int (&out)[NB])
{
    MxBound<NB>::run(rows, nt, bq, out);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int o = __shfl_xor(out[j], 32, 64);      // lanes l and l + 32: different rows of the same column
        out[j] = o < out[j] ? o : out[j];
    }
}

static __device__ __forceinline__ h8v mx_query_fragment(float X, float Y, h4v norm, int hi)
{
    const h4v c = mx_col_coords(X, Y);
    const h4v h = hi ? norm : c;
    return h8v{h.x, h.y, h.z, h.w, h.x, h.y, h.z, h.w};
}

// QT: column tiles of queries per side (32 QT queries); NC: candidates a wave scores at once (every row fragment read from
// LDS feeds QT x NC MFMAs); LIST: the queries are qlist[pair * 2 nq_list + (0 .. nq_list - 1 reference, nq_list .. target)]
// and the result is merged into out_lb by maximum.  a_cap: row tiles per set the launch's LDS layout provides for.
template <int QT, int NC, bool LIST>
__global__ void __launch_bounds__(256, (QT * NC >= 4 ? 2 : 3))
k_bound_mx(const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ work, int n_work_host,
           const int* __restrict__ n_work_dev, int stride, const int32_t* __restrict__ qlist, int nq_list, int a_cap,
           const float* __restrict__ ptx, const float* __restrict__ pty,
           const float* __restrict__ cosv, const float* __restrict__ sinv, float* __restrict__ out_lb)
{
    extern __shared__ __align__(16) unsigned char smem[];
    h8v* s_r = reinterpret_cast<h8v*>(smem);          // [a_cap][64] reference points as row fragments
    h8v* s_t = s_r + a_cap * 64;                      // [a_cap][64] target points as row fragments (as staged, never rotated)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, hi = lane >> 5;
    const int n_work = n_work_dev ? *n_work_dev : n_work_host;
    // LDS byte addresses of this lane's fragment in row tile 0 of either set (a generic pointer's low 32 bits are its LDS offset)
    const unsigned vR = (unsigned)(size_t)smem + lane * 16, vT = vR + (unsigned)a_cap * 1024u;

    for (int wi = (int)gridDim.x == n_work ? xcd_work_index(blockIdx.x, n_work) : (int)blockIdx.x; wi < n_work;
         wi += gridDim.x) {
        const WorkItem w = work[wi];
        const PairDesc pd = pairs[w.pair];
        const int na = pd.n_ref, nb = pd.n_tgt;
        const int nrt = (na + 31) >> 5, ntt = (nb + 31) >> 5;
        const float S = __builtin_ldexpf(1.0f, pd.pad0), inv_s2 = __builtin_ldexpf(1.0f, -2 * pd.pad0);
        const int qa = LIST ? nq_list : (na + stride - 1) / stride;      // <= 32 QT (host)
        const int qb = LIST ? nq_list : (nb + stride - 1) / stride;
        const int32_t* ql = LIST ? qlist + (size_t)w.pair * (2 * nq_list) : nullptr;

        __syncthreads();   // the previous item's readers are done
        mx_stage_rows(s_r, nrt, na, ptx + pd.ref_off, pty + pd.ref_off, S, tid);
        mx_stage_rows(s_t, ntt, nb, ptx + pd.tgt_off, pty + pd.tgt_off, S, tid);
        // this lane's queries, unrotated and scaled, and their norm pieces (candidate-independent); columns past the subset
        // repeat its last query (no effect on the maximum over the subset)
        float ax[QT], ay[QT], bx[QT], by[QT];
        h4v an[QT], bn[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int col = t * 32 + l32;
            const int ia = LIST ? ql[col < qa ? col : qa - 1] : (col < qa ? col : qa - 1) * stride;
            const int ib = LIST ? ql[nq_list + (col < qb ? col : qb - 1)] : (col < qb ? col : qb - 1) * stride;
            ax[t] = S * ptx[pd.ref_off + ia]; ay[t] = S * pty[pd.ref_off + ia];
            bx[t] = S * ptx[pd.tgt_off + ib]; by[t] = S * pty[pd.tgt_off + ib];
            an[t] = mx_col_norm(ax[t], ay[t]); bn[t] = mx_col_norm(bx[t], by[t]);
        }
        const int step = w.pad > 0 ? w.pad : 1;
        float tab_c = 1.0f, tab_s = 0.0f;
        if (wave + 4 * lane < w.cnt) {
            const int al = min(w.a0 + (wave + 4 * lane) * step, pd.n_ang - 1);
            tab_c = cosv[pd.tab_off + al];
            tab_s = sinv[pd.tab_off + al];
        }
        __syncthreads();

        for (int i = 0, k = wave; k < w.cnt; i += NC, k += 4 * NC) {
            // this wave's candidates k, k + 4, ... (NC of them; past the item's end the last one is scored again, not stored)
            float c[NC], sn[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int ij = (k + 4 * j < w.cnt) ? i + j : i;
                c[j] = __shfl(tab_c, ij, 64); sn[j] = __shfl(tab_s, ij, 64);
            }
            h8v bq[QT * NC];
            int m1[QT * NC], m2[QT * NC];
            // pass 1: R^-1 A' against the target as staged.  x' = x c + y s, y' = y c - x s
#pragma unroll
            for (int j = 0; j < NC; ++j)
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    bq[j * QT + t] = mx_query_fragment(__builtin_fmaf(ax[t], c[j], ay[t] * sn[j]), __builtin_fmaf(ay[t], c[j], -(ax[t] * sn[j])), an[t], hi);
            mx_bound_pass<QT * NC>(vT, ntt, bq, m1);
            // pass 2: R B' (the full screen's rotation) against the reference
#pragma unroll
            for (int j = 0; j < NC; ++j)
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    bq[j * QT + t] = mx_query_fragment(__builtin_fmaf(bx[t], c[j], -(by[t] * sn[j])), __builtin_fmaf(bx[t], sn[j], by[t] * c[j]), bn[t], hi);
            mx_bound_pass<QT * NC>(vR, nrt, bq, m2);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                int m = 0;
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    m = m1[j * QT + t] > m ? m1[j * QT + t] : m;
                    m = m2[j * QT + t] > m ? m2[j * QT + t] : m;
                }
                m = wave_max_i32_dpp(m);
                if (lane == 0 && k + 4 * j < w.cnt) {
                    const int a = min(w.a0 + (k + 4 * j) * step, pd.n_ang - 1);
                    float v = __int_as_float(m) * inv_s2;
                    if (LIST) { const float o = out_lb[pd.out_off + a]; v = o > v ? o : v; }   // +inf (ruled out) stays
                    out_lb[pd.out_off + a] = v;
                }
            }
        }
    }
}

// The later rounds bound the candidates the rounds before could not rule out: a few per cent of a pair's list, scattered.
// As queue items (runs inside aligned groups of 8) every run of one to seven candidates staged both sets again -- 34 KB of
// row fragments for a few hundred MFMAs.  Here a workgroup takes one pair (one `split`-th of its candidate list), collects
// the flagged candidates (flags[], written by k_lb_spread / k_lb_keep) with their cos / sin into an LDS list, stages the sets
// ONCE -- and not at all where nothing is flagged -- and its four waves work the list off.
static constexpr int kScanChunk = 1024;      // candidates collected per pass over the list (LDS: 12 KB)

template <int QT, bool LIST>
__global__ void __launch_bounds__(256, 3)
k_bound_mx_scan(const PairDesc* __restrict__ pairs, int n_pairs, int split, const uint8_t* __restrict__ flags, int stride,
                const int32_t* __restrict__ qlist, int nq_list, int a_cap, const float* __restrict__ ptx,
                const float* __restrict__ pty, const float* __restrict__ cosv, const float* __restrict__ sinv,
                float* __restrict__ out_lb, unsigned long long* __restrict__ stats, int stat_slot)
{
    extern __shared__ __align__(16) unsigned char smem[];
    h8v* s_r = reinterpret_cast<h8v*>(smem);
    h8v* s_t = s_r + a_cap * 64;
    int* s_list = reinterpret_cast<int*>(s_t + a_cap * 64);          // [kScanChunk] candidate indices
    float* s_c = reinterpret_cast<float*>(s_list + kScanChunk);        // their cos
    float* s_s = s_c + kScanChunk;                                     // and sin
    __shared__ int s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, hi = lane >> 5;
    const unsigned vR = (unsigned)(size_t)smem + lane * 16, vT = vR + (unsigned)a_cap * 1024u;

    for (int wi = blockIdx.x; wi < n_pairs * split; wi += gridDim.x) {
        const int p = wi / split, part = wi - p * split;
        const PairDesc pd = pairs[p];
        if (pd.n_ang <= 0) continue;
        const int lo = (int)((long long)pd.n_ang * part / split), hi_a = (int)((long long)pd.n_ang * (part + 1) / split);
        const int na = pd.n_ref, nb = pd.n_tgt;
        const int nrt = (na + 31) >> 5, ntt = (nb + 31) >> 5;
        const float S = __builtin_ldexpf(1.0f, pd.pad0), inv_s2 = __builtin_ldexpf(1.0f, -2 * pd.pad0);
        bool staged = false;
        float ax[QT], ay[QT], bx[QT], by[QT];
        h4v an[QT], bn[QT];
        for (int base = lo; base < hi_a; base += kScanChunk) {
            __syncthreads();       // the list's readers of the chunk (or pair) before are done
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            for (int a = base + tid; a < min(base + kScanChunk, hi_a); a += 256)
                if (flags[pd.out_off + a]) {
                    const int k = atomicAdd(&s_cnt, 1);
                    s_list[k] = a; s_c[k] = cosv[pd.tab_off + a]; s_s[k] = sinv[pd.tab_off + a];
                }
            __syncthreads();
            const int n = s_cnt;
            if (n == 0) continue;
            if (stats && tid == 0) atomicAdd(&stats[stat_slot], (unsigned long long)n);
            if (!staged) {
                staged = true;
                mx_stage_rows(s_r, nrt, na, ptx + pd.ref_off, pty + pd.ref_off, S, tid);
                mx_stage_rows(s_t, ntt, nb, ptx + pd.tgt_off, pty + pd.tgt_off, S, tid);
                const int qa = LIST ? nq_list : (na + stride - 1) / stride;
                const int qb = LIST ? nq_list : (nb + stride - 1) / stride;
                const int32_t* ql = LIST ? qlist + (size_t)p * (2 * nq_list) : nullptr;
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const int col = t * 32 + l32;
                    const int ia = LIST ? ql[col < qa ? col : qa - 1] : (col < qa ? col : qa - 1) * stride;
                    const int ib = LIST ? ql[nq_list + (col < qb ? col : qb - 1)] : (col < qb ? col : qb - 1) * stride;
                    ax[t] = S * ptx[pd.ref_off + ia]; ay[t] = S * pty[pd.ref_off + ia];
                    bx[t] = S * ptx[pd.tgt_off + ib]; by[t] = S * pty[pd.tgt_off + ib];
                    an[t] = mx_col_norm(ax[t], ay[t]); bn[t] = mx_col_norm(bx[t], by[t]);
                }
                __syncthreads();
            }
            for (int k = wave; k < n; k += 4) {
                const int a = s_list[k];
                const float c = s_c[k], sn = s_s[k];
                h8v bq[QT];
                int m1[QT], m2[QT];
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    bq[t] = mx_query_fragment(__builtin_fmaf(ax[t], c, ay[t] * sn), __builtin_fmaf(ay[t], c, -(ax[t] * sn)), an[t], hi);
                mx_bound_pass<QT>(vT, ntt, bq, m1);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    bq[t] = mx_query_fragment(__builtin_fmaf(bx[t], c, -(by[t] * sn)), __builtin_fmaf(bx[t], sn, by[t] * c), bn[t], hi);
                mx_bound_pass<QT>(vR, nrt, bq, m2);
                int m = 0;
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    m = m1[t] > m ? m1[t] : m;
                    m = m2[t] > m ? m2[t] : m;
                }
                m = wave_max_i32_dpp(m);
                if (lane == 0) {
                    float v = __int_as_float(m) * inv_s2;
                    if (LIST) { const float o = out_lb[pd.out_off + a]; v = o > v ? o : v; }   // +inf (ruled out) stays
                    out_lb[pd.out_off + a] = v;
                }
            }
        }
    }
}

static constexpr int kLbCandStep = 8;   // first round: every 8th candidate (and the last) gets a bound

// number of candidates the sparse first round scores for a list of n: 0, 8, 16, ... and n - 1
static __host__ __device__ __forceinline__ int lb_sparse_count(int n) { return n <= 0 ? 0 : (n - 1 + kLbCandStep - 1) / kLbCandStep + 1; }

// Per pair: among the candidates scored by the sparse round, the one with the smallest bound
// (first one on ties) is queued for a full screen evaluation; its value is the pair's upper bound.
template <bool SPARSE>
__global__ void __launch_bounds__(256)
k_lb_pick(const PairDesc* __restrict__ pairs, const float* __restrict__ lb32, int32_t* __restrict__ pick_idx,
          WorkItem* __restrict__ items, int* __restrict__ n_items)
{
    __shared__ unsigned long long s_key;
    const int p = blockIdx.x, tid = threadIdx.x;
    const PairDesc pd = pairs[p];
    if (pd.n_ang <= 0) { if (tid == 0) pick_idx[p] = -1; return; }
    if (tid == 0) s_key = ~0ull;
    __syncthreads();
    unsigned long long key = ~0ull;
    const int ne = SPARSE ? lb_sparse_count(pd.n_ang) : pd.n_ang;   // later rounds: every candidate has a bound or +inf
    for (int i = tid; i < ne; i += 256) {
        const int a = SPARSE ? min(i * kLbCandStep, pd.n_ang - 1) : i;
        const unsigned long long k = ((unsigned long long)__float_as_uint(lb32[pd.out_off + a]) << 32) | (unsigned)a;
        key = k < key ? k : key;
    }
    {   // the wave's smallest key (value, then index) by two DPP reductions; one LDS atomic per wave, not per thread
        const unsigned hi = wave_min_u32_dpp((unsigned)(key >> 32));
        const unsigned lo = wave_min_u32_dpp((unsigned)(key >> 32) == hi ? (unsigned)key : 0xffffffffu);
        if ((tid & 63) == 0) atomicMin(&s_key, ((unsigned long long)hi << 32) | lo);
    }
    __syncthreads();
    if (tid == 0) {
        const int a = (int)(s_key & 0xffffffffull);
        pick_idx[p] = a;
        const int slot = atomicAdd(n_items, 1);
        WorkItem w; w.pair = p; w.a0 = a; w.cnt = 1; w.pad = 0;
        items[slot] = w;
    }
}

// After the sparse round.  Rotating the target by a further angle D moves each of its points by the
// chord 2 rho sin(D/2) <= rho_t * sqrt(2 - 2 cos D), so |H(c) - H(c')| <= chord(c, c'): a candidate
// between two scored ones inherits their bounds minus the chord.  If even that exceeds the pair's
// upper bound the candidate is ruled out without being looked at (+inf as its bound); otherwise it
// is queued for a bound of its own, as runs inside its aligned group of 8.
__global__ void __launch_bounds__(256)
k_lb_spread(const PairDesc* __restrict__ pairs, float* __restrict__ lb32, const float* __restrict__ sq32,
            const int32_t* __restrict__ pick_idx, const double* __restrict__ cos64, const double* __restrict__ sin64,
            WorkItem* __restrict__ items, int* __restrict__ n_items, unsigned long long* __restrict__ stats,
            uint8_t* __restrict__ flags)
{
    const int p = blockIdx.x, tid = threadIdx.x;
    const PairDesc pd = pairs[p];
    if (pd.n_ang <= 0) return;
    const int c1 = pick_idx[p], last = pd.n_ang - 1;
    const double ub = sqrt((double)sq32[pd.out_off + c1] + pd.e2) + 2.0 * pd.delta;
    const double* cs = cos64 + pd.tab_off;
    const double* sn = sin64 + pd.tab_off;
    unsigned long long queued = 0;
    for (int g = tid; g * kLbCandStep < pd.n_ang; g += 256) {
        const int lo = g * kLbCandStep, eR = min(lo + kLbCandStep, last);
        const double vL = (double)lb32[pd.out_off + lo] - pd.e2, vR = (double)lb32[pd.out_off + eR] - pd.e2;
        const double hL = sqrt(vL > 0.0 ? vL : 0.0), hR = sqrt(vR > 0.0 ? vR : 0.0);
        int first = -1, lastp = -1;
        for (int a = lo + 1; a < eR; ++a) {
            const double dL = 2.0 - 2.0 * (cs[a] * cs[lo] + sn[a] * sn[lo]);
            const double dR = 2.0 - 2.0 * (cs[a] * cs[eR] + sn[a] * sn[eR]);
            // 4e-8: the cancellation floor of 2 - 2 cos D in f64 (sqrt of two ulps of 2)
            const double chL = pd.rho_t * (sqrt(dL > 0.0 ? dL : 0.0) + 4e-8), chR = pd.rho_t * (sqrt(dR > 0.0 ? dR : 0.0) + 4e-8);
            const double inherited = fmax(hL - chL, hR - chR);
            if (inherited <= ub) { if (first < 0) first = a; lastp = a; if (flags) flags[pd.out_off + a] = 1; }
            else lb32[pd.out_off + a] = __int_as_float(0x7f800000);
        }
        if (first >= 0 && !flags) {      // (flags: k_bound_mx_scan collects the candidates itself)
            const int slot = atomicAdd(n_items, 1);
            WorkItem w; w.pair = p; w.a0 = first; w.cnt = lastp - first + 1; w.pad = 1;
            items[slot] = w;
            queued += (unsigned long long)w.cnt;
        }
    }
    if (stats && queued) atomicAdd(&stats[1], queued);
}

// Per pair: a candidate survives if the smallest exact cost its bound allows does not exceed the
// largest exact cost the picked candidate can have (same interval arithmetic as k_shortlist).
// Survivors are queued as runs inside aligned groups of 8 candidates; every other candidate gets
// +inf as its screened value, which k_shortlist never keeps.
// pick2 (nullable): a second fully screened candidate per pair; the smaller of the two values is the bound.
// FINAL: candidates that are not queued get +inf as their screened value; stats slot 2 counts the queued ones.
template <bool FINAL>
__global__ void __launch_bounds__(256)
k_lb_keep(const PairDesc* __restrict__ pairs, const float* __restrict__ lb32, float* __restrict__ sq32,
          const int32_t* __restrict__ pick_idx, const int32_t* __restrict__ pick2, WorkItem* __restrict__ items,
          int* __restrict__ n_items, unsigned long long* __restrict__ stats, uint8_t* __restrict__ flags)
{
    const int p = blockIdx.x, tid = threadIdx.x;
    const PairDesc pd = pairs[p];
    if (pd.n_ang <= 0) return;
    const int c1 = pick_idx[p], c2 = pick2 ? pick2[p] : c1;
    const float s1 = sq32[pd.out_off + c1], s2 = sq32[pd.out_off + c2];
    const double ub = sqrt((double)(s2 < s1 ? s2 : s1) + pd.e2) + 2.0 * pd.delta;
    unsigned long long queued = 0;
    for (int g = tid; g * 8 < pd.n_ang; g += 256) {
        const int lo = g * 8, hi = (lo + 8 < pd.n_ang) ? lo + 8 : pd.n_ang;
        int first = -1, last = -1;
        for (int a = lo; a < hi; ++a) {
            const double sv = (double)lb32[pd.out_off + a] - pd.e2;
            const bool stays = a == c1 || a == c2 || sqrt(sv > 0.0 ? sv : 0.0) <= ub;
            if (stays) { if (first < 0) first = a; last = a; }
            if (!FINAL && flags) flags[pd.out_off + a] = stays ? 1 : 0;     // every candidate of the pair: no stale flag survives
        }
        if (!FINAL && flags) continue;   // (k_bound_mx_scan collects the flagged candidates itself and counts them)
        if (FINAL)
            for (int a = lo; a < hi; ++a)
                if (a < first || a > last) sq32[pd.out_off + a] = __int_as_float(0x7f800000);
        if (first >= 0) {
            const int slot = atomicAdd(n_items, 1);
            WorkItem w; w.pair = p; w.a0 = first; w.cnt = last - first + 1; w.pad = 0;
            items[slot] = w;
            queued += (unsigned long long)w.cnt;
        }
    }
    if (stats && queued) atomicAdd(&stats[FINAL ? 2 : 3], queued);
}

// The same decision for the matrix-pipe rounds, one thread per candidate.  !FINAL: flags only (k_bound_mx_scan collects the
// flagged candidates itself).  FINAL: the pair's survivors are written to its slice of `klist` (any order) and queued as
// items of <= 8 list entries -- k_screen_mx's workgroup stages the pair's rows once and gives each of its four waves a
// survivor, wherever in the list of candidates they lie.  (As runs inside aligned groups of 8 -- k_lb_keep -- the survivors
// of a pair came in several items of one or two candidates, each a workgroup with two or three waves idle: 0.28 ms of a
// config3 step for 0.6 % of the candidates.)
template <bool FINAL>
__global__ void __launch_bounds__(256)
k_lb_keep_mx(const PairDesc* __restrict__ pairs, const float* __restrict__ lb32, float* __restrict__ sq32,
             const int32_t* __restrict__ pick_idx, const int32_t* __restrict__ pick2, int32_t* __restrict__ klist,
             WorkItem* __restrict__ items, int* __restrict__ n_items, unsigned long long* __restrict__ stats,
             uint8_t* __restrict__ flags)
{
    __shared__ int s_n;
    const int p = blockIdx.x, tid = threadIdx.x;
    const PairDesc pd = pairs[p];
    if (pd.n_ang <= 0) return;
    const int c1 = pick_idx[p], c2 = pick2 ? pick2[p] : c1;
    const float s1 = sq32[pd.out_off + c1], s2 = sq32[pd.out_off + c2];
    const double ub = sqrt((double)(s2 < s1 ? s2 : s1) + pd.e2) + 2.0 * pd.delta;
    if (tid == 0) s_n = 0;
    __syncthreads();              // (also: s1, s2 are read before any thread overwrites a screened value below)
    for (int a = tid; a < pd.n_ang; a += 256) {
        const double sv = (double)lb32[pd.out_off + a] - pd.e2;
        const bool stays = a == c1 || a == c2 || sqrt(sv > 0.0 ? sv : 0.0) <= ub;
        if (!FINAL) flags[pd.out_off + a] = stays ? 1 : 0;
        else if (stays) klist[pd.out_off + atomicAdd(&s_n, 1)] = a;
        else sq32[pd.out_off + a] = __int_as_float(0x7f800000);
    }
    if (FINAL) {
        __syncthreads();
        const int n = s_n;
        for (int k = tid; k * 8 < n; k += 256) {
            const int slot = atomicAdd(n_items, 1);
            WorkItem w; w.pair = p; w.a0 = 8 * k; w.cnt = n - 8 * k < 8 ? n - 8 * k : 8; w.pad = 0;
            items[slot] = w;
        }
        if (tid == 0 && stats && n) atomicAdd(&stats[2], (unsigned long long)n);
    }
}

// The points that decide the picked candidate's Hausdorff distance: the kLbListQ reference points with the
// largest row minima and the kLbListQ target points with the largest column minima (emitted by the pick's
// screen).  Neighbouring candidates have (nearly) the same decisive points, so a bound from these few
// queries is almost exact where the optimum lies.  Sets with fewer points repeat their last choice.
static constexpr int kLbListRP = 1;
static constexpr int kLbListQ = 8 * kLbListRP;

// One WAVE per (pair, side) -- four of them per workgroup: a lane scans its share of the minima, the wave agrees on the
// largest value (first index among equals) with two DPP reductions per choice.  (Round 3's version gave a side to a
// whole workgroup and let its 256 threads meet in one 64-bit LDS atomic per choice: 0.12 ms per config3 step.)
__global__ void __launch_bounds__(256)
k_lb_topk(const PairDesc* __restrict__ pairs, int n_pairs, const int32_t* __restrict__ pick_idx, const float* __restrict__ emit,
          int emit_rows, int emit_cols, int max_n, int32_t* __restrict__ qlist)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* s_val = reinterpret_cast<int*>(smem) + wave * max_n;
    const int slot = blockIdx.x * 4 + wave;            // (pair, side): reference side, target side
    const int p = slot >> 1, side = slot & 1;
    const bool live = p < n_pairs && pairs[p < n_pairs ? p : 0].n_ang > 0 && pick_idx[p < n_pairs ? p : 0] >= 0;
    const PairDesc pd = pairs[live ? p : 0];
    const int n = live ? (side ? pd.n_tgt : pd.n_ref) : 0;
    const float* src = emit + (size_t)(live ? p : 0) * (size_t)(emit_rows + emit_cols) + (side ? emit_rows : 0);
    for (int i = lane; i < n; i += 64) { const int v = __float_as_int(src[i]); s_val[i] = v > 0 ? v : 0; }
    __syncthreads();
    int lastsel = 0;
    for (int k = 0; k < kLbListQ; ++k) {
        int bv = -1, bi = 0x7fffffff;                   // a chosen point is marked -1: nothing left = every value -1
        for (int i = lane; i < n; i += 64) { const int v = s_val[i]; if (v > bv) { bv = v; bi = i; } }
        const int top = wave_max_i32_dpp(bv);
        const int sel = top >= 0 ? -wave_max_i32_dpp(bv == top ? -bi : -0x7fffffff) : lastsel;
        lastsel = sel;
        if (live && lane == 0) { qlist[(size_t)p * (2 * kLbListQ) + side * kLbListQ + k] = sel; if (top >= 0) s_val[sel] = -1; }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------
// Shortlist: per pair, minimum screened cost and every candidate within 2*delta of it.
// One 256-thread workgroup per pair.
// -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_shortlist(const PairDesc* __restrict__ pairs, const float* __restrict__ sq32,
            uint8_t* __restrict__ flag, WorkItem* __restrict__ items, int* __restrict__ n_items)
{
    __shared__ unsigned int s_min;
    const int p = blockIdx.x;
    const PairDesc pd = pairs[p];
    const int tid = threadIdx.x;
    if (tid == 0) s_min = 0x7f800000u;
    __syncthreads();
    unsigned int m = 0x7f800000u;
    for (int a = tid; a < pd.n_ang; a += 256) {
        const unsigned int u = __float_as_uint(sq32[pd.out_off + a]);
        m = u < m ? u : m;
    }
    m = wave_min_u32_dpp(m);
    if ((tid & 63) == 0) atomicMin(&s_min, m);
    __syncthreads();
    // A candidate stays if its smallest possible exact cost does not exceed the smallest
    // upper bound: the screened squared value S is within e2 of the f32-exact one (e2 = 0 for
    // the direct-form kernel) and the f32 representation costs at most delta on the distance.
    const double smin = (double)__uint_as_float(s_min);
    const double thr = sqrt(smin + pd.e2) + 2.0 * pd.delta;
    for (int a = tid; a < pd.n_ang; a += 256) {
        const double sv = (double)sq32[pd.out_off + a] - pd.e2;
        const bool keep = sqrt(sv > 0.0 ? sv : 0.0) <= thr;
        flag[pd.out_off + a] = keep ? 1 : 0;
        if (keep) {
            const int slot = atomicAdd(n_items, 1);
            WorkItem w; w.pair = p; w.a0 = a; w.cnt = 1; w.pad = 0;
            items[slot] = w;
        }
    }
}

// -------------------------------------------------------------------------------------
// Finalize: first index of minimal sqrt'ed exact cost among the re-scored candidates
// (process_utils.rs:72: reduce_with(|a, b| if b.1 < a.1 { b } else { a }), ordered).
// -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_finalize(const PairDesc* __restrict__ pairs, const double* __restrict__ sq64,
           const float* __restrict__ sq32, const uint8_t* __restrict__ flag,
           double* __restrict__ best_cost, int* __restrict__ best_idx,
           int* __restrict__ n_rescored, int* __restrict__ near_cnt, int* __restrict__ near_idx,
           double* __restrict__ all_costs)
{
    __shared__ double s_cost[256];
    __shared__ int s_idx[256];
    __shared__ int s_cnt[256];
    __shared__ int s_near[kMaxNear];
    __shared__ int s_nnear;
    const int p = blockIdx.x;
    const PairDesc pd = pairs[p];
    const int tid = threadIdx.x;
    double bc = __longlong_as_double(0x7ff0000000000000ll);
    int bi = 0x7fffffff, cnt = 0;
    for (int a = tid; a < pd.n_ang; a += 256) {
        const bool f = flag ? (flag[pd.out_off + a] != 0) : true;
        double h;
        if (f) {
            h = sqrt(sq64[pd.out_off + a]);  // process_utils.rs:120 (max before sqrt == sqrt before max)
            ++cnt;
            if (h < bc) { bc = h; bi = a; }  // a increases per thread: first minimum kept
        } else {
            h = sqrt((double)sq32[pd.out_off + a]);
        }
        if (all_costs) all_costs[pd.out_off + a] = h;
    }
    s_cost[tid] = bc; s_idx[tid] = bi; s_cnt[tid] = cnt;
    if (tid == 0) s_nnear = 0;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            const double oc = s_cost[tid + st];
            const int oi = s_idx[tid + st];
            if (oc < s_cost[tid] || (oc == s_cost[tid] && oi < s_idx[tid])) { s_cost[tid] = oc; s_idx[tid] = oi; }
            s_cnt[tid] += s_cnt[tid + st];
        }
        __syncthreads();
    }
    const double gbest = s_cost[0];
    const bool any = s_idx[0] != 0x7fffffff;
    // near-ties: exact-scored candidates within tol2 of the minimum (ascending, first kMaxNear)
    if (near_cnt && any) {
        const double thr = gbest + pd.tol2;
        for (int a = tid; a < pd.n_ang; a += 256) {
            const bool f = flag ? (flag[pd.out_off + a] != 0) : true;
            if (f && sqrt(sq64[pd.out_off + a]) <= thr) {
                const int slot = atomicAdd(&s_nnear, 1);
                if (slot < kMaxNear) s_near[slot] = a;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        best_cost[p] = any ? gbest : __longlong_as_double(0x7ff0000000000000ll);
        best_idx[p] = any ? (s_idx[0] + pd.ang_begin) : -1;
        if (n_rescored) n_rescored[p] = s_cnt[0];
        if (near_cnt) {
            const int n = s_nnear;
            near_cnt[p] = n;
            const int m = n < kMaxNear ? n : kMaxNear;
            for (int i = 1; i < m; ++i) {  // insertion sort of <= 8 entries
                const int v = s_near[i];
                int j = i - 1;
                while (j >= 0 && s_near[j] > v) { s_near[j + 1] = s_near[j]; --j; }
                s_near[j + 1] = v;
            }
            for (int i = 0; i < kMaxNear; ++i) near_idx[p * kMaxNear + i] = i < m ? s_near[i] + pd.ang_begin : -1;
        }
    }
}

// -------------------------------------------------------------------------------------
// Large-set Hausdorff (f64, exact, no rotation): for point sets whose target side does not
// fit the search kernel's LDS budget (e.g. the 10^3..10^4-point CCTA clouds of
// refine_alignment_hausdorff, align_algorithms.rs:400-431).  One workgroup = (pair, block of
// 16*R reference rows); the target columns stream through LDS in chunks, so any Nb works, and
// the row blocks of ONE pair spread over many CUs.  Row minima are complete inside a workgroup
// (it sees every column); column minima are combined across row blocks with 64-bit integer
// atomics in global memory (squared distances are >= +0: integer order == f64 order).
// -------------------------------------------------------------------------------------
struct LargePair {
    int32_t a_off, na, b_off, nb;   // into the f64 point pool
    int32_t col_off;                // into the global column-minimum array
    int32_t pad;
};
struct LargeWork { int32_t pair, row0; };

__global__ void __launch_bounds__(256)
k_large_init(unsigned long long* __restrict__ colmin, long long n_col, unsigned long long* __restrict__ rowmax, int n_pairs)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_col) colmin[i] = 0x7ff0000000000000ull;
    if (i < n_pairs) rowmax[i] = 0ull;
}

// rows per lane: every 16-column step costs one same-address-conflicting LDS atomicMin per lane against
// 7*R fp64 VALU operations, so more rows per lane amortise it (R = 8 -> 16: 6.4 -> see DESIGN 6a)
static constexpr int kLargeR = 16;

// DIRECTED: the rows are every pd.pad-th point of set a only (row r of the subset = point r * pad), no
// column minima are kept, and the block's row maximum goes to g_rowmax[pd.col_off]: the directed
// Hausdorff distance of a SUBSET of a to all of b.  Two such entries per pair (a->b, b->a) give a lower
// bound of the pair's Hausdorff distance in exactly the full kernel's arithmetic (same d^2 bits), at
// 2/pad of its work.
template <int R, bool DIRECTED>
__global__ void __launch_bounds__(256)
k_hausdorff_large(const LargePair* __restrict__ pairs, const LargeWork* __restrict__ work,
                  const double* __restrict__ px, const double* __restrict__ py,
                  unsigned long long* __restrict__ g_colmin, unsigned long long* __restrict__ g_rowmax)
{
    constexpr int NT = 256, NLI = 16, CH = 1024;
    __shared__ double2 s_b[CH];
    __shared__ unsigned long long s_colmin[DIRECTED ? 1 : CH];
    __shared__ unsigned long long s_red;
    const int tid = threadIdx.x, lj = tid & 15, li = tid >> 4;
    const LargeWork w = work[xcd_work_index(blockIdx.x, gridDim.x)];   // a pair's row blocks share one XCD's L2
    const LargePair pd = pairs[w.pair];
    const int nb = pd.nb;
    const int rstep = DIRECTED ? pd.pad : 1;
    const int na = DIRECTED ? (pd.na + rstep - 1) / rstep : pd.na;   // rows of this entry

    double ax[R], ay[R], rmin[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {   // padding rows duplicate the last point (see k_search)
        const int row = w.row0 + r * NLI + li;
        const int rc = (row < na ? row : na - 1) * rstep;
        ax[r] = px[pd.a_off + rc]; ay[r] = py[pd.a_off + rc];
        rmin[r] = __longlong_as_double(0x7ff0000000000000ll);
    }
    if (tid == 0) s_red = 0ull;

    for (int c0 = 0; c0 < nb; c0 += CH) {
        const int n = nb - c0 < CH ? nb - c0 : CH;
        const int np = (n + 15) & ~15;
        __syncthreads();
        for (int j = tid; j < np; j += NT) {
            const int jc = j < n ? j : n - 1;
            s_b[j] = make_double2(px[pd.b_off + c0 + jc], py[pd.b_off + c0 + jc]);
            if (!DIRECTED) s_colmin[j] = 0x7ff0000000000000ull;
        }
        __syncthreads();
        for (int k = 0; k < (np >> 4); ++k) {
            const double2 b0 = s_b[k * 16 + lj];
            double cm = __longlong_as_double(0x7ff0000000000000ll);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double dx = ax[r] - b0.x, dy = ay[r] - b0.y;   // process_utils.rs:105-107
                const double d = dx * dx + dy * dy;
                rmin[r] = dmin2(rmin[r], d);
                if (!DIRECTED) cm = dmin2(cm, d);
            }
            if (!DIRECTED) atomicMin(&s_colmin[k * 16 + lj], (unsigned long long)__double_as_longlong(cm));
        }
        if (DIRECTED) continue;   // uniform: the barrier at the top of the next chunk orders the LDS reuse
        __syncthreads();
        // merge into the pair's column minima.  The global values only ever decrease, so a (possibly
        // stale) plain read that is already <= ours proves the atomic would change nothing; after the
        // first few row blocks most columns skip it (683 -> see DESIGN 6a MB of atomic traffic per launch)
        for (int j = tid; j < n; j += NT) {
            const unsigned long long v = s_colmin[j];
            if (v < g_colmin[pd.col_off + c0 + j]) atomicMin(&g_colmin[pd.col_off + c0 + j], v);
        }
    }
    double rowmax = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double v = lane_min16(rmin[r]);
        rowmax = (v > rowmax && v < __longlong_as_double(0x7ff0000000000000ll)) ? v : rowmax;   // finite minima only (:112-114)
    }
    rowmax = wave_max(rowmax);
    if ((tid & 63) == 0) atomicMax(&s_red, (unsigned long long)__double_as_longlong(rowmax));
    __syncthreads();
    if (tid == 0) atomicMax(&g_rowmax[DIRECTED ? pd.col_off : w.pair], s_red);
}

__global__ void __launch_bounds__(256)
k_large_finish_directed(const unsigned long long* __restrict__ g_rowmax, double* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = sqrt(__longlong_as_double((long long)g_rowmax[i]));
}

__global__ void __launch_bounds__(256)
k_large_finish(const LargePair* __restrict__ pairs, const unsigned long long* __restrict__ g_colmin,
               const unsigned long long* __restrict__ g_rowmax, double* __restrict__ out)
{
    __shared__ unsigned long long s_red;
    const int p = blockIdx.x, tid = threadIdx.x;
    const LargePair pd = pairs[p];
    if (tid == 0) s_red = g_rowmax[p];
    __syncthreads();
    unsigned long long m = 0ull;
    for (int j = tid; j < pd.nb; j += 256) {
        const unsigned long long v = g_colmin[pd.col_off + j];
        m = (v > m && v < 0x7ff0000000000000ull) ? v : m;                   // finite minima only (:112-114)
    }
    atomicMax(&s_red, m);
    __syncthreads();
    if (tid == 0) out[p] = sqrt(__longlong_as_double((long long)s_red));   // process_utils.rs:120, :81
}

hipError_t launch_hausdorff_large(const void* pairs, const void* work, int n_pairs, int n_work, const double* px,
                                  const double* py, void* colmin, long long n_col, void* rowmax, double* out,
                                  hipStream_t s)
{
    const long long n_init = n_col > n_pairs ? n_col : n_pairs;
    hipLaunchKernelGGL(k_large_init, dim3((unsigned)((n_init + 255) / 256)), dim3(256), 0, s,
                       (unsigned long long*)colmin, n_col, (unsigned long long*)rowmax, n_pairs);
    if (n_work > 0)
        hipLaunchKernelGGL((k_hausdorff_large<kLargeR, false>), dim3(n_work), dim3(256), 0, s, (const LargePair*)pairs,
                           (const LargeWork*)work, px, py, (unsigned long long*)colmin, (unsigned long long*)rowmax);
    hipLaunchKernelGGL(k_large_finish, dim3(n_pairs), dim3(256), 0, s, (const LargePair*)pairs,
                       (const unsigned long long*)colmin, (const unsigned long long*)rowmax, out);
    return hipGetLastError();
}
// Lower bounds: `pairs` holds two directed entries per output slot (col_off = slot, pad = stride);
// out[slot] = sqrt(max of the two subset-directed squared distances).
hipError_t launch_hausdorff_large_bound(const void* pairs, const void* work, int n_out, int n_work, const double* px,
                                        const double* py, void* rowmax, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_large_init, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s,
                       (unsigned long long*)rowmax, 0ll, (unsigned long long*)rowmax, n_out);
    if (n_work > 0)
        hipLaunchKernelGGL((k_hausdorff_large<kLargeR, true>), dim3(n_work), dim3(256), 0, s, (const LargePair*)pairs,
                           (const LargeWork*)work, px, py, (unsigned long long*)nullptr, (unsigned long long*)rowmax);
    hipLaunchKernelGGL(k_large_finish_directed, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s,
                       (const unsigned long long*)rowmax, out, n_out);
    return hipGetLastError();
}

int large_rows_per_block() { return 16 * kLargeR; }

// -------------------------------------------------------------------------------------
// launch helpers
// -------------------------------------------------------------------------------------
static constexpr int LDS_CAP = 160 * 1024 - 256;

size_t lds_bytes_f32(int nbp) { return (size_t)nbp * (8 + 8 + 4) + 16; }
size_t lds_bytes_f64(int nbp) { return (size_t)nbp * (16 + 16 + 8) + 16; }
int max_target_points_f32() { return ((LDS_CAP - 16) / 20) & ~15; }
int max_target_points_f64() { return ((LDS_CAP - 16) / 40) & ~15; }

template <typename T, int R, int NLI, bool EXACT, bool MULTI_RB, int MINB = 1>
static hipError_t launch_one(const BatchDev& b, const WorkItem* work, int n_work, const int* n_work_dev,
                             int grid, size_t lds, const T* px, const T* py,
                             const T* cv, const T* sv, T* out, hipStream_t s)
{
    auto kern = k_search<T, R, NLI, EXACT, MULTI_RB, MINB>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NLI * 16), lds, s, b.pairs, work, n_work, n_work_dev,
                       px, py, cv, sv, out);
    return hipGetLastError();
}

// Rows-per-lane variants: the smallest R whose single row block covers the largest
// reference set wins (padding waste is (16*R - na)/na); larger sets loop over row blocks
// of the widest variant.
hipError_t launch_screen_f32(const BatchDev& b, int max_na, int max_nbp, hipStream_t s)
{
    if (b.n_work <= 0) return hipSuccess;
    const size_t lds = lds_bytes_f32(max_nbp);
    const int grid = b.n_work;
#define MM_F32(Rv, MRB) launch_one<float, Rv, 16, false, MRB>(b, b.work, b.n_work, nullptr, grid, lds, b.p32x, b.p32y, \
                                                         b.cos32, b.sin32, b.sq32, s)
    if (max_na <= 16 * 2) return MM_F32(2, false);
    if (max_na <= 16 * 8) return MM_F32(8, false);
    if (max_na <= 16 * 14) return MM_F32(14, false);
    if (max_na <= 16 * 20) return MM_F32(20, false);
    if (max_na <= 16 * 26) return MM_F32(26, false);
    if (max_na <= 16 * 33) return MM_F32(33, false);
    return MM_F32(32, true);
#undef MM_F32
}

size_t lds_bytes_fast(int nbp) { return (size_t)nbp * (16 + 8 + 4) + 16; }
int max_rows_fast() { return 16 * 33; }
int max_target_points_fast() { return ((LDS_CAP - 16) / 28) & ~15; }

// Work from the host-built table (n_dev == nullptr: one workgroup per item -- a persistent grid
// measured slower on the big launch) or from a device queue of at most `cap` items whose length is
// *n_dev (a bounded grid strides over it).
template <int Rv>
static hipError_t launch_fast_r(const BatchDev& b, const WorkItem* work, int n_host, const int* n_dev, int cap,
                                int max_nbp, bool emit, hipStream_t s)
{
    auto kern = emit ? k_screen_fast<Rv, true> : k_screen_fast<Rv, false>;
    const size_t lds = lds_bytes_fast(max_nbp);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int grid = n_dev ? std::min(cap, 256 * 12) : n_host;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, b.pairs, work, n_host, n_dev, b.p32x, b.p32y,
                       b.cos32, b.sin32, b.sq32, emit ? b.emit : nullptr, b.emit_rows, b.emit_cols);
    return hipGetLastError();
}

static hipError_t launch_fast_any(const BatchDev& b, const WorkItem* work, int n_host, const int* n_dev, int cap,
                                  int max_na, int max_nbp, bool emit, hipStream_t s)
{
    if (max_na <= 16 * 8) return launch_fast_r<8>(b, work, n_host, n_dev, cap, max_nbp, emit, s);
    if (max_na <= 16 * 14) return launch_fast_r<14>(b, work, n_host, n_dev, cap, max_nbp, emit, s);
    if (max_na <= 16 * 20) return launch_fast_r<20>(b, work, n_host, n_dev, cap, max_nbp, emit, s);
    if (max_na <= 16 * 26) return launch_fast_r<26>(b, work, n_host, n_dev, cap, max_nbp, emit, s);
    return launch_fast_r<33>(b, work, n_host, n_dev, cap, max_nbp, emit, s);
}

hipError_t launch_screen_fast(const BatchDev& b, int max_na, int max_nbp, hipStream_t s)
{
    if (b.n_work <= 0) return hipSuccess;
    return launch_fast_any(b, b.work, b.n_work, nullptr, 0, max_na, max_nbp, false, s);
}

// ---- bounded screen (k_screen_lb -> pick -> screen the picks -> keep -> screen the survivors) ----
int lb_max_query_points() { return 8 * kLbRP; }
int lb_max_points() { return 4096; }
size_t lds_bytes_lb(int nap, int nbp) { return ((size_t)nap + (size_t)nbp) * 16; }

int lb_mx_max_points() { return 1024; }     // 32 row tiles per set: 64 KB of row fragments

// b.lb_mx_qt column tiles of queries per side (32 queries each), b.lb_mx_nc candidates per wave at once
template <int QT, int NC, bool LIST>
static hipError_t launch_lb_mx_v(const BatchDev& b, const WorkItem* work, int n_host, const int* n_dev, int cap, hipStream_t s)
{
    auto kern = k_bound_mx<QT, NC, LIST>;
    const size_t lds = (size_t)2 * b.lb_mx * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int grid = n_dev ? std::min(cap, 256 * 16) : n_host;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, b.pairs, work, n_host, n_dev, b.lb_stride, b.qlist, kLbListQ, b.lb_mx,
                       b.p32x, b.p32y, b.cos32, b.sin32, b.lb32);
    return hipGetLastError();
}

template <bool LIST>
static hipError_t launch_lb_mx_t(const BatchDev& b, const WorkItem* work, int n_host, const int* n_dev, int cap, hipStream_t s)
{
    const int v = b.lb_mx_qt * 10 + b.lb_mx_nc;
    if (v == 11) return launch_lb_mx_v<1, 1, LIST>(b, work, n_host, n_dev, cap, s);
    if (v == 12) return launch_lb_mx_v<1, 2, LIST>(b, work, n_host, n_dev, cap, s);
    if (v == 21) return launch_lb_mx_v<2, 1, LIST>(b, work, n_host, n_dev, cap, s);
    if (v == 22) return launch_lb_mx_v<2, 2, LIST>(b, work, n_host, n_dev, cap, s);
    return hipErrorInvalidValue;
}

template <int RP, bool LIST>
static hipError_t launch_lb_t(const BatchDev& b, const WorkItem* work, int n_host, const int* n_dev, int cap,
                              int max_nap, int max_nbp, hipStream_t s)
{
    if (b.lb_mx > 0) return launch_lb_mx_t<LIST>(b, work, n_host, n_dev, cap, s);
    auto kern = k_screen_lb<RP, LIST>;
    const size_t lds = lds_bytes_lb(max_nap, max_nbp);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int grid = n_dev ? std::min(cap, 256 * 12) : n_host;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, b.pairs, work, n_host, n_dev, b.lb_stride, b.qlist,
                       b.p32x, b.p32y, b.cos32, b.sin32, b.lb32);
    return hipGetLastError();
}

int lb_candidate_step() { return kLbCandStep; }
int lb_sparse_candidates(int n) { return lb_sparse_count(n); }
int lb_list_queries() { return kLbListQ; }

// round 1: every kLbCandStep-th candidate and the last one (host work list)
hipError_t launch_screen_lb(const BatchDev& b, int max_nap, int max_nbp, hipStream_t s)
{
    if (b.n_work_lb <= 0) return hipSuccess;
    return launch_lb_t<kLbRP, false>(b, b.work_lb, b.n_work_lb, nullptr, 0, max_nap, max_nbp, s);
}

template <bool LIST>
static hipError_t launch_lb_mx_scan(const BatchDev& b, int stat_slot, hipStream_t s)
{
    const size_t lds = (size_t)2 * b.lb_mx * 1024 + (size_t)kScanChunk * 12;
    // two workgroups per pair (halves of its candidate list): 4 000 workgroups for config3, and a pair's staging twice at most
    const int split = 2, grid = std::min(b.n_pairs * split, 256 * 24);
#define MM_SCAN(QT) { auto kern = k_bound_mx_scan<QT, LIST>; \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, b.pairs, b.n_pairs, split, b.flag, b.lb_stride, b.qlist, kLbListQ, b.lb_mx, \
                           b.p32x, b.p32y, b.cos32, b.sin32, b.lb32, b.stats, stat_slot); }
    if (b.lb_mx_qt == 2) MM_SCAN(2) else MM_SCAN(1)
#undef MM_SCAN
    return hipGetLastError();
}

// round 2: the candidates k_lb_spread could not rule out (queue 0, counter [2]; matrix pipe: the flagged ones)
hipError_t launch_screen_lb_queued(const BatchDev& b, int max_nap, int max_nbp, int cap, hipStream_t s)
{
    if (cap <= 0) return hipSuccess;
    if (b.lb_mx > 0) return launch_lb_mx_scan<false>(b, 1, s);
    return launch_lb_t<kLbRP, false>(b, b.items_lb, 0, b.n_items + 2, cap, max_nap, max_nbp, s);
}

// round 3: the survivors of round 2 (queue 1, counter [3]) against the pick's decisive points
hipError_t launch_screen_lb_list(const BatchDev& b, int max_nap, int max_nbp, int cap, hipStream_t s)
{
    if (cap <= 0) return hipSuccess;
    if (b.lb_mx > 0) return launch_lb_mx_scan<true>(b, 3, s);
    return launch_lb_t<kLbListRP, true>(b, b.items_lb + cap, 0, b.n_items + 3, cap, max_nap, max_nbp, s);
}

hipError_t launch_lb_pick(const BatchDev& b, int round, hipStream_t s)
{
    if (b.n_pairs <= 0) return hipSuccess;
    if (round == 0)
        hipLaunchKernelGGL(k_lb_pick<true>, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.lb32, b.pick_idx, b.items_pick,
                           b.n_items + 1);
    else
        hipLaunchKernelGGL(k_lb_pick<false>, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.lb32, b.pick_idx + b.n_pairs,
                           b.items_pick + b.n_pairs, b.n_items + 4);
    return hipGetLastError();
}

// full screen of the picks; round 0 also leaves the row / column minima for k_lb_topk
hipError_t launch_screen_picks(const BatchDev& b, int round, int max_na, int max_nbp, hipStream_t s)
{
    if (b.n_pairs <= 0) return hipSuccess;
    if (b.kept_mx_nct > 0) {
        // bounded search on the matrix pipe: the pick's value (the pair's upper bound) comes from the matrix kernel, and so
        // do, for the first pick, the row / column minima from which k_lb_topk chooses the third round's queries (the
        // generated block's `emit` form) -- no packed-FMA kernel runs under this precision any more (why that matters:
        // profiles/README.md, "packed-FMA kernels beside MFMA kernels")
        if (round == 0) return launch_mx_emit(b, b.items_pick, b.n_items + 1, b.n_pairs, b.kept_mx_nct, b.kept_mx_acap, s);
        return launch_mx_any(b, b.items_pick + b.n_pairs, 0, b.n_items + 4, b.n_pairs, b.kept_mx_nct, 0, b.kept_mx_acap, s, 1);
    }
    if (round == 0) return launch_fast_any(b, b.items_pick, 0, b.n_items + 1, b.n_pairs, max_na, max_nbp, true, s);
    return launch_fast_any(b, b.items_pick + b.n_pairs, 0, b.n_items + 4, b.n_pairs, max_na, max_nbp, false, s);
}

// MM_PRECISION_F32_MATRIX, pairs with a set of fewer than 64 points: no screen at all -- every candidate gets the screened
// value 0, so that the shortlist keeps all of them and the exact f64 kernel scores them (tiny sets: cheap)
__global__ void __launch_bounds__(256) k_screen_none(const PairDesc* __restrict__ pairs, const WorkItem* __restrict__ work, int n_work,
                                                      float* __restrict__ out_sq)
{
    for (int wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const WorkItem w = work[wi];
        const PairDesc pd = pairs[w.pair];
        for (int k = threadIdx.x; k < w.cnt; k += 256) out_sq[pd.out_off + w.a0 + k] = 0.0f;
    }
}

hipError_t launch_screen_none(const BatchDev& b, int work_begin, int n_work, hipStream_t s)
{
    if (n_work <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_screen_none, dim3(std::min(n_work, 4096)), dim3(256), 0, s, b.pairs, b.work + work_begin, n_work, b.sq32);
    return hipGetLastError();
}

hipError_t launch_lb_topk(const BatchDev& b, int max_n, hipStream_t s)
{
    if (b.n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_lb_topk, dim3((2 * b.n_pairs + 3) / 4), dim3(256), (size_t)max_n * 16, s, b.pairs, b.n_pairs, b.pick_idx,
                       b.emit, b.emit_rows, b.emit_cols, max_n, b.qlist);
    return hipGetLastError();
}

hipError_t launch_lb_spread(const BatchDev& b, hipStream_t s)
{
    if (b.n_pairs <= 0) return hipSuccess;
    uint8_t* flags = b.lb_mx > 0 ? b.flag : nullptr;
    if (flags) {       // the candidates that need a bound of their own are flagged (k_bound_mx_scan), not queued
        hipError_t e = hipMemsetAsync(flags, 0, (size_t)b.n_cand, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_lb_spread, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.lb32, b.sq32, b.pick_idx, b.cos64,
                       b.sin64, b.items_lb, b.n_items + 2, b.stats, flags);
    return hipGetLastError();
}

// survivors of the rounds so far: round 2's go to queue 1 (for the list round), the final ones to queue 2
hipError_t launch_lb_keep(const BatchDev& b, int final, int cap, hipStream_t s)
{
    if (b.n_pairs <= 0) return hipSuccess;
    if (b.lb_mx > 0 && b.kept_mx_nct > 0) {
        if (!final)
            hipLaunchKernelGGL(k_lb_keep_mx<false>, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.lb32, b.sq32, b.pick_idx,
                               (const int32_t*)nullptr, b.klist, b.items_lb + cap, b.n_items + 3, b.stats, b.flag);
        else
            hipLaunchKernelGGL(k_lb_keep_mx<true>, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.lb32, b.sq32, b.pick_idx,
                               (const int32_t*)(b.pick_idx + b.n_pairs), b.klist, b.items_lb + 2 * (size_t)cap, b.n_items + 5, b.stats,
                               (uint8_t*)nullptr);
        return hipGetLastError();
    }
    if (!final)
        hipLaunchKernelGGL(k_lb_keep<false>, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.lb32, b.sq32, b.pick_idx,
                           (const int32_t*)nullptr, b.items_lb + cap, b.n_items + 3, b.stats, b.lb_mx > 0 ? b.flag : (uint8_t*)nullptr);
    else
        hipLaunchKernelGGL(k_lb_keep<true>, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.lb32, b.sq32, b.pick_idx,
                           (const int32_t*)(b.pick_idx + b.n_pairs), b.items_lb + 2 * (size_t)cap, b.n_items + 5, b.stats,
                           (uint8_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_screen_kept(const BatchDev& b, int max_na, int max_nbp, int cap, hipStream_t s)
{
    if (cap <= 0) return hipSuccess;
    // every pair of the batch takes the same variant of the matrix-pipe screen: the survivors go through it (<= 8 entries
    // of a pair's list of survivors per queue item, one wave per candidate)
    if (b.kept_mx_nct > 0)
        return launch_mx_any(b, b.items_lb + 2 * (size_t)cap, 0, b.n_items + 5, cap, b.kept_mx_nct, 0, b.kept_mx_acap, s, 4, b.klist);
    return launch_fast_any(b, b.items_lb + 2 * (size_t)cap, 0, b.n_items + 5, cap, max_na, max_nbp, false, s);
}

// Exact f64 kernel variants.  Every variant is a 256-thread workgroup (one wave per SIMD) with a register budget
// of three workgroups per CU:
//  * beside another stream's screen launch only such a workgroup is ever dispatched (3 x 1 wave per SIMD at ~168
//    VGPRs fill every CU; a 512-thread, 2-waves-per-SIMD workgroup never fits the hole a retiring workgroup
//    leaves and waited for the whole 30 ms launch: the re-score of the between stage, 23 us alone, took 30-34 ms
//    in a pipelined driver; stream priority does not help, tools/prio_probe2.hip);
//  * three independent workgroups per CU run out of phase, so the rotation / epilogue / barriers of one overlap
//    the distance loop of the others -- the all-f64 search (MM_PRECISION_F64) went from 108.9 ms (512 threads,
//    R = 17, one workgroup per CU) to 87.1 ms on config3, 0.91 of the measured fp64 issue rate.
// Sets beyond 16 x 14 points loop over row blocks of 16 x R rows; R in {9, 10, 11} is chosen for the least padding
// (521 points = 3 x 176).
template <bool FROM_QUEUE>
static hipError_t launch_f64(const BatchDev& b, int max_na, int max_nbp, int grid, hipStream_t s)
{
    const size_t lds = lds_bytes_f64(max_nbp);
    const WorkItem* work = FROM_QUEUE ? b.items : b.work;
    const int* nd = FROM_QUEUE ? b.n_items : nullptr;
    const int nw = FROM_QUEUE ? 0 : b.n_work;
#define MM_F64(Rv, MRB) launch_one<double, Rv, 16, true, MRB, 3>(b, work, nw, nd, grid, lds, b.p64x, b.p64y, \
                                                            b.cos64, b.sin64, b.sq64, s)
    if (max_na <= 16 * 4) return MM_F64(4, false);
    if (max_na <= 16 * 11) return MM_F64(11, false);
    if (max_na <= 16 * 14) return MM_F64(14, false);
    int best = 11;
    long best_rows = -1;
    for (int r = 11; r >= 9; --r) {
        const long blk = 16L * r, rows = (max_na + blk - 1) / blk * blk;
        if (best_rows < 0 || rows < best_rows) { best_rows = rows; best = r; }
    }
    if (best == 9) return MM_F64(9, true);
    if (best == 10) return MM_F64(10, true);
    return MM_F64(11, true);
#undef MM_F64
}

hipError_t launch_exact_all(const BatchDev& b, int max_na, int max_nbp, hipStream_t s)
{
    if (b.n_work <= 0) return hipSuccess;
    return launch_f64<false>(b, max_na, max_nbp, b.n_work, s);
}

hipError_t launch_rescore(const BatchDev& b, int max_na, int max_nbp, int total_candidates, hipStream_t s)
{
    if (total_candidates <= 0) return hipSuccess;
    // workgroups stride over the device-side queue; bound the grid, every block exits
    int grid = total_candidates < 2048 ? total_candidates : 2048;
    return launch_f64<true>(b, max_na, max_nbp, grid, s);
}

hipError_t launch_shortlist(const BatchDev& b, hipStream_t s)
{
    if (b.n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_shortlist, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.sq32, b.flag, b.items, b.n_items);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------
// Small copies between HBM and pinned host memory as a 256-thread kernel.  Why not hipMemcpyAsync: a copy that
// follows a kernel on its stream is executed by the runtime as a blit kernel of 512-thread workgroups, and a
// 2-waves-per-SIMD workgroup is never dispatched while another stream's screen launch keeps every CU at
// 3 x (1 wave per SIMD, ~168 VGPRs): it waits for the whole 30 ms launch (tools/prio_probe2.hip: k512 12 ms
// against k256 0.13 ms beside the same launch, whatever the stream priorities).  With a pipelined driver that
// stalled the result fetch of the between stage and the staging of the next case.  256 threads fit the hole a
// retiring screen workgroup leaves.  dst/src: device-accessible (HBM or hipHostMalloc), 16-byte aligned.
// -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_copy_small(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, size_t tail_bytes)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail_bytes)
        reinterpret_cast<unsigned char*>(dst + n16)[threadIdx.x] = reinterpret_cast<const unsigned char*>(src + n16)[threadIdx.x];
}

hipError_t launch_copy_small(void* dst, const void* src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    const size_t n16 = bytes / 16, tail = bytes % 16;
    const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((n16 + 255) / 256, 1), 64);
    hipLaunchKernelGGL(k_copy_small, dim3(grid), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n16, tail);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------
// Search sets of the within-pullback search, built from the raw pullback in HBM (SetSrc, mm_device.h).
// The host path did this per case on the CPU (14 ms for 4 x 512 frames) and uploaded four planes; here the
// raw contours go up once as they are and one small kernel writes the planes.  Same arithmetic as the host
// code it replaces, so the pool is bit-identical: index = (i64)((f64)i * ((f64)len / (f64)n)) (contour.rs:51-55),
// coordinate - centroid in f64, one rounding to f32.
// -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_build_sets(const SetSrc* __restrict__ src, const double* __restrict__ raw, float* __restrict__ p32x,
             float* __restrict__ p32y, double* __restrict__ p64x, double* __restrict__ p64y,
             double* __restrict__ rho2_out, double* __restrict__ scale_out)
{
    __shared__ double s_rho[4], s_scale[4];
    const SetSrc s = src[blockIdx.x];
    const int tid = threadIdx.x;
    const int nl = s.lum_len < s.lum_take ? s.lum_len : s.lum_take;
    const double stride_l = (double)s.lum_len / (double)s.lum_take;
    const double stride_c = (double)s.cath_len / (double)s.cath_take;
    double rho2 = 0.0, scale = 0.0;
    for (int k = tid; k < s.n; k += 256) {
        long long at;
        if (k < nl) {
            at = s.lum_at + (s.lum_len <= s.lum_take ? (long long)k : (long long)((double)k * stride_l));
        } else {
            const int kk = k - nl;
            at = s.cath_at + (s.cath_len <= s.cath_take ? (long long)kk : (long long)((double)kk * stride_c));
        }
        double x = raw[3 * at], y = raw[3 * at + 1];
        scale = fmax(scale, fmax(fabs(x), fabs(y)));
        x -= s.cx; y -= s.cy;
        scale = fmax(scale, fmax(fabs(x), fabs(y)));
        p64x[s.dst_off + k] = x; p64y[s.dst_off + k] = y;
        p32x[s.dst_off + k] = (float)x; p32y[s.dst_off + k] = (float)y;
        const double r2 = x * x + y * y;
        rho2 = (r2 <= 1.0e300) ? fmax(rho2, r2) : __longlong_as_double(0x7ff0000000000000ll);   // NaN / inf / overflow:
    }                                                                                             // rho = inf tells the host
    rho2 = wave_max(rho2); scale = wave_max(scale);
    if ((tid & 63) == 0) { s_rho[tid >> 6] = rho2; s_scale[tid >> 6] = scale; }
    __syncthreads();
    if (tid == 0) {
        rho2_out[blockIdx.x] = fmax(fmax(s_rho[0], s_rho[1]), fmax(s_rho[2], s_rho[3]));
        scale_out[blockIdx.x] = fmax(fmax(s_scale[0], s_scale[1]), fmax(s_scale[2], s_scale[3]));
    }
}

hipError_t launch_build_sets(const SetSrc* src, int n_sets, const double* raw, float* p32x, float* p32y, double* p64x,
                             double* p64y, double* rho2, double* scale, hipStream_t s)
{
    if (n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_sets, dim3(n_sets), dim3(256), 0, s, src, raw, p32x, p32y, p64x, p64y, rho2, scale);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------
// Exchange records of a sharded search (see mm_device.h).  One thread per job.
// -------------------------------------------------------------------------------------
struct LocalBest { double cost; int idx; bool uniform; long long abits; };

static __device__ __forceinline__ LocalBest local_best(const PairDesc& pd, int k, const double* __restrict__ best_cost,
                                                       const int* __restrict__ best_idx, const int* __restrict__ near_cnt,
                                                       const int* __restrict__ near_idx, const double* __restrict__ ang64)
{
    LocalBest r;
    r.cost = __longlong_as_double(0x7ff0000000000000ll); r.idx = -1; r.uniform = true; r.abits = 0;
    if (pd.n_ref == 0 || pd.n_tgt == 0) {
        // process_utils.rs:86-88: every cost is 0.0, the first candidate of the slice is its first minimum
        if (pd.n_slice > 0) { r.cost = 0.0; r.idx = pd.ang_begin; r.abits = __double_as_longlong(ang64[pd.tab_off]); }
        return r;
    }
    const int bi = best_idx[k];
    if (bi < 0) return r;
    r.cost = best_cost[k]; r.idx = bi;
    r.abits = __double_as_longlong(ang64[pd.tab_off + (bi - pd.ang_begin)]);
    const int n = near_cnt[k];
    bool ok = n == 1;
    if (!ok && n >= 2 && n <= kMaxNear) {
        ok = true;
        const long long a0 = __double_as_longlong(ang64[pd.tab_off + (near_idx[k * kMaxNear] - pd.ang_begin)]);
        for (int q = 1; q < n; ++q)
            ok = ok && (__double_as_longlong(ang64[pd.tab_off + (near_idx[k * kMaxNear + q] - pd.ang_begin)]) == a0);
    }
    r.uniform = ok;
    return r;
}

__global__ void __launch_bounds__(256)
k_export_cost(const PairDesc* __restrict__ pairs, const int32_t* __restrict__ pair_of_job, int n_jobs,
              const double* __restrict__ best_cost, const int* __restrict__ best_idx, const int* __restrict__ near_cnt,
              const int* __restrict__ near_idx, const double* __restrict__ ang64, double* __restrict__ cost)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_jobs) return;
    const int k = pair_of_job[j];
    double c = __longlong_as_double(0x7ff0000000000000ll);
    if (k >= 0) c = local_best(pairs[k], k, best_cost, best_idx, near_cnt, near_idx, ang64).cost;
    cost[j] = c;
}

__global__ void __launch_bounds__(256)
k_export_keys(const PairDesc* __restrict__ pairs, const int32_t* __restrict__ pair_of_job, int n_jobs,
              const double* __restrict__ best_cost, const int* __restrict__ best_idx, const int* __restrict__ near_cnt,
              const int* __restrict__ near_idx, const double* __restrict__ ang64, const double* __restrict__ gcost,
              long long* __restrict__ keys)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_jobs) return;
    const long long kMax = 0x7fffffffffffffffll, kMin = (long long)0x8000000000000000ull;
    long long k_idx = kMax, k_lo = kMax, k_hi = kMax;
    const int k = pair_of_job[j];
    if (k >= 0) {
        const PairDesc pd = pairs[k];
        const LocalBest lb = local_best(pd, k, best_cost, best_idx, near_cnt, near_idx, ang64);
        const double g = gcost[j];
        if (lb.idx >= 0) {
            if (lb.cost == g) k_idx = (long long)lb.idx;
            if (lb.cost <= g + pd.tol2) {                       // same expression as mm_merge_shards
                k_lo = lb.uniform ? lb.abits : kMin;
                k_hi = lb.uniform ? ~lb.abits : kMin;
            }
        }
    }
    keys[j] = k_idx; keys[n_jobs + j] = k_lo; keys[2 * (size_t)n_jobs + j] = k_hi;
}

hipError_t launch_export_cost(const BatchDev& b, const int32_t* pair_of_job, int n_jobs, double* cost, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_export_cost, dim3((n_jobs + 255) / 256), dim3(256), 0, s, b.pairs, pair_of_job, n_jobs, b.best_cost,
                       b.best_idx, b.near_cnt, b.near_idx, b.ang64, cost);
    return hipGetLastError();
}

hipError_t launch_export_keys(const BatchDev& b, const int32_t* pair_of_job, int n_jobs, const double* gcost,
                              long long* keys, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_export_keys, dim3((n_jobs + 255) / 256), dim3(256), 0, s, b.pairs, pair_of_job, n_jobs, b.best_cost,
                       b.best_idx, b.near_cnt, b.near_idx, b.ang64, gcost, keys);
    return hipGetLastError();
}

hipError_t launch_finalize(const BatchDev& b, int use_flags, hipStream_t s)
{
    if (b.n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize, dim3(b.n_pairs), dim3(256), 0, s, b.pairs, b.sq64, b.sq32,
                       use_flags ? b.flag : nullptr, b.best_cost, b.best_idx, b.n_rescored, b.near_cnt, b.near_idx,
                       b.all_costs);
    return hipGetLastError();
}

}  // namespace mm
