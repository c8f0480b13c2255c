// mm_nn_kernels.hip -- nearest-neighbour squared distances in 3-D, exact f64, for gfx950.
//
// For every query point q of a (query set, point set) pair:  out[q] = min_p |q - p|^2  with
// |q - p|^2 = dx*dx + dy*dy + dz*dz in exactly that order and no contraction (the file is built
// with -ffp-contract=off), i.e. the inner fold of the reference's symmetric_nn_distance and
// find_region_points (src/ccta/adjust_mesh/scale_coronary.rs:142-152, 193-199, 204-210;
// calculate_squared_distance: src/ccta/adjust_mesh.rs:7-12).  min is exact, so the result does not
// depend on the traversal order.
//
// Mapping: one work item = 256 lanes x QPT queries of one pair against a SPAN of kNnSpan chunks of CH
// points; a chunk is staged in LDS as (x, y, z, 0) so that a point is two ds_read broadcasts (all lanes
// read the same address: conflict-free).  The span minima are merged into the output with a 64-bit
// atomicMin on the bit pattern (order-preserving for values >= 0; the output is pre-filled with +inf):
// one atomic per query and 5 120 points, so a search of 82 pairs x 20 000 queries is 13 120 equal items
// (short tail) at 4 atomics per query.  The work list is pair-major and dealt to the XCDs in contiguous
// eighths (xcd order as in mm_kernels.hip), so a pair's sets are fetched from HBM by one XCD.
// Per (query, point): 3 sub + 3 mul + 2 add + 1 min = 9 fp64 VALU operations against 2/QPT LDS reads
// -> fp64-VALU bound.  Sets are SoA f64 in HBM (L2-resident: a set is a few hundred KB).
#include <hip/hip_runtime.h>

#include "mm_device.h"

namespace mm {

struct NnPair { int32_t q_off, nq, p_off, np, out_off, pad; };   // offsets into the point pool / output
struct NnWork { int32_t pair, q0, c0, pad; };                     // queries [q0, q0+256*QPT) x points [c0, c0+kNnSpan*CH)

static constexpr int kNnChunk = 1024;
static constexpr int kNnSpan = 5;

static __device__ __forceinline__ int nn_xcd_work_index(int b, int n)   // see xcd_work_index in mm_kernels.hip
{
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

__global__ void __launch_bounds__(256)
k_nn3_fill(unsigned long long* __restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 0x7ff0000000000000ull;   // +inf
}

template <int QPT>
__global__ void __launch_bounds__(256)
k_nn3_min(const NnPair* __restrict__ pairs, const NnWork* __restrict__ work, int n_work,
          const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ pz,
          unsigned long long* __restrict__ out)
{
    constexpr int NT = 256, CH = kNnChunk;
    __shared__ double4 s_p[CH];
    const int tid = threadIdx.x;
    for (int wi = (int)gridDim.x == n_work ? nn_xcd_work_index(blockIdx.x, n_work) : (int)blockIdx.x; wi < n_work;
         wi += gridDim.x) {
        const NnWork w = work[wi];
        const NnPair pd = pairs[w.pair];
        double qx[QPT], qy[QPT], qz[QPT], m[QPT];
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const int q = w.q0 + k * NT + tid;
            const int qc = q < pd.nq ? q : pd.nq - 1;   // lanes past the end recompute the last query, never stored
            qx[k] = px[pd.q_off + qc]; qy[k] = py[pd.q_off + qc]; qz[k] = pz[pd.q_off + qc];
            m[k] = __builtin_inf();
        }
        const int c_end = pd.np - w.c0 < kNnSpan * CH ? pd.np : w.c0 + kNnSpan * CH;
        for (int c0 = w.c0; c0 < c_end; c0 += CH) {
            const int n = c_end - c0 < CH ? c_end - c0 : CH;
            __syncthreads();   // previous chunk fully consumed
            for (int j = tid; j < n; j += NT)
                s_p[j] = make_double4(px[pd.p_off + c0 + j], py[pd.p_off + c0 + j], pz[pd.p_off + c0 + j], 0.0);
            __syncthreads();
#pragma unroll 4
            for (int j = 0; j < n; ++j) {
                const double4 p = s_p[j];
#pragma unroll
                for (int k = 0; k < QPT; ++k) {
                    const double dx = qx[k] - p.x, dy = qy[k] - p.y, dz = qz[k] - p.z;
                    const double v = dx * dx + dy * dy + dz * dz;
                    m[k] = __builtin_fmin(m[k], v);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const int q = w.q0 + k * NT + tid;
            if (q < pd.nq) atomicMin(&out[pd.out_off + q], (unsigned long long)__double_as_longlong(m[k]));
        }
    }
}

// queries per lane: 1..4 measured within 5 % of each other on MI355X (10.9 / 10.4 / 10.4 / 10.3 ms for the
// 3.3e10-pair search of tools/bench_ccta.py): the kernel is bound by fp64 VALU issue, not by the LDS reads
static constexpr int kNnQpt = 2;
int nn_queries_per_block() { return 256 * kNnQpt; }
int nn_points_per_chunk() { return kNnChunk * kNnSpan; }   // points one work item covers

hipError_t launch_nn3_min(const void* pairs, const void* work, int n_work, const double* px, const double* py,
                          const double* pz, double* out, long long n_out, hipStream_t s)
{
    if (n_out > 0)
        hipLaunchKernelGGL(k_nn3_fill, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s,
                           (unsigned long long*)out, n_out);
    if (n_work > 0)
        hipLaunchKernelGGL(k_nn3_min<kNnQpt>, dim3((unsigned)n_work), dim3(256), 0, s, (const NnPair*)pairs,
                           (const NnWork*)work, n_work, px, py, pz, (unsigned long long*)out);
    return hipGetLastError();
}

}  // namespace mm
