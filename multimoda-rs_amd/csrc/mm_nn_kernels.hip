// mm_nn_kernels.hip -- nearest-neighbour squared distances in 3-D, exact f64, for gfx950.
//
// For every query point q of a (query set, point set) pair:  out[q] = min_p |q - p|^2  with
// |q - p|^2 = dx*dx + dy*dy + dz*dz in exactly that order and no contraction (the file is built
// with -ffp-contract=off), i.e. the inner fold of the reference's symmetric_nn_distance and
// find_region_points (src/ccta/adjust_mesh/scale_coronary.rs:142-152, 193-199, 204-210;
// calculate_squared_distance: src/ccta/adjust_mesh.rs:7-12).  min is exact, so the result does not
// depend on the traversal order.
//
// Mapping: one work item = 256 lanes x QPT queries of one pair against a SPAN of chunks of CH points; a
// chunk is staged in LDS as (x, y, z, 0) so that a point is two ds_read broadcasts (all lanes read the
// same address: conflict-free).  The span minima are merged into the output with a 64-bit atomicMin on the
// bit pattern (order-preserving for values >= 0; the output is pre-filled with +inf).
// Per (query, point): 3 sub + 3 mul + 2 add + 1 min = 9 fp64 VALU operations against 2/QPT LDS reads
// -> fp64-VALU bound.  Sets are SoA f64 in HBM (L2-resident: a set is a few hundred KB).
//
// Pruning (large sets, staged in slabs across their longest axis by the host so that a block of queries and a chunk of points
// are each spatially compact): the host hands every (query block, chunk) item a lower bound lb2 of the
// squared distance between their bounding boxes.  Pass A runs, per query block, the chunk with the smallest
// bound; pass B runs all the others, and an item starts by reading its queries' current minima: if none
// exceeds lb2, no point of the chunk can lower any of them and the item is skipped.  The minimum is exact
// and order-independent, so the result is bit-identical to scanning everything; only work is saved.
// The work lists are pair-major and dealt to the XCDs in contiguous eighths (as in mm_kernels.hip).
#include <hip/hip_runtime.h>

#include "mm_device.h"

namespace mm {

// q_off / p_off: into the point pool; out_off: into the output; qperm_off: into the permutation pool (the
// staged position j of the query set holds original point qperm[qperm_off + j]; -1 = staged in original order)
struct NnPair { int32_t q_off, nq, p_off, np, out_off, qperm_off; };
// queries [q0, q0 + 256 QPT) x points [c0, c0 + n_chunks CH); lb2: see "Pruning" above (pass B only)
struct NnWork { int32_t pair, q0, c0, n_chunks; double lb2; };

static constexpr int kNnChunk = 512;
static constexpr int kNnSpan = 10;

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

// A derived set: pool[dst_off + j] = base point j moved by adj along its unit vector where its flag is set,
// p + unit * x with one rounding for the product and one for the sum (no contraction: this file is built
// with -ffp-contract=off), exactly centerline_based_diameter_morphing (scale_coronary.rs:236-239).
// aux: 7 planes of n_aux doubles -- bx by bz ux uy uz flag.
struct NnMorph { int32_t dst_off, n, aux_off, pad; double adj; };

__global__ void __launch_bounds__(256)
k_nn3_morph(const NnMorph* __restrict__ items, const double* __restrict__ aux, long long n_aux,
            double* __restrict__ px, double* __restrict__ py, double* __restrict__ pz)
{
    const NnMorph it = items[blockIdx.y];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= it.n) return;
    const long long a = it.aux_off + j;
    const bool mv = aux[6 * n_aux + a] != 0.0;
    const double bx = aux[a], by = aux[n_aux + a], bz = aux[2 * n_aux + a];
    const double ux = aux[3 * n_aux + a], uy = aux[4 * n_aux + a], uz = aux[5 * n_aux + a];
    px[it.dst_off + j] = mv ? bx + ux * it.adj : bx;
    py[it.dst_off + j] = mv ? by + uy * it.adj : by;
    pz[it.dst_off + j] = mv ? bz + uz * it.adj : bz;
}

template <int QPT, bool CHECK>
__global__ void __launch_bounds__(256)
k_nn3_min(const NnPair* __restrict__ pairs, const NnWork* __restrict__ work, int n_work,
          const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ pz,
          const int32_t* __restrict__ qperm, unsigned long long* __restrict__ out)
{
    constexpr int NT = 256, CH = kNnChunk;
    __shared__ double4 s_p[CH];
    __shared__ unsigned long long s_max;
    const int tid = threadIdx.x;
    for (int wi = (int)gridDim.x == n_work ? nn_xcd_work_index(blockIdx.x, n_work) : (int)blockIdx.x; wi < n_work;
         wi += gridDim.x) {
        const NnWork w = work[wi];
        const NnPair pd = pairs[w.pair];
        int oi[QPT];   // where this lane's queries report: their original index in the query set
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const int q = w.q0 + k * NT + tid;
            oi[k] = q < pd.nq ? (pd.qperm_off >= 0 ? qperm[pd.qperm_off + q] : q) : -1;
        }
        if (CHECK) {
            // largest current minimum of this block's queries (a stale, larger value only costs work)
            unsigned long long mx = 0ull;
#pragma unroll
            for (int k = 0; k < QPT; ++k)
                if (oi[k] >= 0) { const unsigned long long v = out[pd.out_off + oi[k]]; mx = v > mx ? v : mx; }
            __syncthreads();   // s_max of the previous item is no longer read
            if (tid == 0) s_max = 0ull;
            __syncthreads();
            atomicMax(&s_max, mx);
            __syncthreads();
            if (w.lb2 >= __longlong_as_double((long long)s_max)) continue;   // uniform: nothing here can improve
        }
        double qx[QPT], qy[QPT], qz[QPT], m[QPT];
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const int q = w.q0 + k * NT + tid;
            const int qc = q < pd.nq ? q : pd.nq - 1;   // lanes past the end recompute the last query, never stored
            qx[k] = px[pd.q_off + qc]; qy[k] = py[pd.q_off + qc]; qz[k] = pz[pd.q_off + qc];
            m[k] = __builtin_inf();
        }
        const int c_end = pd.np - w.c0 < w.n_chunks * CH ? pd.np : w.c0 + w.n_chunks * CH;
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
        // the stored values only ever decrease, so a (possibly stale) plain read that is already <= ours
        // proves the atomic would change nothing: most chunks of pass B improve few of their queries
#pragma unroll
        for (int k = 0; k < QPT; ++k)
            if (oi[k] >= 0) {
                const unsigned long long v = (unsigned long long)__double_as_longlong(m[k]);
                if (v < out[pd.out_off + oi[k]]) atomicMin(&out[pd.out_off + oi[k]], v);
            }
    }
}

// Radius counts: cnt[q] += #{p in the chunk : |q - p|^2 <= r2}, the same squared distance, exact f64 -- the
// neighbour counts of clean_up_non_section_points (scale_coronary.rs:342-409; rstar's locate_within_distance keeps
// the points with distance_2 <= the squared radius).  Work items are the (query block, chunk) combinations whose
// bounding boxes come within the radius of each other (host); counts merge with a 32-bit atomicAdd.
template <int QPT>
__global__ void __launch_bounds__(256)
k_nn3_count(const NnPair* __restrict__ pairs, const NnWork* __restrict__ work, int n_work,
            const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ pz,
            const int32_t* __restrict__ qperm, double r2, unsigned int* __restrict__ out)
{
    constexpr int NT = 256, CH = kNnChunk;
    __shared__ double4 s_p[CH];
    const int tid = threadIdx.x;
    for (int wi = (int)gridDim.x == n_work ? nn_xcd_work_index(blockIdx.x, n_work) : (int)blockIdx.x; wi < n_work;
         wi += gridDim.x) {
        const NnWork w = work[wi];
        const NnPair pd = pairs[w.pair];
        int oi[QPT];
        double qx[QPT], qy[QPT], qz[QPT];
        unsigned int c[QPT];
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const int q = w.q0 + k * NT + tid;
            oi[k] = q < pd.nq ? (pd.qperm_off >= 0 ? qperm[pd.qperm_off + q] : q) : -1;
            const int qc = q < pd.nq ? q : pd.nq - 1;
            qx[k] = px[pd.q_off + qc]; qy[k] = py[pd.q_off + qc]; qz[k] = pz[pd.q_off + qc];
            c[k] = 0u;
        }
        const int c_end = pd.np - w.c0 < w.n_chunks * CH ? pd.np : w.c0 + w.n_chunks * CH;
        for (int c0 = w.c0; c0 < c_end; c0 += CH) {
            const int n = c_end - c0 < CH ? c_end - c0 : CH;
            __syncthreads();
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
                    c[k] += v <= r2 ? 1u : 0u;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < QPT; ++k)
            if (oi[k] >= 0 && c[k]) atomicAdd(&out[pd.out_off + oi[k]], c[k]);
    }
}

// queries per lane: 1..4 measured within 5 % of each other on MI355X (10.9 / 10.4 / 10.4 / 10.3 ms for the
// 3.3e10-pair search of tools/bench_ccta.py): the kernel is bound by fp64 VALU issue, not by the LDS reads
static constexpr int kNnQpt = 2;
int nn_queries_per_block() { return 256 * kNnQpt; }
int nn_chunk_points() { return kNnChunk; }
int nn_span_chunks() { return kNnSpan; }   // chunks per work item where nothing is pruned

// Per pair: the sum of its minima in index order, one addition at a time -- the sequential fold of
// symmetric_nn_distance (scale_coronary.rs:193-211), so that a search only has to bring 82 numbers back
// instead of every minimum.  One wave per pair: all lanes fetch a tile into LDS (coalesced), lane 0 folds it
// (the chain of dependent adds is the cost: ~20 k x a few cycles; the LDS reads run ahead of it).
__global__ void __launch_bounds__(64)
k_nn3_sums(const NnPair* __restrict__ pairs, int n_pairs, const double* __restrict__ out, double* __restrict__ sums)
{
    constexpr int T = 2048;
    __shared__ double s_v[T];
    const NnPair pd = pairs[blockIdx.x];
    const double* v = out + pd.out_off;
    double s = 0.0;
    for (int i0 = 0; i0 < pd.nq; i0 += T) {
        const int n = pd.nq - i0 < T ? pd.nq - i0 : T;
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 64) s_v[j] = v[i0 + j];
        __syncthreads();
        if (threadIdx.x == 0) {
            int j = 0;
            for (; j + 8 <= n; j += 8) {
                const double a0 = s_v[j], a1 = s_v[j + 1], a2 = s_v[j + 2], a3 = s_v[j + 3];
                const double a4 = s_v[j + 4], a5 = s_v[j + 5], a6 = s_v[j + 6], a7 = s_v[j + 7];
                s += a0; s += a1; s += a2; s += a3; s += a4; s += a5; s += a6; s += a7;
            }
            for (; j < n; ++j) s += s_v[j];
        }
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = s;
}

hipError_t launch_nn3_sums(const void* pairs, int n_pairs, const double* out, double* sums, hipStream_t s)
{
    if (n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_nn3_sums, dim3((unsigned)n_pairs), dim3(64), 0, s, (const NnPair*)pairs, n_pairs, out, sums);
    return hipGetLastError();
}

hipError_t launch_nn3_morph(const void* items, int n_items, const double* aux, long long n_aux, double* px, double* py,
                            double* pz, hipStream_t s)
{
    if (n_items <= 0) return hipSuccess;
    // grid.x covers the largest set; blocks past a set's end exit
    hipLaunchKernelGGL(k_nn3_morph, dim3((unsigned)((n_aux + 255) / 256), (unsigned)n_items), dim3(256), 0, s,
                       (const NnMorph*)items, aux, n_aux, px, py, pz);
    return hipGetLastError();
}

// out[n_out] (u32, zeroed here) += neighbours within sqrt(r2) over the work items
hipError_t launch_nn3_count(const void* pairs, const void* work, int n_work, const double* px, const double* py,
                            const double* pz, const int32_t* qperm, double r2, unsigned int* out, long long n_out,
                            hipStream_t s)
{
    if (n_out > 0) {
        const hipError_t e = hipMemsetAsync(out, 0, (size_t)n_out * 4, s);
        if (e != hipSuccess) return e;
    }
    if (n_work > 0)
        hipLaunchKernelGGL((k_nn3_count<kNnQpt>), dim3((unsigned)n_work), dim3(256), 0, s, (const NnPair*)pairs,
                           (const NnWork*)work, n_work, px, py, pz, qperm, r2, out);
    return hipGetLastError();
}

// work_a: items that always run (n_a of them); work_b: items that first check their bound (n_b)
hipError_t launch_nn3_min(const void* pairs, const void* work_a, int n_a, const void* work_b, int n_b, const double* px,
                          const double* py, const double* pz, const int32_t* qperm, double* out, long long n_out,
                          hipStream_t s)
{
    if (n_out > 0)
        hipLaunchKernelGGL(k_nn3_fill, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s,
                           (unsigned long long*)out, n_out);
    if (n_a > 0)
        hipLaunchKernelGGL((k_nn3_min<kNnQpt, false>), dim3((unsigned)n_a), dim3(256), 0, s, (const NnPair*)pairs,
                           (const NnWork*)work_a, n_a, px, py, pz, qperm, (unsigned long long*)out);
    if (n_b > 0)
        hipLaunchKernelGGL((k_nn3_min<kNnQpt, true>), dim3((unsigned)n_b), dim3(256), 0, s, (const NnPair*)pairs,
                           (const NnWork*)work_b, n_b, px, py, pz, qperm, (unsigned long long*)out);
    return hipGetLastError();
}

}  // namespace mm
