// mm_device.h -- structures shared by the HIP kernels (mm_kernels.hip) and the host
// engine (mm_engine.cpp).  Internal; the public boundary is include/mm_hausdorff.h.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace mm {

// One (reference set, target set, candidate list) search.  Point sets live in one pool
// (several pairs may share a set: frame i is the target of pair i and the reference of
// pair i+1); cos/sin tables are shared between pairs with identical candidate lists.
struct PairDesc {
    int32_t ref_off, n_ref;   // into the point pool
    int32_t tgt_off, n_tgt;
    int32_t tab_off;          // into the cos/sin tables
    int32_t out_off;          // into the per-candidate outputs
    int32_t n_ang;            // candidates of this pair in this plan (slice)
    int32_t flags;            // MM_SEARCH_SKIP_ZERO
    int32_t ang_full;         // length of the pair's full candidate list (>= n_ang)
    int32_t ang_begin;        // first candidate of the slice this plan owns
    int32_t n_slice;          // candidates in the slice (== n_ang, except for empty-set pairs: n_ang = 0 there)
    int32_t pad0;             // k_screen_mx: scale exponent e (coordinates are multiplied by 2^e); otherwise 0
    double  cx, cy;           // rotation centre (exact kernel)
    double  delta;            // f32 screening error bound (same unit as the costs)
    double  tol2;             // candidates within tol2 of the exact minimum are reported as near-ties
    double  e2;               // absolute error bound of the screened SQUARED value (fast kernel; else 0)
    double  rho_t;            // largest distance of a target point from the rotation centre
};

// Source of one search set of the within-pullback search, built on the device from the raw pullback
// (align_within.rs:173-191: downsample(lumen, S) ++ downsample(catheter, ceil(n_cath * S / len_lumen0)),
// contour.rs:47-58), centred on the frame's centroid.  Raw pool = xyz triples (f64) as the caller holds them.
struct SetSrc {
    int64_t lum_at;           // first lumen point of the frame in the raw pool (point index)
    int64_t cath_at;          // first catheter point of the frame (unused when cath_take == 0)
    int32_t lum_len, lum_take;    // points of the contour, points asked for
    int32_t cath_len, cath_take;
    int32_t dst_off, n;       // into the point pool; n = min(lum_len, lum_take) + min(cath_len, cath_take)
    double  cx, cy;           // the frame's centroid
};

// One workgroup's share: `cnt` consecutive candidates of one pair.
struct WorkItem {
    int32_t pair, a0, cnt, pad;
};

static constexpr int kMaxNear = 8;  // near-tie slots per pair (MM_MAX_NEAR)

struct BatchDev {
    const PairDesc* pairs;
    const WorkItem* work;
    int32_t n_pairs, n_work;
    // point pool: f32 copy relative to the set's centre, f64 copy as given
    const float *p32x, *p32y;
    const double *p64x, *p64y;
    // candidate tables
    const float *cos32, *sin32;
    const double *cos64, *sin64;
    const double *ang64;      // the candidate angles themselves (same indexing): device-side exchange records
    // per-candidate outputs
    float*    sq32;       // squared Hausdorff from the screening kernel
    double*   sq64;       // exact squared Hausdorff (valid where flag != 0 or in exact mode)
    uint8_t*  flag;       // 1 = shortlisted (re-scored in f64); before that, bounded search on the matrix pipe: 1 = needs a bound
    int64_t   n_cand;     // candidates of the level (length of sq32 / lb32 / flag)
    // shortlist queue
    WorkItem* items;      // capacity = total candidates (bounded mode: first the survivors' runs)
    int32_t*  n_items;    // 8 device counters: [0] re-score queue; bounded mode: [1] picks, [2] queue 0 (round 2),
                          // [3] queue 1 (round 3), [4] second picks, [5] queue 2 (survivors)
    // bounded screen (MM_PRECISION_F32_BOUNDED)
    const WorkItem* work_lb;   // bound kernel's work list (more candidates per workgroup)
    int32_t   n_work_lb, lb_stride;
    int32_t   lb_mx;       // > 0: the bounds come from the matrix pipe (k_bound_mx); value = row tiles per set of its LDS layout
    int32_t   lb_mx_qt, lb_mx_nc;   // its variant: column tiles of queries per side (1 | 2), candidates per wave at once (1 | 2)
    int32_t   kept_mx_nct, kept_mx_acap;   // > 0: the picks' and the survivors' screen is k_screen_mx<kept_mx_nct, false> (every pair: that variant)
    float*    lb32;        // per-candidate lower bound of the screened squared value
    int32_t*  pick_idx;    // [2 * n_pairs] per pair: candidate with the smallest bound after round 1 / round 3 (-1: none)
    WorkItem* items_pick;  // [2 * n_pairs] queue entries of the two picks
    WorkItem* items_lb;    // three queues of `runs cap` entries: round 2, round 3, survivors
    int32_t*  klist;       // [n_cand] per pair (at its out_off): the survivors' candidate indices (k_lb_keep_mx)
    float*    emit;        // per pair emit_rows + emit_cols: row / column minima of the first pick
    int32_t   emit_rows, emit_cols;
    int32_t*  qlist;       // per pair 2 * lb_list_queries(): decisive reference / target point indices
    unsigned long long* stats;  // nullable: [1] candidates bounded in round 2, [3] in round 3, [2] fully screened
    // per-pair results
    double*   best_cost;
    int32_t*  best_idx;
    int32_t*  n_rescored;
    int32_t*  near_cnt;   // exact-scored candidates with cost <= best + tol2 (may exceed kMaxNear)
    int32_t*  near_idx;   // [n_pairs * kMaxNear] ascending candidate indices (full-list numbering)
    double*   all_costs;  // optional per-candidate sqrt'ed costs
};

// Launchers (mm_kernels.hip).  All asynchronous on `s`.
hipError_t launch_screen_f32(const BatchDev& b, int max_na, int max_nbp, hipStream_t s);
hipError_t launch_screen_fast(const BatchDev& b, int max_na, int max_nbp, hipStream_t s);
// matrix-pipe screen (MM_PRECISION_F32_MATRIX) over the work items [work_begin, work_begin + n_work) of b.work: every one
// of their pairs must have sets of mx_min_points() .. mx_max_points() points, a target set whose variant (mx_variant) is
// (nct, multi), a reference set of at most a_cap row tiles of 32, and PairDesc::pad0 = the pair's scale exponent
hipError_t launch_screen_mx(const BatchDev& b, int work_begin, int n_work, int nct, int multi, int a_cap, hipStream_t s);
hipError_t launch_screen_none(const BatchDev& b, int work_begin, int n_work, hipStream_t s);   // screened value 0 for every candidate
void       mx_variant(int n_tgt, int* nct, int* multi);
size_t     lds_bytes_mx(int nct, bool multi, int a_cap, int waves);
int        mx_min_points();
int        mx_max_points();
int        max_rows_fast();
int        max_target_points_fast();
// bounded screen: lower bound of every lb_candidate_step()-th candidate -> per-pair pick -> full screen of
// the picks (upper bound) -> spread the bounds to the candidates in between (chord inequality) ->
// lower bound of those still possible -> survivors -> full screen of the survivors (runs of <= 8
// candidates, at most `cap` of them)
hipError_t launch_screen_lb(const BatchDev& b, int max_nap, int max_nbp, hipStream_t s);
hipError_t launch_lb_pick(const BatchDev& b, int round, hipStream_t s);
hipError_t launch_screen_picks(const BatchDev& b, int round, int max_na, int max_nbp, hipStream_t s);
hipError_t launch_lb_spread(const BatchDev& b, hipStream_t s);
hipError_t launch_screen_lb_queued(const BatchDev& b, int max_nap, int max_nbp, int cap, hipStream_t s);
hipError_t launch_lb_keep(const BatchDev& b, int final, int cap, hipStream_t s);
// round 3: the pick's decisive points (largest row / column minima) as queries for the survivors
hipError_t launch_lb_topk(const BatchDev& b, int max_n, hipStream_t s);
hipError_t launch_screen_lb_list(const BatchDev& b, int max_nap, int max_nbp, int cap, hipStream_t s);
int        lb_list_queries();
int        lb_candidate_step();
int        lb_sparse_candidates(int n);   // candidates of a list of n the first round scores
hipError_t launch_screen_kept(const BatchDev& b, int max_na, int max_nbp, int cap, hipStream_t s);
int        lb_max_query_points();   // subset size the bound kernel holds in registers
int        lb_mx_max_points();      // largest set (either side) the matrix-pipe bound kernel stages in LDS
int        lb_max_points();         // largest set (either side) the bound kernel stages in LDS
// large-set Hausdorff (no LDS limit on the set sizes); pairs/work are device arrays of the kernel's
// LargePair {a_off, na, b_off, nb, col_off, pad} / LargeWork {pair, row0} records
hipError_t launch_hausdorff_large(const void* pairs, const void* work, int n_pairs, int n_work, const double* px,
                                  const double* py, void* colmin, long long n_col, void* rowmax, double* out,
                                  hipStream_t s);
// lower bounds of the same pairs: two directed entries per output slot (a->b and b->a; col_off = slot,
// pad = subset stride), out[slot] <= the pair's Hausdorff distance, computed from the same d^2 bits
hipError_t launch_hausdorff_large_bound(const void* pairs, const void* work, int n_out, int n_work, const double* px,
                                        const double* py, void* rowmax, double* out, hipStream_t s);
int        large_rows_per_block();
// 3-D nearest-neighbour squared distances (mm_nn_kernels.hip); pairs/work are device arrays of the kernel's
// NnPair {q_off, nq, p_off, np, out_off, qperm_off} / NnWork {pair, q0, c0, n_chunks, lb2 (f64)} records:
// work_a always runs, work_b items first check their bound against their queries' current minima
hipError_t launch_nn3_min(const void* pairs, const void* work_a, int n_a, const void* work_b, int n_b, const double* px,
                          const double* py, const double* pz, const int32_t* qperm, double* out, long long n_out,
                          hipStream_t s);
// neighbour counts within a radius (same records; every work item runs): out[n_out] u32
hipError_t launch_nn3_count(const void* pairs, const void* work, int n_work, const double* px, const double* py,
                            const double* pz, const int32_t* qperm, double r2, unsigned int* out, long long n_out,
                            hipStream_t s);
// derived sets (NnMorph {dst_off, n, aux_off, pad, adj (f64)}): base point + unit vector * adj where flagged,
// from 7 planes of n_aux doubles (bx by bz ux uy uz flag), written into the SoA point pool
hipError_t launch_nn3_morph(const void* items, int n_items, const double* aux, long long n_aux, double* px, double* py,
                            double* pz, hipStream_t s);
// sums[p] = the minima of pair p added up in index order (sequential f64 fold)
hipError_t launch_nn3_sums(const void* pairs, int n_pairs, const double* out, double* sums, hipStream_t s);
int        nn_queries_per_block();
int        nn_chunk_points();
int        nn_span_chunks();
hipError_t launch_exact_all(const BatchDev& b, int max_na, int max_nbp, hipStream_t s);
// bytes between HBM and pinned host memory by a 256-thread kernel (see k_copy_small: a runtime copy behind a
// kernel is a 512-thread blit that starves beside another stream's screen launch); 16-byte aligned pointers
hipError_t launch_copy_small(void* dst, const void* src, size_t bytes, hipStream_t s);
// search sets built on the device: one workgroup per set; writes the four planes of the point pool and, per
// set, the largest squared distance from the centre (rho2) and the largest coordinate magnitude before / after
// centring (scale)
hipError_t launch_build_sets(const SetSrc* src, int n_sets, const double* raw, float* p32x, float* p32y, double* p64x,
                             double* p64y, double* rho2, double* scale, hipStream_t s);
hipError_t launch_shortlist(const BatchDev& b, hipStream_t s);
hipError_t launch_rescore(const BatchDev& b, int max_na, int max_nbp, int total_candidates, hipStream_t s);
hipError_t launch_finalize(const BatchDev& b, int use_flags, hipStream_t s);
// Device-side exchange records of a sharded search (candidate axis split over ranks), indexed by JOB
// (pair_of_job[j] = this level's pair of job j, or -1 when the job takes no part in the level):
//   export_cost  cost[j] = exact first-minimum cost inside this rank's slice, +inf if it holds no candidate
//                -> all-reduce(MIN) over the ranks gives the global best cost
//   export_keys  given the reduced costs g: keys[j] = first-minimum index if this rank attains g[j], else
//                INT64_MAX; keys[n+j], keys[2n+j] = (angle bits, ~angle bits) of the rank's winner if its
//                minimum lies within the pair's tie tolerance of g[j] and all its near-ties are one angle
//                value, (INT64_MIN, INT64_MIN) if they are not, (INT64_MAX, INT64_MAX) if it is not near
//                -> ONE all-reduce(MIN) of the 3n keys gives the first index of minimal cost over the whole
//                axis (process_utils.rs:72) and min / ~max of the near ranks' angle bits (equal <=> decided)
hipError_t launch_export_cost(const BatchDev& b, const int32_t* pair_of_job, int n_jobs, double* cost, hipStream_t s);
hipError_t launch_export_keys(const BatchDev& b, const int32_t* pair_of_job, int n_jobs, const double* gcost,
                              long long* keys, hipStream_t s);
size_t     lds_bytes_f32(int nbp);
size_t     lds_bytes_f64(int nbp);
int        max_target_points_f32();
int        max_target_points_f64();

}  // namespace mm
