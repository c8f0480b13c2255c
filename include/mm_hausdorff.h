/*
 * mm_hausdorff.h -- C ABI of the MI355X Hausdorff pose-search engine.
 *
 * Drop-in boundary for the three `search_range(cost = hausdorff_distance o rotate)` call
 * sites of yungselm/multimoda-rs (paths relative to the reference checkout):
 *   src/intravascular/processing/align_within.rs:97-119    (within-pullback chain)
 *   src/intravascular/processing/align_between.rs:46-47    (between-pullback search)
 *   src/intravascular/centerline_align/align_algorithms.rs:369-441 (refine, angle x index)
 * and for the metric itself, src/intravascular/processing/process_utils.rs:33-121.
 *
 * Plain pointers and sizes only; every pointer is caller-owned host memory unless the
 * name says `dev`.  All coordinates are f64 SoA (x[], y[]); z never enters the metric
 * (process_utils.rs:105-107).  Functions return 0 on success or a negative mm_status;
 * mm_last_error() returns a thread-local message (the reference surfaces anyhow errors
 * as PyRuntimeError(format!("{e:#}")), binding/functions.rs:228).
 *
 * There is NO CPU fallback behind this ABI: without a HIP device every compute entry
 * point fails with MM_ERR_NO_DEVICE.
 *
 * Threads: an engine (its stream and staging buffers) serves one call at a time; a host that
 * works from several threads (the reference's crossbeam scopes, entry.rs:140-277) gives each its
 * own engine.  Every entry point selects the engine's device for the calling thread, so engines
 * and plans may be used from a thread other than the one that created them.
 */
#ifndef MM_HAUSDORFF_H
#define MM_HAUSDORFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MM_OK               =  0,
    MM_ERR_NO_DEVICE    = -1,  /* no HIP device / HIP runtime error            */
    MM_ERR_INVALID      = -2,  /* bad argument                                 */
    MM_ERR_TOO_LARGE    = -3,  /* a point set exceeds the kernel's LDS budget  */
    MM_ERR_NO_FRAMES    = -4,  /* "Geometry contains no frames"      (align_within.rs:32-34) */
    MM_ERR_NO_POINTS    = -5,  /* "Lumen contours have no points"    (align_within.rs:35-37) */
    MM_ERR_SAMPLE_SIZE  = -6,  /* "sample_size must be > 0"          (align_within.rs:38-40) */
    MM_ERR_HIP          = -7,
    MM_ERR_REF_INDEX    = -8,  /* reference-frame index out of range (Rust would panic)      */
    MM_ERR_INTEGRITY    = -9,  /* a check of check_geometry_integrity failed (integrity_check.rs:8-33);
                                  mm_last_error() holds the reference's message                */
    MM_ERR_COMM         = -10  /* RCCL could not be loaded, or a collective / communicator call failed */
} mm_status;

/* precision of the candidate scoring */
enum {
    MM_PRECISION_F64 = 0, /* every candidate scored in f64 with the reference's exact
                             operation order (bit-identical costs)                        */
    MM_PRECISION_F32 = 1, /* f32 screening of every candidate + f64 exact re-score of all
                             candidates within 2*delta of the f32 minimum; the winner and
                             its cost are bit-identical to MM_PRECISION_F64              */
    MM_PRECISION_F32_FAST = 2, /* same contract; screening in the expanded distance form
                             |a|^2 + |b|^2 - 2ab (25 % fewer packed instructions; absolute error
                             5*2^-24*(rho_a+rho_b)^2 on the squared value -> wider shortlist) */
    MM_PRECISION_F32_BOUNDED = 3, /* same contract (winner and cost bit-identical to MM_PRECISION_F64);
                             before the expanded-form screen every candidate gets a LOWER bound of its
                             Hausdorff distance from every k-th point of either set against all points
                             of the other (2/k of the distance matrix), one full evaluation per pair
                             gives an upper bound, and only candidates whose bound does not exceed it
                             are screened and re-scored.  Candidates ruled out are never the minimum.
                             Screens every candidate like MM_PRECISION_F32_MATRIX when per-candidate costs
                             are requested, the batch is small (mm_engine_set_bound_min_candidates)
                             or a set exceeds the bound kernel's LDS budget                          */
    MM_PRECISION_F32_MATRIX = 4 /* same contract; the screen's squared distances come from the f16 matrix pipe: one
                             v_mfma_f32_32x32x16_f16 per 32 x 32 tile over coordinates split into f16 hi + lo pieces
                             (22 significant bits, fp32 accumulation; absolute error 2^-24*(47 R^2 + 6 rho_a^2 + 27 rho_b^2), R = rho_a+rho_b, on the
                             squared value -> a wider shortlist, same winners and costs after the exact re-score).
                             Chosen per PAIR for sets of 64 .. 2048 points on either side (the reference's sample_size /
                             n_points are user kwargs, binding/functions.rs:144-167); a pair outside that range takes the
                             packed-FMA or the direct-form screen, the other pairs of its batch are not affected   */
    /* CAUTION (round 4, MI355X): the kernels of MM_PRECISION_F32 and MM_PRECISION_F32_FAST -- packed-FMA vector code --
     * returned WRONG screened values in 3 - 15 % of the calls while a kernel that executes MFMAs (this library's matrix-pipe
     * screen on another stream, or any MFMA loop of another process) ran on the chip at the same time; the matrix-pipe
     * kernels, the exact f64 kernels and the small per-pair kernels did not.  Cause unknown (profiles/README.md, tools/
     * probe_mfma_interference.py).  Under MM_PRECISION_F32_MATRIX, _BOUNDED and _F64 every value that decides a result comes
     * from an MFMA or an f64 kernel; they are the defaults of every layer above this header. */
};

/* flags of one search */
enum {
    MM_SEARCH_SKIP_ZERO = 1 /* apply ContourPoint::rotate's `angle == 0.0 -> unchanged`
                               shortcut (contour_point.rs:39-41); set for the within
                               closure, clear for the between closure (align_between.rs:194-209) */
};

typedef struct mm_engine mm_engine; /* one HIP device + stream + grow-only workspaces */

/* ---- runtime ---------------------------------------------------------------------- */
int         mm_device_count(void);
const char* mm_last_error(void);
const char* mm_version(void);

/* device < 0 -> current device.  stream == NULL -> the engine creates its own two streams: a main stream for
 * the searches of device-resident plans (the long launches) and a high-priority side stream for everything short
 * (staging, the between-pullback searches, the per-step searches of the faithful chain), so that a driver which
 * overlaps consecutive cases on two engines does not queue a 20 us kernel behind the other engine's 30 ms launch.
 * Otherwise `stream` is a hipStream_t the caller owns (e.g. torch's current stream), and every kernel of this
 * engine is launched on it.  mm_engine_stream returns the main stream. */
int  mm_engine_create(int device, void* stream, mm_engine** out);
void mm_engine_destroy(mm_engine* e);
int  mm_engine_synchronize(mm_engine* e);
void* mm_engine_stream(mm_engine* e);
/* Order `waiter`'s main stream behind the dominant (long) kernel of the search most recently enqueued on `other`
 * (same device): whatever is enqueued on `waiter` next starts when that launch ends, beside the short tail of
 * `other`'s search (shortlist, re-score, argmin, result copy).  A driver that aligns independent cases back to back
 * uses it to keep the device busy across the hand-over: launch case k+1, then collect case k.  No-op if `other` has
 * not searched yet. */
int  mm_engine_wait_search(mm_engine* waiter, mm_engine* other);
/* The same for a SHARDED search: order `waiter`'s main stream behind the whole exchange of the level most recently
 * enqueued on `other` by mm_within_plan_search_sharded_begin (export kernels, both all-reduces, the copy of the reduced
 * records).  The next case's launch then starts when this case's last copy ends, and no collective ever has to find
 * room beside a launch that fills the device.  No-op if `other` has not enqueued an exchange yet. */
int  mm_engine_wait_exchange(mm_engine* waiter, mm_engine* other);

/* Per-launch timing of the dominant (candidate-scoring) kernel with hipEvents recorded on
 * the engine's stream around every launch.  mm_engine_profile_read synchronizes, returns
 * the number of launches, their summed device time (ms), the pair-distance evaluations
 * (2*Na*Nb per candidate) and candidates they covered, and resets the accumulators. */
int  mm_engine_profile(mm_engine* e, int enable);
int  mm_engine_profile_read(mm_engine* e, int64_t* n_launches, double* ms_total,
                            double* pair_evals, int64_t* candidates);
/* The same measurements per launch (call before mm_engine_profile_read, which resets them):
 * device time in ms and pair-distance evaluations of up to cap launches, in launch order. */
int  mm_engine_profile_launches(mm_engine* e, int64_t cap, float* ms, double* pair_evals,
                                int64_t* n_launches);
/* MM_PRECISION_F32_BOUNDED, accumulated since mm_engine_profile(e, 1): out[0] candidates offered,
 * out[1] candidates given a lower bound in the first (sparse) round, out[2] in the second round,
 * out[3] in the third (decisive-point) round, out[4] candidates that went through the full f32
 * screen (the two per-pair picks not counted). */
int  mm_engine_bound_stats(mm_engine* e, int64_t out[5]);
/* Which kernel screened how many candidates since the engine was created (brute-force levels; the bounded search's
 * rounds are in mm_engine_bound_stats): out[0] direct-form f32, out[1] packed-FMA (expanded form), out[2] matrix pipe with
 * the whole target set per wave (64 .. 544 points), out[3] matrix pipe with the target set in column blocks (545 .. 2048
 * points), out[4] every candidate in exact f64 (MM_PRECISION_F64, or a level whose coordinates no f32 screen can hold).
 * MM_PRECISION_F32_MATRIX chooses per PAIR: only pairs with a set of fewer than 64 or more than 2048 points (or a radius
 * beyond 1e+-30) show up under out[0] / out[1]. */
int  mm_engine_screen_stats(mm_engine* e, int64_t out[5]);
/* MM_PRECISION_F32_BOUNDED runs its bound rounds only on batches of at least n candidates (default
 * 16384): a dozen dependent launches cost more than screening a small batch outright. 0 = always. */
int  mm_engine_set_bound_min_candidates(mm_engine* e, int64_t n);
/* MM_PRECISION_F32_BOUNDED computes its lower bounds, its picks and its survivors on the f16 matrix pipe when every pair of
 * the batch has sets of 64 .. 544 points and the target sets share one column-tile count (default, on != 0; a batch of other
 * shapes is screened outright on the matrix pipe, without bound rounds); on == 0 keeps the packed-FMA kernels of rounds 1-3
 * (the A/B switch of bench.py's bounded_search leg).  Same winners and costs either way. */
int  mm_engine_set_bound_matrix(mm_engine* e, int on);

/* TEST HOOK (nothing in the product calls it): the lower bound MM_PRECISION_F32_BOUNDED's bound kernels give every
 * candidate of one search -- out_lb2[i] <= (exact cost of candidate i)^2 up to *e2 (error bound of the kernel's squared
 * values) and *delta (of the distance); *stride = every stride-th point of either set was a query.  matrix != 0: the
 * matrix-pipe kernel (sets of 64 .. 544 points), 0: the packed-FMA kernel. */
int  mm_lower_bounds(mm_engine* e, const double* rx, const double* ry, int nr, const double* tx, const double* ty, int nt,
                     double cx, double cy, const double* angles, int n_angles, int flags, int matrix, float* out_lb2,
                     double* e2, double* delta, int* stride);
/* TEST HOOK (nothing in the product calls it): what the first pick of the matrix-pipe bounded search leaves for the choice
 * of the third round's queries -- for ONE candidate angle the squared distance from every reference point to its nearest
 * rotated target point (row_min2[nr]) and from every rotated target point to its nearest reference point (col_min2[nt]),
 * as the f16 hi+lo matrix kernel computes them (each within *e2 of the exact squared value), and the screened squared
 * Hausdorff value (*value2 = the maximum over both).  Sets of 64 .. 528 points. */
int  mm_pick_minima(mm_engine* e, const double* rx, const double* ry, int nr, const double* tx, const double* ty, int nt,
                    double cx, double cy, double angle, int flags, float* row_min2, float* col_min2, float* value2, double* e2);

/* ---- the metric: hausdorff_distance (process_utils.rs:78-82) ------------------------ */
/* f64-exact on the device; empty set on either side -> 0.0 (process_utils.rs:86-88). */
int mm_hausdorff_2d(mm_engine* e,
                    const double* ax, const double* ay, int na,
                    const double* bx, const double* by, int nb,
                    double* out);

/* Batched metric: out[p] = hausdorff_distance(A_p, B_p), f64-exact.  This is the call at
 * centerline_align/align_algorithms.rs:431 for every (index, angle) candidate of
 * refine_alignment_hausdorff at once.  first_min (nullable) receives the index of the first
 * strict minimum (`if hausdorff_dist < min_hausdorff`, :433-437; candidates are in the
 * caller's enumeration order), -1 if n_pairs == 0. */
int mm_hausdorff_batch(mm_engine* e, int n_pairs,
                       const int64_t* a_off, const double* ax, const double* ay,
                       const int64_t* b_off, const double* bx, const double* by,
                       double* out, int32_t* first_min);

/* Host helpers of refine_alignment_hausdorff (align_algorithms.rs:339-451), exact:
 *   mm_refine_angles            `angle = init - range; while angle <= init + range { ..; angle += step }`
 *                               (accumulated, :386-387,439); returns the count
 *   mm_filter_points_in_region  indices of the points inside the +-5 mm box around two
 *                               centerline points (:454-505); xyz triples
 *   mm_refine_downsample_count  clamp(ceil(|filtered| / (M*F) * M), 1, M)  (:415-418)        */
int64_t mm_refine_angles(double initial, double range, double step, double* out, int64_t cap);
int64_t mm_filter_points_in_region(const double* xyz, int64_t n, const double* start_xyz,
                                   const double* end_xyz, int64_t* out_idx, int64_t cap);
int64_t mm_refine_downsample_count(int64_t n_filtered, int64_t n_points_per_frame, int64_t n_frames);

/* ---- candidate enumeration of search_range (process_utils.rs:43-67) ---------------- */
/* Host-side, exact: writes up to `cap` wrapped angles, returns their count. When the
 * reference returns early (step <= 0, or stop <= start) *degenerate = 1 and
 * *early_value is the value it returns. has_center = 0 <=> center_angle = None.  A list of more than 2^24
 * candidates (a step far below the range; below the spacing of the doubles the reference would not terminate) is
 * refused: MM_ERR_TOO_LARGE, here and in every search that enumerates its candidates from (step, range).
 * mm_refine_angles likewise stops with MM_ERR_TOO_LARGE after 2^22 angles. */
int64_t mm_search_angles(double step_deg, double range_deg, int has_center, double center,
                         double limes_deg, double* out, int64_t cap,
                         int* degenerate, double* early_value);

/* ---- one search: search_range(|a| hausdorff(ref, rotate(tgt, a, centre))) ----------- */
/* `angles` is the host-generated candidate list (radians, already wrapped). Returns the
 * FIRST index of minimal cost (process_utils.rs:72 ordered reduce). all_costs (nullable,
 * n_angles) receives sqrt'ed costs: exact for re-scored candidates, f32-derived for the
 * rest when precision == MM_PRECISION_F32. */
int mm_best_rotation(mm_engine* e,
                     const double* rx, const double* ry, int nr,
                     const double* tx, const double* ty, int nt,
                     double cx, double cy,
                     const double* angles, int n_angles, int flags, int precision,
                     double* best_angle, double* best_cost, int* best_idx,
                     double* all_costs);

/* ---- batched searches (one launch sequence for all pairs) ---------------------------- */
/* Pair p uses reference points  ref_[xy][ref_off[p] .. ref_off[p+1]),
 *             target points     tgt_[xy][tgt_off[p] .. tgt_off[p+1]),
 *             candidates        angles[ang_off[p] .. ang_off[p+1]),
 *             rotation centre   (cx[p], cy[p]),  flags[p] (nullable -> 0).
 * Outputs (n_pairs each): best_idx (index into the pair's own candidate list),
 * best_angle, best_cost; n_rescored (nullable) = f64 re-scores done for the pair;
 * all_costs (nullable, ang_off[n_pairs] entries). */
int mm_best_rotation_batch(mm_engine* e, int n_pairs,
                           const int64_t* ref_off, const double* ref_x, const double* ref_y,
                           const int64_t* tgt_off, const double* tgt_x, const double* tgt_y,
                           const int64_t* ang_off, const double* angles,
                           const double* cx, const double* cy, const int32_t* flags,
                           int precision,
                           int32_t* best_idx, double* best_angle, double* best_cost,
                           int32_t* n_rescored, double* all_costs);

/* ---- device-resident plan: upload once, run many times ------------------------------ */
/* Same arguments as mm_best_rotation_batch; the batch is staged into HBM at creation.
 * angle_begin/angle_end restrict every pair's candidate list to the slice
 * [angle_begin, min(angle_end, n_angles_p)) -- the shard one rank owns when the candidate
 * axis is split across GPUs; pass 0 / INT32_MAX for everything. */
typedef struct mm_plan mm_plan;
int  mm_plan_create(mm_engine* e, int n_pairs,
                    const int64_t* ref_off, const double* ref_x, const double* ref_y,
                    const int64_t* tgt_off, const double* tgt_x, const double* tgt_y,
                    const int64_t* ang_off, const double* angles,
                    const double* cx, const double* cy, const int32_t* flags,
                    int precision, int32_t angle_begin, int32_t angle_end,
                    mm_plan** out);
/* The same with the point sets given once and referenced by index: pair p searches reference
 * set ref_set[p] against target set tgt_set[p] (a set may serve any number of pairs, e.g. every
 * frame of a pullback against a window of neighbouring frames).  Set s is x/y[set_off[s] ..
 * set_off[s+1]); its f32 copy is taken relative to (set_cx[s], set_cy[s]), which must equal the
 * rotation centre of every pair that uses it.  shared_angles != 0: every pair uses the one
 * candidate list angles[0 .. ang_off[1]) (ang_off has 2 entries).  want_costs != 0 keeps the
 * per-candidate costs for mm_plan_fetch(all_costs). */
int  mm_plan_create_indexed(mm_engine* e, int n_sets, const int64_t* set_off, const double* x,
                            const double* y, const double* set_cx, const double* set_cy,
                            int n_pairs, const int32_t* ref_set, const int32_t* tgt_set,
                            const int64_t* ang_off, const double* angles, int shared_angles,
                            const double* cx, const double* cy, const int32_t* flags,
                            int precision, int want_costs, mm_plan** out);
void mm_plan_destroy(mm_plan* p);
/* Enqueue the whole search (screen -> shortlist -> exact re-score -> argmin) on the
 * engine's stream; asynchronous. */
int  mm_plan_run(mm_plan* p);
/* Enqueue only the dominant screening kernel (for roofline timing with HIP events). */
int  mm_plan_run_screen_only(mm_plan* p);
/* Wait for the stream and copy results to the host (same meaning as the batch call;
 * best_idx is relative to the pair's FULL candidate list, -1 if the slice was empty). */
int  mm_plan_fetch(mm_plan* p, int32_t* best_idx, double* best_angle, double* best_cost,
                   int32_t* n_rescored, double* all_costs);
/* Device pointers of the per-pair results (n_pairs entries each) for collectives issued
 * by the caller (torch.distributed / RCCL): best cost (f64) and best index (i32). */
int  mm_plan_result_dev(mm_plan* p, void** best_cost_dev, void** best_idx_dev);
/* Average device time (ms) of `iters` back-to-back runs, measured with hipEvents on the
 * engine's stream; screen_only selects mm_plan_run_screen_only. */
int  mm_plan_time(mm_plan* p, int iters, int screen_only, float* ms_avg);
/* Work accounting of one run: candidates screened, pair-distance evaluations the
 * reference algorithm performs for them (2*Na*Nb each), bytes staged in HBM. */
int  mm_plan_stats(mm_plan* p, int64_t* n_candidates, double* pair_evals, int64_t* hbm_bytes);

/* ---- host orchestration mirroring the reference's L2 drivers -------------------------- */
/* Flat (CSR) mirror of Geometry/Frame (types/native/geometry.rs:9-12, frame.rs:8-15);
 * all arrays caller-owned and updated in place.  Points are AoS xyz triples. */
typedef struct {
    int32_t   n_frames;
    uint32_t* id;          /* [F]   Frame.id                           */
    uint32_t* lumen_id;    /* [F]   Frame.lumen.id                     */
    uint32_t* orig_frame;  /* [F]   Frame.lumen.original_frame         */
    double*   centroid;    /* [F*3] Frame.centroid                     */
    int64_t*  lumen_off;   /* [F+1]                                    */
    double*   lumen;       /* xyz triples                              */
    int32_t   has_catheter;/* frames[0].extras contains Catheter       */
    int64_t*  cath_off;    /* [F+1] or NULL                            */
    double*   cath;
    int64_t*  extra_off;   /* [F+1] or NULL: all other extras contours */
    double*   extra;
    uint8_t*  has_ref;     /* [F]   Frame.reference_point.is_some()    */
    double*   ref;         /* [F*3]                                    */
    double*   lumen_centroid; /* [F*3] or NULL: Frame.lumen.centroid (Contour.centroid).  Frame::translate recomputes
                            * it from the points (frame.rs:19-20), Frame::rotate leaves it alone (frame.rs:40-63) --
                            * so after the chain it is the mean of a frame's lumen BEFORE the step's last rotation;
                            * mm_frame_translate / the chain / mm_align_between maintain it when it is given.      */
} mm_geometry;

/* AlignLog (align_within.rs:14-22), returned to Python as 7-tuples (functions.rs:26-40) */
typedef struct {
    uint32_t contour_id;
    uint32_t matched_to;
    double   rot_deg;
    double   tx, ty;
    double   cx, cy;
} mm_alignlog;

/* align_frames_in_geometry, lines 24-134 (align_within.rs): the sequential chain for
 * n_geoms pullbacks advanced in lockstep (the reference runs them in 4 crossbeam threads,
 * entry.rs:140-203); every step is one batched device search.  logs[g] must hold
 * geoms[g]->n_frames - 1 entries.  mode: 0 = faithful chain (every step's candidates are
 * scored on the chain state), 1 = decoupled screening of all frame pairs in one launch
 * followed by the exact chain pass (same results). */
int mm_align_within(mm_engine* e, int n_geoms, mm_geometry** geoms,
                    double step_deg, double range_deg, int bruteforce, int64_t sample_size,
                    int precision, int mode, mm_alignlog** logs, int64_t* pose_evals);

/* The decoupled mode split in two, so that the point sets are resident in HBM before the
 * search is timed: create() validates, builds the centred search sets of the original frames
 * and stages them; run() scores every (frame pair, candidate) in one launch sequence, then
 * walks the chain exactly (mutating the geometries passed to create()) and writes the logs.
 * A plan runs once (the walk consumes the original frames).  n_unresolved (nullable) = chain
 * steps whose winner had to be re-searched on the chain state. */
typedef struct mm_within_plan mm_within_plan;
int  mm_within_plan_create(mm_engine* e, int n_geoms, mm_geometry** geoms,
                           double step_deg, double range_deg, int bruteforce, int64_t sample_size,
                           int precision, mm_within_plan** out);
int  mm_within_plan_run(mm_within_plan* p, mm_alignlog** logs, int64_t* pose_evals,
                        int64_t* n_unresolved);
void mm_within_plan_destroy(mm_within_plan* p);
/* Diagnostics: search set `set` (sets are numbered geometry by geometry, frame by frame) as staged in HBM -- the
 * centred f64 coordinates, their f32 copy and rho (largest distance from the centre, as the error bounds use it).
 * The sets are built on the device from the raw contours (k_build_sets); MM_HOST_SETS=1 in the environment selects
 * the host-side construction instead (the checker).  Returns the set's size (at most cap points are copied), -1 on
 * error. */
int64_t mm_within_plan_fetch_set(mm_within_plan* p, int32_t set, double* x64, double* y64, float* x32, float* y32,
                                 int64_t cap, double* rho);

/* The same search with the candidate axis sharded over `world` ranks (one process per GPU).
 * run() == for every level { level_local; merge over ranks; level_commit }; walk.  Between
 * level_local and level_commit the caller exchanges the per-job arrays between ranks
 * (torch.distributed all_gather over RCCL/xGMI) and merges them with mm_merge_shards, which
 * is host-only and also serves world == 1.  Every rank then holds the same winners and walks
 * the chain redundantly (deterministic host f64).
 *   set_shard    before the first level: this rank's share of every candidate list is
 *                [n*rank/world, n*(rank+1)/world)
 *   dims         number of jobs (frame pairs, all pullbacks), levels, and per-job tie tolerance
 *   level_local  arrays of n_jobs: exact first minimum inside the slice (cost, idx, angle),
 *                uniform = all near-ties of the slice are one angle value, active = takes part
 *   level_commit ok[j] != 0 -> winner `angle[j]`; 0 -> re-searched on the chain state in walk */
int  mm_within_plan_set_shard(mm_within_plan* p, int rank, int world);
/* create + set_shard in one (level 0 is staged once, for the rank's slice) */
int  mm_within_plan_create_sharded(mm_engine* e, int n_geoms, mm_geometry** geoms,
                                   double step_deg, double range_deg, int bruteforce, int64_t sample_size,
                                   int precision, int rank, int world, mm_within_plan** out);
int  mm_within_plan_dims(mm_within_plan* p, int32_t* n_jobs, int32_t* n_levels, double* tol);
int  mm_within_plan_level_local(mm_within_plan* p, int level, double* cost, int32_t* uniform,
                                double* angle, int32_t* idx, int32_t* active);
/* level_local == level_launch (below; asynchronous) + level_collect (waits, copies the per-job records out) */
int  mm_within_plan_level_collect(mm_within_plan* p, int level, double* cost, int32_t* uniform,
                                  double* angle, int32_t* idx, int32_t* active);
int  mm_within_plan_level_commit(mm_within_plan* p, int level, const uint8_t* ok, const double* angle);
int  mm_within_plan_walk(mm_within_plan* p, mm_alignlog** logs, int64_t* pose_evals,
                         int64_t* n_unresolved);
/* The exchange on the DEVICE (all-reduce(MIN) of the per-shard best score, then all-reduce(MIN) of
 * the index masked by score == global min).  level_launch enqueues the level on this rank's slice and leaves the
 * per-pair results in HBM; the two export calls write job-indexed records into caller-owned DEVICE buffers with
 * small kernels on the engine's stream (mm_engine_stream: issue the collectives stream-ordered after them, e.g.
 * torch.cuda.ExternalStream); commit_dev copies the two REDUCED records to the host once and commits the level.
 *   export_cost  cost_dev[n_jobs] f64: exact first-minimum cost of the slice, +inf if it holds no candidate
 *                                                                          -> all_reduce(MIN) -> gcost_dev
 *   export_keys  keys_dev[3 * n_jobs] i64, given gcost_dev:
 *                  [j]            first-minimum index if this rank attains gcost[j], else INT64_MAX
 *                  [n+j], [2n+j]  angle bits and ~angle bits of this rank's winner if its minimum lies within
 *                                 the tie tolerance of gcost[j] and all its near-ties are one angle value;
 *                                 INT64_MIN twice if they are not; INT64_MAX twice if it is not near
 *                                                                          -> ONE all_reduce(MIN) of all 3n
 *   commit_dev   winner = candidate keys[j] (the first index of minimal cost over the whole axis,
 *                process_utils.rs:72); decided iff keys[n+j] == ~keys[2n+j] (every near rank uniform, all of one
 *                angle value) -- the rule of mm_merge_shards; otherwise the step is re-searched on the chain
 *                state in walk.  Identical on every rank. */
int  mm_within_plan_level_launch(mm_within_plan* p, int level);
int  mm_within_plan_level_export_cost(mm_within_plan* p, int level, double* cost_dev);
int  mm_within_plan_level_export_keys(mm_within_plan* p, int level, const double* gcost_dev, int64_t* keys_dev);
int  mm_within_plan_level_commit_dev(mm_within_plan* p, int level, const double* gcost_dev, const int64_t* keys_dev);
/* ---- multi-GPU: the shard grid and the collective behind the C ABI ------------------------------------------
 * north_star: "the candidate-pose x frame-pair grid shards naturally across the 8 GPUs of one node with an RCCL
 * allreduce over xGMI of the per-shard best score".  A shard is a tile of that grid: `pair_blocks` x `cand_slices`
 * ranks, rank r = pair block r / cand_slices, candidate slice r % cand_slices.  Rank r scores, for the jobs
 * (frame pairs, all pullbacks) [J*pb/pair_blocks, J*(pb+1)/pair_blocks), the candidates
 * [n*cs/cand_slices, n*(cs+1)/cand_slices) of each list; every other (job, candidate) is some other rank's.  The
 * exchange is the same for every grid: a rank exports +inf / INT64_MAX for what it does not own, and the two
 * all-reduces(MIN) give every rank the first index of minimal cost over the whole axis (process_utils.rs:69-74).
 * pair_blocks = 1 is the pure candidate-axis split (mm_within_plan_create_sharded); cand_slices = 1 shards the frame
 * pairs only (each pair's exact re-score then runs on one rank instead of on every rank).
 *   mm_shard_grid   the default tile shape for `world` ranks and n_jobs frame pairs: as many pair blocks as leave
 *                   every rank at least 64 frame pairs, the rest of the factor on the candidate axis */
int  mm_shard_grid(int world, int64_t n_jobs, int* pair_blocks, int* cand_slices);
int  mm_within_plan_create_grid(mm_engine* e, int n_geoms, mm_geometry** geoms,
                                double step_deg, double range_deg, int bruteforce, int64_t sample_size,
                                int precision, int rank, int pair_blocks, int cand_slices, mm_within_plan** out);
int  mm_within_plan_set_shard_grid(mm_within_plan* p, int rank, int pair_blocks, int cand_slices);
/* Staging follows the tile: a plan CREATED on a tile with pair_blocks > 1 copies to the device, and builds search sets
 * of, only the frames its pair block reads (frames i-1 and i of its jobs: its share of the pullbacks plus one halo
 * frame per boundary) -- 1 / pair_blocks of the host-side copy, the PCIe transfer and the pool.  Its tolerance (tol of
 * mm_within_plan_dims) is then defined for its own jobs only and reported as 0 for the others; a host-side merge takes
 * the largest over the ranks (the device-side exchange already uses the owner's).  Moving such a plan to another pair
 * block with mm_within_plan_set_shard_grid stages that block's frames again.
 *   mm_within_plan_staged   raw contour points copied to the device / points in the plan's set pool */
int  mm_within_plan_staged(mm_within_plan* p, int64_t* raw_points, int64_t* set_points);
/* mm_within_plan_walk for SOME of the plan's pullbacks: take[g] != 0 (n_geoms bytes).  After a sharded search every rank
 * holds every winner, so the chain walks -- pure host work, one pullback independent of the other (entry.rs:140-203: four
 * threads) -- need not be repeated on every rank: rank r walks the pullbacks g with g mod world == r and broadcasts their
 * logs and coordinates (mm_comm_broadcast; multimoda_rs_amd.distributed.finish_sharded).  Pullbacks not taken are left
 * untouched, their logs unwritten; pose_evals / n_unresolved count the taken ones. */
int  mm_within_plan_walk_geoms(mm_within_plan* p, const uint8_t* take, mm_alignlog** logs, int64_t* pose_evals, int64_t* n_unresolved);
/* TIMING ONLY (bench.py's single-process rehearsal of a rank of a larger job): with `on` != 0 the sharded entry points accept
 * a world = 1 communicator for a plan whose (rank, world) is a tile of a larger grid.  The reduced records then hold this
 * tile alone -- the result is NOT an alignment, and mm_within_plan_walk / _run_sharded report n_unresolved = -1 to say so.
 * Off by default; there is no environment variable behind it. */
int  mm_within_plan_set_timing_rehearsal(mm_within_plan* p, int on);

/* The communicator: RCCL (librccl.so.1, loaded at run time; MM_RCCL_LIB overrides the path; a process that already
 * holds a copy -- torch ships one -- shares it).  One process per GPU:
 *   rank 0:      mm_comm_unique_id(id)                      -> MM_COMM_ID_BYTES bytes, handed to every rank by the host
 *   every rank:  mm_comm_init_rank(id, rank, world, device, &comm)       (collective, like ncclCommInitRank)
 * A communicator serves one thread at a time; collectives are enqueued on the stream given (the plan's engine
 * stream inside mm_within_plan_search_sharded), in the same order on every rank. */
typedef struct mm_comm mm_comm;
#define MM_COMM_ID_BYTES 128
int  mm_comm_unique_id(void* id);
int  mm_comm_init_rank(const void* id, int rank, int world, int device, mm_comm** out);
void mm_comm_destroy(mm_comm* c);
int  mm_comm_rank(const mm_comm* c);
int  mm_comm_world(const mm_comm* c);
int  mm_comm_version(void);   /* RCCL's version code, -1 if RCCL cannot be loaded */
/* all-reduce(MIN) in place on device memory, enqueued on `stream` (a hipStream_t) */
int  mm_comm_all_reduce_min_f64(mm_comm* c, double* dev, int64_t n, void* stream);
int  mm_comm_all_reduce_min_i64(mm_comm* c, int64_t* dev, int64_t n, void* stream);
/* broadcast of `bytes` bytes of device memory from rank `root`, in place, enqueued on `stream`: how a host that walks pullback g
 * on rank g mod world only (mm_within_plan_walk_geoms) hands its logs and coordinates to the other ranks */
int  mm_comm_broadcast(mm_comm* c, void* dev, int64_t bytes, int root, void* stream);

/* The sharded search with the exchange inside the library: for every level
 *   level_launch -> export_cost -> ncclAllReduce(MIN, f64 x jobs) -> export_keys -> ncclAllReduce(MIN, i64 x 3 jobs)
 *   -> commit_dev
 * on the engine's stream, one host synchronisation per level.  The plan must have been created for
 * (mm_comm_rank, mm_comm_world) = (rank, pair_blocks * cand_slices).  A level already enqueued by
 * mm_within_plan_level_launch (a driver that queues the next case's launch early) is not launched again.
 * run_sharded = search_sharded + walk: every rank ends with the same logs and geometry (the walk is host f64). */
int  mm_within_plan_search_sharded(mm_within_plan* p, mm_comm* c);
/* search_sharded split for a driver that aligns independent cases back to back: begin enqueues level 0 completely
 * (launch unless already launched, exports, both all-reduces, the copy of the reduced records) and returns without
 * waiting; the driver then queues the next case's launch behind it (mm_engine_wait_exchange + mm_within_plan_level_launch
 * on another engine) and calls mm_within_plan_search_sharded, which collects level 0 and runs any further levels.
 * The collectives are issued in the same order on every rank: begin(case k), then begin(case k+1), ... */
int  mm_within_plan_search_sharded_begin(mm_within_plan* p, mm_comm* c);
int  mm_within_plan_run_sharded(mm_within_plan* p, mm_comm* c, mm_alignlog** logs, int64_t* pose_evals,
                                int64_t* n_unresolved);

/* arrays cost/uniform/angle/idx are [world][n] rank-major; tol [n] (nullable); outputs [n] */
int  mm_merge_shards(int world, int n, const double* cost, const int32_t* uniform,
                     const double* angle, const int32_t* idx, const double* tol,
                     uint8_t* ok, double* out_angle, int32_t* out_idx, double* out_cost);

/* align_between_geometries (align_between.rs:11-68) for n_pairs independent (a,b) pairs
 * (entry.rs:206-277 runs two at a time); b is moved onto a.  best_rotation[p] receives
 * the searched angle (radians). */
int mm_align_between(mm_engine* e, int n_pairs, mm_geometry** a, mm_geometry** b,
                     double rot_deg, double step_rot_deg, int64_t sample_size,
                     int precision, double* best_rotation, int64_t* pose_evals);

/* Sets exactly as the chain builds them for frame i (align_within.rs:45-59,173-191):
 * downsample(lumen, S) ++ downsample(catheter, ceil(n_cath*S/len_lumen0)).  Writes up to
 * cap points to out_x/out_y, returns the count. */
int64_t mm_catheter_lumen_vec(const mm_geometry* g, int32_t frame, int64_t sample_size,
                              double* out_x, double* out_y, int64_t cap);

/* extract_geometry_points_with_frame_info (align_between.rs:154-178) */
int64_t mm_extract_between_points(const mm_geometry* g, int64_t sample_size,
                                  double* out_x, double* out_y, int64_t cap);

/* Frame::translate / Frame::rotate (frame.rs:17-64) on frame i of a flat geometry. */
void mm_frame_translate(mm_geometry* g, int32_t i, double dx, double dy, double dz);
void mm_frame_rotate(mm_geometry* g, int32_t i, double angle, double cx, double cy);

/* read_contour_data (src/intravascular/io/input.rs:172-194) for the regular case: headerless rows of
 * exactly four plain decimal numbers `frame<delim>x<delim>y<delim>z`, LF or CRLF line ends, the first a u32
 * written as the reference's u32::from_str takes it (optional '+', decimal digits only, below 2^32).  Every number is converted with correct rounding (what Rust's
 * str::parse::<f64> and the csv crate deliver).  Returns the number of rows written to out (4 doubles
 * each, at most cap rows are stored), or -1 if the text is not of that regular form (quotes, blank or
 * ragged lines, other characters, non-finite values): the caller then reads it row by row, skipping
 * invalid rows like the reference.  Host only, no device. */
int64_t mm_parse_contour_table(const char* text, int64_t len, char delim, double* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
