/*
 * mm_oracle.h -- CPU oracle for the Hausdorff pose search (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C, f64, line-by-line restatement of the reference algorithm
 * (yungselm/multimoda-rs, paths relative to the reference checkout):
 *   src/intravascular/processing/process_utils.rs:33-121   search_range, hausdorff_distance
 *   src/types/native/contour_point.rs:28-53                translate / rotate
 *   src/types/native/contour.rs:47-58                      downsample_contour_points
 *   src/types/native/frame.rs:17-64                        Frame::translate / Frame::rotate
 *   src/intravascular/processing/align_within.rs:24-134,173-247
 *   src/intravascular/processing/align_between.rs:11-68,95-271
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported baseline -- never as the product path.
 *
 * Parity status: PINNED by the reference's own known-answer tests
 * (process_utils.rs:130-547, align_within.rs:791-887, align_between.rs:280-373);
 * see tests/test_oracle_kat.py.  The reference (Rust) cannot be compiled in this
 * image (no cargo/rustc), so there is no oracle/_ref build.
 *
 * Must be compiled with -ffp-contract=off (Rust never fuses a*b+c).
 */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* contour_point.rs:55-68 -- only the coordinates matter on this path. */
typedef struct { double x, y, z; } orc_point;

/* align_within.rs:14-22 AlignLog */
typedef struct {
    uint32_t contour_id;
    uint32_t matched_to;
    double   rot_deg;
    double   tx, ty;
    double   cx, cy;
} orc_alignlog;

/* Flat (CSR) mirror of Geometry/Frame (geometry.rs:9-12, frame.rs:8-15).  All arrays
 * are caller-owned and mutated in place. "extra" carries every non-catheter extras
 * contour (eem, calcification, ...) so that frame transforms touch them too. */
typedef struct {
    int32_t    n_frames;
    uint32_t*  id;          /* [F]   Frame.id                                   */
    uint32_t*  lumen_id;    /* [F]   Frame.lumen.id                             */
    uint32_t*  orig_frame;  /* [F]   Frame.lumen.original_frame                 */
    double*    centroid;    /* [F*3] Frame.centroid                             */
    int64_t*   lumen_off;   /* [F+1]                                            */
    orc_point* lumen;
    int32_t    has_catheter;/* frames[0].extras contains Catheter               */
    int64_t*   cath_off;    /* [F+1] (may be NULL if !has_catheter)             */
    orc_point* cath;
    int64_t*   extra_off;   /* [F+1] or NULL                                    */
    orc_point* extra;
    uint8_t*   has_ref;     /* [F]   Frame.reference_point.is_some()            */
    orc_point* ref;         /* [F]                                              */
    double*    lumen_centroid; /* [F*3] or NULL: Frame.lumen.centroid -- recomputed by Frame::translate
                             * (frame.rs:19-20), untouched by Frame::rotate (frame.rs:40-63)        */
} orc_geometry;

/* ---- process_utils.rs ------------------------------------------------------------ */
double orc_directed_hausdorff(const orc_point* a, size_t na, const orc_point* b, size_t nb);
double orc_hausdorff(const orc_point* s1, size_t n1, const orc_point* s2, size_t n2);
/* SoA convenience wrapper (x,y only; z is ignored by the metric anyway). */
double orc_hausdorff_xy(const double* ax, const double* ay, size_t na,
                        const double* bx, const double* by, size_t nb);

typedef double (*orc_cost_fn)(double angle, void* ctx);

/* Candidate enumeration of search_range (process_utils.rs:43-67).
 * Returns the number of angles written (<= cap).  *degenerate is set to 1 when the
 * reference returns early (step <= 0 or stop <= start); *early_value then holds the
 * value it returns.  Angles are the *wrapped* values the cost function receives. */
size_t orc_search_angles(double step_deg, double range_deg, int has_center, double center,
                         double limes_deg, double* out, size_t cap,
                         int* degenerate, double* early_value);

/* search_range (process_utils.rs:33-75).  n_threads > 1 evaluates candidates in
 * parallel (OpenMP), like the reference's rayon par_iter; the result does not depend
 * on n_threads (ordered first-minimum reduction). */
double orc_search_range(orc_cost_fn f, void* ctx, double step_deg, double range_deg,
                        int has_center, double center, double limes_deg, int n_threads);

/* ---- point / set helpers ---------------------------------------------------------- */
orc_point orc_rotate_point(orc_point p, double angle, double cx, double cy); /* contour_point.rs:38-52 */
size_t orc_downsample(const orc_point* pts, size_t len, size_t n, orc_point* out); /* contour.rs:47-58 */

/* Cost of one candidate as the within-pullback closure computes it
 * (align_within.rs:100-104, 200-206): rotate target by `angle` about (cx,cy) with the
 * angle==0 shortcut, then hausdorff(reference, rotated). */
double orc_cost_within(const orc_point* ref, size_t nr, const orc_point* tgt, size_t nt,
                       double angle, double cx, double cy);
/* Between-pullback closure (align_between.rs:189-216): no shortcut. */
double orc_cost_between(const orc_point* ref, size_t nr, const orc_point* tgt, size_t nt,
                        double angle, double cx, double cy);

/* find_best_rotation (align_within.rs:193-247) / find_best_rotation_between
 * (align_between.rs:219-257): hierarchical coarse->fine search. between != 0 selects
 * the between-closure. */
double orc_find_best_rotation(const orc_point* ref, size_t nr, const orc_point* tgt, size_t nt,
                              double step_deg, double range_deg, double cx, double cy,
                              int between, int n_threads);
/* Brute-force branch of align_within.rs:97-110. */
double orc_bruteforce_rotation(const orc_point* ref, size_t nr, const orc_point* tgt, size_t nt,
                               double step_deg, double range_deg, double cx, double cy,
                               int n_threads);

/* Count of cost evaluations the reference performs for one search (for metric
 * accounting): hierarchical if !bruteforce. */
size_t orc_count_evals(double step_deg, double range_deg, int bruteforce);

/* ---- frame transforms (frame.rs:17-64) -------------------------------------------- */
void orc_frame_translate(orc_geometry* g, int32_t i, double dx, double dy, double dz);
void orc_frame_rotate(orc_geometry* g, int32_t i, double angle, double cx, double cy);

/* catheter_lumen_vec_from_frames (align_within.rs:173-191). out must hold
 * sample_size_lumen + sample_size_catheter points (or the full contours if shorter). */
size_t orc_catheter_lumen_vec(const orc_geometry* g, int32_t i, size_t sample_size_lumen,
                              int has_sc, size_t sample_size_catheter, orc_point* out);

/* align_frames_in_geometry, lines 24-134 only (the chain; post-steps are not part of
 * this path).  logs must hold n_frames-1 entries.  Returns 0, or
 *  -1 "Geometry contains no frames", -2 "Lumen contours have no points",
 *  -3 "sample_size must be > 0". */
int orc_align_within_chain(orc_geometry* g, double step_deg, double range_deg,
                           int bruteforce, size_t sample_size, orc_alignlog* logs,
                           int n_threads);

/* geometry.rs:42-69 */
size_t orc_find_proximal_end_idx(const orc_geometry* g);
int    orc_find_ref_frame_idx(const orc_geometry* g, size_t* out);

/* extract_geometry_points_with_frame_info (align_between.rs:154-178): returns count. */
size_t orc_extract_between_points(const orc_geometry* g, size_t sample_size,
                                  orc_point* out, size_t cap);

/* align_between_geometries (align_between.rs:11-68) without the prints / pair clone.
 * best_rotation_out receives the searched angle (radians). */
int orc_align_between(orc_geometry* a, orc_geometry* b, double rot_deg, double step_rot_deg,
                      size_t sample_size, double* best_rotation_out, int n_threads);

/* ---- refine_alignment_hausdorff (centerline_align/align_algorithms.rs:339-451) -------- */
/* accumulated angle enumeration, :386-387,439 */
size_t orc_refine_angles(double initial, double range, double step, double* out, size_t cap);
/* filter_points_in_region, :454-505; returns count, writes indices */
size_t orc_filter_points_in_region(const orc_point* pts, size_t n, const orc_point* start,
                                   const orc_point* end, int64_t* out_idx, size_t cap);
/* n_downsample, :415-418 */
size_t orc_refine_downsample_count(size_t n_filtered, size_t n_points_per_frame, size_t n_frames);

#ifdef __cplusplus
}
#endif
#endif
