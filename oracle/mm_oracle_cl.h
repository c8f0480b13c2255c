/*
 * mm_oracle_cl.h -- CPU oracle for the centerline placement + three-point search + Hausdorff
 * refinement (TEST INFRASTRUCTURE ONLY; same rules as mm_oracle.h).
 *
 * Plain-C f64 restatement of (paths relative to the reference checkout):
 *   src/types/native/centerline.rs:14-62                        from_contour_points, find_reference_cl_point_idx
 *   src/intravascular/centerline_align/preprocessing.rs:16-280  preprocess_centerline
 *   src/types/native/contour.rs:368-405                         sort_contour_points
 *   src/types/native/geometry.rs:241-250                        rotate_geometry
 *   src/intravascular/centerline_align/align_algorithms.rs:66-535
 *   src/intravascular/centerline_align/align.rs:63-285, 381-595
 *
 * Third-party arithmetic on this path: nalgebra 0.35.0 (Cargo.lock) -- Vector3::{norm, angle,
 * cross, normalize}, Unit::new_normalize, Rotation3::from_axis_angle, Rotation3 * Vector3.
 * nalgebra is not vendored in the reference checkout; the formulas below restate its published
 * source (base/norm.rs, base/matrix.rs `angle`, geometry/rotation_specialization.rs
 * `from_axis_angle`, base/blas.rs `gemv`: y_i = (r_i0 v0 + r_i1 v1) + r_i2 v2).
 *
 * Parity status: the Hausdorff selection (strict `<`, candidate order) is pinned by the same KATs
 * as mm_oracle.h; the 3-D placement is LOOSELY PINNED -- the reference's own tests
 * (align_algorithms.rs:573-934, preprocessing.rs:288-604, centerline.rs:989-1021) hold identity /
 * 90 degree / straight-line cases with 1e-6..1e-12 tolerances, all reproduced in
 * tests/test_oracle_cl_kat.py; operation order inside nalgebra is restated from memory of its
 * source, not checked against a build ("placement parity loosely pinned").
 */
#ifndef MM_ORACLE_CL_H
#define MM_ORACLE_CL_H

#include "mm_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* centerline_point.rs:4-11 */
typedef struct {
    double   x, y, z;        /* contour_point */
    double   tx, ty, tz;     /* tangent       */
    double   radius;
    uint32_t branch_id;
    uint32_t pad_;
} orc_clpoint;

/* align_algorithms.rs:65-71 FrameTransformation (rotation row-major) */
typedef struct {
    double t[3];
    double r[9];
    double pivot[3];
} orc_frame_tf;

/* A Geometry as the placement code sees it: the flat geometry plus the lumen contour's own
 * (possibly stale) centroid and the boundaries of the individual extras contours. */
typedef struct {
    orc_geometry* g;
    uint8_t*  has_lumen_centroid;  /* [F] Frame.lumen.centroid.is_some(); NULL = all None      */
    double*   lumen_centroid;      /* [F*3]                                                   */
    int32_t   n_extra_kinds;       /* K: extras contours per frame inside g->extra (not cath) */
    int64_t*  extra_kind_off;      /* [F*K+1] CSR over (frame,kind) into g->extra, or NULL    */
    /* ContourPoint.aortic of the lumen and the Wall contour: they travel with their points when a contour is
     * sorted (contour.rs:385-390 moves whole ContourPoints); align_walls reads the wall's (align.rs:385-407) */
    uint8_t*  lumen_aortic;        /* [lumen points] or NULL = all false                      */
    uint8_t*  wall_aortic;         /* [wall points, frame by frame] or NULL = all false       */
    int32_t   wall_kind1;          /* 1 + index of the Wall contour among the K kinds; 0 = no Wall contours */
} orc_clgeom;

/* centerline.rs:14-42.  Returns 0, or -1 when the reference would panic (n == 1). */
int    orc_centerline_from_points(const orc_point* pts, size_t n, orc_clpoint* out);
/* centerline.rs:51-62 */
size_t orc_cl_find_ref_idx(const orc_clpoint* cl, size_t n, const double ref[3]);
/* preprocessing.rs:16-108.  Returns the number of resampled points (written up to cap), or
 * -1 "Centerline has no branch-0 points", -3 "Reference mesh has no frames". */
int64_t orc_preprocess_centerline(const orc_clpoint* cl, size_t n, const orc_geometry* ref_mesh,
                                  orc_clpoint* out, size_t cap, double* spacing);

void   orc_sort_contour_points(orc_point* pts, size_t n);               /* contour.rs:368-405   */
void   orc_sort_contour_points_flags(orc_point* pts, uint8_t* flags, size_t n);  /* flags (nullable) follow their points */
/* align_walls (align.rs:381-595): with anomalous != 0 and >= 2 frames in geoms[0], the Wall contour of every frame
 * of every geometry is rotated about the lumen normal onto the parallel-transported direction of frame 0's wall.
 * PARITY UNPINNED: the reference holds no test of it. */
void   orc_align_walls(orc_clgeom** geoms, int n_geoms, int anomalous);
void   orc_rotate_geometry(orc_clgeom* g, double angle);                /* geometry.rs:241-250  */
void   orc_newell_normal(const orc_point* pts, size_t n, const double c[3], double out[3]); /* :206-235 */
void   orc_align_frame(const orc_point* pts, size_t n, int has_centroid, const double centroid[3],
                       const orc_clpoint* clp, orc_frame_tf* tf);       /* :128-173 */
orc_point orc_tf_apply(const orc_frame_tf* tf, orc_point p);            /* :74-93   */
/* apply_transformations (:511-535 with get_transformations :96-126): geoms[0] is the primary
 * geometry.  Returns the number of frames that received a transformation. */
size_t orc_apply_transformations(orc_clgeom** geoms, int n_geoms, const orc_clpoint* cl, size_t ncl,
                                 const double ref_pt[3]);
/* Rotation3::from_axis_angle(&Unit::new_normalize(axis), angle), row-major */
void   orc_rotation_from_axis_angle(const double axis[3], double angle, double r[9]);
void   orc_rotate_contour_around_centroid(orc_point* pts, size_t n, int has_centroid,
                                          const double centroid[3], double angle); /* :238-259 */
/* best_rotation_three_point (:263-336) */
double orc_best_rotation_three_point(const orc_point* pts, size_t n, int has_centroid,
                                     const double centroid[3], uint32_t index_reference,
                                     const double p_main[3], const double p_ccw[3],
                                     const double p_cw[3], double angle_step,
                                     const orc_clpoint* clp);
/* refine_alignment_hausdorff (:339-451).  all_costs (nullable, cap entries) receives the
 * Hausdorff value of every evaluated candidate in evaluation order; *n_evals their number. */
int orc_refine_alignment_hausdorff(orc_clgeom** geoms, int n_geoms, const orc_clpoint* cl, size_t ncl,
                                   size_t initial_cl_ref_idx, double initial_rotation,
                                   const orc_point* points, size_t n_points,
                                   double angle_search_range, double angle_step,
                                   size_t index_search_range,
                                   double* best_angle, size_t* best_idx, double* min_hausdorff,
                                   double* all_costs, size_t cap, size_t* n_evals);

/* align.rs:63-124 / 126-166 / 169-285 (write = false).  Geometries are transformed in place.
 * Return 0, -1..-3 from preprocessing, -4 no reference frame, -5 missing reference point. */
int orc_align_three_point(const orc_clpoint* cl, size_t ncl, orc_clgeom** geoms, int n_geoms,
                          uint32_t ref_point_index, const double p_main[3], const double p_ccw[3],
                          const double p_cw[3], double angle_step, int align_wall_anomalous,
                          double* spacing, double* total_rotation);
int orc_align_manual(const orc_clpoint* cl, size_t ncl, orc_clgeom** geoms, int n_geoms,
                     double rotation_angle_deg, const double ref_pt[3], int align_wall_anomalous,
                     double* spacing, double* total_rotation);
int orc_align_combined(const orc_clpoint* cl, size_t ncl, orc_clgeom** geoms, int n_geoms,
                       uint32_t ref_point_index, const double p_main[3], const double p_ccw[3],
                       const double p_cw[3], const orc_point* points, size_t n_points,
                       double angle_step, double refine_angle_range, size_t refine_index_range,
                       int align_wall_anomalous, double* spacing, double* total_rotation,
                       size_t* refined_idx);

#ifdef __cplusplus
}
#endif
#endif
