/*
 * mm_centerline.h -- C ABI of the centerline placement path: three-point initial rotation,
 * frame placement on the centerline and the Hausdorff refinement grid (the third call site of
 * the Hausdorff search).  Same conventions as mm_hausdorff.h:
 * plain pointers and sizes, caller-owned arrays updated in place, 0 or a negative MM_ERR_* code,
 * message in mm_last_error().
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   src/types/native/centerline.rs:14-62                         Centerline::from_contour_points,
 *                                                                find_reference_cl_point_idx
 *   src/intravascular/centerline_align/preprocessing.rs:16-108   preprocess_centerline
 *   src/types/native/contour.rs:368-405                          Contour::sort_contour_points
 *   src/types/native/geometry.rs:241-250                         Geometry::rotate_geometry
 *   src/intravascular/centerline_align/align_algorithms.rs:96-126,511-535  apply_transformations
 *   src/intravascular/centerline_align/align_algorithms.rs:263-336         best_rotation_three_point
 *   src/intravascular/centerline_align/align_algorithms.rs:339-451         refine_alignment_hausdorff
 *   src/intravascular/centerline_align/align.rs:63-285           align_three_point_rs, align_manual_rs,
 *                                                                align_combined_rs (write = false)
 * Python entry points that bind them: multimodars/_processing.py:1010-1300
 * (binding src/intravascular/binding/align.rs:83-460).
 *
 * The geometry transforms are host f64 (they are O(points)); every Hausdorff evaluation of the
 * refinement grid runs on the device through the same exact-f64 kernels as mm_hausdorff_batch.
 */
#ifndef MM_CENTERLINE_H
#define MM_CENTERLINE_H

#include "mm_hausdorff.h"

#ifdef __cplusplus
extern "C" {
#endif

/* CenterlinePoint (types/native/centerline_point.rs:4-11), 64 bytes */
typedef struct {
    double   x, y, z;        /* contour_point coordinates */
    double   tx, ty, tz;     /* tangent                   */
    double   radius;
    uint32_t branch_id;      /* 0 = main vessel           */
    uint32_t pad_;
} mm_clpoint;

/* A Geometry as the placement code reads it: mm_geometry plus the lumen contour's own centroid
 * (Contour.centroid, which Frame::rotate leaves stale -- align_frame reads it, :130-135) and the
 * boundaries of the individual extras contours (each is sorted on its own after a rotation). */
typedef struct {
    mm_geometry* g;
    uint8_t*  has_lumen_centroid;  /* [F] Frame.lumen.centroid.is_some(); NULL = all None       */
    double*   lumen_centroid;      /* [F*3]                                                    */
    int32_t   n_extra_kinds;       /* K extras contours per frame inside g->extra (catheter excluded) */
    int64_t*  extra_kind_off;      /* [F*K+1] CSR over (frame, kind) into g->extra; NULL = one contour per frame */
    /* ContourPoint.aortic of the lumen and of the Wall contour (nullable = all false).  A sort moves whole
     * ContourPoints (contour.rs:385-390), so these are permuted with their points by every function here that
     * sorts contours; mm_align_walls reads the wall's (align.rs:385-407). */
    uint8_t*  lumen_aortic;        /* [lumen points]                                                          */
    uint8_t*  wall_aortic;         /* [wall points, frame by frame]                                           */
    int32_t   wall_kind1;          /* 1 + index of the Wall contour among the K kinds; 0 = no Wall contours   */
} mm_cl_geometry;

/* Centerline::from_contour_points (centerline.rs:14-42): tangents = normalised forward
 * differences, last point repeats its predecessor's.  xyz: n triples.  n == 1 is an error (the
 * reference panics). */
int     mm_centerline_from_points(const double* xyz, int64_t n, mm_clpoint* out);
/* find_reference_cl_point_idx (centerline.rs:51-62): first point of minimal 3-D distance. */
int64_t mm_centerline_find_ref_idx(const mm_clpoint* cl, int64_t n, const double ref[3]);
/* preprocess_centerline (preprocessing.rs:16-108): branch 0 only, descending z, resampled at
 * the mean spacing of ref_mesh's frame centroids.  Returns the number of points (writes up to
 * cap of them; call with cap = 0 to size the buffer) or a negative error. */
int64_t mm_centerline_preprocess(const mm_clpoint* cl, int64_t n, const mm_geometry* ref_mesh,
                                 mm_clpoint* out, int64_t cap, double* spacing);

/* Contour::sort_contour_points (contour.rs:368-405) on n xyz triples, in place. */
int     mm_sort_contour_points(double* xyz, int64_t n);
/* Geometry::rotate_geometry (geometry.rs:241-250): every frame about its own centroid, then
 * every contour re-sorted; angle == 0.0 returns without sorting. */
int     mm_rotate_geometry(mm_cl_geometry* g, double angle);
/* apply_transformations (align_algorithms.rs:511-535 with get_transformations :96-126 and
 * align_frame :128-173): geoms[0] is the primary geometry the transformations are computed from;
 * all n_geoms (1 or 2: Geometry or GeometryPair) receive them.  Returns the number of frames
 * placed (frames past the end of the centerline stay untouched) or a negative error. */
int64_t mm_apply_transformations(mm_cl_geometry** geoms, int n_geoms, const mm_clpoint* cl, int64_t ncl,
                                 const double ref_pt[3]);
/* best_rotation_three_point (align_algorithms.rs:263-336): sweep 0..2pi in accumulated steps of
 * angle_step over the reference frame's lumen (n xyz triples, point_index == position). */
int     mm_best_rotation_three_point(const double* lumen_xyz, int64_t n, int has_centroid,
                                     const double centroid[3], uint32_t index_reference,
                                     const double p_main[3], const double p_ccw[3], const double p_cw[3],
                                     double angle_step, const mm_clpoint* clp, double* best_angle);

/* refine_alignment_hausdorff (align_algorithms.rs:339-451).  The (index shift x angle) grid is
 * rebuilt on the host (rotate + sort + place + downsample per candidate, candidates in
 * parallel), then every candidate's hausdorff_distance(filtered CCTA points, placed frames) is
 * evaluated on the device in one batch; the first minimum in (index asc, angle asc) order wins
 * (strict `<`, :433).  all_costs (nullable, cap entries) receives the evaluated candidates'
 * costs in that order, *n_evals their number. */
int     mm_refine_alignment_hausdorff(mm_engine* e, mm_cl_geometry** geoms, int n_geoms,
                                      const mm_clpoint* cl, int64_t ncl, int64_t initial_cl_ref_idx,
                                      double initial_rotation, const double* points_xyz, int64_t n_points,
                                      double angle_search_range, double angle_step, int64_t index_search_range,
                                      double* best_angle, int64_t* best_idx, double* min_hausdorff,
                                      double* all_costs, int64_t cap, int64_t* n_evals);

/* align_walls (align.rs:381-595): with anomalous != 0 and at least two frames in geoms[0], the Wall contour of every
 * frame (but the first) of every geometry is rotated about its lumen normal so that its aortic side -- or, without
 * aortic flags, its major axis -- follows the direction of frame 0's wall, parallel-transported along the vessel.
 * Lumen and all other contours stay.  The reference holds no test of this function: parity is against
 * oracle/mm_oracle_cl.c's restatement only ("parity unpinned"). */
int     mm_align_walls(mm_cl_geometry** geoms, int n_geoms, int anomalous);

/* align_three_point_rs / align_manual_rs / align_combined_rs (align.rs:63-124, 126-166, 169-285)
 * with write = false.  geoms (1 or 2) are transformed in place; angles in radians except
 * rotation_angle_deg; *total_rotation is in radians (the binding converts, align.rs:125).
 * align_wall_anomalous != 0: mm_align_walls as the last step (align.rs:105-107,147-149,266-268). */
int     mm_align_three_point(const mm_clpoint* cl, int64_t ncl, mm_cl_geometry** geoms, int n_geoms,
                             uint32_t ref_point_index, const double p_main[3], const double p_ccw[3],
                             const double p_cw[3], double angle_step, int align_wall_anomalous,
                             double* spacing, double* total_rotation);
int     mm_align_manual(const mm_clpoint* cl, int64_t ncl, mm_cl_geometry** geoms, int n_geoms,
                        double rotation_angle_deg, const double ref_pt[3], int align_wall_anomalous,
                        double* spacing, double* total_rotation);
int     mm_align_combined(mm_engine* e, const mm_clpoint* cl, int64_t ncl, mm_cl_geometry** geoms, int n_geoms,
                          uint32_t ref_point_index, const double p_main[3], const double p_ccw[3],
                          const double p_cw[3], const double* points_xyz, int64_t n_points,
                          double angle_step, double refine_angle_range, int64_t refine_index_range,
                          int align_wall_anomalous, double* spacing, double* total_rotation,
                          int64_t* refined_idx, int64_t* n_evals);

#ifdef __cplusplus
}
#endif
#endif
