/*
 * mm_ccta.h -- C ABI of the CCTA diameter search: 41 radial scalings of a
 * vessel region scored by the symmetric RMS nearest-neighbour distance to a reference cloud, in
 * 3-D.  Same conventions as mm_hausdorff.h.  Points are xyz triples (f64), caller-owned.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   src/ccta/adjust_mesh/scale_coronary.rs:8-63     centerline_based_wall_diameter_optimization
 *   src/ccta/adjust_mesh/scale_coronary.rs:65-88    centerline_based_aortic_diameter_optimization
 *   src/ccta/adjust_mesh/scale_coronary.rs:90-131   centerline_based_diameter_optimization
 *   src/ccta/adjust_mesh/scale_coronary.rs:133-183  find_region_points
 *   src/ccta/adjust_mesh/scale_coronary.rs:188-216  symmetric_nn_distance
 *   src/ccta/adjust_mesh/scale_coronary.rs:218-261  centerline_based_diameter_morphing
 *   src/ccta/adjust_mesh/scale_coronary.rs:263-340  find_points_by_cl_region_rs, find_cl_points_in_range
 *   src/ccta/adjust_mesh/scale_coronary.rs:342-409  clean_up_non_section_points
 * Python entry points that bind them: src/ccta/binding/ccta_py.rs:263-481
 * (adjust_diameter_centerline_morphing_simple, find_proximal_distal_scaling, find_aortic_scaling,
 * find_aortic_wall_scaling), wrapped by multimodars/ccta/scaling.py.
 *
 * All nearest-neighbour minima are computed on the device in exact f64 (mm_nn_kernels.hip); the
 * per-point minima are summed on the host in index order (the reference's rayon sum has no fixed
 * order; the sequential one is among those it can produce).
 */
#ifndef MM_CCTA_H
#define MM_CCTA_H

#include "mm_centerline.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MM_CCTA_SCALING_STEPS 41   /* -2.0 .. 2.0 mm in steps of 0.1 (scale_coronary.rs:70-73) */

/* min_p |q - p|^2 for every query q of every (query set, point set) pair.  Sets: n_sets CSR ranges
 * of xyz triples; pair k reads sets q_set[k] / p_set[k] and writes sets[q_set[k]].n values at
 * out + out_off[k].  An empty point set yields +inf. */
int     mm_nn_min_sq_batch(mm_engine* e, int n_sets, const int64_t* set_off, const double* xyz,
                           int n_pairs, const int32_t* q_set, const int32_t* p_set,
                           const int64_t* out_off, double* out);
/* symmetric_nn_distance (:188-216); +inf if either set is empty */
int     mm_symmetric_nn_distance(mm_engine* e, const double* a_xyz, int64_t na, const double* b_xyz, int64_t nb,
                                 double* out);
/* centerline_based_diameter_morphing (:218-261), host f64 */
int     mm_diameter_morphing(const mm_clpoint* cl, int64_t ncl, const double* pts_xyz, int64_t n,
                             double diameter_adjustment_mm, double* out_xyz);
/* find_region_points (:133-183): selected (nearest n_points, sorted by distance then index) and
 * remaining (input order); buffers hold n triples each.  Returns the number selected. */
int64_t mm_find_region_points(mm_engine* e, const double* anomalous_xyz, int64_t n, const double* reference_xyz,
                              int64_t nr, int64_t n_points, double* selected_xyz, double* remaining_xyz);
/* centerline_based_aortic_diameter_optimization (:65-88).  all_dist (nullable) receives the
 * MM_CCTA_SCALING_STEPS distances.  *best = f64::MAX if every distance is +inf (empty input). */
int     mm_aortic_diameter_optimization(mm_engine* e, const double* intramural_xyz, int64_t ni,
                                        const double* reference_xyz, int64_t nr, const mm_clpoint* cl, int64_t ncl,
                                        double* best, double* all_dist);
/* centerline_based_diameter_optimization (:90-131) */
int     mm_diameter_optimization(mm_engine* e, const double* anomalous_xyz, int64_t n, int64_t n_proximal,
                                 int64_t n_distal, const mm_clpoint* cl, int64_t ncl,
                                 const double* proximal_reference_xyz, int64_t npr,
                                 const double* distal_reference_xyz, int64_t ndr,
                                 double* proximal_best, double* distal_best);
/* centerline_based_wall_diameter_optimization (:8-63), host f64 */
int     mm_wall_diameter_optimization(const mm_clpoint* cl, int64_t ncl, const double ref_pt[3],
                                      const double* aortic_xyz, int64_t na, double* out);

/* clean_up_non_section_points (:342-409; Python: clean_outlier_points, ccta_py.rs:345-358): for every point of
 * `cleanup`, the neighbours within neighborhood_radius among the reference points and among the other cleanup
 * points are counted on the device (exact f64, |q - p|^2 <= r^2 like rstar's locate_within_distance);
 * to_reference[i] = 1 where ref / (ref + self) >= min_neighbor_ratio: the point joins the reference set
 * (appended in input order), 0: it stays. */
int     mm_clean_outlier_points(mm_engine* e, const double* cleanup_xyz, int64_t nc, const double* reference_xyz,
                                int64_t nr, double neighborhood_radius, double min_neighbor_ratio,
                                uint8_t* to_reference);
/* find_points_by_cl_region_rs (:263-312; Python: find_points_by_cl_region, ccta_py.rs:304-319).  cl_frame_index
 * (nullable -> 0, 1, 2, ...) = contour_point.frame_index of every centerline point.  label[i]:
 *   0 proximal, 1 distal, 2 between (anomalous section), 3 proximal moved to between by the first clean-up,
 *   4 distal moved to between by the second.  The reference's three vectors are: proximal = labels 0 in input
 *   order; distal = labels 1 in input order; between = labels 2 in input order, then 3 in input order, then 4. */
int     mm_find_points_by_cl_region(mm_engine* e, const mm_clpoint* cl, const uint32_t* cl_frame_index, int64_t ncl,
                                    const double* frame_centroids_xyz, int64_t n_frames, const double* points_xyz,
                                    int64_t n, uint8_t* label);

#ifdef __cplusplus
}
#endif
#endif
