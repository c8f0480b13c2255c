/*
 * mm_build.h -- C ABI of the geometry builder and of the bookkeeping around the searches:
 * host f64 in the reference's operation order, no device work.  Same conventions as mm_hausdorff.h.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   src/intravascular/io/build.rs:9-205            build_geometry_from_inputdata (from an InputData)
 *   src/intravascular/io/integrity_check.rs:8-256  check_geometry_integrity, the builder's last step
 *   src/types/native/contour.rs:158-224,368-405    build_contour_with_mapping, compute_centroid, sort_contour_points
 *   src/types/native/frame.rs:69-82,163-204        set_value(id), create_catheter_points
 *   src/types/native/geometry.rs:42-59,72-155,325-381  find_proximal_end_idx, reorder_frames,
 *                                                  ensure_proximal_at_position_zero
 * The Python layer's io.build_geometry_from_inputdata calls this; tests/test_refbuild.py compares the result with
 * an independent pure-Python restatement of the same reference code on every fixture directory, bit for bit.
 */
#ifndef MM_BUILD_H
#define MM_BUILD_H

#include "mm_hausdorff.h"

#ifdef __cplusplus
extern "C" {
#endif

/* types/native/record.rs: one row of the records file.  phase: 0 = "D", 1 = "S", 2 = anything else. */
typedef struct {
    uint32_t frame;
    uint8_t  phase;
    uint8_t  has_m1, has_m2;     /* measurement_1 / measurement_2 are Option<f64> */
    uint8_t  pad_;
    double   m1, m2;
} mm_record;

typedef struct mm_built mm_built;   /* a built Geometry, owned by the library */

/* build_geometry_from_inputdata(Some(input_data), ..) (build.rs:9-205).  Point arrays are rows of four doubles
 * [frame_index, x, y, z] (the numpy_to_inputdata contract, multimodars/_converters.py:204-437); eem / calcification /
 * sidebranch may be NULL (None).  lumen_aortic (nullable): ContourPoint.aortic per lumen row.  records == NULL
 * (n_records ignored) <=> InputData.record is None.  n_points == 0 -> no catheter contours (build.rs:152). */
int  mm_build_geometry(const double* lumen4, int64_t n_lumen, const uint8_t* lumen_aortic,
                       const double* eem4, int64_t n_eem, const double* calc4, int64_t n_calc,
                       const double* side4, int64_t n_side, const double ref4[4],
                       const mm_record* records, int64_t n_records, int diastole,
                       double image_center_x, double image_center_y, double radius, uint32_t n_points,
                       mm_built** out);
/* The same builder without its last step, check_geometry_integrity (build.rs:199, integrity_check.rs:8-33): frames
 * whose contours differ in point count and a reference point no frame carries are let through.  NOT the reference's
 * behaviour (mm_build_geometry fails with MM_ERR_INTEGRITY and the reference's message) -- for hosts that validate
 * their contours themselves, and for this repository's builder tests on ragged input. */
int  mm_build_geometry_lenient(const double* lumen4, int64_t n_lumen, const uint8_t* lumen_aortic,
                       const double* eem4, int64_t n_eem, const double* calc4, int64_t n_calc,
                       const double* side4, int64_t n_side, const double ref4[4],
                       const mm_record* records, int64_t n_records, int diastole,
                       double image_center_x, double image_center_y, double radius, uint32_t n_points,
                       mm_built** out);
/* sizes for the caller's arrays: frames, lumen points, catheter points, points of the other extras
 * (eem, calcification, sidebranch -- in that order inside a frame) */
int  mm_built_dims(const mm_built* b, int32_t* n_frames, int64_t* n_lumen, int64_t* n_cath, int64_t* n_extra);
/* Copy the geometry into caller-allocated arrays: dst as in mm_geometry (id, lumen_id, orig_frame, centroid,
 * lumen_off, lumen, cath_off + cath if any catheter contour exists, extra_off + extra if n_extra > 0, has_ref, ref).
 * extra_counts [F*3]: points of eem / calcification / sidebranch per frame.  aortic / pulmonary thickness [F] with
 * their has_ flags (Contour.aortic_thickness / pulmonary_thickness of the lumen).  lumen_aortic_out (nullable)
 * [n_lumen]: the per-point flags in final order.  dst->has_catheter is set. */
int  mm_built_export(const mm_built* b, mm_geometry* dst, int64_t* extra_counts, double* aortic_thickness,
                     uint8_t* has_aortic, double* pulmonary_thickness, uint8_t* has_pulmonary,
                     uint8_t* lumen_aortic_out);
void mm_built_destroy(mm_built* b);

/* Contour::compute_centroid (contour.rs:213-224) of n CSR contours of xyz triples: sequential sums / count, the
 * reference's fold; out [n*3] (a contour without points yields zeros).  Contours are independent: worker pool. */
int  mm_contour_centroids(const double* xyz, const int64_t* off, int64_t n_contours, double* out);

/* ---- the bookkeeping around the searches on a frame list (csrc/mm_frames.cpp) -----------------------------------
 * Reference interfaces replaced:
 *   src/intravascular/processing/align_within.rs:136-160  the post-steps of align_frames_in_geometry
 *        (fill_holes :330-653, is_anomalous_coronary :249-254, angle_ref_point_to_right :256-314, rotate_geometry
 *         geometry.rs:241-250, assign_aortic :316-328, wall::create_wall_frames wall.rs:7-213, smooth_frames
 *         geometry.rs:165-239)
 *   src/intravascular/processing/postprocessing.rs:12-476  postprocess_geom_pair (same-rate check, resampling,
 *        z-translation, trim_geom_pair, adjust_walls_anomalous_geom_pair)
 * A frame list changes its shape (holes get frames, walls get contours, pairs get trimmed), so it is a
 * library-owned object filled from and exported to flat arrays. */
typedef struct {
    mm_geometry g;                  /* extras per frame in the order eem, calcification, sidebranch, wall            */
    int64_t*  extra_counts;         /* [F*4] points of those four kinds per frame                                    */
    uint8_t*  has_lumen_centroid;   /* [F] Frame.lumen.centroid.is_some(); NULL on input = all Some if the next is set */
    double*   lumen_centroid;       /* [F*3]; NULL on input = all None                                               */
    double*   aortic_thickness;     /* [F] Contour.aortic_thickness of the lumen ...                                 */
    uint8_t*  has_aortic;           /* [F] ... and whether it is Some                                                */
    double*   pulmonary_thickness;
    uint8_t*  has_pulmonary;
    uint8_t*  lumen_aortic;         /* [lumen points] ContourPoint.aortic; NULL on input = all false                 */
    uint8_t*  wall_aortic;          /* [wall points]; NULL on input = all false                                      */
} mm_flat_geometry;

typedef struct mm_frames mm_frames;
int  mm_frames_from_flat(const mm_flat_geometry* in, mm_frames** out);
int  mm_frames_dims(const mm_frames* f, int32_t* n_frames, int64_t* n_lumen, int64_t* n_cath, int64_t* n_extra,
                    int64_t* n_wall);
/* every array of `out` caller-allocated to the sizes of mm_frames_dims (cath / extra arrays only where > 0) */
int  mm_frames_export(const mm_frames* f, mm_flat_geometry* out);
void mm_frames_destroy(mm_frames* f);
/* The post-steps of align_frames_in_geometry (align_within.rs:136-160) on the chain's result.  ref_idx = the
 * reference frame index taken BEFORE the chain (:42-44).  Lumen contour centroids are carried like the reference
 * carries them: untouched by the rotation (Frame::rotate, frame.rs:40-63), averaged / interpolated into frames that
 * fill holes, recomputed from the points only by smoothing (geometry.rs:204) -- i.e. with smooth == 0 they are what
 * the chain's last Frame::translate left (mm_geometry.lumen_centroid, maintained by mm_align_within). */
int  mm_frames_finish_within(mm_frames* f, int64_t ref_idx, int smooth, int* anomalous);
/* postprocess_geom_pair(pair, tolerance, anomalous) (postprocessing.rs:12-87), both lists replaced */
int  mm_frames_postprocess_pair(mm_frames* a, mm_frames* b, double tolerance, int anomalous);

#ifdef __cplusplus
}
#endif
#endif
