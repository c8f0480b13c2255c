/*
 * mm_build.h -- C ABI of the geometry builder and of the bookkeeping around the searches:
 * host f64 in the reference's operation order, no device work.  Same conventions as mm_hausdorff.h.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   src/intravascular/io/build.rs:9-205            build_geometry_from_inputdata (from an InputData)
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

#ifdef __cplusplus
}
#endif
#endif
