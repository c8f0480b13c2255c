"""ctypes binding of ``libmm_hausdorff.so`` (the C ABI declared in ``include/mm_hausdorff.h``).

There is no CPU fallback: if the shared library is missing, or no HIP device is visible
when an :class:`Engine` is created, this module raises.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import weakref
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MM_LIB_PATH: another build of the same library (tools/asan_host.sh runs the CPU suite on an AddressSanitizer build
# of the host code); never a fallback -- a path that does not exist fails like a missing in-tree build
LIB_PATH = os.environ.get("MM_LIB_PATH") or os.path.join(_HERE, "lib", "libmm_hausdorff.so")

MM_PRECISION_F64 = 0
MM_PRECISION_F32 = 1
MM_PRECISION_F32_FAST = 2
MM_PRECISION_F32_BOUNDED = 3
MM_PRECISION_F32_MATRIX = 4
MM_SEARCH_SKIP_ZERO = 1

EXPORTS = [
    "mm_device_count", "mm_last_error", "mm_version",
    "mm_engine_create", "mm_engine_destroy", "mm_engine_synchronize", "mm_engine_stream", "mm_engine_wait_search",
    "mm_engine_profile", "mm_engine_profile_read", "mm_engine_profile_launches", "mm_engine_bound_stats", "mm_engine_screen_stats", "mm_engine_set_bound_matrix", "mm_lower_bounds", "mm_pick_minima",
    "mm_engine_set_bound_min_candidates",
    "mm_hausdorff_2d", "mm_hausdorff_batch", "mm_refine_angles", "mm_filter_points_in_region",
    "mm_refine_downsample_count", "mm_search_angles", "mm_best_rotation", "mm_best_rotation_batch",
    "mm_plan_create", "mm_plan_create_indexed", "mm_plan_destroy", "mm_plan_run", "mm_plan_run_screen_only", "mm_plan_fetch",
    "mm_plan_result_dev", "mm_plan_time", "mm_plan_stats",
    "mm_align_within", "mm_align_between", "mm_within_plan_create", "mm_within_plan_create_sharded", "mm_within_plan_run", "mm_within_plan_destroy", "mm_within_plan_fetch_set",
    "mm_within_plan_set_shard", "mm_within_plan_dims", "mm_within_plan_level_local", "mm_within_plan_level_commit",
    "mm_within_plan_walk", "mm_within_plan_level_collect", "mm_within_plan_level_launch", "mm_within_plan_level_export_cost",
    "mm_within_plan_level_export_keys", "mm_within_plan_level_commit_dev", "mm_merge_shards", "mm_catheter_lumen_vec", "mm_extract_between_points",
    "mm_frame_translate", "mm_frame_rotate", "mm_parse_contour_table",
    "mm_shard_grid", "mm_within_plan_create_grid", "mm_within_plan_set_shard_grid", "mm_comm_unique_id", "mm_comm_init_rank",
    "mm_comm_destroy", "mm_comm_rank", "mm_comm_world", "mm_comm_version", "mm_comm_all_reduce_min_f64",
    "mm_comm_all_reduce_min_i64", "mm_within_plan_search_sharded", "mm_within_plan_run_sharded",
    "mm_within_plan_search_sharded_begin", "mm_engine_wait_exchange", "mm_within_plan_set_timing_rehearsal", "mm_within_plan_walk_geoms", "mm_comm_broadcast", "mm_within_plan_staged",
]
# include/mm_centerline.h
EXPORTS_CENTERLINE = [
    "mm_centerline_from_points", "mm_centerline_find_ref_idx", "mm_centerline_preprocess",
    "mm_sort_contour_points", "mm_rotate_geometry", "mm_apply_transformations", "mm_best_rotation_three_point",
    "mm_refine_alignment_hausdorff", "mm_align_three_point", "mm_align_manual", "mm_align_combined", "mm_align_walls",
]


class MMGeometry(C.Structure):
    """``mm_geometry`` (include/mm_hausdorff.h)."""
    _fields_ = [
        ("n_frames", C.c_int32),
        ("id", C.c_void_p),
        ("lumen_id", C.c_void_p),
        ("orig_frame", C.c_void_p),
        ("centroid", C.c_void_p),
        ("lumen_off", C.c_void_p),
        ("lumen", C.c_void_p),
        ("has_catheter", C.c_int32),
        ("cath_off", C.c_void_p),
        ("cath", C.c_void_p),
        ("extra_off", C.c_void_p),
        ("extra", C.c_void_p),
        ("has_ref", C.c_void_p),
        ("ref", C.c_void_p),
        ("lumen_centroid", C.c_void_p),
    ]


# include/mm_build.h
EXPORTS_BUILD = ["mm_build_geometry", "mm_build_geometry_lenient", "mm_built_dims", "mm_built_export", "mm_built_destroy", "mm_contour_centroids",
                 "mm_frames_from_flat", "mm_frames_dims", "mm_frames_export", "mm_frames_destroy",
                 "mm_frames_finish_within", "mm_frames_postprocess_pair"]


class MMRecord(C.Structure):
    """``mm_record`` (include/mm_build.h)."""
    _fields_ = [("frame", C.c_uint32), ("phase", C.c_uint8), ("has_m1", C.c_uint8), ("has_m2", C.c_uint8),
                ("pad_", C.c_uint8), ("m1", C.c_double), ("m2", C.c_double)]


class MMFlatGeometry(C.Structure):
    """``mm_flat_geometry`` (include/mm_build.h)."""
    _fields_ = [("g", MMGeometry), ("extra_counts", C.c_void_p), ("has_lumen_centroid", C.c_void_p),
                ("lumen_centroid", C.c_void_p), ("aortic_thickness", C.c_void_p), ("has_aortic", C.c_void_p),
                ("pulmonary_thickness", C.c_void_p), ("has_pulmonary", C.c_void_p), ("lumen_aortic", C.c_void_p),
                ("wall_aortic", C.c_void_p)]


# include/mm_ccta.h
EXPORTS_CCTA = [
    "mm_nn_min_sq_batch", "mm_symmetric_nn_distance", "mm_diameter_morphing", "mm_find_region_points",
    "mm_aortic_diameter_optimization", "mm_diameter_optimization", "mm_wall_diameter_optimization",
    "mm_clean_outlier_points", "mm_find_points_by_cl_region",
]


class MMClGeometry(C.Structure):
    """``mm_cl_geometry`` (include/mm_centerline.h)."""
    _fields_ = [
        ("g", C.POINTER(MMGeometry)),
        ("has_lumen_centroid", C.c_void_p),
        ("lumen_centroid", C.c_void_p),
        ("n_extra_kinds", C.c_int32),
        ("extra_kind_off", C.c_void_p),
        ("lumen_aortic", C.c_void_p),
        ("wall_aortic", C.c_void_p),
        ("wall_kind1", C.c_int32),
    ]


class MMAlignLog(C.Structure):
    _fields_ = [
        ("contour_id", C.c_uint32),
        ("matched_to", C.c_uint32),
        ("rot_deg", C.c_double),
        ("tx", C.c_double),
        ("ty", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
    ]


_lib = None
_engines = weakref.WeakSet()


def _close_all_engines():
    """Interpreter shutdown: release plans and engines while the HIP runtime is still alive
    (destructors running after its teardown abort the process)."""
    for e in list(_engines):
        try:
            e.close()
        except Exception:
            pass


atexit.register(_close_all_engines)


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's) but link it by its unversioned name, so when this
    library is loaded BEFORE torch the loader does not recognise the system runtime as the one torch
    asks for and maps a second HIP runtime, which then finds no GPU ("No HIP GPUs are available") --
    and device pointers / streams could not be shared with torch.distributed (RCCL) anyway.  If torch
    is installed and not yet imported, map its runtime first (by path, without importing torch); our
    library then binds to it by SONAME and a later `import torch` reuses the same mapping.
    MM_HIP_RUNTIME=system keeps the system runtime (for processes that never import torch)."""
    import sys
    if "torch" in sys.modules or os.environ.get("MM_HIP_RUNTIME") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
    except Exception:
        pass            # fall back to the system runtime; torch-free use is unaffected


def lib():
    """Load the HIP extension; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). There is no CPU fallback."
        )
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    P, I, D = C.c_void_p, C.c_int, C.c_double
    I64, I32 = C.c_int64, C.c_int32
    L.mm_device_count.restype = I
    L.mm_last_error.restype = C.c_char_p
    L.mm_version.restype = C.c_char_p
    L.mm_engine_create.restype = I
    L.mm_engine_create.argtypes = [I, P, C.POINTER(P)]
    L.mm_engine_destroy.restype = None
    L.mm_engine_destroy.argtypes = [P]
    L.mm_engine_synchronize.restype = I
    L.mm_engine_synchronize.argtypes = [P]
    L.mm_engine_stream.restype = P
    L.mm_engine_stream.argtypes = [P]
    L.mm_engine_wait_search.restype = I
    L.mm_engine_wait_search.argtypes = [P, P]
    L.mm_engine_profile.restype = I
    L.mm_engine_profile.argtypes = [P, I]
    L.mm_engine_profile_read.restype = I
    L.mm_engine_profile_read.argtypes = [P, C.POINTER(I64), C.POINTER(D), C.POINTER(D), C.POINTER(I64)]
    L.mm_engine_profile_launches.restype = I
    L.mm_engine_profile_launches.argtypes = [P, I64, P, P, C.POINTER(I64)]
    L.mm_engine_bound_stats.restype = I
    L.mm_engine_bound_stats.argtypes = [P, P]
    L.mm_pick_minima.restype = I
    L.mm_pick_minima.argtypes = [P, P, P, I, P, P, I, D, D, D, I, P, P, P, P]
    L.mm_lower_bounds.restype = I
    L.mm_lower_bounds.argtypes = [P, P, P, I, P, P, I, D, D, P, I, I, I, P, P, P, P]
    L.mm_engine_set_bound_matrix.restype = I
    L.mm_engine_set_bound_matrix.argtypes = [P, I]
    L.mm_engine_screen_stats.restype = I
    L.mm_engine_screen_stats.argtypes = [P, P]
    L.mm_parse_contour_table.restype = I64
    L.mm_parse_contour_table.argtypes = [C.c_char_p, I64, C.c_char, P, I64]
    L.mm_engine_set_bound_min_candidates.restype = I
    L.mm_engine_set_bound_min_candidates.argtypes = [P, I64]
    L.mm_hausdorff_2d.restype = I
    L.mm_hausdorff_2d.argtypes = [P, P, P, I, P, P, I, C.POINTER(D)]
    L.mm_hausdorff_batch.restype = I
    L.mm_hausdorff_batch.argtypes = [P, I, P, P, P, P, P, P, P, C.POINTER(I32)]
    L.mm_refine_angles.restype = I64
    L.mm_refine_angles.argtypes = [D, D, D, P, I64]
    L.mm_filter_points_in_region.restype = I64
    L.mm_filter_points_in_region.argtypes = [P, I64, P, P, P, I64]
    L.mm_refine_downsample_count.restype = I64
    L.mm_refine_downsample_count.argtypes = [I64, I64, I64]
    L.mm_search_angles.restype = I64
    L.mm_search_angles.argtypes = [D, D, I, D, D, P, I64, C.POINTER(I), C.POINTER(D)]
    L.mm_best_rotation.restype = I
    L.mm_best_rotation.argtypes = [P, P, P, I, P, P, I, D, D, P, I, I, I, C.POINTER(D), C.POINTER(D),
                                   C.POINTER(I), P]
    batch_args = [P, I, P, P, P, P, P, P, P, P, P, P, P, I]
    L.mm_best_rotation_batch.restype = I
    L.mm_best_rotation_batch.argtypes = batch_args + [P, P, P, P, P]
    L.mm_plan_create.restype = I
    L.mm_plan_create.argtypes = batch_args + [I32, I32, C.POINTER(P)]
    L.mm_plan_create_indexed.restype = I
    L.mm_plan_create_indexed.argtypes = [P, I, P, P, P, P, P, I, P, P, P, P, I, P, P, P, I, I, C.POINTER(P)]
    L.mm_plan_destroy.restype = None
    L.mm_plan_destroy.argtypes = [P]
    L.mm_plan_run.restype = I
    L.mm_plan_run.argtypes = [P]
    L.mm_plan_run_screen_only.restype = I
    L.mm_plan_run_screen_only.argtypes = [P]
    L.mm_plan_fetch.restype = I
    L.mm_plan_fetch.argtypes = [P, P, P, P, P, P]
    L.mm_plan_result_dev.restype = I
    L.mm_plan_result_dev.argtypes = [P, C.POINTER(P), C.POINTER(P)]
    L.mm_plan_time.restype = I
    L.mm_plan_time.argtypes = [P, I, I, C.POINTER(C.c_float)]
    L.mm_plan_stats.restype = I
    L.mm_plan_stats.argtypes = [P, C.POINTER(I64), C.POINTER(D), C.POINTER(I64)]
    L.mm_align_within.restype = I
    L.mm_align_within.argtypes = [P, I, P, D, D, I, I64, I, I, P, C.POINTER(I64)]
    L.mm_within_plan_create.restype = I
    L.mm_within_plan_create.argtypes = [P, I, P, D, D, I, I64, I, C.POINTER(P)]
    L.mm_within_plan_create_sharded.restype = I
    L.mm_within_plan_create_sharded.argtypes = [P, I, P, D, D, I, I64, I, I, I, C.POINTER(P)]
    L.mm_within_plan_run.restype = I
    L.mm_within_plan_run.argtypes = [P, P, C.POINTER(I64), C.POINTER(I64)]
    L.mm_within_plan_destroy.restype = None
    L.mm_within_plan_destroy.argtypes = [P]
    L.mm_within_plan_set_shard.restype = I
    L.mm_within_plan_set_shard.argtypes = [P, I, I]
    L.mm_within_plan_staged.restype = I
    L.mm_within_plan_staged.argtypes = [P, C.POINTER(I64), C.POINTER(I64)]
    L.mm_within_plan_dims.restype = I
    L.mm_within_plan_dims.argtypes = [P, C.POINTER(I32), C.POINTER(I32), P]
    L.mm_shard_grid.restype = I
    L.mm_shard_grid.argtypes = [I, I64, C.POINTER(I), C.POINTER(I)]
    L.mm_within_plan_create_grid.restype = I
    L.mm_within_plan_create_grid.argtypes = [P, I, P, D, D, I, I64, I, I, I, I, C.POINTER(P)]
    L.mm_within_plan_set_shard_grid.restype = I
    L.mm_within_plan_set_shard_grid.argtypes = [P, I, I, I]
    L.mm_within_plan_walk_geoms.restype = I
    L.mm_within_plan_walk_geoms.argtypes = [P, P, P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.mm_comm_broadcast.restype = I
    L.mm_comm_broadcast.argtypes = [P, P, I64, I, P]
    L.mm_within_plan_set_timing_rehearsal.restype = I
    L.mm_within_plan_set_timing_rehearsal.argtypes = [P, I]
    L.mm_comm_unique_id.restype = I
    L.mm_comm_unique_id.argtypes = [P]
    L.mm_comm_init_rank.restype = I
    L.mm_comm_init_rank.argtypes = [P, I, I, I, C.POINTER(P)]
    L.mm_comm_destroy.restype = None
    L.mm_comm_destroy.argtypes = [P]
    L.mm_comm_rank.restype = I
    L.mm_comm_rank.argtypes = [P]
    L.mm_comm_world.restype = I
    L.mm_comm_world.argtypes = [P]
    L.mm_comm_version.restype = I
    L.mm_comm_version.argtypes = []
    L.mm_comm_all_reduce_min_f64.restype = I
    L.mm_comm_all_reduce_min_f64.argtypes = [P, P, I64, P]
    L.mm_comm_all_reduce_min_i64.restype = I
    L.mm_comm_all_reduce_min_i64.argtypes = [P, P, I64, P]
    L.mm_within_plan_search_sharded.restype = I
    L.mm_within_plan_search_sharded.argtypes = [P, P]
    L.mm_within_plan_search_sharded_begin.restype = I
    L.mm_within_plan_search_sharded_begin.argtypes = [P, P]
    L.mm_engine_wait_exchange.restype = I
    L.mm_engine_wait_exchange.argtypes = [P, P]
    L.mm_within_plan_run_sharded.restype = I
    L.mm_within_plan_run_sharded.argtypes = [P, P, P, C.POINTER(I64), C.POINTER(I64)]
    L.mm_within_plan_level_local.restype = I
    L.mm_within_plan_level_local.argtypes = [P, I, P, P, P, P, P]
    L.mm_within_plan_level_collect.restype = I
    L.mm_within_plan_level_collect.argtypes = [P, I, P, P, P, P, P]
    L.mm_within_plan_level_commit.restype = I
    L.mm_within_plan_level_commit.argtypes = [P, I, P, P]
    L.mm_within_plan_fetch_set.restype = I64
    L.mm_within_plan_fetch_set.argtypes = [P, I32, P, P, P, P, I64, C.POINTER(D)]
    L.mm_within_plan_level_launch.restype = I
    L.mm_within_plan_level_launch.argtypes = [P, I]
    L.mm_within_plan_level_export_cost.restype = I
    L.mm_within_plan_level_export_cost.argtypes = [P, I, P]
    L.mm_within_plan_level_export_keys.restype = I
    L.mm_within_plan_level_export_keys.argtypes = [P, I, P, P]
    L.mm_within_plan_level_commit_dev.restype = I
    L.mm_within_plan_level_commit_dev.argtypes = [P, I, P, P]
    L.mm_within_plan_walk.restype = I
    L.mm_within_plan_walk.argtypes = [P, P, C.POINTER(I64), C.POINTER(I64)]
    L.mm_merge_shards.restype = I
    L.mm_merge_shards.argtypes = [I, I, P, P, P, P, P, P, P, P, P]
    L.mm_align_between.restype = I
    L.mm_align_between.argtypes = [P, I, P, P, D, D, I64, I, P, C.POINTER(I64)]
    L.mm_catheter_lumen_vec.restype = I64
    L.mm_catheter_lumen_vec.argtypes = [C.POINTER(MMGeometry), I32, I64, P, P, I64]
    L.mm_extract_between_points.restype = I64
    L.mm_extract_between_points.argtypes = [C.POINTER(MMGeometry), I64, P, P, I64]
    L.mm_frame_translate.restype = None
    L.mm_frame_translate.argtypes = [C.POINTER(MMGeometry), I32, D, D, D]
    L.mm_frame_rotate.restype = None
    L.mm_frame_rotate.argtypes = [C.POINTER(MMGeometry), I32, D, D, D]
    # include/mm_centerline.h
    U32 = C.c_uint32
    L.mm_centerline_from_points.restype = I
    L.mm_centerline_from_points.argtypes = [P, I64, P]
    L.mm_centerline_find_ref_idx.restype = I64
    L.mm_centerline_find_ref_idx.argtypes = [P, I64, P]
    L.mm_centerline_preprocess.restype = I64
    L.mm_centerline_preprocess.argtypes = [P, I64, C.POINTER(MMGeometry), P, I64, C.POINTER(D)]
    L.mm_sort_contour_points.restype = I
    L.mm_sort_contour_points.argtypes = [P, I64]
    L.mm_rotate_geometry.restype = I
    L.mm_rotate_geometry.argtypes = [C.POINTER(MMClGeometry), D]
    L.mm_apply_transformations.restype = I64
    L.mm_apply_transformations.argtypes = [P, I, P, I64, P]
    L.mm_best_rotation_three_point.restype = I
    L.mm_best_rotation_three_point.argtypes = [P, I64, I, P, U32, P, P, P, D, P, C.POINTER(D)]
    L.mm_refine_alignment_hausdorff.restype = I
    L.mm_refine_alignment_hausdorff.argtypes = [P, P, I, P, I64, I64, D, P, I64, D, D, I64, C.POINTER(D),
                                                C.POINTER(I64), C.POINTER(D), P, I64, C.POINTER(I64)]
    L.mm_align_three_point.restype = I
    L.mm_align_three_point.argtypes = [P, I64, P, I, U32, P, P, P, D, I, C.POINTER(D), C.POINTER(D)]
    L.mm_align_walls.restype = I
    L.mm_align_walls.argtypes = [P, I, I]
    L.mm_align_manual.restype = I
    L.mm_align_manual.argtypes = [P, I64, P, I, D, P, I, C.POINTER(D), C.POINTER(D)]
    L.mm_align_combined.restype = I
    L.mm_align_combined.argtypes = [P, P, I64, P, I, U32, P, P, P, P, I64, D, D, I64, I, C.POINTER(D), C.POINTER(D),
                                    C.POINTER(I64), C.POINTER(I64)]
    # include/mm_build.h
    L.mm_build_geometry.restype = I
    L.mm_build_geometry.argtypes = [P, I64, P, P, I64, P, I64, P, I64, P, P, I64, I, D, D, D, C.c_uint32, C.POINTER(P)]
    L.mm_build_geometry_lenient.restype = I
    L.mm_build_geometry_lenient.argtypes = L.mm_build_geometry.argtypes
    L.mm_built_dims.restype = I
    L.mm_built_dims.argtypes = [P, C.POINTER(I32), C.POINTER(I64), C.POINTER(I64), C.POINTER(I64)]
    L.mm_built_export.restype = I
    L.mm_built_export.argtypes = [P, C.POINTER(MMGeometry), P, P, P, P, P, P]
    L.mm_built_destroy.restype = None
    L.mm_built_destroy.argtypes = [P]
    L.mm_contour_centroids.restype = I
    L.mm_contour_centroids.argtypes = [P, P, I64, P]
    L.mm_frames_from_flat.restype = I
    L.mm_frames_from_flat.argtypes = [C.POINTER(MMFlatGeometry), C.POINTER(P)]
    L.mm_frames_dims.restype = I
    L.mm_frames_dims.argtypes = [P, C.POINTER(I32), C.POINTER(I64), C.POINTER(I64), C.POINTER(I64), C.POINTER(I64)]
    L.mm_frames_export.restype = I
    L.mm_frames_export.argtypes = [P, C.POINTER(MMFlatGeometry)]
    L.mm_frames_destroy.restype = None
    L.mm_frames_destroy.argtypes = [P]
    L.mm_frames_finish_within.restype = I
    L.mm_frames_finish_within.argtypes = [P, I64, I, C.POINTER(I)]
    L.mm_frames_postprocess_pair.restype = I
    L.mm_frames_postprocess_pair.argtypes = [P, P, D, I]
    # include/mm_ccta.h
    L.mm_nn_min_sq_batch.restype = I
    L.mm_nn_min_sq_batch.argtypes = [P, I, P, P, I, P, P, P, P]
    L.mm_symmetric_nn_distance.restype = I
    L.mm_symmetric_nn_distance.argtypes = [P, P, I64, P, I64, C.POINTER(D)]
    L.mm_diameter_morphing.restype = I
    L.mm_diameter_morphing.argtypes = [P, I64, P, I64, D, P]
    L.mm_find_region_points.restype = I64
    L.mm_find_region_points.argtypes = [P, P, I64, P, I64, I64, P, P]
    L.mm_aortic_diameter_optimization.restype = I
    L.mm_aortic_diameter_optimization.argtypes = [P, P, I64, P, I64, P, I64, C.POINTER(D), P]
    L.mm_diameter_optimization.restype = I
    L.mm_diameter_optimization.argtypes = [P, P, I64, I64, I64, P, I64, P, I64, P, I64, C.POINTER(D), C.POINTER(D)]
    L.mm_clean_outlier_points.restype = I
    L.mm_clean_outlier_points.argtypes = [P, P, I64, P, I64, D, D, P]
    L.mm_find_points_by_cl_region.restype = I
    L.mm_find_points_by_cl_region.argtypes = [P, P, P, I64, P, I64, P, I64, P]
    L.mm_wall_diameter_optimization.restype = I
    L.mm_wall_diameter_optimization.argtypes = [P, I64, P, P, I64, C.POINTER(D)]
    _lib = L
    return L


def last_error() -> str:
    return lib().mm_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = ""):
    """Map a negative status to RuntimeError, like the reference maps anyhow errors to
    PyRuntimeError (binding/functions.rs:228)."""
    if rc != 0:
        raise RuntimeError(f"{what + ': ' if what else ''}{last_error()} (mm_status {rc})")


def device_count() -> int:
    return int(lib().mm_device_count())


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _xy(a) -> np.ndarray:
    """array-like of points -> (n, >=2) f64 array ((0, 2) when empty)."""
    a = np.asarray(a, dtype=np.float64)
    if a.size == 0:
        return np.zeros((0, 2), dtype=np.float64)
    if a.ndim != 2 or a.shape[1] < 2:
        raise ValueError("point sets must be (n, 2) or (n, 3) arrays")
    return a


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def search_angles(step_deg: float, range_deg: float, center: Optional[float] = None,
                  limes_deg: Optional[float] = None):
    """Candidate list of ``search_range`` (process_utils.rs:43-67), host-exact.
    Returns (angles, degenerate, early_value)."""
    if limes_deg is None:
        limes_deg = range_deg
    deg, early = C.c_int(0), C.c_double(0.0)
    hc, c = (0, 0.0) if center is None else (1, float(center))
    n = lib().mm_search_angles(step_deg, range_deg, hc, c, limes_deg, None, 0, C.byref(deg), C.byref(early))
    if n < 0:
        check(int(n), "search_angles")
    out = np.empty(int(n), dtype=np.float64)
    if n:
        lib().mm_search_angles(step_deg, range_deg, hc, c, limes_deg, _ptr(out), n, C.byref(deg), C.byref(early))
    return out, bool(deg.value), early.value


def refine_angles(initial: float, search_range: float, step: float) -> np.ndarray:
    """Accumulated angle enumeration of refine_alignment_hausdorff (align_algorithms.rs:386-439)."""
    n = lib().mm_refine_angles(initial, search_range, step, None, 0)
    if n < 0:
        check(int(n), "refine_angles")
    out = np.empty(int(n), dtype=np.float64)
    if n:
        lib().mm_refine_angles(initial, search_range, step, _ptr(out), n)
    return out


def filter_points_in_region(points_xyz, start_xyz, end_xyz) -> np.ndarray:
    """Indices of the points inside the +-5 mm bounding box of two centerline points
    (align_algorithms.rs:454-505)."""
    pts = np.ascontiguousarray(points_xyz, dtype=np.float64).reshape(-1, 3)
    s = np.ascontiguousarray(start_xyz, dtype=np.float64)
    e = np.ascontiguousarray(end_xyz, dtype=np.float64)
    idx = np.empty(pts.shape[0], dtype=np.int64)
    m = lib().mm_filter_points_in_region(_ptr(pts), pts.shape[0], _ptr(s), _ptr(e), _ptr(idx), pts.shape[0])
    return idx[:m].copy()


def contour_centroids(xyz: np.ndarray, off: np.ndarray) -> np.ndarray:
    """Contour::compute_centroid (contour.rs:213-224) of CSR contours: sequential sums / count (``mm_contour_centroids``)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    off = np.ascontiguousarray(off, dtype=np.int64)
    out = np.zeros((off.shape[0] - 1, 3), dtype=np.float64)
    check(lib().mm_contour_centroids(_ptr(xyz), _ptr(off), off.shape[0] - 1, _ptr(out)), "contour_centroids")
    return out


def refine_downsample_count(n_filtered: int, n_points_per_frame: int, n_frames: int) -> int:
    """align_algorithms.rs:415-418"""
    return int(lib().mm_refine_downsample_count(n_filtered, n_points_per_frame, n_frames))


class Batch:
    """Host-side description of a batch of searches (SoA f64 + CSR offsets)."""

    def __init__(self, refs: Sequence[np.ndarray], tgts: Sequence[np.ndarray], angle_lists: Sequence[np.ndarray],
                 centres: Sequence[Sequence[float]], flags: Optional[Sequence[int]] = None):
        n = len(refs)
        assert len(tgts) == n and len(angle_lists) == n and len(centres) == n
        self.n_pairs = n

        def pack(sets):
            arrs = [_xy(s) for s in sets]
            off = np.zeros(n + 1, dtype=np.int64)
            off[1:] = np.cumsum([a.shape[0] for a in arrs])
            xs = np.empty(int(off[-1]), dtype=np.float64)
            ys = np.empty(int(off[-1]), dtype=np.float64)
            for i, a in enumerate(arrs):
                xs[off[i]:off[i + 1]] = a[:, 0]
                ys[off[i]:off[i + 1]] = a[:, 1]
            return off, xs, ys

        self.ref_off, self.ref_x, self.ref_y = pack(refs)
        self.tgt_off, self.tgt_x, self.tgt_y = pack(tgts)
        self.ang_off = np.zeros(n + 1, dtype=np.int64)
        self.ang_off[1:] = np.cumsum([len(a) for a in angle_lists])
        self.angles = (np.concatenate([_f64(a) for a in angle_lists]) if n and self.ang_off[-1] > 0
                       else np.zeros(0, dtype=np.float64))
        c = np.asarray(centres, dtype=np.float64).reshape(n, 2) if n else np.zeros((0, 2))
        self.cx = np.ascontiguousarray(c[:, 0])
        self.cy = np.ascontiguousarray(c[:, 1])
        self.flags = np.ascontiguousarray(np.zeros(n, dtype=np.int32) if flags is None
                                          else np.asarray(flags, dtype=np.int32))

    def _args(self):
        return [self.n_pairs, _ptr(self.ref_off), _ptr(self.ref_x), _ptr(self.ref_y),
                _ptr(self.tgt_off), _ptr(self.tgt_x), _ptr(self.tgt_y),
                _ptr(self.ang_off), _ptr(self.angles), _ptr(self.cx), _ptr(self.cy), _ptr(self.flags)]


class Comm:
    """``mm_comm``: the RCCL communicator owned by the library (include/mm_hausdorff.h, "multi-GPU").  Rank 0 makes
    the 128-byte id (``Comm.unique_id()``), the host hands it to every rank, every rank constructs ``Comm(id, rank,
    world, device)`` -- a collective call, like ncclCommInitRank."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(lib().mm_comm_unique_id(buf), "mm_comm_unique_id")
        return buf.raw

    def __init__(self, uid: bytes, rank: int, world: int, device: int = -1):
        if len(uid) != Comm.ID_BYTES:
            raise ValueError("the communicator id is 128 bytes")
        self.handle = C.c_void_p()
        check(lib().mm_comm_init_rank(C.create_string_buffer(uid, Comm.ID_BYTES), int(rank), int(world), int(device),
                                      C.byref(self.handle)), "mm_comm_init_rank")
        self.rank, self.world = int(rank), int(world)

    def all_reduce_min_f64(self, dev_ptr: int, n: int, stream: int):
        check(lib().mm_comm_all_reduce_min_f64(self.handle, C.c_void_p(dev_ptr), int(n), C.c_void_p(stream)),
              "mm_comm_all_reduce_min_f64")

    def broadcast(self, dev_ptr: int, nbytes: int, root: int, stream: int):
        check(lib().mm_comm_broadcast(self.handle, C.c_void_p(dev_ptr), int(nbytes), int(root), C.c_void_p(stream)), "mm_comm_broadcast")

    def all_reduce_min_i64(self, dev_ptr: int, n: int, stream: int):
        check(lib().mm_comm_all_reduce_min_i64(self.handle, C.c_void_p(dev_ptr), int(n), C.c_void_p(stream)),
              "mm_comm_all_reduce_min_i64")

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            lib().mm_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_grid(world: int, n_jobs: int):
    """``mm_shard_grid``: the default (pair_blocks, cand_slices) for `world` ranks and n_jobs frame pairs."""
    pb, cs = C.c_int(0), C.c_int(0)
    check(lib().mm_shard_grid(int(world), int(n_jobs), C.byref(pb), C.byref(cs)), "mm_shard_grid")
    return int(pb.value), int(cs.value)


class Engine:
    """One HIP device + stream (``mm_engine``)."""

    def __init__(self, device: int = -1, stream: Optional[int] = None):
        self._children = weakref.WeakSet()   # plans staged on this engine: closed before the engine
        self._h = C.c_void_p()
        check(lib().mm_engine_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)),
              "mm_engine_create")
        _engines.add(self)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            for child in list(self._children):
                child.close()
            lib().mm_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return int(lib().mm_engine_stream(self._h) or 0)

    def synchronize(self):
        check(lib().mm_engine_synchronize(self._h), "mm_engine_synchronize")

    def wait_search(self, other: "Engine"):
        """Whatever this engine enqueues next on its main stream starts when `other`'s most recent search launch
        ends (``mm_engine_wait_search``)."""
        check(lib().mm_engine_wait_search(self._h, other._h), "mm_engine_wait_search")

    def wait_exchange(self, other: "Engine"):
        """Whatever this engine enqueues next on its main stream starts when the sharded level most recently enqueued on
        `other` (WithinPlan.search_sharded_begin) has finished its exchange (``mm_engine_wait_exchange``)."""
        check(lib().mm_engine_wait_exchange(self._h, other._h), "mm_engine_wait_exchange")

    def profile(self, enable: bool = True):
        """hipEvent timing around every launch of the scoring kernel (mm_engine_profile)."""
        check(lib().mm_engine_profile(self._h, int(enable)), "mm_engine_profile")

    def profile_launches(self, cap: int = 4096):
        """Per-launch (ms, pair-distance evaluations) of the scoring kernel since profiling was enabled;
        call before profile_read(), which resets them."""
        ms = np.zeros(cap, dtype=np.float32)
        pe = np.zeros(cap, dtype=np.float64)
        n = C.c_int64(0)
        check(lib().mm_engine_profile_launches(self._h, cap, _ptr(ms), _ptr(pe), C.byref(n)), "mm_engine_profile_launches")
        k = min(int(n.value), cap)
        return ms[:k].astype(np.float64), pe[:k]

    def set_bound_min_candidates(self, n: int):
        """MM_PRECISION_F32_BOUNDED uses its bound rounds only on batches of at least n candidates (default 16384)."""
        check(lib().mm_engine_set_bound_min_candidates(self._h, int(n)), "mm_engine_set_bound_min_candidates")

    def lower_bounds(self, ref, tgt, angles, centre, skip_zero=True, matrix=True):
        """TEST HOOK (``mm_lower_bounds``): the bounded search's lower bound of every candidate's SQUARED cost, with the
        kernel's e2, delta and query stride."""
        ref = np.ascontiguousarray(ref, dtype=np.float64); tgt = np.ascontiguousarray(tgt, dtype=np.float64)
        rx, ry = np.ascontiguousarray(ref[:, 0]), np.ascontiguousarray(ref[:, 1])
        tx, ty = np.ascontiguousarray(tgt[:, 0]), np.ascontiguousarray(tgt[:, 1])
        ang = np.ascontiguousarray(angles, dtype=np.float64)
        out = np.zeros(len(ang), dtype=np.float32)
        e2, delta, stride = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        check(lib().mm_lower_bounds(self._h, _ptr(rx), _ptr(ry), len(rx), _ptr(tx), _ptr(ty), len(tx), float(centre[0]),
                                    float(centre[1]), _ptr(ang), len(ang), MM_SEARCH_SKIP_ZERO if skip_zero else 0, int(bool(matrix)),
                                    _ptr(out), C.byref(e2), C.byref(delta), C.byref(stride)), "mm_lower_bounds")
        return out, e2.value, delta.value, stride.value

    def pick_minima(self, ref, tgt, angle, centre, skip_zero=True):
        """TEST HOOK (``mm_pick_minima``): squared row minima, squared column minima, screened squared value and e2 of one
        candidate, as the first pick of the matrix-pipe bounded search leaves them."""
        ref = np.ascontiguousarray(ref, dtype=np.float64); tgt = np.ascontiguousarray(tgt, dtype=np.float64)
        rx, ry = np.ascontiguousarray(ref[:, 0]), np.ascontiguousarray(ref[:, 1])
        tx, ty = np.ascontiguousarray(tgt[:, 0]), np.ascontiguousarray(tgt[:, 1])
        rows, cols = np.zeros(len(rx), dtype=np.float32), np.zeros(len(tx), dtype=np.float32)
        val, e2 = np.zeros(1, dtype=np.float32), C.c_double(0.0)
        check(lib().mm_pick_minima(self._h, _ptr(rx), _ptr(ry), len(rx), _ptr(tx), _ptr(ty), len(tx), float(centre[0]),
                                   float(centre[1]), float(angle), MM_SEARCH_SKIP_ZERO if skip_zero else 0, _ptr(rows), _ptr(cols),
                                   _ptr(val), C.byref(e2)), "mm_pick_minima")
        return rows, cols, float(val[0]), e2.value

    def set_bound_matrix(self, on: bool):
        """MM_PRECISION_F32_BOUNDED: bounds and survivors on the matrix pipe (default) or on the packed-FMA kernels."""
        check(lib().mm_engine_set_bound_matrix(self._h, int(on)), "mm_engine_set_bound_matrix")

    def screen_stats(self):
        """Candidates screened since the engine was created, by kernel (``mm_engine_screen_stats``)."""
        out = np.zeros(5, dtype=np.int64)
        check(lib().mm_engine_screen_stats(self._h, _ptr(out)), "mm_engine_screen_stats")
        return {"direct_f32": int(out[0]), "packed_fma": int(out[1]), "matrix": int(out[2]), "matrix_blocks": int(out[3]),
                "exact_f64": int(out[4])}

    def bound_stats(self):
        """MM_PRECISION_F32_BOUNDED since profile(True): candidates offered, lower-bounded in rounds 1-3 and fully
        screened (include/mm_hausdorff.h)."""
        out = np.zeros(5, dtype=np.int64)
        check(lib().mm_engine_bound_stats(self._h, _ptr(out)), "mm_engine_bound_stats")
        return {"offered": int(out[0]), "bounded_round1": int(out[1]), "bounded_round2": int(out[2]),
                "bounded_round3": int(out[3]), "screened": int(out[4])}

    def profile_read(self):
        n, ms, pe, ca = C.c_int64(0), C.c_double(0.0), C.c_double(0.0), C.c_int64(0)
        check(lib().mm_engine_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(pe), C.byref(ca)),
              "mm_engine_profile_read")
        return {"launches": int(n.value), "ms": float(ms.value), "pair_evals": float(pe.value),
                "candidates": int(ca.value)}

    # -- metric ----------------------------------------------------------------------
    def hausdorff(self, a, b) -> float:
        """``hausdorff_distance`` (process_utils.rs:78-82), f64-exact on the device."""
        a, b = _xy(a), _xy(b)
        ax, ay = _f64(a[:, 0]), _f64(a[:, 1])
        bx, by = _f64(b[:, 0]), _f64(b[:, 1])
        out = C.c_double(0.0)
        check(lib().mm_hausdorff_2d(self._h, _ptr(ax), _ptr(ay), len(ax), _ptr(bx), _ptr(by), len(bx), C.byref(out)),
              "mm_hausdorff_2d")
        return out.value

    def hausdorff_batch(self, pairs):
        """``hausdorff_distance`` for a list of (A, B) point-set pairs in one launch; returns
        (costs, index of the first strict minimum) -- the evaluation + selection of
        refine_alignment_hausdorff (align_algorithms.rs:431-437)."""
        n = len(pairs)
        A = [_xy(a) for a, _ in pairs]
        B = [_xy(b) for _, b in pairs]

        def pack(sets):
            off = np.zeros(n + 1, dtype=np.int64)
            off[1:] = np.cumsum([s.shape[0] for s in sets])
            x = np.concatenate([s[:, 0] for s in sets]) if n else np.zeros(0)
            y = np.concatenate([s[:, 1] for s in sets]) if n else np.zeros(0)
            return off, _f64(x), _f64(y)

        ao, ax, ay = pack(A)
        bo, bx, by = pack(B)
        out = np.zeros(n, dtype=np.float64)
        fm = C.c_int32(-1)
        check(lib().mm_hausdorff_batch(self._h, n, _ptr(ao), _ptr(ax), _ptr(ay), _ptr(bo), _ptr(bx), _ptr(by),
                                       _ptr(out), C.byref(fm)), "mm_hausdorff_batch")
        return out, int(fm.value)

    # -- one search ------------------------------------------------------------------
    def best_rotation(self, ref, tgt, angles, centre, skip_zero=True, precision=MM_PRECISION_F32_MATRIX,
                      return_costs=False):
        ref, tgt = _xy(ref), _xy(tgt)
        rx, ry = _f64(ref[:, 0]), _f64(ref[:, 1])
        tx, ty = _f64(tgt[:, 0]), _f64(tgt[:, 1])
        ang = _f64(angles)
        costs = np.empty(len(ang), dtype=np.float64) if return_costs else None
        ba, bc, bi = C.c_double(0.0), C.c_double(0.0), C.c_int(-1)
        check(lib().mm_best_rotation(self._h, _ptr(rx), _ptr(ry), len(rx), _ptr(tx), _ptr(ty), len(tx),
                                     float(centre[0]), float(centre[1]), _ptr(ang), len(ang),
                                     MM_SEARCH_SKIP_ZERO if skip_zero else 0, precision,
                                     C.byref(ba), C.byref(bc), C.byref(bi), _ptr(costs)), "mm_best_rotation")
        if return_costs:
            return bi.value, ba.value, bc.value, costs
        return bi.value, ba.value, bc.value

    # -- batch -----------------------------------------------------------------------
    def best_rotation_batch(self, batch: Batch, precision=MM_PRECISION_F32_MATRIX, return_costs=False):
        n = batch.n_pairs
        bidx = np.full(n, -1, dtype=np.int32)
        bang = np.zeros(n, dtype=np.float64)
        bcost = np.zeros(n, dtype=np.float64)
        nres = np.zeros(n, dtype=np.int32)
        costs = np.zeros(int(batch.ang_off[-1]), dtype=np.float64) if return_costs else None
        check(lib().mm_best_rotation_batch(self._h, *batch._args(), precision, _ptr(bidx), _ptr(bang), _ptr(bcost),
                                           _ptr(nres), _ptr(costs)), "mm_best_rotation_batch")
        out = {"best_idx": bidx, "best_angle": bang, "best_cost": bcost, "n_rescored": nres}
        if return_costs:
            out["costs"] = costs
        return out

    def plan(self, batch: Batch, precision=MM_PRECISION_F32_MATRIX, angle_begin=0, angle_end=2**31 - 1) -> "Plan":
        return Plan(self, batch, precision, angle_begin, angle_end)


class IndexedBatch:
    """Point sets given once, pairs referencing them by index, one shared candidate list
    (``mm_plan_create_indexed``)."""

    def __init__(self, sets: Sequence[np.ndarray], set_centres, ref_set, tgt_set, angles, flags=None):
        arrs = [_xy(s) for s in sets]
        self.n_sets = len(arrs)
        self.set_off = np.zeros(self.n_sets + 1, dtype=np.int64)
        self.set_off[1:] = np.cumsum([a.shape[0] for a in arrs])
        self.x = _f64(np.concatenate([a[:, 0] for a in arrs])) if arrs else np.zeros(0)
        self.y = _f64(np.concatenate([a[:, 1] for a in arrs])) if arrs else np.zeros(0)
        c = np.asarray(set_centres, dtype=np.float64).reshape(self.n_sets, 2)
        self.set_cx, self.set_cy = _f64(c[:, 0]), _f64(c[:, 1])
        self.ref_set = np.ascontiguousarray(ref_set, dtype=np.int32)
        self.tgt_set = np.ascontiguousarray(tgt_set, dtype=np.int32)
        self.n_pairs = int(self.ref_set.shape[0])
        self.angles = _f64(angles)
        self.ang_off = np.array([0, len(self.angles)], dtype=np.int64)
        self.cx = _f64(self.set_cx[self.tgt_set])       # pair centre = centre of its sets
        self.cy = _f64(self.set_cy[self.tgt_set])
        self.flags = np.ascontiguousarray(np.zeros(self.n_pairs, dtype=np.int32) if flags is None
                                          else np.asarray(flags, dtype=np.int32))


class Plan:
    """Device-resident batch (``mm_plan``): upload once, run many times."""

    def __init__(self, engine: Engine, batch, precision, angle_begin=0, angle_end=2**31 - 1, want_costs=True):
        self.engine = engine
        self.batch = batch
        self._h = C.c_void_p()
        engine._children.add(self)
        if isinstance(batch, IndexedBatch):
            b = batch
            check(lib().mm_plan_create_indexed(engine.handle, b.n_sets, _ptr(b.set_off), _ptr(b.x), _ptr(b.y),
                                               _ptr(b.set_cx), _ptr(b.set_cy), b.n_pairs, _ptr(b.ref_set),
                                               _ptr(b.tgt_set), _ptr(b.ang_off), _ptr(b.angles), 1, _ptr(b.cx),
                                               _ptr(b.cy), _ptr(b.flags), precision, int(bool(want_costs)),
                                               C.byref(self._h)), "mm_plan_create_indexed")
            self._n_costs = b.n_pairs * len(b.angles)
            return
        self._n_costs = int(batch.ang_off[-1])
        check(lib().mm_plan_create(engine.handle, *batch._args(), precision, int(angle_begin), int(angle_end),
                                   C.byref(self._h)), "mm_plan_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mm_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, screen_only=False):
        f = lib().mm_plan_run_screen_only if screen_only else lib().mm_plan_run
        check(f(self._h), "mm_plan_run")

    def fetch(self, return_costs=False):
        n = self.batch.n_pairs
        bidx = np.full(n, -1, dtype=np.int32)
        bang = np.zeros(n, dtype=np.float64)
        bcost = np.zeros(n, dtype=np.float64)
        nres = np.zeros(n, dtype=np.int32)
        costs = np.full(self._n_costs, np.nan, dtype=np.float64) if return_costs else None
        check(lib().mm_plan_fetch(self._h, _ptr(bidx), _ptr(bang), _ptr(bcost), _ptr(nres), _ptr(costs)),
              "mm_plan_fetch")
        out = {"best_idx": bidx, "best_angle": bang, "best_cost": bcost, "n_rescored": nres}
        if return_costs:
            out["costs"] = costs
        return out

    def result_dev_ptrs(self):
        c, i = C.c_void_p(), C.c_void_p()
        check(lib().mm_plan_result_dev(self._h, C.byref(c), C.byref(i)), "mm_plan_result_dev")
        return int(c.value or 0), int(i.value or 0)

    def time(self, iters=10, screen_only=False) -> float:
        ms = C.c_float(0.0)
        check(lib().mm_plan_time(self._h, iters, int(screen_only), C.byref(ms)), "mm_plan_time")
        return float(ms.value)

    def stats(self):
        n, pe, hb = C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        check(lib().mm_plan_stats(self._h, C.byref(n), C.byref(pe), C.byref(hb)), "mm_plan_stats")
        return {"candidates": int(n.value), "pair_evals": float(pe.value), "hbm_bytes": int(hb.value)}
