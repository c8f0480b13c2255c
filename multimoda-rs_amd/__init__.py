"""multimoda-rs_amd -- MI355X-native Hausdorff pose-search engine.

One hot path of yungselm/multimoda-rs (the brute-force / coarse-to-fine Hausdorff rotation
search of ``src/intravascular``) rebuilt for gfx950: hand-written HIP kernels behind a
C ABI (``include/mm_hausdorff.h``), host orchestration in C++, and this thin Python layer
that mirrors the reference's ``mm.from_array_*`` / ``mm.from_file_*`` entry points for
that path.  Import as ``import multimoda_rs_amd as mm`` (root-level shim), since the
package directory name contains a hyphen.
"""
from __future__ import annotations

from . import _native
from ._native import (MM_PRECISION_F32, MM_PRECISION_F32_BOUNDED, MM_PRECISION_F32_FAST, MM_PRECISION_F32_MATRIX, MM_PRECISION_F64, MM_SEARCH_SKIP_ZERO, Batch, Comm, Engine,
                      IndexedBatch, Plan,
                      device_count, filter_points_in_region, refine_angles, refine_downsample_count, search_angles)
from .geometry import (FlatGeometry, WithinPlan, align_between, align_within, between_points, catheter_points,
                       contour_centroid, search_set)
from .io import InputData, Record, build_geometry_from_inputdata, numpy_to_inputdata, process_directory
from .api import (GeometryPair, align_frames_in_geometries, from_array_doublepair, from_array_full,
                  from_array_single, from_array_singlepair, from_file_doublepair, from_file_full,
                  from_file_single, from_file_singlepair)
from .centerline import (Centerline, align_combined, align_manual, align_three_point, numpy_to_centerline, read_centerline_vtp,
                         preprocess_centerline)
from . import centerline
from . import ccta
from .ccta import (adjust_diameter_centerline_morphing_simple, clean_outlier_points, find_aorta_scaling,
                   find_aortic_scaling, find_aortic_wall_scaling, find_distal_and_proximal_scaling,
                   find_points_by_cl_region, find_proximal_distal_scaling)
from .convert import numpy_to_geometry, to_array
from .export import to_obj
from . import export
from .extension import ShiftRotationSearch
from .synth import synthetic_case, synthetic_pullback

__version__ = "0.1.0"

__all__ = [
    "Engine", "Comm", "Batch", "Plan", "FlatGeometry", "device_count", "search_angles", "refine_angles",
    "filter_points_in_region", "refine_downsample_count",
    "align_within", "align_between", "WithinPlan", "search_set", "between_points",
    "from_file_full", "from_file_doublepair", "from_file_singlepair", "from_file_single",
    "from_array_full", "from_array_doublepair", "from_array_singlepair", "from_array_single",
    "InputData", "Record", "numpy_to_inputdata", "build_geometry_from_inputdata", "process_directory",
    "GeometryPair", "align_frames_in_geometries",
    "ShiftRotationSearch", "to_array", "numpy_to_geometry", "to_obj", "export",
    "Centerline", "numpy_to_centerline", "read_centerline_vtp", "preprocess_centerline", "align_three_point", "align_manual",
    "align_combined", "centerline",
    "ccta", "adjust_diameter_centerline_morphing_simple", "find_proximal_distal_scaling", "find_aortic_scaling",
    "find_aortic_wall_scaling", "find_distal_and_proximal_scaling", "find_aorta_scaling",
    "find_points_by_cl_region", "clean_outlier_points",
    "synthetic_case", "synthetic_pullback", "catheter_points", "contour_centroid",
    "MM_PRECISION_F32", "MM_PRECISION_F32_BOUNDED", "MM_PRECISION_F32_FAST", "MM_PRECISION_F32_MATRIX", "MM_PRECISION_F64", "MM_SEARCH_SKIP_ZERO",
]
