// mm_device.h -- structures shared by the HIP kernels (mm_kernels.hip) and the host
// engine (mm_engine.cpp).  Internal; the public boundary is include/mm_hausdorff.h.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace mm {

// One (reference set, target set, candidate list) search.
struct PairDesc {
    int32_t ref_off, n_ref;   // into the ref SoA arrays
    int32_t tgt_off, n_tgt;   // into the tgt SoA arrays
    int32_t ang_off, n_ang;   // into cos/sin tables and the per-candidate outputs
    int32_t flags;            // MM_SEARCH_SKIP_ZERO
    int32_t ang_full;         // length of the pair's full candidate list (>= n_ang)
    int32_t ang_begin;        // first candidate of the slice this plan owns
    int32_t pad;
    double  cx, cy;           // rotation centre (exact kernel)
    double  delta;            // f32 screening error bound (same unit as the costs)
};

// One workgroup's share: `cnt` consecutive candidates of one pair.
struct WorkItem {
    int32_t pair, a0, cnt, pad;
};

struct BatchDev {
    const PairDesc* pairs;
    const WorkItem* work;
    int32_t n_pairs, n_work;
    // f32 screening inputs (coordinates relative to the rotation centre)
    const float *ref32x, *ref32y, *tgt32x, *tgt32y, *cos32, *sin32;
    // f64 exact inputs (absolute coordinates)
    const double *ref64x, *ref64y, *tgt64x, *tgt64y, *cos64, *sin64;
    // per-candidate outputs
    float*    sq32;       // squared Hausdorff from the screening kernel
    double*   sq64;       // exact squared Hausdorff (valid where flag != 0 or in exact mode)
    uint8_t*  flag;       // 1 = shortlisted (re-scored in f64)
    // shortlist queue
    WorkItem* items;      // capacity = total candidates
    int32_t*  n_items;    // device counter
    // per-pair results
    double*   best_cost;
    int32_t*  best_idx;
    int32_t*  n_rescored;
    double*   all_costs;  // optional per-candidate sqrt'ed costs
};

// Launchers (mm_kernels.hip).  All asynchronous on `s`.
struct KernelConfig { int r; int nli; size_t lds; };
hipError_t launch_screen_f32(const BatchDev& b, int max_na, int max_nbp, hipStream_t s);
hipError_t launch_exact_all(const BatchDev& b, int max_na, int max_nbp, hipStream_t s);
hipError_t launch_shortlist(const BatchDev& b, hipStream_t s);
hipError_t launch_rescore(const BatchDev& b, int max_na, int max_nbp, int total_candidates, hipStream_t s);
hipError_t launch_finalize(const BatchDev& b, int use_flags, hipStream_t s);
size_t     lds_bytes_f32(int nbp);
size_t     lds_bytes_f64(int nbp);
int        max_target_points_f32();
int        max_target_points_f64();
const char* screen_kernel_name();

}  // namespace mm
