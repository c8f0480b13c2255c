// mm_engine.h -- internal host-side classes behind include/mm_hausdorff.h.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mm_hausdorff.h"
#include "mm_device.h"

namespace mm {

extern thread_local std::string g_last_error;
int set_error(int code, const std::string& msg);
int hip_error(hipError_t e, const char* what);

// One device + one stream + grow-only staging buffers (pinned host, device) reused by the
// transient plans behind mm_best_rotation_batch, so the per-call cost in the sequential
// chain is one H2D copy, the kernel launches and one D2H copy.
struct Engine {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* host_buf = nullptr; size_t host_cap = 0;
    void* dev_buf = nullptr;  size_t dev_cap = 0;
    // profiling of the scoring kernel (mm_engine_profile*)
    bool profile = false;
    std::vector<hipEvent_t> events;   // pairs: [2k] before, [2k+1] after launch k
    size_t launches = 0;
    double prof_pair_evals = 0.0;
    int64_t prof_candidates = 0;
    int profile_begin();
    int profile_end(double pair_evals, int64_t candidates);
    int ensure_host(size_t bytes);
    int ensure_dev(size_t bytes);
};

struct Plan {
    Engine* eng = nullptr;
    int P = 0, W = 0;
    int64_t A = 0;
    int precision = MM_PRECISION_F32;
    bool transient = false;
    int max_na = 1, max_nbp = 16;
    double pair_evals = 0.0;
    size_t in_bytes = 0, total_bytes = 0;
    size_t off_best_cost = 0, off_best_idx = 0, off_n_rescored = 0, off_all_costs = 0;
    unsigned char* blob = nullptr;
    BatchDev dev{};
    std::vector<PairDesc> host_pairs;
    std::vector<WorkItem> host_work;
    std::vector<double> host_angles;       // slice-local candidate angles
    std::vector<double> first_angle;       // per pair: first candidate of the slice (trivial pairs)
    std::vector<int64_t> user_ang_off;     // caller's candidate offsets (for all_costs scatter)
    int32_t slice_end = INT32_MAX;
    std::vector<uint8_t> trivial;          // pair has an empty set: every cost is 0.0

    int build(Engine* e, int n_pairs, const int64_t* ref_off, const double* ref_x, const double* ref_y,
              const int64_t* tgt_off, const double* tgt_x, const double* tgt_y,
              const int64_t* ang_off, const double* angles, const double* cx, const double* cy,
              const int32_t* flags, int precision, int32_t angle_begin, int32_t angle_end,
              bool want_costs, bool transient);
    int run(bool screen_only);
    int fetch(int32_t* best_idx, double* best_angle, double* best_cost, int32_t* n_rescored, double* all_costs);
    ~Plan();
};

}  // namespace mm
