// mm_engine.h -- internal host-side classes behind include/mm_hausdorff.h.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <array>
#include <atomic>
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
    struct Buf { void* p = nullptr; size_t cap = 0; };
    int device = 0;
    hipStream_t stream = nullptr;   // main stream: the searches of device-resident plans (the long launches)
    bool own_stream = false;
    // high-priority side stream for everything short: staging of resident plans, transient batches (the
    // between-pullback searches, the per-step searches of the faithful chain).  Two engines share one GPU in a
    // pipelined driver; without the priority a 23 us between-stage kernel queues behind the other engine's
    // 30 ms launch.  Equals `stream` when the caller supplied the stream.
    hipStream_t aux = nullptr;
    bool own_aux = false;
    // recorded on `stream` right after the dominant kernel of a resident plan's search: another engine's stream can
    // wait for it (mm_engine_wait_search) and start ITS search exactly when this one's long launch ends, while the
    // short tail (shortlist, re-score, argmin, fetch) and the host work of this case proceed beside it
    hipEvent_t search_done = nullptr;
    bool search_done_recorded = false;
    // recorded on `stream` behind the last copy of a sharded level's exchange (export kernels, collectives, records to
    // pinned memory): mm_engine_wait_exchange orders another engine's next launch behind the whole level
    hipEvent_t tail_done = nullptr;
    bool tail_done_recorded = false;
    Buf host_pts, host_lvl;   // pinned staging: point pool / level (+ results)
    Buf host_pof;             // pinned: job -> pair map of a sharded level (WithinPlan::upload_pair_of_job)
    hipEvent_t pof_done = nullptr; bool pof_busy = false;   // its upload in flight
    Buf dev_pts, dev_lvl;     // device buffers of transient plans
    Buf dev_raw;              // raw pullbacks of a within-plan while its search sets are built on the device
    int ensure(Buf& b, size_t bytes, bool host);
    // Device blobs of resident plans come from a small per-engine cache: hipFree waits for EVERY stream of the
    // device (a plan closed beside another engine's 30 ms launch stalled its thread for the whole launch), and
    // a stream of cases asks for the same sizes again and again.  blob_release keeps the block for the next
    // blob_alloc of a similar size (at most kBlobCache blocks; the rest, and everything at destroy, is freed).
    static constexpr int kBlobCache = 8;
    std::vector<Buf> blob_cache;
    int blob_alloc(void** p, size_t bytes, size_t* cap);
    void blob_release(void* p, size_t cap);
    int sync_all();                 // both streams
    // grow-only pageable scratch for host-side set construction (refinement grid): a fresh 50 MB
    // std::vector per call costs ~10 ms of zero-fill and page faults
    std::vector<double> scratch[3];
    double* scratch_f64(int slot, size_t n) { if (scratch[slot].size() < n) scratch[slot].resize(n); return scratch[slot].data(); }
    // profiling of the scoring kernel (mm_engine_profile*)
    bool profile = false;
    std::vector<hipEvent_t> events;   // pairs: [2k] before, [2k+1] after launch k
    size_t launches = 0;
    double prof_pair_evals = 0.0;
    std::vector<double> launch_pair_evals;   // per launch, same indexing as the event pairs
    int64_t prof_candidates = 0;
    // bounded screen (MM_PRECISION_F32_BOUNDED) while profiling: candidates offered / bounded in round 1 (host
    // counts), device accumulators [1] bounded in round 2, [2] fully screened
    int64_t bound_min_candidates = 16384;   // smaller batches skip the bound rounds (mm_engine_set_bound_min_candidates)
    int64_t bound_offered = 0, bound_round1 = 0;
    bool bound_matrix = true;          // MM_PRECISION_F32_BOUNDED: bounds and survivors on the matrix pipe (mm_engine_set_bound_matrix)
    int bound_matrix_qt = 1, bound_matrix_nc = 1;
    bool bound_matrix_kept = false;    // survivors through k_screen_mx (measured slower than the packed-FMA screen: off)   // k_bound_mx's variant: query tiles per side, candidates per wave
    // candidates screened since the engine was created, by kernel: [0] direct-form f32, [1] packed FMA, [2] matrix pipe
    // (whole target set per wave), [3] matrix pipe (target set in column blocks), [4] exact f64 for every candidate
    // (mm_engine_screen_stats; atomics: levels are staged from several host threads)
    std::atomic<int64_t> screened[5] = {};
    unsigned long long* dev_stats = nullptr;
    int profile_begin(hipStream_t s);
    int profile_end(hipStream_t s, double pair_evals, int64_t candidates);
};

// ---- internal batch description (the C ABI wrappers translate into this) --------------
struct SetRef {            // one point set: SoA f64 as given + the centre its f32 copy is relative to
    const double* x; const double* y; int32_t n; double cx, cy;
};
struct PairSpec {          // one search
    int32_t ref_set, tgt_set;      // indices into the set list
    double cx, cy;                 // rotation centre (must equal the centre of both sets' f32 copies)
    int32_t flags;
    const double* angles; int32_t n_angles;   // candidate list (pairs passing the same pointer share tables)
    double tie_tol;                // near-tie tolerance on the exact cost (0 -> exact ties only)
    double delta_extra;            // added to the f32 screening bound
    int32_t slice_begin = 0, slice_end = INT32_MAX;   // this pair's share of the candidate axis
};
struct Comm;               // mm_comm.cpp: the RCCL communicator behind mm_comm_*

struct BatchResult {
    std::vector<int32_t> best_idx, n_rescored, near_cnt, near_idx;  // near_idx: kMaxNear per pair
    std::vector<double> best_cost;
};

// A staged batch: point pool (uploaded once) + a re-stageable level (descriptors, candidate
// tables, outputs).  Transient plans borrow the engine's grow-only buffers.
struct Plan {
    Engine* eng = nullptr;
    bool transient = false;
    hipStream_t stream = nullptr;   // run / fetch: the engine's aux stream for transient plans, its main stream otherwise
    int precision = MM_PRECISION_F32;
    // sets
    std::vector<int32_t> set_off, set_len;
    std::vector<double> set_rho;   // max distance of a point from the set's centre
    int64_t n_points = 0;
    unsigned char* pts_blob = nullptr; size_t pts_bytes = 0; bool own_pts = false; size_t pts_cap = 0;
    size_t o32x = 0, o32y = 0, o64x = 0, o64y = 0;   // planes of the pool inside pts_blob
    // level
    int P = 0, W = 0;
    int64_t A = 0;            // candidates in this plan (sum of slices)
    int64_t T = 0;            // cos/sin table entries
    int max_na = 1, max_nbp = 16;
    double pair_evals = 0.0;
    unsigned char* lvl_blob = nullptr; size_t lvl_cap = 0; bool own_lvl = false;
    size_t lvl_in_bytes = 0, lvl_bytes = 0;
    size_t off_best_cost = 0, res_bytes = 0, off_all_costs = 0;
    size_t r_best_idx = 0, r_n_rescored = 0, r_near_cnt = 0, r_near_idx = 0;  // offsets inside the result block
    int32_t slice_end = INT32_MAX;
    BatchDev dev{};
    std::vector<PairDesc> host_pairs;
    std::vector<WorkItem> host_work;
    std::vector<double> host_tables;          // distinct candidate lists (slice-local), dev tables mirror them
    std::vector<uint8_t> trivial;             // pair has an empty set: every cost is 0.0
    bool want_costs = false;
    bool use_fast = false;                    // expanded-form screening kernel selected
    bool use_mx = false;                      // matrix-pipe screening kernel selected (MM_PRECISION_F32_MATRIX)
    // MM_PRECISION_F32_MATRIX: the work list is grouped by screen variant, one launch per group.  kind 3 = no screen (a set of fewer than 64 points: every candidate scored exactly); kind 2 = k_screen_mx
    // <nct, multi> with LDS for a_cap row tiles; kind 1 / 0 = the pairs outside its range (fewer than 64 or more than 2048
    // points, radii f16 cannot scale): packed-FMA screen / direct-form f32 screen
    struct ScreenGroup { int kind, nct, multi, a_cap, work_begin, work_count; };
    std::vector<ScreenGroup> groups;
    bool use_lb = false;                      // lower-bound pass in front of the screen (MM_PRECISION_F32_BOUNDED)
    int W_lb = 0, lb_stride = 0, lb_runs_cap = 0, max_nt = 1;
    int lb_mx_tiles = 0, kept_nct = 0, kept_acap = 0;   // bounded search on the matrix pipe (BatchDev::lb_mx, kept_mx_*)
    double lb_pair_evals = 0.0;               // pair-distances of the first bound round
    int64_t lb_sparse_total = 0;              // candidates the first bound round scores
    std::vector<WorkItem> host_work_lb;

    // pool layout + allocation for sets of the given sizes, no data (the caller fills it on the device and
    // sets set_rho); stage_sets = alloc_pool + host conversion + one H2D copy
    int alloc_pool(Engine* e, const std::vector<int32_t>& lens, bool transient);
    // `st` (nullable -> this->stream): the stream the staging copies go to
    int stage_sets(Engine* e, const std::vector<SetRef>& sets, bool transient, hipStream_t st = nullptr);
    int stage_level(const std::vector<PairSpec>& pairs, int precision, int32_t angle_begin, int32_t angle_end,
                    bool want_costs, hipStream_t st = nullptr);
    int run(bool screen_only);
    int mark_search_done();   // resident plans: record Engine::search_done behind the dominant kernel
    int fetch(BatchResult& out, double* all_costs_plan_order);
    size_t hbm_bytes() const { return pts_bytes + lvl_bytes; }
    std::vector<int32_t> pair_slice_end;      // per pair: end of the candidate slice this plan owns
    int32_t slice_hi(int p) const { return pair_slice_end[(size_t)p]; }
    double angle_of(int p, int32_t idx) const;
    ~Plan();
};

// One-shot batch on the engine's transient buffers (build, run, fetch).
int run_batch(Engine* e, const std::vector<SetRef>& sets, const std::vector<PairSpec>& pairs, int precision,
              BatchResult& out);

// hausdorff_distance of set pairs (indices into `sets`), exact f64 on the device; shared sets are
// staged once.  out[pairs.size()].
int hausdorff_sets(Engine* e, const std::vector<SetRef>& sets, const std::vector<std::array<int32_t, 2>>& pairs,
                   double* out);

// First index of minimal hausdorff_distance over the pairs (strict '<' in pair order) and its value, with
// lower bounds ruling pairs out where every pair runs on the streaming kernel; n_exact = pairs evaluated.
int hausdorff_sets_first_min(Engine* e, const std::vector<SetRef>& sets, const std::vector<std::array<int32_t, 2>>& pairs,
                             int32_t* best, double* best_cost, int64_t* n_exact);

}  // namespace mm
