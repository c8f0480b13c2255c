// mm_host.cpp -- host orchestration above the device search: candidate enumeration,
// frame transforms, set construction, the within-pullback chain and the between-pullback
// alignment.  Everything here is exact f64 host arithmetic in the reference's operation
// order (compiled with -ffp-contract=off); the only device work is the batched search
// (mm_best_rotation_batch).  Reference lines are cited per function
// (paths relative to the reference checkout, yungselm/multimoda-rs).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "mm_engine.h"
#include "mm_pool.h"
#include "mm_trace.h"

#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>

namespace mm {

// cos and sin of the same angle: the reference computes both in one function, which LLVM on
// x86_64-linux-gnu lowers to one glibc `sincos` call; sincos is not bit-identical to separate
// sin()/cos() for every argument, so the pair is requested explicitly ("sincos").
static inline void sin_cos(double x, double& s, double& c) { ::sincos(x, &s, &c); }

static constexpr double kPi = 3.14159265358979323846264338327950288;

static inline double deg2rad(double d) { return d * (kPi / 180.0); }  // f64::to_radians
static inline double rad2deg(double r) { return r * (180.0 / kPi); }  // f64::to_degrees

static inline double wrap_pi(double a)  // ((a + PI).rem_euclid(2 PI)) - PI, process_utils.rs:66
{
    double r = std::fmod(a + kPi, 2.0 * kPi);
    if (r < 0.0) r += std::fabs(2.0 * kPi);
    return r - kPi;
}

// set by enumerate_angles when a candidate list would exceed 2^24 entries; the entry points turn it into an error
static thread_local bool tl_enum_overflow = false;
static int enum_overflow_error()
{
    if (!tl_enum_overflow) return MM_OK;
    tl_enum_overflow = false;
    return set_error(MM_ERR_TOO_LARGE, "more than 2^24 candidates for one search (step far below the range)");
}

// process_utils.rs:43-67.  Returns false when the reference returns early (`early`).
static bool enumerate_angles(double step_deg, double range_deg, bool has_center, double center_in,
                             double limes_deg, std::vector<double>& out, double& early)
{
    out.clear();
    const double range_rad = deg2rad(range_deg);
    const double step_rad = deg2rad(step_deg);
    if (step_rad <= 0.0) { early = has_center ? center_in : 0.0; return false; }
    const double center = has_center ? center_in : 0.0;
    const double limes = deg2rad(limes_deg);
    const double start = std::fmax(center - range_rad, -limes);
    const double stop = std::fmin(center + range_rad, limes);
    if (stop <= start) { early = center; return false; }
    const double sf = std::ceil((stop - start) / step_rad);
    // `as usize` saturates (NaN -> 0).  More than 2^24 candidates for ONE search (a step of 2e-5 degrees over +-180) is
    // refused: the reference would allocate the list or, below the spacing of the doubles, never terminate
    if (sf >= 16777216.0) { tl_enum_overflow = true; early = center; return false; }
    size_t steps = !(sf > 0.0) ? 0 : (size_t)sf;
    steps = std::max<size_t>(steps, 1);
    out.reserve(steps + 1);
    for (size_t i = 0; i <= steps; ++i) {
        const double a = start + (double)i * step_rad;
        if (!(a <= stop)) break;
        out.push_back(wrap_pi(a));
    }
    early = center;
    return true;
}

// One level of the coarse->fine ladder (align_within.rs:208-246, align_between.rs:219-257).
struct Level { double step, range; };

static std::vector<Level> search_levels(double step_deg, double range_deg, bool bruteforce)
{
    std::vector<Level> lv;
    if (bruteforce || (step_deg >= 1.0 && step_deg <= INFINITY)) {
        lv.push_back({step_deg, range_deg});
        return lv;
    }
    const double r5 = range_deg > 5.0 ? 5.0 : range_deg;
    lv.push_back({1.0, range_deg});
    if (step_deg >= 0.1 && step_deg < 1.0) {
        lv.push_back({step_deg, r5});
    } else if (step_deg >= 0.01 && step_deg < 0.1) {
        lv.push_back({0.1, r5});
        lv.push_back({step_deg, range_deg > 10.0 * step_deg ? 10.0 * step_deg : range_deg});
    } else {  // includes NaN, like Rust's `_` arm
        lv.push_back({0.1, r5});
        lv.push_back({0.01, range_deg > 0.1 ? 0.1 : range_deg});
        lv.push_back({step_deg, range_deg > 10.0 * step_deg ? 10.0 * step_deg : range_deg});
    }
    return lv;
}

// -------------------------------------------------------------------------------------
// point helpers on flat geometries (xyz triples)
// -------------------------------------------------------------------------------------
static inline void rotate_xy(double& x, double& y, double angle, double cx, double cy)
{
    // contour_point.rs:38-52
    if (angle == 0.0) return;
    const double rx = x - cx, ry = y - cy;
    double co, si;
    sin_cos(angle, si, co);
    x = rx * co - ry * si + cx;
    y = rx * si + ry * co + cy;
}

static void span_translate(double* p, int64_t lo, int64_t hi, double dx, double dy, double dz)
{
    for (int64_t k = lo; k < hi; ++k) { p[3 * k] += dx; p[3 * k + 1] += dy; p[3 * k + 2] += dz; }
}

static void span_rotate(double* p, int64_t lo, int64_t hi, double angle, double cx, double cy)
{
    // contour_point.rs:38-52 applied to a span; the reference evaluates cos/sin per point, with
    // the same argument every time -- hoisted here (identical values, ~40 ns saved per point)
    if (angle == 0.0) return;
    double co, si;
    sin_cos(angle, si, co);
    for (int64_t k = lo; k < hi; ++k) {
        const double rx = p[3 * k] - cx, ry = p[3 * k + 1] - cy;
        p[3 * k] = rx * co - ry * si + cx;
        p[3 * k + 1] = rx * si + ry * co + cy;
    }
}

// contour.rs:47-58: evenly strided subset; appends (x,y) to the SoA vectors
static void downsample_append(const double* pts, int64_t len, int64_t n, std::vector<double>& ox,
                              std::vector<double>& oy)
{
    if (len <= n) {
        for (int64_t i = 0; i < len; ++i) { ox.push_back(pts[3 * i]); oy.push_back(pts[3 * i + 1]); }
        return;
    }
    const double stride = (double)len / (double)n;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t idx = (int64_t)((double)i * stride);
        ox.push_back(pts[3 * idx]); oy.push_back(pts[3 * idx + 1]);
    }
}

struct SampleSpec { int64_t lumen; bool has_cath; int64_t cath; };

// align_within.rs:45-59
static SampleSpec sample_spec(const mm_geometry* g, int64_t sample_size)
{
    const int64_t len0 = g->lumen_off[1] - g->lumen_off[0];
    const double ratio = (double)sample_size / (double)len0;
    SampleSpec s{sample_size, false, 0};
    if (g->has_catheter && g->cath_off) {
        const int64_t c0 = g->cath_off[1] - g->cath_off[0];
        s.has_cath = true;
        const double cs = std::ceil((double)c0 * ratio);                 // finite: len0 > 0 is checked by every caller
        s.cath = !(cs > 0.0) ? 0 : (cs >= 9.2e18 ? INT64_MAX : (int64_t)cs);
    }
    return s;
}

// align_within.rs:173-191
static void frame_search_set(const mm_geometry* g, int32_t i, const SampleSpec& s, std::vector<double>& ox,
                             std::vector<double>& oy)
{
    downsample_append(g->lumen + 3 * g->lumen_off[i], g->lumen_off[i + 1] - g->lumen_off[i], s.lumen, ox, oy);
    if (s.has_cath && g->cath_off)
        downsample_append(g->cath + 3 * g->cath_off[i], g->cath_off[i + 1] - g->cath_off[i], s.cath, ox, oy);
}

// geometry.rs:42-69
static size_t ref_or_proximal(const mm_geometry* g)
{
    for (int32_t i = 0; i < g->n_frames; ++i)
        if (g->has_ref && g->has_ref[i]) return (size_t)g->id[i];
    const int32_t n = g->n_frames;
    if (n == 0) return 0;
    if (n == 1) return (size_t)g->lumen_id[0];
    return (size_t)((g->orig_frame[0] > g->orig_frame[n - 1]) ? g->lumen_id[0] : g->lumen_id[n - 1]);
}

// -------------------------------------------------------------------------------------
// A batch of independent searches that advance level by level: every level is one device
// launch sequence over all still-active searches.
// -------------------------------------------------------------------------------------
struct SearchJob {
    std::vector<double> rx, ry, tx, ty;
    double cx = 0.0, cy = 0.0;
    int32_t flags = 0;
    double result = 0.0;  // chosen angle (radians)
};

static int run_searches(Engine* e, std::vector<SearchJob>& jobs, double step_deg, double range_deg,
                        bool bruteforce, int precision, int64_t* pose_evals)
{
    const std::vector<Level> levels = search_levels(step_deg, range_deg, bruteforce);
    const int J = (int)jobs.size();
    if (J == 0) return MM_OK;
    std::vector<double> centre(J, 0.0);
    std::vector<std::vector<double>> lists(J);
    for (size_t l = 0; l < levels.size(); ++l) {
        std::vector<int> active;
        for (int j = 0; j < J; ++j) {
            double early = 0.0;
            const bool ok = enumerate_angles(levels[l].step, levels[l].range, l > 0, centre[j], range_deg,
                                             lists[j], early);
            if (int erc = enum_overflow_error()) return erc;
            if (!ok) { centre[j] = early; continue; }  // search_range returned early
            active.push_back(j);
        }
        if (active.empty()) continue;
        std::vector<SetRef> sets;
        std::vector<PairSpec> pairs;
        for (int j : active) {
            const SearchJob& sj = jobs[j];
            const int32_t sid = (int32_t)sets.size();
            sets.push_back(SetRef{sj.rx.data(), sj.ry.data(), (int32_t)sj.rx.size(), sj.cx, sj.cy});
            sets.push_back(SetRef{sj.tx.data(), sj.ty.data(), (int32_t)sj.tx.size(), sj.cx, sj.cy});
            pairs.push_back(PairSpec{sid, sid + 1, sj.cx, sj.cy, sj.flags, lists[j].data(), (int32_t)lists[j].size(), 0.0, 0.0});
            if (pose_evals) *pose_evals += (int64_t)lists[j].size();
        }
        BatchResult res;
        int rc = run_batch(e, sets, pairs, precision, res);
        if (rc) return rc;
        for (size_t k = 0; k < active.size(); ++k) centre[active[k]] = lists[active[k]][res.best_idx[k]];
    }
    for (int j = 0; j < J; ++j) jobs[j].result = centre[j];
    return MM_OK;
}

// -------------------------------------------------------------------------------------
// Decoupled within-pullback search ("screen all pairs at once, then walk the chain").
//
// In exact arithmetic the cost of step i does not depend on the chain state: frame i-1
// and frame i have both been rotated by the same cumulative angle about a common centre
// (align_within.rs:79-90), and a rigid motion preserves the Hausdorff distance.  So every
// (frame pair, candidate) of every pullback can be scored in ONE launch on the original
// frames, each centred on its own centroid.  In floating point the chain-state cost and
// the decoupled cost differ by rounding only; we bound that difference by
// eps = 2^-42 * scale (scale = largest coordinate magnitude involved, >= 40x the worst
// case of the handful of roundings involved) and keep the reference's result exactly:
//   * screening keeps every candidate within 2*(delta_f32 + eps) of the f32 minimum,
//   * those are re-scored in f64 on the decoupled sets,
//   * a pair is RESOLVED if exactly one candidate is within 2*eps of the f64 minimum (or
//     all such candidates are the same angle value, e.g. the duplicated -pi when
//     range == limes == 180 deg: equal angles give equal chain costs and the lowest index
//     wins, process_utils.rs:72);
//   * anything else is searched again on the chain state in the walk below, exactly as
//     mode 0 does.
// The walk itself (pre-rotation, translation, rotation, logs) is the reference's.
// -------------------------------------------------------------------------------------
struct WithinPlan {
    Engine* e = nullptr;
    int n_geoms = 0;
    std::vector<mm_geometry*> geoms;
    double step_deg = 0, range_deg = 0;
    bool bruteforce = false;
    int64_t sample_size = 0;
    int precision = MM_PRECISION_F32;
    std::vector<SampleSpec> spec;
    int32_t max_frames = 0;
    // decoupled stage
    std::vector<std::vector<double>> sx, sy;  // centred search sets, one per (geom, frame)
    std::vector<int32_t> set_base;            // first set id of each geometry
    std::vector<double> eps;                  // per geometry
    std::vector<int> job_geom, job_frame;     // job j = (geometry, frame i >= 1)
    std::vector<int32_t> job_base;            // first job of each geometry
    std::vector<Level> levels;
    std::vector<double> level0;               // shared candidate list of level 0
    bool level0_ok = false; double level0_early = 0.0;
    Plan plan;
    bool staged = false, level0_staged = false;
    std::vector<PairSpec> lvl_pairs;
    std::vector<int> lvl_active;
    std::vector<std::vector<double>> lists;   // per-job candidate lists of levels >= 1

    // search state (persists across the level_local / commit calls)
    std::vector<double> centre;
    std::vector<uint8_t> resolved;
    std::vector<int64_t> evals;
    // this plan's tile of the (frame pair x candidate) grid: world = grid_p * grid_c ranks, rank = pair block
    // rank / grid_c, candidate slice rank % grid_c (grid_p = 1: the pure candidate-axis split)
    int rank = 0, world = 1, grid_p = 1, grid_c = 1;
    // what build_sets_device staged: every frame, or (a plan created on a tile with several pair blocks) only the frames
    // block sets_pb of sets_gp reads -- the other sets are empty and their pairs trivial with an empty slice
    bool sets_partial = false;
    int sets_pb = 0, sets_gp = 1;
    int64_t staged_points = 0;
    bool owns(int j) const
    {
        const int64_t J = (int64_t)job_geom.size(), pb = rank / grid_c;
        return (int64_t)j >= J * pb / grid_p && (int64_t)j < J * (pb + 1) / grid_p;
    }
    bool searched = false;
    int launched_level = -1;          // level most recently enqueued by level_launch
    // exchange records of search_sharded (device; from the engine's blob cache)
    unsigned char* d_xrec = nullptr; size_t xrec_cap = 0;
    // device-side exchange (level_launch / export_* / commit_dev): job -> pair of the current level
    int32_t* d_pair_of_job = nullptr;
    int pof_level = -1;               // level whose map d_pair_of_job holds

    size_t pof_cap = 0;
    ~WithinPlan()
    {   // not hipFree: it waits for the whole device
        if (d_pair_of_job) e->blob_release(d_pair_of_job, pof_cap);
        if (d_xrec) e->blob_release(d_xrec, xrec_cap);
    }
    int upload_pair_of_job(size_t l);
    int search_sharded(Comm* c);
    int search_sharded_begin(Comm* c);
    int search_sharded_end(Comm* c);
    int prepare();
    int build_sets_host(int32_t n_sets);
    int build_sets_device(int32_t n_sets);
    int restage_sets();
    int level_launch(size_t l);
    int level_local(size_t l, double* cost, int32_t* uniform, double* angle, int32_t* idx, int32_t* active);
    int level_collect(size_t l, double* cost, int32_t* uniform, double* angle, int32_t* idx, int32_t* active);
    int level_commit(size_t l, const uint8_t* ok, const double* angle);
    int level_export_cost(size_t l, double* cost_dev);
    int level_export_keys(size_t l, const double* gcost_dev, long long* keys_dev);
    int level_commit_dev(size_t l, const double* gcost_dev, const long long* keys_dev);
    int level_copy_records(const double* gcost_dev, const long long* keys_dev);
    int level_commit_records(size_t l);
    int exchange_enqueue(Comm* c, size_t l);
    bool xchg_pending = false;        // search_sharded_begin has enqueued level 0's exchange
    bool rehearsal = false;           // mm_within_plan_set_timing_rehearsal: timing only, the result is not an alignment
    void build_level_pairs(size_t l, const std::vector<double>& centres, const std::vector<uint8_t>& take,
                           std::vector<PairSpec>& pairs, std::vector<int>& active, std::vector<double>* centre_out);
    int search();
    int walk(mm_alignlog** logs, int64_t* pose_evals, int64_t* n_unresolved);
    std::vector<uint8_t> walk_take;   // mm_within_plan_walk_geoms: the pullbacks to walk (empty: all)
    bool taken(int g) const { return walk_take.empty() || walk_take[(size_t)g] != 0; }
};

// The search sets built on the HOST and uploaded as four planes (the path of round 1; kept behind
// MM_HOST_SETS=1 as the checker of build_sets_device: tests compare the two pools bit for bit).
int WithinPlan::build_sets_host(int32_t n_sets)
{
    sx.assign((size_t)n_sets, {}); sy.assign((size_t)n_sets, {});
    std::vector<double> set_scale((size_t)n_sets, 0.0);
    std::vector<int> set_geom((size_t)n_sets);
    for (int g = 0; g < n_geoms; ++g)
        for (int32_t i = 0; i < geoms[g]->n_frames; ++i) set_geom[(size_t)(set_base[g] + i)] = g;
    {
        TraceTimer t_sets("prepare: search sets (host)");
        parallel_for(n_sets, [&](int s) {           // sets are independent: build them over the worker pool
            const int g = set_geom[(size_t)s];
            const int32_t i = s - set_base[g];
            const mm_geometry* G = geoms[g];
            std::vector<double>&x = sx[(size_t)s], &y = sy[(size_t)s];
            frame_search_set(G, i, spec[g], x, y);
            const double cx = G->centroid[3 * i], cy = G->centroid[3 * i + 1];
            double scale = 0.0;
            for (size_t k = 0; k < x.size(); ++k) {
                scale = std::max(scale, std::max(std::fabs(x[k]), std::fabs(y[k])));
                x[k] -= cx; y[k] -= cy;
                scale = std::max(scale, std::max(std::fabs(x[k]), std::fabs(y[k])));
            }
            set_scale[(size_t)s] = scale;
        });
    }
    for (int g = 0; g < n_geoms; ++g) {
        double scale = 0.0;
        for (int32_t i = 0; i < geoms[g]->n_frames; ++i) scale = std::max(scale, set_scale[(size_t)(set_base[g] + i)]);
        // chain-state coordinates stay within |frame-0 centroid| + radius: 4x covers it amply
        eps[g] = std::ldexp(4.0 * scale + 1.0, -42);
    }
    std::vector<SetRef> sets(sx.size());
    for (size_t s = 0; s < sx.size(); ++s) sets[s] = SetRef{sx[s].data(), sy[s].data(), (int32_t)sx[s].size(), 0.0, 0.0};
    TraceTimer t("prepare: stage sets (host)");
    return plan.stage_sets(e, sets, /*transient=*/false, e->aux);   // staging: the high-priority side stream
}

// The search sets built on the DEVICE: the raw contours (xyz f64, as the caller holds them) go up once through
// pinned staging, k_build_sets writes the four planes of the pool, and 16 bytes per set (rho^2, scale) come back.
// Everything on the engine's high-priority side stream, so staging the next case does not wait behind another
// engine's search.
int WithinPlan::build_sets_device(int32_t n_sets)
{
    TraceTimer t_all("prepare: sets on the device");
    std::vector<SetSrc> src((size_t)n_sets);
    std::vector<int32_t> lens((size_t)n_sets);
    std::vector<int64_t> lum_base(n_geoms), cath_base(n_geoms);
    // The frames this rank's tile reads: frames i-1 and i of every job (g, i) of its pair block -- a contiguous run of
    // frames per geometry.  With one pair block (and on a plan that may still be re-tiled) that is every frame.
    std::vector<int32_t> f0(n_geoms, 0), f1(n_geoms, -1);
    sets_partial = grid_p > 1; sets_gp = grid_p; sets_pb = rank / grid_c;
    if (sets_partial) {
        for (int g = 0; g < n_geoms; ++g) f0[g] = INT32_MAX;
        for (int j = 0; j < (int)job_geom.size(); ++j)
            if (owns(j)) {
                const int g = job_geom[j];
                f0[g] = std::min(f0[g], job_frame[j] - 1); f1[g] = std::max(f1[g], job_frame[j]);
            }
    } else {
        for (int g = 0; g < n_geoms; ++g) f1[g] = geoms[g]->n_frames - 1;
    }
    int64_t raw_pts = 0;
    for (int g = 0; g < n_geoms; ++g) {       // the bases are those of frame 0, so that base + off[i] addresses frame i
        const mm_geometry* G = geoms[g];
        const bool hc = spec[g].has_cath && G->cath_off;
        lum_base[g] = cath_base[g] = 0;
        if (f1[g] < f0[g]) continue;
        lum_base[g] = raw_pts - G->lumen_off[f0[g]]; raw_pts += G->lumen_off[f1[g] + 1] - G->lumen_off[f0[g]];
        if (hc) { cath_base[g] = raw_pts - G->cath_off[f0[g]]; raw_pts += G->cath_off[f1[g] + 1] - G->cath_off[f0[g]]; }
    }
    for (int g = 0; g < n_geoms; ++g) {
        const mm_geometry* G = geoms[g];
        const bool hc = spec[g].has_cath && G->cath_off;
        for (int32_t i = 0; i < G->n_frames; ++i) {
            SetSrc& d = src[(size_t)(set_base[g] + i)];
            const int64_t ll = G->lumen_off[i + 1] - G->lumen_off[i], cl = hc ? G->cath_off[i + 1] - G->cath_off[i] : 0;
            if (ll < 0 || cl < 0 || ll > INT32_MAX || cl > INT32_MAX || spec[g].lumen > INT32_MAX || spec[g].cath > INT32_MAX)
                return set_error(MM_ERR_INVALID, "contour too long");
            if (i < f0[g] || i > f1[g]) {       // not read by this tile: an empty set
                d = SetSrc{};
                d.lum_take = d.cath_take = 1;
                lens[(size_t)(set_base[g] + i)] = 0;
                continue;
            }
            d.lum_at = lum_base[g] + G->lumen_off[i];
            d.cath_at = hc ? cath_base[g] + G->cath_off[i] : 0;
            d.lum_len = (int32_t)ll; d.lum_take = (int32_t)spec[g].lumen;
            d.cath_len = (int32_t)cl; d.cath_take = hc ? (int32_t)std::max<int64_t>(spec[g].cath, 0) : 0;
            d.n = std::min(d.lum_len, d.lum_take) + std::min(d.cath_len, d.cath_take);
            d.cx = G->centroid[3 * i]; d.cy = G->centroid[3 * i + 1];
            lens[(size_t)(set_base[g] + i)] = d.n;
        }
    }
    staged_points = raw_pts;
    TraceTimer t_alloc("prepare:   descriptors + pool");
    int rc = plan.alloc_pool(e, lens, /*transient=*/false);
    if (rc) return rc;
    for (int32_t s = 0; s < n_sets; ++s) src[(size_t)s].dst_off = plan.set_off[(size_t)s];
    t_alloc.stop();

    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t raw_bytes = (size_t)raw_pts * 24, o_src = up(raw_bytes), o_out = up(o_src + (size_t)n_sets * sizeof(SetSrc));
    const size_t total = up(o_out + (size_t)n_sets * 16);
    if ((rc = e->ensure(e->host_pts, total, true))) return rc;
    if ((rc = e->ensure(e->dev_raw, total, false))) return rc;
    unsigned char* h = (unsigned char*)e->host_pts.p;
    unsigned char* d = (unsigned char*)e->dev_raw.p;
    {
        TraceTimer t_cp("prepare:   pageable -> pinned");
        // pageable -> pinned, 1 MiB pieces over the worker pool (the runtime's own pageable path is one thread)
        struct Piece { const double* from; size_t at, bytes; };
        std::vector<Piece> pieces;
        auto add = [&](const double* p, int64_t at_pts, int64_t n_pts) {
            const size_t bytes = (size_t)n_pts * 24, step = (size_t)1 << 20;
            for (size_t o = 0; o < bytes; o += step)
                pieces.push_back(Piece{(const double*)((const unsigned char*)p + o), (size_t)at_pts * 24 + o, std::min(step, bytes - o)});
        };
        for (int g = 0; g < n_geoms; ++g) {
            const mm_geometry* G = geoms[g];
            if (f1[g] < f0[g]) continue;
            const int64_t l0 = G->lumen_off[f0[g]], l1 = G->lumen_off[f1[g] + 1];
            add(G->lumen + 3 * l0, lum_base[g] + l0, l1 - l0);
            if (spec[g].has_cath && G->cath_off) {
                const int64_t c0 = G->cath_off[f0[g]], c1 = G->cath_off[f1[g] + 1];
                add(G->cath + 3 * c0, cath_base[g] + c0, c1 - c0);
            }
        }
        parallel_for((int)pieces.size(), [&](int k) { std::memcpy(h + pieces[(size_t)k].at, pieces[(size_t)k].from, pieces[(size_t)k].bytes); });
        std::memcpy(h + o_src, src.data(), (size_t)n_sets * sizeof(SetSrc));
    }
    TraceTimer t_dev("prepare:   H2D + build kernel + sync");
    hipStream_t st = e->aux;
    hipError_t he = hipMemcpyAsync(d, h, o_out, hipMemcpyHostToDevice, st);
    if (he != hipSuccess) return hip_error(he, "raw pullbacks H2D");
    double *d_rho2 = (double*)(d + o_out), *d_scale = d_rho2 + n_sets;
    he = launch_build_sets((const SetSrc*)(d + o_src), n_sets, (const double*)d, (float*)(plan.pts_blob + plan.o32x),
                           (float*)(plan.pts_blob + plan.o32y), (double*)(plan.pts_blob + plan.o64x),
                           (double*)(plan.pts_blob + plan.o64y), d_rho2, d_scale, st);
    if (he != hipSuccess) return hip_error(he, "build_sets kernel launch");
    he = launch_copy_small(h + o_out, d + o_out, (size_t)n_sets * 16, st);   // not hipMemcpyAsync: see k_copy_small
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess) return hip_error(he, "build_sets");
    const double *h_rho2 = (const double*)(h + o_out), *h_scale = h_rho2 + n_sets;
    for (int32_t s = 0; s < n_sets; ++s) plan.set_rho[(size_t)s] = std::sqrt(h_rho2[s]) * (1.0 + 1e-12);
    for (int g = 0; g < n_geoms; ++g) {
        double scale = 0.0, rho = 0.0;
        for (int32_t i = 0; i < geoms[g]->n_frames; ++i) {
            scale = std::max(scale, h_scale[set_base[g] + i]);
            rho = std::max(rho, plan.set_rho[(size_t)(set_base[g] + i)]);
        }
        // chain-state coordinates stay within |frame-0 centroid| + radius.  A tile that staged only its own frames has
        // not seen frame 0: it adds that bound explicitly (the ranks of one pair block hold the same frames and so the
        // same eps; a job's tolerance is its owners', mm_within_plan_dims reports 0 for the others)
        if (sets_partial)
            scale = std::max(scale, std::max(std::fabs(geoms[g]->centroid[0]), std::fabs(geoms[g]->centroid[1])) + rho);
        eps[g] = std::ldexp(4.0 * scale + 1.0, -42);
    }
    return MM_OK;
}

// The sets again, for the tile the plan has been moved to (mm_within_plan_set_shard_grid on a plan that staged one pair
// block's frames).
int WithinPlan::restage_sets()
{
    hipError_t he = hipStreamSynchronize(e->aux);
    if (he == hipSuccess) he = hipStreamSynchronize(plan.stream);
    if (he != hipSuccess) return hip_error(he, "restage_sets");
    if (plan.pts_blob && plan.own_pts) { e->blob_release(plan.pts_blob, plan.pts_cap); plan.pts_blob = nullptr; }
    return build_sets_device(set_base[n_geoms]);
}

int WithinPlan::prepare()
{
    spec.resize(n_geoms);
    max_frames = 0;
    for (int g = 0; g < n_geoms; ++g) {
        const mm_geometry* G = geoms[g];
        if (!G || G->n_frames <= 0) return set_error(MM_ERR_NO_FRAMES, "Geometry contains no frames");
        if (G->lumen_off[1] - G->lumen_off[0] <= 0) return set_error(MM_ERR_NO_POINTS, "Lumen contours have no points");
        if (sample_size <= 0) return set_error(MM_ERR_SAMPLE_SIZE, "sample_size must be > 0");
        spec[g] = sample_spec(G, sample_size);
        max_frames = std::max(max_frames, G->n_frames);
    }
    levels = search_levels(step_deg, range_deg, bruteforce);
    level0_ok = enumerate_angles(levels[0].step, levels[0].range, false, 0.0, range_deg, level0, level0_early);
    if (int erc = enum_overflow_error()) return erc;

    // centred search sets of the ORIGINAL frames + per-geometry rounding scale
    set_base.assign(n_geoms + 1, 0); job_base.assign(n_geoms + 1, 0);
    job_geom.clear(); job_frame.clear();
    eps.assign(n_geoms, 0.0);
    int32_t n_sets = 0;
    for (int g = 0; g < n_geoms; ++g) {
        set_base[g] = n_sets;
        job_base[g] = (int32_t)job_geom.size();
        for (int32_t i = 1; i < geoms[g]->n_frames; ++i) { job_geom.push_back(g); job_frame.push_back(i); }
        n_sets += geoms[g]->n_frames;
    }
    set_base[n_geoms] = n_sets;
    job_base[n_geoms] = (int32_t)job_geom.size();
    static const bool host_sets = std::getenv("MM_HOST_SETS") != nullptr;   // the previous host-side construction (checker)
    int rc = host_sets ? build_sets_host(n_sets) : build_sets_device(n_sets);
    if (rc) return rc;
    // level 0 has no centre: its candidate list, descriptors and tables are known now
    if (level0_ok) {
        TraceTimer t("prepare: stage level 0");
        build_level_pairs(0, std::vector<double>(), std::vector<uint8_t>(job_geom.size(), 1), lvl_pairs, lvl_active, nullptr);
        if ((rc = plan.stage_level(lvl_pairs, precision, 0, INT32_MAX, false, e->aux))) return rc;
        level0_staged = true;
    }
    staged = true;
    return MM_OK;
}

// PairSpecs of one level for the jobs that are still resolved; `lists` holds per-job
// candidate lists for levels >= 1 (level 0 shares one list).
void WithinPlan::build_level_pairs(size_t l, const std::vector<double>& centres, const std::vector<uint8_t>& take,
                                   std::vector<PairSpec>& pairs, std::vector<int>& active, std::vector<double>* centre_out)
{
    const int J = (int)job_geom.size();
    pairs.clear(); active.clear();
    if (l > 0 && (int)lists.size() != J) lists.assign(J, std::vector<double>());
    for (int j = 0; j < J; ++j) {
        if (!take[j]) continue;
        const double* lp; int32_t ln;
        if (l == 0) {
            if (!level0_ok) { if (centre_out) (*centre_out)[j] = level0_early; continue; }
            lp = level0.data(); ln = (int32_t)level0.size();
        } else {
            double early = 0.0;
            const bool ok = enumerate_angles(levels[l].step, levels[l].range, true, centres[j], range_deg, lists[j], early);
            if (!ok) {                              // (an overflow leaves tl_enum_overflow set: the callers check it)
                if (centre_out) (*centre_out)[j] = early;
                continue;
            }
            lp = lists[j].data(); ln = (int32_t)lists[j].size();
        }
        const int g = job_geom[j], i = job_frame[j];
        const int32_t sid = set_base[g] + i;
        PairSpec sp{sid - 1, sid, 0.0, 0.0, MM_SEARCH_SKIP_ZERO, lp, ln, 2.0 * eps[g], eps[g]};
        // this rank's tile: the jobs of its pair block, and of their lists its slice of the candidate axis; a job of
        // another block keeps its place in the level (every rank commits every job) with an empty slice
        const int cs = rank % grid_c;
        const bool mine = owns(j);
        sp.slice_begin = mine ? (int32_t)((int64_t)ln * cs / grid_c) : 0;
        sp.slice_end = mine ? (int32_t)((int64_t)ln * (cs + 1) / grid_c) : 0;
        pairs.push_back(sp);
        active.push_back(j);
    }
}

// One level over this rank's candidate slice.  Per job (n_jobs entries): active = the job
// takes part in this level; cost/idx/angle = exact first minimum inside the slice (+inf/-1
// if the slice is empty); uniform = every candidate of the slice within the tie tolerance
// of that minimum is the same angle value.
// Stage (if needed) and enqueue level l over this rank's candidate slice; results stay in HBM.
int WithinPlan::level_launch(size_t l)
{
    const int J = (int)job_geom.size();
    if (l == 0) { centre.assign(J, 0.0); resolved.assign(J, 1); evals.assign(J, 0); searched = true; }
    int rc;
    if (!(l == 0 && level0_staged)) {
        TraceTimer t("within: stage level");
        build_level_pairs(l, centre, resolved, lvl_pairs, lvl_active, &centre);
        if ((rc = enum_overflow_error())) return rc;
        if (lvl_active.empty()) return MM_OK;
        if ((rc = plan.stage_level(lvl_pairs, precision, 0, INT32_MAX, false))) return rc;
    }
    if (lvl_active.empty()) return MM_OK;
    if (world > 1)     // job -> pair map of the level for the device-side exchange, uploaded ahead of the search
        if ((rc = upload_pair_of_job(l))) return rc;
    launched_level = (int)l;
    return plan.run(false);
}

// The job -> pair map goes up from PINNED memory (the engine's, grow-only): a copy from a pageable vector is staged
// synchronously by the runtime, and on a stream that waits for another engine's launch (the look-ahead of a pipelined
// driver) that would hold the host for the whole launch.  The buffer is free again at the level's synchronisation.
int WithinPlan::upload_pair_of_job(size_t l)
{
    const int J = (int)job_geom.size();
    // (several plans may share an engine -- shard plans driven in lockstep by a test: wait for the previous upload)
    if (e->pof_busy) { (void)hipEventSynchronize(e->pof_done); e->pof_busy = false; }
    if (int rc = e->ensure(e->host_pof, (size_t)std::max(J, 1) * 4, true)) return rc;
    int32_t* h = (int32_t*)e->host_pof.p;
    for (int j = 0; j < J; ++j) h[j] = -1;
    for (size_t k = 0; k < lvl_active.size(); ++k) h[(size_t)lvl_active[k]] = (int32_t)k;
    if (!d_pair_of_job) {
        if (int arc = e->blob_alloc((void**)&d_pair_of_job, (size_t)std::max(J, 1) * 4, &pof_cap)) return arc;
    }
    // (the 256-thread copy kernel, not hipMemcpyAsync: behind a stream wait the runtime's copy is a 512-thread blit)
    const hipError_t hc = launch_copy_small(d_pair_of_job, h, (size_t)J * 4, plan.stream);
    if (hc != hipSuccess) return hip_error(hc, "copy(pair_of_job)");
    if (!e->pof_done && hipEventCreateWithFlags(&e->pof_done, hipEventDisableTiming) != hipSuccess) e->pof_done = nullptr;
    if (e->pof_done && hipEventRecord(e->pof_done, plan.stream) == hipSuccess) e->pof_busy = true;
    else (void)hipStreamSynchronize(plan.stream);
    pof_level = (int)l;
    return MM_OK;
}

int WithinPlan::level_local(size_t l, double* cost, int32_t* uniform, double* angle, int32_t* idx, int32_t* active)
{
    int rc;
    {
        TraceTimer t("within: search kernels (launch)");
        if ((rc = level_launch(l))) return rc;
    }
    return level_collect(l, cost, uniform, angle, idx, active);
}

// The fetch half of level_local: waits for the level enqueued by level_launch and fills the per-job records.
int WithinPlan::level_collect(size_t l, double* cost, int32_t* uniform, double* angle, int32_t* idx, int32_t* active)
{
    (void)l;
    if (!searched) return set_error(MM_ERR_INVALID, "level_collect before level_launch");
    const int J = (int)job_geom.size();
    for (int j = 0; j < J; ++j) { cost[j] = INFINITY; uniform[j] = 1; angle[j] = 0.0; idx[j] = -1; active[j] = 0; }
    int rc;
    BatchResult res;
    {
        TraceTimer t("within: search kernels (wait + fetch)");
        if (lvl_active.empty()) return MM_OK;
        if ((rc = plan.fetch(res, nullptr))) return rc;
    }
    for (size_t k = 0; k < lvl_active.size(); ++k) {
        const int j = lvl_active[k];
        const double* lp = lvl_pairs[k].angles;
        active[j] = 1;
        if (res.best_idx[k] < 0) continue;  // empty slice
        cost[j] = res.best_cost[k]; idx[j] = res.best_idx[k]; angle[j] = lp[res.best_idx[k]];
        const int32_t n = res.near_cnt[k];
        bool ok = (n == 1);
        if (!ok && n >= 2 && n <= kMaxNear) {
            ok = true;
            const double a0 = lp[res.near_idx[k * kMaxNear]];
            for (int q = 1; q < n; ++q) ok = ok && (std::memcmp(&a0, &lp[res.near_idx[k * kMaxNear + q]], 8) == 0);
        }
        uniform[j] = ok ? 1 : 0;
    }
    return MM_OK;
}

// ---- device-side exchange -------------------------------------------------------------
// After level_launch the per-pair results of this rank's slice are in HBM.  export_cost / export_keys
// write job-indexed exchange records into caller-owned DEVICE buffers (kernels on the plan's stream), the
// caller all-reduces them over the ranks (RCCL), and commit_dev copies the two reduced records to the
// host once and commits the level.  No host round trip between the search and the collectives.
int WithinPlan::level_export_cost(size_t l, double* cost_dev)
{
    (void)l;
    if (!searched) return set_error(MM_ERR_INVALID, "level_export_cost before level_launch");
    const int J = (int)job_geom.size();
    if (J == 0) return MM_OK;
    if (pof_level != (int)l || lvl_active.empty())     // world == 1 (tests), or a level without active jobs
        if (int rc = upload_pair_of_job(l)) return rc;
    const hipError_t he = launch_export_cost(plan.dev, d_pair_of_job, J, cost_dev, plan.stream);
    return he == hipSuccess ? MM_OK : hip_error(he, "export_cost kernel launch");
}

int WithinPlan::level_export_keys(size_t l, const double* gcost_dev, long long* keys_dev)
{
    (void)l;
    if (!searched || !d_pair_of_job) return set_error(MM_ERR_INVALID, "level_export_keys before level_export_cost");
    const hipError_t he = launch_export_keys(plan.dev, d_pair_of_job, (int)job_geom.size(), gcost_dev, keys_dev, plan.stream);
    return he == hipSuccess ? MM_OK : hip_error(he, "export_keys kernel launch");
}

int WithinPlan::level_commit_dev(size_t l, const double* gcost_dev, const long long* keys_dev)
{
    if (int rc = level_copy_records(gcost_dev, keys_dev)) return rc;
    return level_commit_records(l);
}

// reduced records -> pinned host memory (the engine's level staging buffer: free between stage_level and fetch),
// enqueued on the plan's stream behind the collectives; Engine::tail_done marks the end of the level's device work
int WithinPlan::level_copy_records(const double* gcost_dev, const long long* keys_dev)
{
    const int J = (int)job_geom.size();
    if (J == 0) return MM_OK;
    const size_t o_keys = ((size_t)J * 8 + 255) / 256 * 256;
    int rc = e->ensure(e->host_lvl, o_keys + (size_t)J * 24, true);
    if (rc) return rc;
    unsigned char* hp = (unsigned char*)e->host_lvl.p;
    hipError_t he = launch_copy_small(hp, gcost_dev, (size_t)J * 8, plan.stream);
    if (he == hipSuccess) he = launch_copy_small(hp + o_keys, keys_dev, (size_t)J * 24, plan.stream);
    if (he != hipSuccess) return hip_error(he, "exchange records D2H");
    if (!e->tail_done && hipEventCreateWithFlags(&e->tail_done, hipEventDisableTiming) != hipSuccess) e->tail_done = nullptr;
    e->tail_done_recorded = e->tail_done && hipEventRecord(e->tail_done, plan.stream) == hipSuccess;
    return MM_OK;
}

// wait for the records, decode them and commit the level (identical on every rank)
int WithinPlan::level_commit_records(size_t l)
{
    const int J = (int)job_geom.size();
    if (J == 0) return MM_OK;
    const size_t o_keys = ((size_t)J * 8 + 255) / 256 * 256;
    unsigned char* hp = (unsigned char*)e->host_lvl.p;
    const hipError_t he = hipStreamSynchronize(plan.stream);
    if (he != hipSuccess) return hip_error(he, "exchange records D2H");
    const double* h_gcost = (const double*)hp;
    const long long* h_keys = (const long long*)(hp + o_keys);
    std::vector<uint8_t> ok((size_t)J, 1);
    std::vector<double> angle((size_t)J, 0.0);
    for (size_t k = 0; k < lvl_active.size(); ++k) {
        const int j = lvl_active[k];
        if (!(h_gcost[(size_t)j] < INFINITY)) continue;            // no rank held a candidate (mm_merge_shards: ok = 1)
        const long long gi = h_keys[(size_t)j], lo = h_keys[(size_t)J + j], hi = ~h_keys[2 * (size_t)J + j];
        if (gi < 0 || gi >= (long long)lvl_pairs[k].n_angles)
            return set_error(MM_ERR_INVALID, "exchange: reduced winner index out of range (ranks disagree on the candidate lists)");
        angle[(size_t)j] = lvl_pairs[k].angles[gi];                // every rank holds the full list on the host
        ok[(size_t)j] = (lo == hi) ? 1 : 0;                        // all near ranks uniform and of one angle value
        long long abits; std::memcpy(&abits, &angle[(size_t)j], 8);
        if (ok[(size_t)j] && abits != lo)
            return set_error(MM_ERR_INVALID, "exchange: reduced angle differs from the winner's list entry");
    }
    return level_commit(l, ok.data(), angle.data());
}

// Record the merged (all ranks) outcome of level l: ok[j] != 0 -> the level's winner is
// `angle[j]`; ok[j] == 0 -> the step is re-searched on the chain state during the walk.
int WithinPlan::level_commit(size_t l, const uint8_t* ok, const double* angle)
{
    if (!searched) return set_error(MM_ERR_INVALID, "level_commit before level_local");
    for (size_t k = 0; k < lvl_active.size(); ++k) {
        const int j = lvl_active[k];
        evals[j] += lvl_pairs[k].n_angles;
        if (ok[j]) centre[j] = angle[j];
        else resolved[j] = 0;
    }
    (void)l;
    return MM_OK;
}

int WithinPlan::search()
{
    const int J = (int)job_geom.size();
    std::vector<double> cost(J), angle(J), tol(J), out_angle(J), out_cost(J);
    std::vector<int32_t> uniform(J), idx(J), active(J), out_idx(J);
    std::vector<uint8_t> ok(J);
    for (int j = 0; j < J; ++j) tol[j] = 2.0 * eps[job_geom[j]];
    for (size_t l = 0; l < levels.size(); ++l) {
        int rc = level_local(l, cost.data(), uniform.data(), angle.data(), idx.data(), active.data());
        if (rc) return rc;
        if ((rc = mm_merge_shards(1, J, cost.data(), uniform.data(), angle.data(), idx.data(), tol.data(), ok.data(),
                                  out_angle.data(), out_idx.data(), out_cost.data())))
            return rc;
        if ((rc = level_commit(l, ok.data(), out_angle.data()))) return rc;
    }
    return MM_OK;
}

// The search over this rank's tile with the exchange inside the library (include/mm_hausdorff.h, "multi-GPU"):
// per level the tile's results stay in HBM, two small kernels write the job-indexed records, RCCL reduces them in
// place on the engine's stream, one copy + one synchronisation bring the reduced records to the host.
int comm_all_reduce_min(Comm* c, void* dev, int64_t n, bool is_f64, hipStream_t st);
int comm_rank(const Comm* c);
int comm_world(const Comm* c);

int WithinPlan::exchange_enqueue(Comm* c, size_t l)
{
    // mm_within_plan_set_timing_rehearsal (timing only, bench.py's MM_BENCH_REHEARSE_WORLD): one process plays a rank of a
    // larger job on a world = 1 communicator -- the reduced records then hold this rank's tile alone, the result is not an
    // alignment and walk() says so (n_unresolved = -1)
    if ((comm_world(c) != world || comm_rank(c) != rank) && !(rehearsal && comm_world(c) == 1))
        return set_error(MM_ERR_INVALID, "search_sharded: the plan's (rank, world) is not the communicator's");
    const int J = (int)job_geom.size();
    const size_t o_keys = ((size_t)std::max(J, 1) * 8 + 255) / 256 * 256;
    if (!d_xrec)
        if (int rc = e->blob_alloc((void**)&d_xrec, o_keys + (size_t)std::max(J, 1) * 24, &xrec_cap)) return rc;
    double* xc = (double*)d_xrec;
    long long* xk = (long long*)(d_xrec + o_keys);
    int rc;
    if (launched_level != (int)l && (rc = level_launch(l))) return rc;   // (level 0 may have been queued early)
    if ((rc = level_export_cost(l, xc))) return rc;
    if (J > 0 && (rc = comm_all_reduce_min(c, xc, J, true, plan.stream))) return rc;
    if (J > 0 && (rc = level_export_keys(l, xc, xk))) return rc;
    if (J > 0 && (rc = comm_all_reduce_min(c, xk, 3 * (int64_t)J, false, plan.stream))) return rc;
    return level_copy_records(xc, xk);
}

// Level 0 up to and including the copy of the reduced records, all enqueued, nothing waited for: a driver that aligns
// independent cases back to back queues the NEXT case's launch behind this (mm_engine_wait_exchange) before it
// collects this one -- the device goes from this level's last copy straight into the next launch, and no collective
// ever runs beside a long kernel.
int WithinPlan::search_sharded_begin(Comm* c)
{
    if (xchg_pending) return set_error(MM_ERR_INVALID, "search_sharded_begin called twice");
    if (levels.empty()) return MM_OK;
    if (int rc = exchange_enqueue(c, 0)) return rc;
    xchg_pending = true;
    return MM_OK;
}

int WithinPlan::search_sharded_end(Comm* c)
{
    for (size_t l = 0; l < levels.size(); ++l) {
        int rc;
        if (!(l == 0 && xchg_pending) && (rc = exchange_enqueue(c, l))) return rc;
        xchg_pending = false;
        if ((rc = level_commit_records(l))) return rc;
    }
    return MM_OK;
}

int WithinPlan::search_sharded(Comm* c)
{
    return search_sharded_end(c);     // (after search_sharded_begin: collects level 0 and runs the remaining levels)
}

int WithinPlan::walk(mm_alignlog** logs, int64_t* pose_evals, int64_t* n_unresolved)
{
    if (!searched) return set_error(MM_ERR_INVALID, "walk before the search levels were run");
    TraceTimer tw("within: chain walk");
    // Fast path: every step was decided by the one-shot search -> the pullbacks are
    // independent pure-host chains; walk them concurrently, one thread per pullback, like the
    // reference's crossbeam scope (entry.rs:140-203).
    bool all_resolved = true;
    for (size_t j = 0; j < resolved.size(); ++j) all_resolved = all_resolved && (resolved[j] != 0 || !taken(job_geom[j]));
    if (all_resolved) {
        // Pass 1 (serial, O(frames)): the chain couples frames only through the frame centroids.
        // Rotating a frame about its own centroid leaves that centroid bit for bit
        // (x = c - c = 0 -> 0*cos - 0*sin + c), so after step i the centroid is c + (p - c) with p the
        // final centroid of frame i-1; cumulative angles are a prefix sum.  Pass 2 (parallel over
        // frames, worker pool): the same Frame::rotate / translate / rotate calls as the sequential
        // walk, with those arguments -- every frame touches only its own data.
        struct Step { int g; int32_t i; double cum, tx, ty, cx, cy, best; };
        std::vector<Step> steps;
        for (int g = 0; g < n_geoms; ++g) {
            if (!taken(g)) continue;
            const mm_geometry* G = geoms[g];
            double cumulative = 0.0;
            double pcx = G->n_frames > 0 ? G->centroid[0] : 0.0, pcy = G->n_frames > 0 ? G->centroid[1] : 0.0;
            for (int32_t i = 1; i < G->n_frames; ++i) {
                const double c0x = G->centroid[3 * i], c0y = G->centroid[3 * i + 1];
                const double tx = pcx - c0x, ty = pcy - c0y;                 // align_within.rs:84-88
                const double cx = c0x + tx, cy = c0y + ty;                   // frame.rs:35-36
                const double best = centre[job_base[g] + (i - 1)];
                steps.push_back(Step{g, i, cumulative, tx, ty, cx, cy, best});
                cumulative += best;                                          // :123
                pcx = cx; pcy = cy;
            }
        }
        constexpr int kChunk = 16;
        const int n_chunks = (int)((steps.size() + kChunk - 1) / kChunk);
        parallel_for(n_chunks, [&](int c) {
            const size_t lo = (size_t)c * kChunk, hi = std::min(steps.size(), lo + kChunk);
            for (size_t k = lo; k < hi; ++k) {
                const Step& st = steps[k];
                mm_geometry* G = geoms[st.g];
                const int32_t i = st.i;
                if (st.cum != 0.0)  // align_within.rs:79-82
                    mm_frame_rotate(G, i, st.cum, G->centroid[3 * i], G->centroid[3 * i + 1]);
                mm_frame_translate(G, i, st.tx, st.ty, 0.0);                 // :90
                mm_frame_rotate(G, i, st.best, st.cx, st.cy);                // :121-122
                if (logs && logs[st.g]) {
                    mm_alignlog& L = logs[st.g][i - 1];
                    L.contour_id = G->id[i]; L.matched_to = G->id[i - 1];
                    L.rot_deg = rad2deg(st.best); L.tx = st.tx; L.ty = st.ty;
                    L.cx = G->centroid[3 * i]; L.cy = G->centroid[3 * i + 1];
                }
            }
        });
        if (pose_evals) for (size_t j = 0; j < evals.size(); ++j) if (taken(job_geom[j])) *pose_evals += evals[j];
        return MM_OK;
    }
    std::vector<double> cumulative(n_geoms, 0.0);
    for (int32_t i = 1; i < max_frames; ++i) {
        std::vector<SearchJob> jobs;      // unresolved steps: searched on the chain state
        std::vector<int> job_owner;
        struct Step { int g; double tx, ty, cx, cy, best; bool pending; };
        std::vector<Step> steps;
        for (int g = 0; g < n_geoms; ++g) {
            mm_geometry* G = geoms[g];
            if (i >= G->n_frames || !taken(g)) continue;
            const double pcx = G->centroid[3 * (i - 1)], pcy = G->centroid[3 * (i - 1) + 1];
            if (cumulative[g] != 0.0)  // align_within.rs:79-82
                mm_frame_rotate(G, i, cumulative[g], G->centroid[3 * i], G->centroid[3 * i + 1]);
            const double tx = pcx - G->centroid[3 * i], ty = pcy - G->centroid[3 * i + 1];  // :84-88
            mm_frame_translate(G, i, tx, ty, 0.0);                                            // :90
            Step st{g, tx, ty, G->centroid[3 * i], G->centroid[3 * i + 1], 0.0, false};
            const int j = job_base[g] + (i - 1);
            if (resolved[j]) {
                st.best = centre[j];
                if (pose_evals) *pose_evals += evals[j];
            } else {
                SearchJob job;
                frame_search_set(G, i, spec[g], job.tx, job.ty);      // :92-93
                frame_search_set(G, i - 1, spec[g], job.rx, job.ry);  // :94-95
                job.cx = st.cx; job.cy = st.cy; job.flags = MM_SEARCH_SKIP_ZERO;
                jobs.push_back(std::move(job));
                job_owner.push_back((int)steps.size());
                st.pending = true;
                if (n_unresolved) ++*n_unresolved;
            }
            steps.push_back(st);
        }
        if (!jobs.empty()) {
            int rc = run_searches(e, jobs, step_deg, range_deg, bruteforce, precision, pose_evals);
            if (rc) return rc;
            for (size_t k = 0; k < jobs.size(); ++k) steps[job_owner[k]].best = jobs[k].result;
        }
        for (const Step& st : steps) {
            mm_geometry* G = geoms[st.g];
            mm_frame_rotate(G, i, st.best, st.cx, st.cy);  // :121-122
            cumulative[st.g] += st.best;                   // :123
            if (logs && logs[st.g]) {
                mm_alignlog& L = logs[st.g][i - 1];
                L.contour_id = G->id[i]; L.matched_to = G->id[i - 1];
                L.rot_deg = rad2deg(st.best); L.tx = st.tx; L.ty = st.ty;
                L.cx = G->centroid[3 * i]; L.cy = G->centroid[3 * i + 1];
            }
        }
    }
    return MM_OK;
}

}  // namespace mm

using namespace mm;

// The HIP device is a per-thread setting: a host that calls from several threads (the reference's
// crossbeam scopes, a pipelined driver) must get the engine's device whichever thread it is on.
static int select_device(Engine* e)
{
    const hipError_t he = hipSetDevice(e->device);
    return he == hipSuccess ? MM_OK : hip_error(he, "hipSetDevice");
}

extern "C" {

int64_t mm_search_angles(double step_deg, double range_deg, int has_center, double center, double limes_deg,
                         double* out, int64_t cap, int* degenerate, double* early_value)
{
    std::vector<double> v;
    double early = 0.0;
    const bool ok = enumerate_angles(step_deg, range_deg, has_center != 0, center, limes_deg, v, early);
    if (int erc = enum_overflow_error()) return erc;
    if (degenerate) *degenerate = ok ? 0 : 1;
    if (early_value) *early_value = ok ? 0.0 : early;
    if (out) for (int64_t i = 0; i < (int64_t)v.size() && i < cap; ++i) out[i] = v[(size_t)i];
    return (int64_t)v.size();
}

// align_algorithms.rs:386-387,439: accumulated angle enumeration of the refine grid
int64_t mm_refine_angles(double initial, double range, double step, double* out, int64_t cap)
{
    int64_t n = 0;
    if (!(step > 0.0)) return 0;  // the reference would loop forever
    double angle = initial - range;
    while (angle <= initial + range) {
        if (out && n < cap) out[n] = angle;
        ++n;
        angle += step;
        // a step below the spacing of the doubles at `angle` never advances (the reference would not terminate), and a
        // grid of more than 2^22 angles is refused by the refinement anyway
        if (n > ((int64_t)1 << 22)) return set_error(MM_ERR_TOO_LARGE, "mm_refine_angles: more than 2^22 angles");
    }
    return n;
}

// align_algorithms.rs:454-505
int64_t mm_filter_points_in_region(const double* xyz, int64_t n, const double* s, const double* e,
                                   int64_t* out_idx, int64_t cap)
{
    if (n < 0 || (n > 0 && !xyz) || !s || !e) return set_error(MM_ERR_INVALID, "mm_filter_points_in_region: bad arguments");
    const double margin = 5.0;
    double lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(s[k], e[k]) - margin; hi[k] = std::fmax(s[k], e[k]) + margin; }
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double *p = xyz + 3 * i;
        if (p[0] >= lo[0] && p[0] <= hi[0] && p[1] >= lo[1] && p[1] <= hi[1] && p[2] >= lo[2] && p[2] <= hi[2]) {
            if (out_idx && m < cap) out_idx[m] = i;
            ++m;
        }
    }
    return m;
}

// align_algorithms.rs:415-418
int64_t mm_refine_downsample_count(int64_t n_filtered, int64_t n_points_per_frame, int64_t n_frames)
{
    const double ratio = (double)n_filtered / ((double)n_points_per_frame * (double)n_frames);
    const double nd = std::ceil(ratio * (double)n_points_per_frame);
    // Rust's `as usize` saturates: NaN (0/0 with no frames or no points) -> 0, +inf -> usize::MAX; then clamp(1, M)
    int64_t n = !(nd > 0.0) ? 0 : (nd >= 9.2e18 ? INT64_MAX : (int64_t)nd);
    if (n < 1) n = 1;
    if (n > n_points_per_frame) n = n_points_per_frame;
    return n;
}

// One plain decimal number [+-]digits[.digits][(e|E)[+-]digits] starting at p (end = text end); on success
// *val is its correctly rounded double and the position after it is returned, else nullptr.
// Clinger's fast path: a mantissa below 2^53 times or divided by an exactly representable power of ten
// (10^0..10^22) is ONE correctly rounded IEEE operation; everything else goes through strtod.
static const char* parse_decimal(const char* p, const char* end, double* val)
{
    static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                      1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* start = p;
    bool neg = false;
    if (p < end && (*p == '+' || *p == '-')) { neg = *p == '-'; ++p; }
    uint64_t m = 0;
    int digits = 0, sig = 0, e10 = 0;
    bool exact = true;
    auto take = [&](char c, bool frac) {
        ++digits;
        if (sig < 19) { m = m * 10 + (uint64_t)(c - '0'); if (m) ++sig; if (frac) --e10; }
        else { if (c != '0') exact = false; if (!frac) ++e10; }
    };
    while (p < end && *p >= '0' && *p <= '9') take(*p++, false);
    if (p < end && *p == '.') {
        ++p;
        while (p < end && *p >= '0' && *p <= '9') take(*p++, true);
    }
    if (digits == 0) return nullptr;
    if (p < end && (*p == 'e' || *p == 'E')) {
        const char* q = p + 1;
        bool eneg = false;
        if (q < end && (*q == '+' || *q == '-')) { eneg = *q == '-'; ++q; }
        if (q >= end || *q < '0' || *q > '9') return nullptr;
        int ex = 0;
        while (q < end && *q >= '0' && *q <= '9') { if (ex < 100000) ex = ex * 10 + (*q - '0'); ++q; }
        e10 += eneg ? -ex : ex;
        p = q;
    }
    if (exact && m <= ((uint64_t)1 << 53) && e10 >= -22 && e10 <= 22) {
        double v = (double)m;
        v = e10 < 0 ? v / kPow10[-e10] : v * kPow10[e10];
        *val = neg ? -v : v;
        return p;
    }
    char buf[128];
    const size_t n = (size_t)(p - start);
    if (n >= sizeof(buf)) return nullptr;
    std::memcpy(buf, start, n);
    buf[n] = 0;
    char* stop = nullptr;
    *val = std::strtod(buf, &stop);
    return stop == buf + n ? p : nullptr;
}

int64_t mm_parse_contour_table(const char* text, int64_t len, char delim, double* out, int64_t cap)
{
    if (!text || len < 0 || (cap > 0 && !out)) { set_error(MM_ERR_INVALID, "mm_parse_contour_table: bad arguments"); return -1; }
    const char *p = text, *end = text + len;
    int64_t rows = 0;
    while (p < end) {
        double v[4];
        {   // field 0 is a u32 in the reference (ContourPoint.frame_index): optional '+', decimal digits, nothing else
            // ("1.0", "1e0", "-0" do not deserialise there); anything else is left to the row reader
            const char* q = p;
            if (q < end && *q == '+') ++q;
            const char* d0 = q;
            while (q < end && *q >= '0' && *q <= '9') ++q;
            if (q == d0 || q >= end || *q != delim) return -1;
        }
        for (int f = 0; f < 4; ++f) {
            p = parse_decimal(p, end, &v[f]);
            if (!p || !std::isfinite(v[f])) return -1;
            if (f < 3) { if (p >= end || *p != delim) return -1; ++p; }
        }
        if (p < end) {                      // line end: LF or CRLF
            if (*p == '\r') { ++p; if (p >= end || *p != '\n') return -1; }
            if (*p != '\n') return -1;
            ++p;
        }
        if (v[0] < 0.0 || v[0] != std::floor(v[0]) || v[0] > 4294967295.0) return -1;   // frame index: u32
        if (rows < cap) { out[4 * rows] = v[0]; out[4 * rows + 1] = v[1]; out[4 * rows + 2] = v[2]; out[4 * rows + 3] = v[3]; }
        ++rows;
    }
    return rows;
}

// Contour::compute_centroid (contour.rs:213-224) of frame i's lumen into g->lumen_centroid: sequential sums / n
static void lumen_mean(mm_geometry* g, int32_t i)
{
    const int64_t lo = g->lumen_off[i], hi = g->lumen_off[i + 1];
    if (hi <= lo) return;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int64_t k = lo; k < hi; ++k) { sx += g->lumen[3 * k]; sy += g->lumen[3 * k + 1]; sz += g->lumen[3 * k + 2]; }
    const double n = (double)(hi - lo);
    g->lumen_centroid[3 * i] = sx / n; g->lumen_centroid[3 * i + 1] = sy / n; g->lumen_centroid[3 * i + 2] = sz / n;
}

// frame.rs:17-38
void mm_frame_translate(mm_geometry* g, int32_t i, double dx, double dy, double dz)
{
    if (!g || i < 0 || i >= g->n_frames || !g->lumen_off || !g->lumen || !g->centroid) return;   // nothing to move
    span_translate(g->lumen, g->lumen_off[i], g->lumen_off[i + 1], dx, dy, dz);
    if (g->cath_off) span_translate(g->cath, g->cath_off[i], g->cath_off[i + 1], dx, dy, dz);
    if (g->extra_off) span_translate(g->extra, g->extra_off[i], g->extra_off[i + 1], dx, dy, dz);
    if (g->has_ref && g->has_ref[i]) { g->ref[3 * i] += dx; g->ref[3 * i + 1] += dy; g->ref[3 * i + 2] += dz; }
    g->centroid[3 * i] += dx; g->centroid[3 * i + 1] += dy; g->centroid[3 * i + 2] += dz;
    if (g->lumen_centroid) lumen_mean(g, i);   // frame.rs:19-20: self.lumen.compute_centroid()
}

// frame.rs:40-63
void mm_frame_rotate(mm_geometry* g, int32_t i, double angle, double cx, double cy)
{
    if (angle == 0.0) return;
    if (!g || i < 0 || i >= g->n_frames || !g->lumen_off || !g->lumen || !g->centroid) return;
    span_rotate(g->lumen, g->lumen_off[i], g->lumen_off[i + 1], angle, cx, cy);
    if (g->cath_off) span_rotate(g->cath, g->cath_off[i], g->cath_off[i + 1], angle, cx, cy);
    if (g->extra_off) span_rotate(g->extra, g->extra_off[i], g->extra_off[i + 1], angle, cx, cy);
    if (g->has_ref && g->has_ref[i]) rotate_xy(g->ref[3 * i], g->ref[3 * i + 1], angle, cx, cy);
    const double x = g->centroid[3 * i] - cx, y = g->centroid[3 * i + 1] - cy;
    double co, si;
    sin_cos(angle, si, co);
    g->centroid[3 * i] = x * co - y * si + cx;
    g->centroid[3 * i + 1] = x * si + y * co + cy;
}

int64_t mm_catheter_lumen_vec(const mm_geometry* g, int32_t frame, int64_t sample_size, double* out_x,
                              double* out_y, int64_t cap)
{
    if (!g || frame < 0 || frame >= g->n_frames || sample_size <= 0) return 0;
    if (g->lumen_off[1] - g->lumen_off[0] <= 0) return 0;
    std::vector<double> x, y;
    frame_search_set(g, frame, sample_spec(g, sample_size), x, y);
    for (int64_t i = 0; i < (int64_t)x.size() && i < cap; ++i) { out_x[i] = x[(size_t)i]; out_y[i] = y[(size_t)i]; }
    return (int64_t)x.size();
}

// align_between.rs:154-178
static void between_points(const mm_geometry* g, int64_t sample_size, std::vector<double>& x, std::vector<double>& y)
{
    const int64_t total = g->lumen_off[g->n_frames] - g->lumen_off[0];
    const double ratio = (double)sample_size / (double)total;
    for (int32_t i = 0; i < g->n_frames; ++i) {
        const int64_t len = g->lumen_off[i + 1] - g->lumen_off[i];
        const double fd = std::ceil((double)len * ratio);
        int64_t fs = !(fd > 0.0) ? 0 : (fd >= 9.2e18 ? INT64_MAX : (int64_t)fd);
        fs = std::max<int64_t>(fs, 1);
        downsample_append(g->lumen + 3 * g->lumen_off[i], len, fs, x, y);
    }
}

int64_t mm_extract_between_points(const mm_geometry* g, int64_t sample_size, double* out_x, double* out_y, int64_t cap)
{
    if (!g || g->n_frames <= 0) return 0;
    std::vector<double> x, y;
    between_points(g, sample_size, x, y);
    if (out_x && out_y)
        for (int64_t i = 0; i < (int64_t)x.size() && i < cap; ++i) { out_x[i] = x[(size_t)i]; out_y[i] = y[(size_t)i]; }
    return (int64_t)x.size();
}

// align_within.rs:24-134 for n_geoms pullbacks in lockstep.
int mm_align_within(mm_engine* eh, int n_geoms, mm_geometry** geoms, double step_deg, double range_deg,
                    int bruteforce, int64_t sample_size, int precision, int mode, mm_alignlog** logs,
                    int64_t* pose_evals)
{
    Engine* e = reinterpret_cast<Engine*>(eh);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    if (n_geoms <= 0 || !geoms) return set_error(MM_ERR_INVALID, "no geometries");
    if (mode != 0 && mode != 1) return set_error(MM_ERR_INVALID, "mm_align_within: mode must be 0 (chain) or 1 (decoupled)");
    if (int drc = select_device(e)) return drc;
    if (mode == 1) {
        mm_within_plan* wp = nullptr;
        int rc = mm_within_plan_create(eh, n_geoms, geoms, step_deg, range_deg, bruteforce, sample_size, precision, &wp);
        if (rc) return rc;
        rc = mm_within_plan_run(wp, logs, pose_evals, nullptr);
        mm_within_plan_destroy(wp);
        return rc;
    }
    if (pose_evals) *pose_evals = 0;
    int32_t max_frames = 0;
    std::vector<SampleSpec> spec(n_geoms);
    for (int g = 0; g < n_geoms; ++g) {
        const mm_geometry* G = geoms[g];
        if (!G || G->n_frames <= 0) return set_error(MM_ERR_NO_FRAMES, "Geometry contains no frames");
        if (G->lumen_off[1] - G->lumen_off[0] <= 0) return set_error(MM_ERR_NO_POINTS, "Lumen contours have no points");
        if (sample_size <= 0) return set_error(MM_ERR_SAMPLE_SIZE, "sample_size must be > 0");
        spec[g] = sample_spec(G, sample_size);
        max_frames = std::max(max_frames, G->n_frames);
    }
    std::vector<double> cumulative(n_geoms, 0.0);
    for (int32_t i = 1; i < max_frames; ++i) {
        std::vector<SearchJob> jobs;
        std::vector<int> owner;
        std::vector<double> txs, tys;
        for (int g = 0; g < n_geoms; ++g) {
            mm_geometry* G = geoms[g];
            if (i >= G->n_frames) continue;
            const double pcx = G->centroid[3 * (i - 1)], pcy = G->centroid[3 * (i - 1) + 1];
            if (cumulative[g] != 0.0)  // :79-82
                mm_frame_rotate(G, i, cumulative[g], G->centroid[3 * i], G->centroid[3 * i + 1]);
            const double tx = pcx - G->centroid[3 * i], ty = pcy - G->centroid[3 * i + 1];  // :84-88
            mm_frame_translate(G, i, tx, ty, 0.0);                                            // :90
            SearchJob job;
            frame_search_set(G, i, spec[g], job.tx, job.ty);      // testing_points   :92-93
            frame_search_set(G, i - 1, spec[g], job.rx, job.ry);  // reference_points :94-95
            job.cx = G->centroid[3 * i]; job.cy = G->centroid[3 * i + 1];
            job.flags = MM_SEARCH_SKIP_ZERO;
            jobs.push_back(std::move(job));
            owner.push_back(g); txs.push_back(tx); tys.push_back(ty);
        }
        int rc = run_searches(e, jobs, step_deg, range_deg, bruteforce != 0, precision, pose_evals);
        if (rc) return rc;
        for (size_t k = 0; k < jobs.size(); ++k) {
            const int g = owner[k];
            mm_geometry* G = geoms[g];
            const double best = jobs[k].result;
            mm_frame_rotate(G, i, best, jobs[k].cx, jobs[k].cy);  // :121-122
            cumulative[g] += best;                                // :123
            if (logs && logs[g]) {
                mm_alignlog& L = logs[g][i - 1];
                L.contour_id = G->id[i]; L.matched_to = G->id[i - 1];
                L.rot_deg = rad2deg(best); L.tx = txs[k]; L.ty = tys[k];
                L.cx = G->centroid[3 * i]; L.cy = G->centroid[3 * i + 1];
            }
        }
    }
    return MM_OK;
}

int mm_within_plan_create(mm_engine* eh, int n_geoms, mm_geometry** geoms, double step_deg, double range_deg,
                          int bruteforce, int64_t sample_size, int precision, mm_within_plan** out)
{
    return mm_within_plan_create_sharded(eh, n_geoms, geoms, step_deg, range_deg, bruteforce, sample_size, precision, 0, 1, out);
}

int mm_within_plan_create_sharded(mm_engine* eh, int n_geoms, mm_geometry** geoms, double step_deg, double range_deg,
                                  int bruteforce, int64_t sample_size, int precision, int rank, int world,
                                  mm_within_plan** out)
{
    return mm_within_plan_create_grid(eh, n_geoms, geoms, step_deg, range_deg, bruteforce, sample_size, precision, rank, 1,
                                      world, out);
}

// Default tile shape: frame pairs first (a pair's exact re-score then runs on one rank, and the screen keeps its
// full-size workgroups), the candidate axis when the pairs are few.
int mm_shard_grid(int world, int64_t n_jobs, int* pair_blocks, int* cand_slices)
{
    if (world <= 0 || n_jobs < 0 || !pair_blocks || !cand_slices) return set_error(MM_ERR_INVALID, "mm_shard_grid: bad argument");
    int p = 1;
    for (int d = 1; d <= world; ++d)
        if (world % d == 0 && n_jobs / d >= 64) p = d;
    *pair_blocks = p; *cand_slices = world / p;
    return MM_OK;
}

int mm_within_plan_create_grid(mm_engine* eh, int n_geoms, mm_geometry** geoms, double step_deg, double range_deg,
                               int bruteforce, int64_t sample_size, int precision, int rank, int pair_blocks,
                               int cand_slices, mm_within_plan** out)
{
    Engine* e = reinterpret_cast<Engine*>(eh);
    if (!e || !out) return set_error(MM_ERR_INVALID, "engine/out == NULL");
    *out = nullptr;
    if (n_geoms <= 0 || !geoms) return set_error(MM_ERR_INVALID, "no geometries");
    if (pair_blocks <= 0 || cand_slices <= 0 || pair_blocks > (1 << 15) || cand_slices > (1 << 15))
        return set_error(MM_ERR_INVALID, "bad shard");
    const int world = pair_blocks * cand_slices;
    if (rank < 0 || rank >= world) return set_error(MM_ERR_INVALID, "bad shard");
    if (int drc = select_device(e)) return drc;
    WithinPlan* wp = new WithinPlan();
    wp->rank = rank; wp->world = world; wp->grid_p = pair_blocks; wp->grid_c = cand_slices;
    wp->e = e; wp->n_geoms = n_geoms; wp->geoms.assign(geoms, geoms + n_geoms);
    wp->step_deg = step_deg; wp->range_deg = range_deg; wp->bruteforce = bruteforce != 0;
    wp->sample_size = sample_size; wp->precision = precision;
    int rc = wp->prepare();
    if (rc) { delete wp; return rc; }
    *out = reinterpret_cast<mm_within_plan*>(wp);
    return MM_OK;
}

int mm_within_plan_run(mm_within_plan* h, mm_alignlog** logs, int64_t* pose_evals, int64_t* n_unresolved)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp) return set_error(MM_ERR_INVALID, "within plan == NULL");
    if (pose_evals) *pose_evals = 0;
    if (n_unresolved) *n_unresolved = 0;
    if (int drc = select_device(wp->e)) return drc;
    int rc = wp->search();
    if (rc) return rc;
    return wp->walk(logs, pose_evals, n_unresolved);
}

int mm_within_plan_set_shard(mm_within_plan* h, int rank, int world)
{
    return mm_within_plan_set_shard_grid(h, rank, 1, world);
}

int mm_within_plan_set_shard_grid(mm_within_plan* h, int rank, int pair_blocks, int cand_slices)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || pair_blocks <= 0 || cand_slices <= 0 || pair_blocks > (1 << 15) || cand_slices > (1 << 15))
        return set_error(MM_ERR_INVALID, "bad shard");
    const int world = pair_blocks * cand_slices;
    if (rank < 0 || rank >= world) return set_error(MM_ERR_INVALID, "bad shard");
    if (wp->searched) return set_error(MM_ERR_INVALID, "shard must be set before the first level");
    if (int drc = select_device(wp->e)) return drc;
    if (rank != wp->rank || pair_blocks != wp->grid_p || cand_slices != wp->grid_c) {
        wp->rank = rank; wp->world = world; wp->grid_p = pair_blocks; wp->grid_c = cand_slices; wp->level0_staged = false;
        // a plan created on a tile holds that pair block's frames only: another block needs its own (a plan created
        // unsharded holds every frame and is re-tiled in place)
        if (wp->sets_partial && (pair_blocks != wp->sets_gp || rank / cand_slices != wp->sets_pb)) {
            if (int rc = wp->restage_sets()) return rc;
        }
        if (wp->level0_ok) {  // re-stage level 0 for the new slice now, not inside the search
            wp->build_level_pairs(0, std::vector<double>(), std::vector<uint8_t>(wp->job_geom.size(), 1), wp->lvl_pairs,
                                  wp->lvl_active, nullptr);
            int rc = wp->plan.stage_level(wp->lvl_pairs, wp->precision, 0, INT32_MAX, false, wp->e->aux);
            if (rc) return rc;
            wp->level0_staged = true;
        }
    }
    return MM_OK;
}

int mm_within_plan_staged(mm_within_plan* h, int64_t* raw_points, int64_t* set_points)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp) return set_error(MM_ERR_INVALID, "within plan == NULL");
    if (raw_points) *raw_points = wp->staged_points;
    if (set_points) *set_points = wp->plan.n_points;
    return MM_OK;
}

int mm_within_plan_dims(mm_within_plan* h, int32_t* n_jobs, int32_t* n_levels, double* tol)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp) return set_error(MM_ERR_INVALID, "within plan == NULL");
    const int J = (int)wp->job_geom.size();
    if (n_jobs) *n_jobs = J;
    if (n_levels) *n_levels = (int32_t)wp->levels.size();
    // (a plan that staged only its pair block's frames knows the tolerance of its own jobs only: 0 for the others, the
    // exchange takes a job's tolerance from its owners -- distributed.merge_level)
    if (tol) for (int j = 0; j < J; ++j) tol[j] = (wp->sets_partial && !wp->owns(j)) ? 0.0 : 2.0 * wp->eps[wp->job_geom[j]];
    return MM_OK;
}

int mm_within_plan_level_local(mm_within_plan* h, int level, double* cost, int32_t* uniform, double* angle,
                               int32_t* idx, int32_t* active)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || level < 0 || (size_t)level >= wp->levels.size()) return set_error(MM_ERR_INVALID, "bad level");
    if (int drc = select_device(wp->e)) return drc;
    return wp->level_local((size_t)level, cost, uniform, angle, idx, active);
}

int64_t mm_within_plan_fetch_set(mm_within_plan* h, int32_t set, double* x64, double* y64, float* x32, float* y32,
                                 int64_t cap, double* rho)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || set < 0 || (size_t)set >= wp->plan.set_len.size()) { set_error(MM_ERR_INVALID, "bad set index"); return -1; }
    if (select_device(wp->e)) return -1;
    const Plan& pl = wp->plan;
    const int64_t n = pl.set_len[(size_t)set], o = pl.set_off[(size_t)set], m = std::min<int64_t>(n, cap);
    if (rho) *rho = pl.set_rho[(size_t)set];
    hipError_t he = hipSuccess;
    if (m > 0 && x64) he = hipMemcpy(x64, pl.dev.p64x + o, (size_t)m * 8, hipMemcpyDeviceToHost);
    if (he == hipSuccess && m > 0 && y64) he = hipMemcpy(y64, pl.dev.p64y + o, (size_t)m * 8, hipMemcpyDeviceToHost);
    if (he == hipSuccess && m > 0 && x32) he = hipMemcpy(x32, pl.dev.p32x + o, (size_t)m * 4, hipMemcpyDeviceToHost);
    if (he == hipSuccess && m > 0 && y32) he = hipMemcpy(y32, pl.dev.p32y + o, (size_t)m * 4, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { hip_error(he, "fetch_set"); return -1; }
    return n;
}

int mm_within_plan_level_launch(mm_within_plan* h, int level)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || level < 0 || (size_t)level >= wp->levels.size()) return set_error(MM_ERR_INVALID, "bad level");
    if (int drc = select_device(wp->e)) return drc;
    return wp->level_launch((size_t)level);
}

int mm_within_plan_level_export_cost(mm_within_plan* h, int level, double* cost_dev)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || level < 0 || (size_t)level >= wp->levels.size() || !cost_dev) return set_error(MM_ERR_INVALID, "bad level / buffer");
    if (int drc = select_device(wp->e)) return drc;
    return wp->level_export_cost((size_t)level, cost_dev);
}

int mm_within_plan_level_export_keys(mm_within_plan* h, int level, const double* gcost_dev, int64_t* keys_dev)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || level < 0 || (size_t)level >= wp->levels.size() || !gcost_dev || !keys_dev)
        return set_error(MM_ERR_INVALID, "bad level / buffer");
    if (int drc = select_device(wp->e)) return drc;
    return wp->level_export_keys((size_t)level, gcost_dev, (long long*)keys_dev);
}

int mm_within_plan_level_commit_dev(mm_within_plan* h, int level, const double* gcost_dev, const int64_t* keys_dev)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || level < 0 || (size_t)level >= wp->levels.size() || !gcost_dev || !keys_dev)
        return set_error(MM_ERR_INVALID, "bad level / buffer");
    if (int drc = select_device(wp->e)) return drc;
    return wp->level_commit_dev((size_t)level, gcost_dev, (const long long*)keys_dev);
}

int mm_within_plan_level_collect(mm_within_plan* h, int level, double* cost, int32_t* uniform, double* angle,
                                 int32_t* idx, int32_t* active)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || level < 0 || (size_t)level >= wp->levels.size()) return set_error(MM_ERR_INVALID, "bad level");
    if (int drc = select_device(wp->e)) return drc;
    return wp->level_collect((size_t)level, cost, uniform, angle, idx, active);
}

int mm_within_plan_level_commit(mm_within_plan* h, int level, const uint8_t* ok, const double* angle)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || level < 0 || (size_t)level >= wp->levels.size()) return set_error(MM_ERR_INVALID, "bad level");
    return wp->level_commit((size_t)level, ok, angle);
}

int mm_within_plan_search_sharded(mm_within_plan* h, mm_comm* ch)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || !ch) return set_error(MM_ERR_INVALID, "within plan / communicator == NULL");
    if (int drc = select_device(wp->e)) return drc;
    return wp->search_sharded(reinterpret_cast<Comm*>(ch));
}

int mm_within_plan_search_sharded_begin(mm_within_plan* h, mm_comm* ch)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || !ch) return set_error(MM_ERR_INVALID, "within plan / communicator == NULL");
    if (int drc = select_device(wp->e)) return drc;
    return wp->search_sharded_begin(reinterpret_cast<Comm*>(ch));
}

int mm_within_plan_run_sharded(mm_within_plan* h, mm_comm* ch, mm_alignlog** logs, int64_t* pose_evals, int64_t* n_unresolved)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || !ch) return set_error(MM_ERR_INVALID, "within plan / communicator == NULL");
    if (pose_evals) *pose_evals = 0;
    if (n_unresolved) *n_unresolved = 0;
    if (int drc = select_device(wp->e)) return drc;
    if (int rc = wp->search_sharded(reinterpret_cast<Comm*>(ch))) return rc;
    const int rc = wp->walk(logs, pose_evals, n_unresolved);
    if (wp->rehearsal && n_unresolved) *n_unresolved = -1;
    return rc;
}

int mm_within_plan_walk(mm_within_plan* h, mm_alignlog** logs, int64_t* pose_evals, int64_t* n_unresolved)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp) return set_error(MM_ERR_INVALID, "within plan == NULL");
    if (pose_evals) *pose_evals = 0;
    if (n_unresolved) *n_unresolved = 0;
    if (int drc = select_device(wp->e)) return drc;   // unresolved steps search on the chain state
    const int rc = wp->walk(logs, pose_evals, n_unresolved);
    if (wp->rehearsal && n_unresolved) *n_unresolved = -1;    // not an alignment: the records held one tile of a larger job
    return rc;
}

int mm_within_plan_walk_geoms(mm_within_plan* h, const uint8_t* take, mm_alignlog** logs, int64_t* pose_evals, int64_t* n_unresolved)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp || !take) return set_error(MM_ERR_INVALID, "within plan / take == NULL");
    if (pose_evals) *pose_evals = 0;
    if (n_unresolved) *n_unresolved = 0;
    if (int drc = select_device(wp->e)) return drc;
    wp->walk_take.assign(take, take + wp->n_geoms);
    const int rc = wp->walk(logs, pose_evals, n_unresolved);
    wp->walk_take.clear();
    if (wp->rehearsal && n_unresolved) *n_unresolved = -1;
    return rc;
}

int mm_within_plan_set_timing_rehearsal(mm_within_plan* h, int on)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp) return set_error(MM_ERR_INVALID, "within plan == NULL");
    wp->rehearsal = on != 0;
    return MM_OK;
}

// Merge per-shard results of `world` ranks (arrays are [world][n], rank-major).  For job j
// the winner is the first index of minimal cost over the whole candidate axis; ok[j] = the
// winner is decided, i.e. every shard whose minimum lies within tol[j] of the global minimum
// reports a uniform near-set and all those shards agree on the angle value bit for bit.
int mm_merge_shards(int world, int n, const double* cost, const int32_t* uniform, const double* angle,
                    const int32_t* idx, const double* tol, uint8_t* ok, double* out_angle, int32_t* out_idx,
                    double* out_cost)
{
    if (world <= 0 || n < 0) return set_error(MM_ERR_INVALID, "mm_merge_shards: bad sizes");
    if (n > 0 && (!cost || !uniform || !angle || !idx || !tol || !ok || !out_angle || !out_idx || !out_cost))
        return set_error(MM_ERR_INVALID, "mm_merge_shards: NULL array");
    for (int j = 0; j < n; ++j) {
        double gbest = INFINITY;
        for (int r = 0; r < world; ++r) gbest = std::fmin(gbest, cost[(size_t)r * n + j]);
        ok[j] = 1; out_angle[j] = 0.0; out_idx[j] = -1; out_cost[j] = gbest;
        if (!(gbest < INFINITY)) continue;  // no shard had a candidate
        // first index of minimal cost (process_utils.rs:72), shards own increasing index ranges
        for (int r = 0; r < world; ++r) {
            const size_t k = (size_t)r * n + j;
            if (cost[k] == gbest && (out_idx[j] < 0 || idx[k] < out_idx[j])) { out_idx[j] = idx[k]; out_angle[j] = angle[k]; }
        }
        const double thr = gbest + (tol ? tol[j] : 0.0);
        for (int r = 0; r < world; ++r) {
            const size_t k = (size_t)r * n + j;
            if (idx[k] < 0 || !(cost[k] <= thr)) continue;
            if (!uniform[k] || std::memcmp(&angle[k], &out_angle[j], 8) != 0) ok[j] = 0;
        }
    }
    return MM_OK;
}

void mm_within_plan_destroy(mm_within_plan* h)
{
    WithinPlan* wp = reinterpret_cast<WithinPlan*>(h);
    if (!wp) return;
    (void)wp->e->sync_all();
    delete wp;
}

// align_between.rs:11-68 for n_pairs independent (a, b) pairs.
int mm_align_between(mm_engine* eh, int n_pairs, mm_geometry** a, mm_geometry** b, double rot_deg,
                     double step_rot_deg, int64_t sample_size, int precision, double* best_rotation,
                     int64_t* pose_evals)
{
    Engine* e = reinterpret_cast<Engine*>(eh);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    if (n_pairs <= 0 || !a || !b) return set_error(MM_ERR_INVALID, "no geometry pairs");
    if (pose_evals) *pose_evals = 0;
    if (int drc = select_device(e)) return drc;
    TraceTimer t0("between: total");
    std::vector<SearchJob> jobs(n_pairs);
    std::vector<std::array<double, 3>> a_ref(n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        mm_geometry *A = a[p], *B = b[p];
        if (!A || !B || A->n_frames <= 0 || B->n_frames <= 0) return set_error(MM_ERR_NO_FRAMES, "Geometry contains no frames");
        const size_t ia = ref_or_proximal(A), ib = ref_or_proximal(B);  // :19-24
        if (ia >= (size_t)A->n_frames || ib >= (size_t)B->n_frames)
            return set_error(MM_ERR_REF_INDEX, "reference frame index out of range");
    }
    // The pairs are independent (the reference runs them in a crossbeam scope, entry.rs:206-277).
    // The reference makes three passes over B: translate (:40), rotate (:50), translate (:68).
    // Per coordinate these are the same roundings in the same order whether done as three passes
    // or one, so the search sets are built from the original frames plus the first translation
    // (the only thing the search needs) and the three steps are fused into ONE pass afterwards,
    // split over a few threads per pair.
    std::vector<std::array<double, 3>> t1(n_pairs);
    auto prep = [&](int p) {
        mm_geometry *A = a[p], *B = b[p];
        const size_t ia = ref_or_proximal(A), ib = ref_or_proximal(B);
        a_ref[p] = {A->centroid[3 * ia], A->centroid[3 * ia + 1], A->centroid[3 * ia + 2]};
        t1[p] = {a_ref[p][0] - B->centroid[3 * ib], a_ref[p][1] - B->centroid[3 * ib + 1],
                 a_ref[p][2] - B->centroid[3 * ib + 2]};                             // :33-37
        const int64_t s = std::max<int64_t>(sample_size, 500);                      // :43-44
        SearchJob& job = jobs[p];
        between_points(A, s, job.rx, job.ry);
        between_points(B, s, job.tx, job.ty);
        for (double& v : job.tx) v += t1[p][0];                                      // :40 on the sampled points
        for (double& v : job.ty) v += t1[p][1];
        // :260-271 centroid of the reference sample
        double sx = 0.0, sy = 0.0;
        for (double v : job.rx) sx += v;
        for (double v : job.ry) sy += v;
        if (!job.rx.empty()) { job.cx = sx / (double)job.rx.size(); job.cy = sy / (double)job.rx.size(); }
        job.flags = 0;  // no angle==0 shortcut in this closure (:194-209)
    };
    for (int p = 0; p < n_pairs; ++p) prep(p);
    int rc;
    {
        TraceTimer t("between: searches");
        rc = run_searches(e, jobs, step_rot_deg, rot_deg, /*bruteforce=*/false, precision, pose_evals);  // :46-47
    }
    if (rc) return rc;
    TraceTimer t2("between: apply");
    struct Motion { double dx, dy, dz, co, si, cx, cy, fx, fy, fz; };
    std::vector<Motion> mo(n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        mm_geometry *A = a[p], *B = b[p];
        Motion& m = mo[p];
        m.dx = t1[p][0]; m.dy = t1[p][1]; m.dz = t1[p][2];
        sin_cos(jobs[p].result, m.si, m.co);                                         // :96-97
        m.cx = a_ref[p][0]; m.cy = a_ref[p][1];
        // the final translation needs B's reference-frame centroid after translate + rotate (:53-66)
        const size_t ia = ref_or_proximal(A), ib = ref_or_proximal(B);
        double bx = B->centroid[3 * ib] + m.dx, by = B->centroid[3 * ib + 1] + m.dy, bz = B->centroid[3 * ib + 2] + m.dz;
        {
            const double tx = bx - m.cx, ty = by - m.cy;
            const double rx = tx * m.co - ty * m.si, ry = tx * m.si + ty * m.co;
            bx = rx + m.cx; by = ry + m.cy;
        }
        m.fx = A->centroid[3 * ia] - bx; m.fy = A->centroid[3 * ia + 1] - by; m.fz = A->centroid[3 * ia + 2] - bz;
        if (best_rotation) best_rotation[p] = jobs[p].result;
    }
    auto move_frames = [&](int p, int32_t f0, int32_t f1) {
        mm_geometry* B = b[p];
        const Motion m = mo[p];
        auto mv = [&](double* q) {
            double x = q[0] + m.dx, y = q[1] + m.dy, z = q[2] + m.dz;                // :40   Frame::translate
            const double tx = x - m.cx, ty = y - m.cy;                               // :100-107 rotate_point
            const double rx = tx * m.co - ty * m.si, ry = tx * m.si + ty * m.co;
            x = rx + m.cx; y = ry + m.cy;
            q[0] = x + m.fx; q[1] = y + m.fy; q[2] = z + m.fz;                       // :68   Frame::translate
        };
        for (int32_t i = f0; i < f1; ++i) {
            for (int64_t k = B->lumen_off[i]; k < B->lumen_off[i + 1]; ++k) mv(B->lumen + 3 * k);
            mv(B->centroid + 3 * i);
            if (B->cath_off) for (int64_t k = B->cath_off[i]; k < B->cath_off[i + 1]; ++k) mv(B->cath + 3 * k);
            if (B->extra_off) for (int64_t k = B->extra_off[i]; k < B->extra_off[i + 1]; ++k) mv(B->extra + 3 * k);
            if (B->has_ref && B->has_ref[i]) mv(B->ref + 3 * i);
            if (B->lumen_centroid) lumen_mean(B, i);   // the move ends with Frame::translate (:68): recomputed
        }
    };
    {
        // chunks of frames over the worker pool; small geometries stay on the calling thread
        struct Part { int p; int32_t f0, f1; };
        std::vector<Part> parts;
        for (int p = 0; p < n_pairs; ++p) {
            const int32_t F = b[p]->n_frames;
            const int32_t step = F >= 64 ? 32 : std::max<int32_t>(F, 1);
            for (int32_t f0 = 0; f0 < F; f0 += step) parts.push_back(Part{p, f0, std::min<int32_t>(F, f0 + step)});
        }
        parallel_for((int)parts.size(), [&](int k) { move_frames(parts[(size_t)k].p, parts[(size_t)k].f0, parts[(size_t)k].f1); });
    }
    return MM_OK;
}

}  // extern "C"
