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

namespace mm {

static constexpr double kPi = 3.14159265358979323846264338327950288;

static inline double deg2rad(double d) { return d * (kPi / 180.0); }  // f64::to_radians
static inline double rad2deg(double r) { return r * (180.0 / kPi); }  // f64::to_degrees

static inline double wrap_pi(double a)  // ((a + PI).rem_euclid(2 PI)) - PI, process_utils.rs:66
{
    double r = std::fmod(a + kPi, 2.0 * kPi);
    if (r < 0.0) r += std::fabs(2.0 * kPi);
    return r - kPi;
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
    size_t steps = sf <= 0.0 ? 0 : (size_t)sf;
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
    const double co = std::cos(angle), si = std::sin(angle);
    x = rx * co - ry * si + cx;
    y = rx * si + ry * co + cy;
}

static void span_translate(double* p, int64_t lo, int64_t hi, double dx, double dy, double dz)
{
    for (int64_t k = lo; k < hi; ++k) { p[3 * k] += dx; p[3 * k + 1] += dy; p[3 * k + 2] += dz; }
}

static void span_rotate(double* p, int64_t lo, int64_t hi, double angle, double cx, double cy)
{
    for (int64_t k = lo; k < hi; ++k) rotate_xy(p[3 * k], p[3 * k + 1], angle, cx, cy);
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
        s.cath = (int64_t)std::ceil((double)c0 * ratio);
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
    std::vector<double> centre(J, 0.0);
    std::vector<std::vector<double>> lists(J);
    for (size_t l = 0; l < levels.size(); ++l) {
        std::vector<int> active;
        for (int j = 0; j < J; ++j) {
            double early = 0.0;
            const bool ok = enumerate_angles(levels[l].step, levels[l].range, l > 0, centre[j], range_deg,
                                             lists[j], early);
            if (!ok) { centre[j] = early; continue; }  // search_range returned early
            active.push_back(j);
        }
        if (active.empty()) continue;
        std::vector<int64_t> ro{0}, to{0}, ao{0};
        std::vector<double> rx, ry, tx, ty, ang, cxs, cys;
        std::vector<int32_t> fl;
        for (int j : active) {
            const SearchJob& s = jobs[j];
            rx.insert(rx.end(), s.rx.begin(), s.rx.end()); ry.insert(ry.end(), s.ry.begin(), s.ry.end());
            tx.insert(tx.end(), s.tx.begin(), s.tx.end()); ty.insert(ty.end(), s.ty.begin(), s.ty.end());
            ang.insert(ang.end(), lists[j].begin(), lists[j].end());
            ro.push_back((int64_t)rx.size()); to.push_back((int64_t)tx.size()); ao.push_back((int64_t)ang.size());
            cxs.push_back(s.cx); cys.push_back(s.cy); fl.push_back(s.flags);
            if (pose_evals) *pose_evals += (int64_t)lists[j].size();
        }
        std::vector<int32_t> bidx(active.size());
        std::vector<double> bang(active.size()), bcost(active.size());
        int rc = mm_best_rotation_batch(reinterpret_cast<mm_engine*>(e), (int)active.size(), ro.data(), rx.data(),
                                        ry.data(), to.data(), tx.data(), ty.data(), ao.data(), ang.data(),
                                        cxs.data(), cys.data(), fl.data(), precision, bidx.data(), bang.data(),
                                        bcost.data(), nullptr, nullptr);
        if (rc) return rc;
        for (size_t k = 0; k < active.size(); ++k) centre[active[k]] = lists[active[k]][bidx[k]];
    }
    for (int j = 0; j < J; ++j) jobs[j].result = centre[j];
    return MM_OK;
}

}  // namespace mm

using namespace mm;

extern "C" {

int64_t mm_search_angles(double step_deg, double range_deg, int has_center, double center, double limes_deg,
                         double* out, int64_t cap, int* degenerate, double* early_value)
{
    std::vector<double> v;
    double early = 0.0;
    const bool ok = enumerate_angles(step_deg, range_deg, has_center != 0, center, limes_deg, v, early);
    if (degenerate) *degenerate = ok ? 0 : 1;
    if (early_value) *early_value = ok ? 0.0 : early;
    if (out) for (int64_t i = 0; i < (int64_t)v.size() && i < cap; ++i) out[i] = v[(size_t)i];
    return (int64_t)v.size();
}

// frame.rs:17-38
void mm_frame_translate(mm_geometry* g, int32_t i, double dx, double dy, double dz)
{
    span_translate(g->lumen, g->lumen_off[i], g->lumen_off[i + 1], dx, dy, dz);
    if (g->cath_off) span_translate(g->cath, g->cath_off[i], g->cath_off[i + 1], dx, dy, dz);
    if (g->extra_off) span_translate(g->extra, g->extra_off[i], g->extra_off[i + 1], dx, dy, dz);
    if (g->has_ref && g->has_ref[i]) { g->ref[3 * i] += dx; g->ref[3 * i + 1] += dy; g->ref[3 * i + 2] += dz; }
    g->centroid[3 * i] += dx; g->centroid[3 * i + 1] += dy; g->centroid[3 * i + 2] += dz;
}

// frame.rs:40-63
void mm_frame_rotate(mm_geometry* g, int32_t i, double angle, double cx, double cy)
{
    if (angle == 0.0) return;
    span_rotate(g->lumen, g->lumen_off[i], g->lumen_off[i + 1], angle, cx, cy);
    if (g->cath_off) span_rotate(g->cath, g->cath_off[i], g->cath_off[i + 1], angle, cx, cy);
    if (g->extra_off) span_rotate(g->extra, g->extra_off[i], g->extra_off[i + 1], angle, cx, cy);
    if (g->has_ref && g->has_ref[i]) rotate_xy(g->ref[3 * i], g->ref[3 * i + 1], angle, cx, cy);
    const double x = g->centroid[3 * i] - cx, y = g->centroid[3 * i + 1] - cy;
    const double co = std::cos(angle), si = std::sin(angle);
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
        int64_t fs = (int64_t)std::ceil((double)len * ratio);
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
    if (mode != 0) return set_error(MM_ERR_INVALID, "mm_align_within: mode 1 (decoupled) is not available yet");
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

// align_between.rs:11-68 for n_pairs independent (a, b) pairs.
int mm_align_between(mm_engine* eh, int n_pairs, mm_geometry** a, mm_geometry** b, double rot_deg,
                     double step_rot_deg, int64_t sample_size, int precision, double* best_rotation,
                     int64_t* pose_evals)
{
    Engine* e = reinterpret_cast<Engine*>(eh);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    if (n_pairs <= 0 || !a || !b) return set_error(MM_ERR_INVALID, "no geometry pairs");
    if (pose_evals) *pose_evals = 0;
    std::vector<SearchJob> jobs(n_pairs);
    std::vector<std::array<double, 3>> a_ref(n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        mm_geometry *A = a[p], *B = b[p];
        if (!A || !B || A->n_frames <= 0 || B->n_frames <= 0) return set_error(MM_ERR_NO_FRAMES, "Geometry contains no frames");
        const size_t ia = ref_or_proximal(A), ib = ref_or_proximal(B);  // :19-24
        if (ia >= (size_t)A->n_frames || ib >= (size_t)B->n_frames)
            return set_error(MM_ERR_REF_INDEX, "reference frame index out of range");
        a_ref[p] = {A->centroid[3 * ia], A->centroid[3 * ia + 1], A->centroid[3 * ia + 2]};
        const double dx = a_ref[p][0] - B->centroid[3 * ib], dy = a_ref[p][1] - B->centroid[3 * ib + 1],
                     dz = a_ref[p][2] - B->centroid[3 * ib + 2];                    // :33-37
        for (int32_t i = 0; i < B->n_frames; ++i) mm_frame_translate(B, i, dx, dy, dz);  // :40
        const int64_t s = std::max<int64_t>(sample_size, 500);                      // :43-44
        SearchJob& job = jobs[p];
        between_points(A, s, job.rx, job.ry);
        between_points(B, s, job.tx, job.ty);
        // :260-271 centroid of the reference sample
        double sx = 0.0, sy = 0.0;
        for (double v : job.rx) sx += v;
        for (double v : job.ry) sy += v;
        if (!job.rx.empty()) { job.cx = sx / (double)job.rx.size(); job.cy = sy / (double)job.rx.size(); }
        job.flags = 0;  // no angle==0 shortcut in this closure (:194-209)
    }
    int rc = run_searches(e, jobs, step_rot_deg, rot_deg, /*bruteforce=*/false, precision, pose_evals);  // :46-47
    if (rc) return rc;
    for (int p = 0; p < n_pairs; ++p) {
        mm_geometry *A = a[p], *B = b[p];
        const double best = jobs[p].result;
        // :95-145 rotate the whole of B about A's reference-frame centroid (no shortcut)
        const double co = std::cos(best), si = std::sin(best);
        const double cx = a_ref[p][0], cy = a_ref[p][1];
        auto rot = [&](double& x, double& y) {
            const double tx = x - cx, ty = y - cy;
            const double rx = tx * co - ty * si, ry = tx * si + ty * co;
            x = rx + cx; y = ry + cy;
        };
        for (int32_t i = 0; i < B->n_frames; ++i) {
            for (int64_t k = B->lumen_off[i]; k < B->lumen_off[i + 1]; ++k) rot(B->lumen[3 * k], B->lumen[3 * k + 1]);
            rot(B->centroid[3 * i], B->centroid[3 * i + 1]);
            if (B->cath_off) for (int64_t k = B->cath_off[i]; k < B->cath_off[i + 1]; ++k) rot(B->cath[3 * k], B->cath[3 * k + 1]);
            if (B->extra_off) for (int64_t k = B->extra_off[i]; k < B->extra_off[i + 1]; ++k) rot(B->extra[3 * k], B->extra[3 * k + 1]);
            if (B->has_ref && B->has_ref[i]) rot(B->ref[3 * i], B->ref[3 * i + 1]);
        }
        const size_t ia = ref_or_proximal(A), ib = ref_or_proximal(B);  // :53-58
        const double fx = A->centroid[3 * ia] - B->centroid[3 * ib], fy = A->centroid[3 * ia + 1] - B->centroid[3 * ib + 1],
                     fz = A->centroid[3 * ia + 2] - B->centroid[3 * ib + 2];  // :60-66
        for (int32_t i = 0; i < B->n_frames; ++i) mm_frame_translate(B, i, fx, fy, fz);  // :68
        if (best_rotation) best_rotation[p] = best;
    }
    return MM_OK;
}

}  // extern "C"
