// mm_centerline.cpp -- centerline placement path (include/mm_centerline.h): three-point initial
// rotation, per-frame placement on the resampled centerline, and the Hausdorff refinement grid.
// Geometry transforms are exact f64 host arithmetic in the reference's operation order (compiled
// with -ffp-contract=off); the grid's Hausdorff evaluations run on the device (hausdorff_sets).
// Reference lines are cited per function (paths relative to the reference checkout).
//
// Small-vector arithmetic follows nalgebra 0.35 (the reference's Cargo.lock): dot = (x0 y0 + x1 y1)
// + x2 y2, norm = sqrt(dot), Matrix::angle = acos(clamp(dot / (|a||b|))), from_axis_angle as in
// geometry/rotation_specialization.rs, matrix * vector accumulated column by column (blas gemv).
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/mm_centerline.h"
#include "mm_engine.h"
#include "mm_pool.h"
#include "mm_sort.h"
#include "mm_trace.h"

namespace mm {
namespace {

constexpr double kTau = 6.283185307179586476925286766559;
constexpr double kPiCl = 3.14159265358979323846264338327950288;

inline void sin_cos(double x, double& s, double& c) { ::sincos(x, &s, &c); }

struct Vec3 {
    double v[3];
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};
inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const Vec3& a) { return std::sqrt(dot(a, a)); }
inline Vec3 cross(const Vec3& a, const Vec3& b)
{
    return Vec3{{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}};
}
inline double angle_between(const Vec3& a, const Vec3& b)
{
    const double prod = dot(a, b), n1 = norm(a), n2 = norm(b);
    if (n1 == 0.0 || n2 == 0.0) return 0.0;
    double c = prod / (n1 * n2);
    c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
    return std::acos(c);
}

struct Mat3 {
    double m[9];  // row-major
    static Mat3 identity() { return Mat3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
    // Rotation3::from_axis_angle(&Unit::new_normalize(axis), angle)
    static Mat3 axis_angle(const Vec3& axis, double angle)
    {
        if (angle == 0.0) return identity();
        const double n = norm(axis);
        const double ux = axis[0] / n, uy = axis[1] / n, uz = axis[2] / n;
        const double sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
        double s, c;
        sin_cos(angle, s, c);
        const double omc = 1.0 - c;
        return Mat3{{sqx + (1.0 - sqx) * c, ux * uy * omc - uz * s, ux * uz * omc + uy * s,
                     ux * uy * omc + uz * s, sqy + (1.0 - sqy) * c, uy * uz * omc - ux * s,
                     ux * uz * omc - uy * s, uy * uz * omc + ux * s, sqz + (1.0 - sqz) * c}};
    }
    Vec3 operator*(const Vec3& x) const
    {
        Vec3 o;
        for (int i = 0; i < 3; ++i) {
            double y = m[3 * i] * x[0];
            y = m[3 * i + 1] * x[1] + y;
            y = m[3 * i + 2] * x[2] + y;
            o[i] = y;
        }
        return o;
    }
};

// FrameTransformation (align_algorithms.rs:65-94)
struct FrameTf {
    Vec3 t; Mat3 r; Vec3 pivot;
    inline void apply(double* p) const
    {
        const Vec3 rel{{(p[0] + t[0]) - pivot[0], (p[1] + t[1]) - pivot[1], (p[2] + t[2]) - pivot[2]}};
        const Vec3 rot = r * rel;
        p[0] = pivot[0] + rot[0]; p[1] = pivot[1] + rot[1]; p[2] = pivot[2] + rot[2];
    }
    void apply_span(double* xyz, int64_t lo, int64_t hi) const
    {
        for (int64_t i = lo; i < hi; ++i) apply(xyz + 3 * i);
    }
};

Vec3 mean_of(const double* xyz, int64_t n)
{
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int64_t i = 0; i < n; ++i) sx += xyz[3 * i];
    for (int64_t i = 0; i < n; ++i) sy += xyz[3 * i + 1];
    for (int64_t i = 0; i < n; ++i) sz += xyz[3 * i + 2];
    return Vec3{{sx / (double)n, sy / (double)n, sz / (double)n}};
}

// calculate_normal (align_algorithms.rs:206-235): Newell's method about `c`
Vec3 newell_normal(const double* xyz, int64_t n, const Vec3& c)
{
    if (n < 3) return Vec3{{0.0, 0.0, 1.0}};
    Vec3 acc{{0.0, 0.0, 0.0}};
    for (int64_t i = 0; i < n; ++i) {
        const double* cur = xyz + 3 * i;
        const double* nxt = xyz + 3 * ((i + 1) % n);
        acc[0] += (cur[1] - c[1]) * (nxt[2] - c[2]) - (cur[2] - c[2]) * (nxt[1] - c[1]);
        acc[1] += (cur[2] - c[2]) * (nxt[0] - c[0]) - (cur[0] - c[0]) * (nxt[2] - c[2]);
        acc[2] += (cur[0] - c[0]) * (nxt[1] - c[1]) - (cur[1] - c[1]) * (nxt[0] - c[0]);
    }
    const double nn = norm(acc);
    if (nn > 1e-12) return Vec3{{acc[0] / nn, acc[1] / nn, acc[2] / nn}};
    return Vec3{{0.0, 0.0, 1.0}};
}

// align_frame (align_algorithms.rs:128-173)
FrameTf align_frame(const double* xyz, int64_t n, bool has_c, const double* c_in, const mm_clpoint& clp)
{
    const Vec3 c = has_c ? Vec3{{c_in[0], c_in[1], c_in[2]}} : mean_of(xyz, n);
    FrameTf tf;
    tf.t = Vec3{{clp.x - c[0], clp.y - c[1], clp.z - c[2]}};
    const Vec3 cur = newell_normal(xyz, n, c), des{{clp.tx, clp.ty, clp.tz}};
    const double ang = angle_between(cur, des);
    tf.r = Mat3::identity();
    if (!(std::fabs(ang) < 1e-6)) {
        const Vec3 axis = cross(cur, des);
        if (!(norm(axis) < 1e-6)) tf.r = Mat3::axis_angle(axis, ang);
    }
    tf.pivot = Vec3{{clp.x, clp.y, clp.z}};
    return tf;
}

// Contour::sort_contour_points (contour.rs:368-405).  `key`/`perm`/`tmp` are caller scratch.
struct SortScratch { std::vector<double> key, tmp; std::vector<int32_t> perm; std::vector<uint8_t> tf; };
void sort_contour(double* xyz, int64_t n, SortScratch& sc, uint8_t* flags = nullptr)   // flags: ContourPoint.aortic
{
    if (n == 0) return;
    double sx = 0.0, sy = 0.0;
    for (int64_t i = 0; i < n; ++i) { sx += xyz[3 * i]; sy += xyz[3 * i + 1]; }
    const double cx = sx / (double)n, cy = sy / (double)n;
    sc.key.resize((size_t)n); sc.perm.resize((size_t)n); sc.tmp.resize((size_t)n * 3);
    for (int64_t i = 0; i < n; ++i) sc.key[i] = std::atan2(xyz[3 * i + 1] - cy, xyz[3 * i] - cx);
    stable_argsort(sc.key.data(), n, sc.perm.data());
    // highest y moves to the front; Iterator::max_by keeps the last of equal maxima
    int64_t start = 0;
    for (int64_t i = 1; i < n; ++i)
        if (!(xyz[3 * sc.perm[i] + 1] < xyz[3 * sc.perm[start] + 1])) start = i;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t src = sc.perm[(size_t)((i + start) % n)];
        sc.tmp[3 * i] = xyz[3 * src]; sc.tmp[3 * i + 1] = xyz[3 * src + 1]; sc.tmp[3 * i + 2] = xyz[3 * src + 2];
    }
    std::memcpy(xyz, sc.tmp.data(), (size_t)n * 24);
    if (flags) {
        sc.tf.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) sc.tf[(size_t)i] = flags[sc.perm[(size_t)((i + start) % n)]];
        std::memcpy(flags, sc.tf.data(), (size_t)n);
    }
}

// ContourPoint::rotate (contour_point.rs:38-52) with the sin/cos pair hoisted out of the loop
inline void rotate_xy_span(double* xyz, int64_t lo, int64_t hi, double s, double c, double cx, double cy)
{
    for (int64_t i = lo; i < hi; ++i) {
        const double x = xyz[3 * i] - cx, y = xyz[3 * i + 1] - cy;
        xyz[3 * i] = x * c - y * s + cx;
        xyz[3 * i + 1] = x * s + y * c + cy;
    }
}

int64_t find_ref_idx(const mm_clpoint* cl, int64_t n, const double ref[3])
{
    int64_t best = 0;
    double best_d = INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        const double dx = cl[i].x - ref[0], dy = cl[i].y - ref[1], dz = cl[i].z - ref[2];
        const double d = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (d < best_d) { best_d = d; best = i; }
    }
    return best;
}

// preprocessing.rs:162-242
mm_clpoint interpolate_at(const std::vector<mm_clpoint>& cl, const std::vector<double>& cum, double target)
{
    const size_t n = cl.size();
    // binary_search_by: Ok(i) -> i, Err(0) -> 0, Err(pos) -> pos - 1
    const size_t ub = (size_t)(std::upper_bound(cum.begin(), cum.end(), target) - cum.begin());
    const size_t idx = ub == 0 ? 0 : ub - 1;
    if (idx >= n - 1) { mm_clpoint o = cl[n - 1]; o.branch_id = 0; return o; }
    const mm_clpoint &p0 = cl[idx], &p1 = cl[idx + 1];
    const double s0 = cum[idx], s1 = cum[idx + 1], denom = s1 - s0;
    const double t = std::fabs(denom) < 1e-12 ? 0.0 : (target - s0) / denom;
    mm_clpoint o{};
    o.x = p0.x + t * (p1.x - p0.x);
    o.y = p0.y + t * (p1.y - p0.y);
    o.z = p0.z + t * (p1.z - p0.z);
    const Vec3 t0{{p0.tx, p0.ty, p0.tz}}, t1{{p1.tx, p1.ty, p1.tz}};
    Vec3 tg{{0.0, 0.0, 0.0}};
    if (norm(t0) > 0.0 || norm(t1) > 0.0) {
        for (int k = 0; k < 3; ++k) tg[k] = t0[k] * (1.0 - t) + t1[k] * t;
        const double tn = norm(tg);
        if (tn > 1e-12) { tg[0] /= tn; tg[1] /= tn; tg[2] /= tn; }
        else tg = Vec3{{0.0, 0.0, 0.0}};
    }
    o.tx = tg[0]; o.ty = tg[1]; o.tz = tg[2];
    o.radius = p0.radius * (1.0 - t) + p1.radius * t;
    o.branch_id = 0;
    return o;
}

int preprocess(const mm_clpoint* cl_in, int64_t n_in, const mm_geometry* mesh, std::vector<mm_clpoint>& out,
               double& spacing)
{
    std::vector<mm_clpoint> cl;
    cl.reserve((size_t)std::max<int64_t>(n_in, 0));
    for (int64_t i = 0; i < n_in; ++i) if (cl_in[i].branch_id == 0) cl.push_back(cl_in[i]);  // :23-27
    if (cl.empty()) return set_error(MM_ERR_INVALID, "Couldn't resample the centerline: Centerline has no branch-0 points");
    if (cl.front().z < cl.back().z) std::reverse(cl.begin(), cl.end());                        // :39-47
    if (!mesh || mesh->n_frames <= 0)
        return set_error(MM_ERR_NO_FRAMES, "Couldn't resample the centerline: Reference mesh has no frames");
    bool has_mean = false;
    double mean = 0.0;
    if (mesh->n_frames >= 2) {                                                                 // :244-280
        double sum = 0.0;
        for (int32_t i = 0; i + 1 < mesh->n_frames; ++i) {
            const double* a = mesh->centroid + 3 * i;
            const double* b = a + 3;
            const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
            sum += std::sqrt(dx * dx + dy * dy + dz * dz);
        }
        mean = sum / (double)(mesh->n_frames - 1);
        has_mean = std::isfinite(mean) && mean > 1e-12;
    }
    std::vector<double> cum(cl.size());                                                        // :110-126
    cum[0] = 0.0;
    for (size_t i = 1; i < cl.size(); ++i) {
        const double dx = cl[i].x - cl[i - 1].x, dy = cl[i].y - cl[i - 1].y, dz = cl[i].z - cl[i - 1].z;
        cum[i] = cum[i - 1] + std::sqrt(dx * dx + dy * dy + dz * dz);
    }
    const double total = cum.back();
    bool ok = false;                                                                           // :128-143
    if (has_mean) { spacing = mean; ok = true; }
    else if (cl.size() >= 2) {
        const double fb = total / (double)(cl.size() - 1);
        if (std::isfinite(fb) && fb > 1e-12) { spacing = fb; ok = true; }
    }
    if (!ok) { out = cl; spacing = 0.0; return MM_OK; }                                        // :70-73
    std::vector<double> s_new;                                                                 // :145-160
    for (double s = 0.0; s <= total + 1e-9; s += spacing) s_new.push_back(s);
    if (!s_new.empty() && s_new.back() > total + 1e-6) s_new.back() = total;
    out.clear(); out.reserve(s_new.size());
    for (double s : s_new) out.push_back(interpolate_at(cl, cum, s));
    return MM_OK;
}

inline bool lumen_centroid_of(const mm_cl_geometry* cg, int32_t i, const double*& c)
{
    if (cg->has_lumen_centroid && cg->lumen_centroid && cg->has_lumen_centroid[i]) { c = cg->lumen_centroid + 3 * i; return true; }
    c = nullptr;
    return false;
}

int check_geoms(mm_cl_geometry** geoms, int n_geoms)
{
    if (!geoms || n_geoms < 1 || n_geoms > 2) return set_error(MM_ERR_INVALID, "expected one geometry or a geometry pair");
    for (int g = 0; g < n_geoms; ++g) {
        if (!geoms[g] || !geoms[g]->g) return set_error(MM_ERR_INVALID, "geometry == NULL");
        const mm_geometry* G = geoms[g]->g;
        if (G->n_frames < 0 || !G->lumen_off || !G->centroid) return set_error(MM_ERR_INVALID, "malformed geometry");
        if (geoms[g]->extra_kind_off && geoms[g]->n_extra_kinds <= 0)
            return set_error(MM_ERR_INVALID, "extra_kind_off given with n_extra_kinds <= 0");
        if (geoms[g]->wall_kind1 < 0 || geoms[g]->wall_kind1 > (geoms[g]->extra_kind_off ? geoms[g]->n_extra_kinds : 1))
            return set_error(MM_ERR_INVALID, "wall_kind1 is not one of the extras kinds");
    }
    return MM_OK;
}

// get_transformations (align_algorithms.rs:96-126)
void frame_transforms(const mm_cl_geometry* prim, const mm_clpoint* cl, int64_t ncl, const double ref_pt[3],
                      std::vector<FrameTf>& tfs)
{
    const mm_geometry* g = prim->g;
    const int64_t ref_idx = find_ref_idx(cl, ncl, ref_pt);
    tfs.clear();
    for (int32_t i = 0; i < g->n_frames; ++i) {
        const int64_t k = ref_idx + i;
        if (k >= ncl) continue;  // "Centerline index out of bounds": the frame keeps its place
        const double* c;
        const bool hc = lumen_centroid_of(prim, i, c);
        tfs.push_back(align_frame(g->lumen + 3 * g->lumen_off[i], g->lumen_off[i + 1] - g->lumen_off[i], hc, c, cl[k]));
    }
}

// apply_transforms_to_geometry (align_algorithms.rs:521-535)
void apply_transforms(mm_cl_geometry* cg, const std::vector<FrameTf>& tfs)
{
    mm_geometry* g = cg->g;
    for (int32_t i = 0; i < g->n_frames && (size_t)i < tfs.size(); ++i) {
        const FrameTf& tf = tfs[(size_t)i];
        tf.apply_span(g->lumen, g->lumen_off[i], g->lumen_off[i + 1]);
        const double* c;
        const bool hc = lumen_centroid_of(cg, i, c);
        if (hc) tf.apply(cg->lumen_centroid + 3 * i);                                          // :186-201
        if (g->cath_off) tf.apply_span(g->cath, g->cath_off[i], g->cath_off[i + 1]);
        if (g->extra_off) tf.apply_span(g->extra, g->extra_off[i], g->extra_off[i + 1]);
        if (g->has_ref && g->has_ref[i]) tf.apply(g->ref + 3 * i);                             // :529-531
        for (int k = 0; k < 3; ++k) g->centroid[3 * i + k] = hc ? cg->lumen_centroid[3 * i + k] : 0.0;  // :532
    }
}

// The Wall contour of frame i: its span in g->extra and its offset into wall_aortic (`wall_at`[i], a prefix sum).
struct WallIndex {
    std::vector<int64_t> at;   // [F+1] wall points before frame i
    int32_t K = 1, k = -1;
    explicit WallIndex(const mm_cl_geometry* cg)
    {
        const mm_geometry* g = cg->g;
        if (!cg->wall_kind1 || !g->extra_off) return;
        K = cg->extra_kind_off ? cg->n_extra_kinds : 1;
        k = cg->wall_kind1 - 1;
        at.assign((size_t)g->n_frames + 1, 0);
        for (int32_t i = 0; i < g->n_frames; ++i) at[(size_t)i + 1] = at[(size_t)i] + (hi(cg, i) - lo(cg, i));
    }
    bool any() const { return k >= 0; }
    int64_t lo(const mm_cl_geometry* cg, int32_t i) const
    { return cg->extra_kind_off ? cg->extra_kind_off[(int64_t)i * K + k] : cg->g->extra_off[i]; }
    int64_t hi(const mm_cl_geometry* cg, int32_t i) const
    { return cg->extra_kind_off ? cg->extra_kind_off[(int64_t)i * K + k + 1] : cg->g->extra_off[i + 1]; }
};

void rotate_geometry(mm_cl_geometry* cg, double angle)
{
    if (angle == 0.0) return;                                                                  // geometry.rs:242-244
    mm_geometry* g = cg->g;
    const WallIndex wi(cg);
    auto wall_flags = [&](int32_t i) -> uint8_t* {
        return (wi.any() && cg->wall_aortic) ? cg->wall_aortic + wi.at[(size_t)i] : nullptr;
    };
    constexpr int kFrames = 8;   // frames are independent: chunks of them over the worker pool
    parallel_for((g->n_frames + kFrames - 1) / kFrames, [&](int c) {
        SortScratch sc;
        for (int32_t i = c * kFrames; i < std::min<int32_t>(g->n_frames, (c + 1) * kFrames); ++i) {
            mm_frame_rotate(g, i, angle, g->centroid[3 * i], g->centroid[3 * i + 1]);          // :246-247
            sort_contour(g->lumen + 3 * g->lumen_off[i], g->lumen_off[i + 1] - g->lumen_off[i], sc,   // frame.rs:123-129
                         cg->lumen_aortic ? cg->lumen_aortic + g->lumen_off[i] : nullptr);
            if (g->cath_off) sort_contour(g->cath + 3 * g->cath_off[i], g->cath_off[i + 1] - g->cath_off[i], sc);
            if (g->extra_off) {
                if (cg->extra_kind_off) {
                    const int32_t K = cg->n_extra_kinds;
                    for (int32_t k = 0; k < K; ++k) {
                        const int64_t lo = cg->extra_kind_off[(int64_t)i * K + k], hi = cg->extra_kind_off[(int64_t)i * K + k + 1];
                        sort_contour(g->extra + 3 * lo, hi - lo, sc, k == wi.k ? wall_flags(i) : nullptr);
                    }
                } else {
                    sort_contour(g->extra + 3 * g->extra_off[i], g->extra_off[i + 1] - g->extra_off[i], sc,
                                 wi.k == 0 ? wall_flags(i) : nullptr);
                }
            }
        }
    });
}

// ---- align_walls (align.rs:381-595): wall twist compensation -----------------------------------------------------
struct WallView { double* p; const uint8_t* aortic; int64_t n; };

// aortic_centroid_direction (:385-407)
bool aortic_direction(const WallView& w, const double* c, Vec3& out)
{
    int64_t m = 0;
    if (w.aortic) for (int64_t i = 0; i < w.n; ++i) m += w.aortic[i] ? 1 : 0;
    if (m == 0) return false;
    const double n = (double)m;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int64_t i = 0; i < w.n; ++i) if (w.aortic[i]) sx += w.p[3 * i];
    for (int64_t i = 0; i < w.n; ++i) if (w.aortic[i]) sy += w.p[3 * i + 1];
    for (int64_t i = 0; i < w.n; ++i) if (w.aortic[i]) sz += w.p[3 * i + 2];
    out = Vec3{{sx / n - c[0], sy / n - c[1], sz / n - c[2]}};
    return !(norm(out) < 1e-9);
}

// wall_major_axis (:410-437): the farthest pair, first maximum in (i, j) order
bool major_axis(const WallView& w, Vec3& out)
{
    if (w.n < 2) return false;
    double best = 0.0;
    int64_t fa = 0, fb = 0;
    for (int64_t i = 0; i < w.n; ++i) {
        const double xi = w.p[3 * i], yi = w.p[3 * i + 1], zi = w.p[3 * i + 2];
        for (int64_t j = i + 1; j < w.n; ++j) {
            const double dx = xi - w.p[3 * j], dy = yi - w.p[3 * j + 1], dz = zi - w.p[3 * j + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 > best) { best = d2; fa = i; fb = j; }
        }
    }
    out = Vec3{{w.p[3 * fb] - w.p[3 * fa], w.p[3 * fb + 1] - w.p[3 * fa + 1], w.p[3 * fb + 2] - w.p[3 * fa + 2]}};
    return !(norm(out) < 1e-9);
}

// project_onto_plane_normalized (:465-472)
bool project_normalized(const Vec3& v, const Vec3& t, Vec3& out)
{
    const double k = dot(v, t);
    const Vec3 p{{v[0] - t[0] * k, v[1] - t[1] * k, v[2] - t[2] * k}};
    const double n = norm(p);
    if (n < 1e-9) return false;
    out = Vec3{{p[0] / n, p[1] / n, p[2] / n}};
    return true;
}

// parallel_transport (:476-492)
Vec3 parallel_transport(const Vec3& v, const Vec3& tf, const Vec3& tt)
{
    const double ang = angle_between(tf, tt);
    if (ang < 1e-9) return v;
    const Vec3 axis = cross(tf, tt);
    if (norm(axis) < 1e-9) {                       // anti-parallel tangents: half a turn about a perpendicular
        Vec3 perp = std::fabs(tf[0]) < 0.9 ? Vec3{{1.0 - tf[0] * tf[0], 0.0 - tf[1] * tf[0], 0.0 - tf[2] * tf[0]}}
                                           : Vec3{{0.0 - tf[0] * tf[1], 1.0 - tf[1] * tf[1], 0.0 - tf[2] * tf[1]}};
        const double n = norm(perp);
        perp = Vec3{{perp[0] / n, perp[1] / n, perp[2] / n}};
        return Mat3::axis_angle(perp, kPiCl) * v;
    }
    return Mat3::axis_angle(axis, ang) * v;
}

// signed_angle_around_axis (:495-497)
inline double signed_angle(const Vec3& from, const Vec3& to, const Vec3& axis)
{
    return std::atan2(dot(cross(from, to), axis), dot(from, to));
}

// align_walls_on_geometry (:507-584).  The lumen normals (Newell about the frame centroid, :440-461) and the wall
// directions of the frames are independent of each other and of the walk: they are computed over the worker pool
// (the farthest-pair scan is the O(n^2) part), then the transported direction is walked frame by frame.
void align_walls_on_geometry(mm_cl_geometry* cg)
{
    mm_geometry* g = cg->g;
    const int32_t F = g->n_frames;
    if (F < 1) return;
    const WallIndex wi(cg);
    if (!wi.any()) return;                                        // no frame has a Wall contour (:512-515)
    auto wall_of = [&](int32_t i) {
        const int64_t lo = wi.lo(cg, i), hi = wi.hi(cg, i);
        return WallView{g->extra + 3 * lo, cg->wall_aortic ? cg->wall_aortic + wi.at[(size_t)i] : nullptr, hi - lo};
    };
    std::vector<Vec3> normal((size_t)F), dir((size_t)F);
    std::vector<uint8_t> kind((size_t)F, 0);                      // 0 = no usable direction, 1 = aortic side, 2 = major axis
    parallel_for(F, [&](int i) {
        const double* c = g->centroid + 3 * i;
        normal[(size_t)i] = newell_normal(g->lumen + 3 * g->lumen_off[i], g->lumen_off[i + 1] - g->lumen_off[i],
                                          Vec3{{c[0], c[1], c[2]}});
        const WallView w = wall_of(i);
        if (w.n <= 0) return;
        if (aortic_direction(w, c, dir[(size_t)i])) kind[(size_t)i] = 1;
        else if (major_axis(w, dir[(size_t)i])) kind[(size_t)i] = 2;
    });
    if (wall_of(0).n <= 0 || !kind[0]) return;                    // :512-520
    Vec3 u;
    if (!project_normalized(dir[0], normal[0], u)) return;        // :521-524
    for (int32_t i = 1; i < F; ++i) {
        const Vec3& tc = normal[(size_t)i];
        u = parallel_transport(u, normal[(size_t)i - 1], tc);     // :531 (kept when the projection below fails)
        Vec3 pu;
        if (!project_normalized(u, tc, pu)) continue;             // :532-535
        u = pu;
        if (!kind[(size_t)i]) continue;                           // no wall / no direction (:540-550)
        Vec3 v;
        if (!project_normalized(dir[(size_t)i], tc, v)) continue; // :552-555
        double ang;
        if (kind[(size_t)i] == 1) ang = signed_angle(v, u, tc);   // :557-559
        else {                                                    // :560-569 the sign of a major axis is ambiguous
            const double a1 = signed_angle(v, u, tc), a2 = signed_angle(Vec3{{-v[0], -v[1], -v[2]}}, u, tc);
            ang = std::fabs(a1) <= std::fabs(a2) ? a1 : a2;
        }
        if (std::fabs(ang) < 1e-6) continue;                      // :571-573
        const Mat3 r = Mat3::axis_angle(tc, ang);                 // :575-587
        const double* c = g->centroid + 3 * i;
        const WallView w = wall_of(i);
        for (int64_t k = 0; k < w.n; ++k) {
            double* q = w.p + 3 * k;
            const Vec3 rot = r * Vec3{{q[0] - c[0], q[1] - c[1], q[2] - c[2]}};
            q[0] = c[0] + rot[0]; q[1] = c[1] + rot[1]; q[2] = c[2] + rot[2];
        }
    }
}

// align_walls (:589-595)
void align_walls(mm_cl_geometry** geoms, int n_geoms, bool anomalous)
{
    if (!anomalous || geoms[0]->g->n_frames < 2) return;
    for (int g = 0; g < n_geoms; ++g) align_walls_on_geometry(geoms[g]);
}

// best_rotation_three_point (align_algorithms.rs:263-336)
double three_point(const double* lumen, int64_t n, bool has_c, const double* c_in, uint32_t index_reference,
                   const double pm[3], const double pc[3], const double pw[3], double step, const mm_clpoint& clp)
{
    double best_angle = 0.0, min_err = DBL_MAX;
    std::vector<double> tmp((size_t)n * 3);
    const int64_t i_cw = n / 2;
    for (double angle = 0.0; angle < kTau; angle += step) {
        std::memcpy(tmp.data(), lumen, (size_t)n * 24);
        // rotate_contour_around_centroid (:238-259): about the contour's normal through its centroid
        const Vec3 c = has_c ? Vec3{{c_in[0], c_in[1], c_in[2]}} : mean_of(tmp.data(), n);
        const Mat3 r = Mat3::axis_angle(newell_normal(tmp.data(), n, c), angle);
        for (int64_t i = 0; i < n; ++i) {
            double* p = tmp.data() + 3 * i;
            const Vec3 rot = r * Vec3{{p[0] - c[0], p[1] - c[1], p[2] - c[2]}};
            p[0] = c[0] + rot[0]; p[1] = c[1] + rot[1]; p[2] = c[2] + rot[2];
        }
        const FrameTf tf = align_frame(tmp.data(), n, has_c, c_in, clp);                        // :294
        // only the three landmark points are read afterwards (:301-311)
        double q[3][3];
        const int64_t pick[3] = {(int64_t)index_reference, 0, i_cw};
        const double* tgt[3] = {pm, pc, pw};
        double err = 0.0;
        for (int k = 0; k < 3; ++k) {
            std::memcpy(q[k], tmp.data() + 3 * pick[k], 24);
            tf.apply(q[k]);
            const Vec3 d{{tgt[k][0] - q[k][0], tgt[k][1] - q[k][1], tgt[k][2] - q[k][2]}};
            const double dist = norm(d);
            err = k == 0 ? dist * dist : err + dist * dist;                                    // :326
        }
        if (err < min_err) { min_err = err; best_angle = angle; }
    }
    return best_angle;
}

// filter_points_in_region (align_algorithms.rs:454-505): x,y of the points inside the box
void filter_region(const double* pts, int64_t n, const mm_clpoint& a, const mm_clpoint& b,
                   std::vector<double>& fx, std::vector<double>& fy)
{
    const double margin = 5.0;
    const double min_x = std::fmin(a.x, b.x) - margin, max_x = std::fmax(a.x, b.x) + margin;
    const double min_y = std::fmin(a.y, b.y) - margin, max_y = std::fmax(a.y, b.y) + margin;
    const double min_z = std::fmin(a.z, b.z) - margin, max_z = std::fmax(a.z, b.z) + margin;
    fx.clear(); fy.clear();
    for (int64_t i = 0; i < n; ++i) {
        const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (x >= min_x && x <= max_x && y >= min_y && y <= max_y && z >= min_z && z <= max_z) { fx.push_back(x); fy.push_back(y); }
    }
}

struct Candidate { int32_t group; double angle; int64_t cl_idx; int64_t off, n; };

// refine_alignment_hausdorff (align_algorithms.rs:339-451)
int refine(Engine* e, mm_cl_geometry** geoms, const mm_clpoint* cl, int64_t ncl, int64_t initial_idx,
           double initial_rotation, const double* pts, int64_t n_pts, double ang_range, double ang_step,
           int64_t idx_range, double& best_angle, int64_t& best_idx, double& min_h,
           double* all_costs, int64_t cap, int64_t* n_evals)
{
    const mm_cl_geometry* prim = geoms[0];
    const mm_geometry* g = prim->g;
    const int64_t F = g->n_frames;
    best_angle = initial_rotation; best_idx = initial_idx; min_h = DBL_MAX;
    if (n_evals) *n_evals = 0;
    if (F <= 0) return set_error(MM_ERR_NO_FRAMES, "refine_alignment_hausdorff: geometry has no frames");
    if (!(ang_step > 0.0)) return set_error(MM_ERR_INVALID, "refine_alignment_hausdorff: angle_step must be > 0");
    const int64_t m = g->lumen_off[1] - g->lumen_off[0];                                       // :412
    TraceTimer tt_ref("refine: total");

    // candidate list in the reference's evaluation order; only the primary geometry's lumen
    // enters the cost (:411-431), so only that is rebuilt per candidate
    struct Group { int64_t cl_idx; std::vector<double> fx, fy; int64_t n_down; };
    std::vector<Group> groups;
    std::vector<Candidate> cands;
    const int64_t lo = idx_range == 0 ? 0 : -idx_range, hi = idx_range == 0 ? 0 : idx_range;  // :363-367
    int64_t flat_total = 0;
    for (int64_t d = lo; d <= hi; ++d) {
        const int64_t cur = initial_idx + d;
        if (cur < 0) continue;                                                                 // :371-373
        if (cur + F >= ncl) continue;                                                          // :376-378
        Group grp;
        grp.cl_idx = cur;
        filter_region(pts, n_pts, cl[cur], cl[cur + F - 1], grp.fx, grp.fy);                   // :400-404
        if (grp.fx.empty()) continue;                                                          // :406-409 (every angle skipped)
        const double ratio = (double)grp.fx.size() / ((double)m * (double)F);                  // :415-418
        const double nd = std::ceil(ratio * (double)m);
        int64_t n_down = !(nd > 0.0) ? 0 : (nd >= 9.2e18 ? INT64_MAX : (int64_t)nd);          // `as usize` saturates
        n_down = std::max<int64_t>(1, std::min<int64_t>(n_down, m));
        grp.n_down = n_down;
        int64_t per_cand = 0;
        for (int64_t f = 0; f < F; ++f) {
            const int64_t len = g->lumen_off[f + 1] - g->lumen_off[f];
            per_cand += (n_down < m) ? std::min(len, n_down) : len;                            // :420-428, contour.rs:47-58
        }
        const int32_t gi = (int32_t)groups.size();
        groups.push_back(std::move(grp));
        for (double a = initial_rotation - ang_range; a <= initial_rotation + ang_range; a += ang_step) {  // :386-387, 439
            cands.push_back(Candidate{gi, a, cur, flat_total, per_cand});
            flat_total += per_cand;
            if (cands.size() > (size_t)1 << 22) return set_error(MM_ERR_TOO_LARGE, "refine grid has more than 2^22 candidates");
        }
    }
    if (cands.empty()) return MM_OK;
    if (flat_total > (int64_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "refine grid exceeds 2^30 points");

    // Rebuild the placed frames of every candidate on the host.  rotate_by_best_rotation (:395,
    // geometry.rs:241-250: rotate every frame about its centroid, re-sort its points) depends on the
    // angle only, not on the index shift, so it is done once per (angle, frame) and shared by the
    // 2R+1 index shifts; the placement (align_frame + apply, :394-398) and the downsampling are per
    // candidate.  Both phases run over the worker pool.
    double* flat_x = e->scratch_f64(0, (size_t)flat_total);
    double* flat_y = e->scratch_f64(1, (size_t)flat_total);
    std::vector<double> angles;
    for (double a = initial_rotation - ang_range; a <= initial_rotation + ang_range; a += ang_step) angles.push_back(a);
    const int64_t NL = g->lumen_off[F];
    const size_t n_ang = angles.size();
    double* rotated = e->scratch_f64(2, n_ang * (size_t)NL * 3);
    {
        TraceTimer tt("refine: rotate + sort");
        parallel_for((int)(n_ang * (size_t)F), [&](int job) {
            const size_t ai = (size_t)job / (size_t)F;
            const int64_t f = (int64_t)((size_t)job % (size_t)F);
            const int64_t len = g->lumen_off[f + 1] - g->lumen_off[f];
            double* p = rotated + (ai * (size_t)NL + (size_t)g->lumen_off[f]) * 3;
            std::memcpy(p, g->lumen + 3 * g->lumen_off[f], (size_t)len * 24);
            const double ang = angles[ai];
            if (ang != 0.0) {                                                                 // geometry.rs:242-244
                double s, c;
                sin_cos(ang, s, c);
                rotate_xy_span(p, 0, len, s, c, g->centroid[3 * f], g->centroid[3 * f + 1]);
                SortScratch sc;
                sort_contour(p, len, sc);
            }
        });
    }
    {
        TraceTimer tt("refine: place + sample");
        constexpr int64_t kFrames = 8;   // frames per job
        const int64_t fchunks = (F + kFrames - 1) / kFrames;
        // write offset of frame f inside a candidate's flat set
        std::vector<std::vector<int64_t>> frame_off(groups.size(), std::vector<int64_t>((size_t)F + 1, 0));
        for (size_t gi = 0; gi < groups.size(); ++gi)
            for (int64_t f = 0; f < F; ++f) {
                const int64_t len = g->lumen_off[f + 1] - g->lumen_off[f];
                frame_off[gi][(size_t)f + 1] = frame_off[gi][(size_t)f] + ((groups[gi].n_down < m) ? std::min(len, groups[gi].n_down) : len);
            }
        parallel_for((int)(cands.size() * (size_t)fchunks), [&](int job) {
            const size_t ci = (size_t)job / (size_t)fchunks;
            const int64_t f0 = (int64_t)((size_t)job % (size_t)fchunks) * kFrames, f1 = std::min(F, f0 + kFrames);
            const Candidate& cd = cands[ci];
            const Group& grp = groups[(size_t)cd.group];
            const size_t ai = ci % n_ang;   // candidates are (group-major, angle-minor)
            for (int64_t f = f0; f < f1; ++f) {
                const int64_t len = g->lumen_off[f + 1] - g->lumen_off[f];
                const double* p = rotated + (ai * (size_t)NL + (size_t)g->lumen_off[f]) * 3;
                // cl_segment = points[cur .. cur+F), ref_pt = its first point (:381-392): frame f -> cl[cur + f]
                const double* lc;
                const bool hc = lumen_centroid_of(prim, (int32_t)f, lc);
                const FrameTf tf = align_frame(p, len, hc, lc, cl[cd.cl_idx + f]);
                int64_t w = cd.off + frame_off[(size_t)cd.group][(size_t)f];
                auto emit = [&](int64_t src) {
                    double q[3] = {p[3 * src], p[3 * src + 1], p[3 * src + 2]};
                    tf.apply(q);
                    flat_x[(size_t)w] = q[0]; flat_y[(size_t)w] = q[1]; ++w;
                };
                if (grp.n_down < m && len > grp.n_down) {                                     // downsample_contour_points
                    const double stepf = (double)len / (double)grp.n_down;
                    for (int64_t i = 0; i < grp.n_down; ++i) emit((int64_t)((double)i * stepf));
                } else {
                    for (int64_t i = 0; i < len; ++i) emit(i);
                }
            }
        });
    }

    // one device batch: hausdorff_distance(filtered points of the group, placed frames), x,y only (:431)
    std::vector<SetRef> sets;
    std::vector<std::array<int32_t, 2>> pr;
    sets.reserve(groups.size() + cands.size());
    for (const Group& grp : groups) sets.push_back(SetRef{grp.fx.data(), grp.fy.data(), (int32_t)grp.fx.size(), 0.0, 0.0});
    for (const Candidate& cd : cands) {
        pr.push_back({cd.group, (int32_t)sets.size()});
        sets.push_back(SetRef{flat_x + cd.off, flat_y + cd.off, (int32_t)cd.n, 0.0, 0.0});
    }
    int rc;
    if (all_costs) {
        std::vector<double> cost(cands.size());
        {
            TraceTimer tt("refine: device batch");
            rc = hausdorff_sets(e, sets, pr, cost.data());
        }
        if (rc) return rc;
        for (size_t ci = 0; ci < cands.size(); ++ci) {
            if ((int64_t)ci < cap) all_costs[ci] = cost[ci];
            if (cost[ci] < min_h) { min_h = cost[ci]; best_angle = cands[ci].angle; best_idx = cands[ci].cl_idx; }  // :433-437
        }
    } else {
        // only the winner is asked for: the same strict-'<' first minimum, most candidates ruled out by
        // lower bounds instead of evaluated (hausdorff_sets_first_min)
        int32_t bi = -1; double bc = INFINITY;
        {
            TraceTimer tt("refine: device batch");
            rc = hausdorff_sets_first_min(e, sets, pr, &bi, &bc, nullptr);
        }
        if (rc) return rc;
        if (bi >= 0 && bc < min_h) { min_h = bc; best_angle = cands[(size_t)bi].angle; best_idx = cands[(size_t)bi].cl_idx; }
    }
    if (n_evals) *n_evals = (int64_t)cands.size();
    return MM_OK;
}

int find_ref_frame(const mm_geometry* g, int64_t& idx)
{
    for (int32_t i = 0; i < g->n_frames; ++i)
        if (g->has_ref && g->has_ref[i]) {                                                     // geometry.rs:62-69
            idx = (int64_t)g->id[i];
            if (idx >= g->n_frames) return set_error(MM_ERR_REF_INDEX, "reference frame id used as index is out of range");
            return MM_OK;
        }
    return set_error(MM_ERR_INVALID, "Couldn't find ref frame idx: No reference point found in any frame");
}

int three_point_initial(const std::vector<mm_clpoint>& rcl, mm_cl_geometry** geoms, uint32_t ref_point_index,
                        const double pm[3], const double pc[3], const double pw[3], double step,
                        int64_t& cl_ref_idx, double& rot)
{
    const mm_cl_geometry* prim = geoms[0];
    const mm_geometry* g = prim->g;
    int64_t ref_idx = 0;
    int rc = find_ref_frame(g, ref_idx);                                                       // align.rs:82-85
    if (rc) return rc;
    if (!(g->has_ref && g->has_ref[ref_idx])) return set_error(MM_ERR_INVALID, "missing reference point");  // :86-89
    const int64_t n = g->lumen_off[ref_idx + 1] - g->lumen_off[ref_idx];
    if ((int64_t)ref_point_index >= n || n == 0)
        return set_error(MM_ERR_INVALID, "reference point index is not a point of the reference frame's lumen");
    if (!(step > 0.0)) return set_error(MM_ERR_INVALID, "angle_step must be > 0");
    cl_ref_idx = find_ref_idx(rcl.data(), (int64_t)rcl.size(), pm);                            // :90
    const double* lc;
    const bool hc = lumen_centroid_of(prim, (int32_t)ref_idx, lc);
    rot = three_point(g->lumen + 3 * g->lumen_off[ref_idx], n, hc, lc, ref_point_index, pm, pc, pw, step,
                      rcl[(size_t)cl_ref_idx]);                                                // :92-100
    return MM_OK;
}

}  // namespace
}  // namespace mm

using namespace mm;

extern "C" {

int mm_centerline_from_points(const double* xyz, int64_t n, mm_clpoint* out)
{
    if (n < 0 || (n > 0 && (!xyz || !out))) return set_error(MM_ERR_INVALID, "mm_centerline_from_points: bad arguments");
    if (n == 1) return set_error(MM_ERR_INVALID, "a centerline needs at least two points");
    for (int64_t i = 0; i < n; ++i) {
        mm_clpoint& o = out[i];
        o.x = xyz[3 * i]; o.y = xyz[3 * i + 1]; o.z = xyz[3 * i + 2];
        if (i + 1 < n) {                                                                       // centerline.rs:20-22
            const Vec3 d{{xyz[3 * (i + 1)] - o.x, xyz[3 * (i + 1) + 1] - o.y, xyz[3 * (i + 1) + 2] - o.z}};
            const double nn = norm(d);
            o.tx = d[0] / nn; o.ty = d[1] / nn; o.tz = d[2] / nn;
        } else { o.tx = out[i - 1].tx; o.ty = out[i - 1].ty; o.tz = out[i - 1].tz; }          // :23-24
        o.radius = 0.0; o.branch_id = 0; o.pad_ = 0;
    }
    return MM_OK;
}

int64_t mm_centerline_find_ref_idx(const mm_clpoint* cl, int64_t n, const double ref[3])
{
    if (!cl || !ref || n <= 0) return 0;
    return find_ref_idx(cl, n, ref);
}

int64_t mm_centerline_preprocess(const mm_clpoint* cl, int64_t n, const mm_geometry* ref_mesh, mm_clpoint* out,
                                 int64_t cap, double* spacing)
{
    if (n < 0 || (n > 0 && !cl)) return set_error(MM_ERR_INVALID, "mm_centerline_preprocess: bad arguments");
    std::vector<mm_clpoint> res;
    double sp = 0.0;
    int rc = preprocess(cl, n, ref_mesh, res, sp);
    if (rc) return rc;
    if (spacing) *spacing = sp;
    if (out) for (int64_t i = 0; i < (int64_t)res.size() && i < cap; ++i) out[i] = res[(size_t)i];
    return (int64_t)res.size();
}

int mm_sort_contour_points(double* xyz, int64_t n)
{
    if (n < 0 || (n > 0 && !xyz)) return set_error(MM_ERR_INVALID, "mm_sort_contour_points: bad arguments");
    SortScratch sc;
    sort_contour(xyz, n, sc);
    return MM_OK;
}

int mm_rotate_geometry(mm_cl_geometry* g, double angle)
{
    mm_cl_geometry* arr[1] = {g};
    int rc = check_geoms(arr, 1);
    if (rc) return rc;
    rotate_geometry(g, angle);
    return MM_OK;
}

int mm_align_walls(mm_cl_geometry** geoms, int n_geoms, int anomalous)
{
    int rc = check_geoms(geoms, n_geoms);
    if (rc) return rc;
    align_walls(geoms, n_geoms, anomalous != 0);
    return MM_OK;
}

int64_t mm_apply_transformations(mm_cl_geometry** geoms, int n_geoms, const mm_clpoint* cl, int64_t ncl,
                                 const double ref_pt[3])
{
    int rc = check_geoms(geoms, n_geoms);
    if (rc) return rc;
    if (!cl || ncl <= 0 || !ref_pt) return set_error(MM_ERR_INVALID, "mm_apply_transformations: empty centerline");
    std::vector<FrameTf> tfs;
    frame_transforms(geoms[0], cl, ncl, ref_pt, tfs);
    for (int g = 0; g < n_geoms; ++g) apply_transforms(geoms[g], tfs);
    return (int64_t)tfs.size();
}

int mm_best_rotation_three_point(const double* lumen_xyz, int64_t n, int has_centroid, const double centroid[3],
                                 uint32_t index_reference, const double p_main[3], const double p_ccw[3],
                                 const double p_cw[3], double angle_step, const mm_clpoint* clp, double* best_angle)
{
    if (!lumen_xyz || n <= 0 || !p_main || !p_ccw || !p_cw || !clp || !best_angle || (has_centroid && !centroid))
        return set_error(MM_ERR_INVALID, "mm_best_rotation_three_point: bad arguments");
    if ((int64_t)index_reference >= n) return set_error(MM_ERR_INVALID, "index_reference is not a point of the contour");
    if (!(angle_step > 0.0)) return set_error(MM_ERR_INVALID, "angle_step must be > 0");
    *best_angle = three_point(lumen_xyz, n, has_centroid != 0, centroid, index_reference, p_main, p_ccw, p_cw,
                              angle_step, *clp);
    return MM_OK;
}

int mm_refine_alignment_hausdorff(mm_engine* h, mm_cl_geometry** geoms, int n_geoms, const mm_clpoint* cl,
                                  int64_t ncl, int64_t initial_cl_ref_idx, double initial_rotation,
                                  const double* points_xyz, int64_t n_points, double angle_search_range,
                                  double angle_step, int64_t index_search_range, double* best_angle,
                                  int64_t* best_idx, double* min_hausdorff, double* all_costs, int64_t cap,
                                  int64_t* n_evals)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    int rc = check_geoms(geoms, n_geoms);
    if (rc) return rc;
    if (!cl || ncl <= 0 || n_points < 0 || (n_points > 0 && !points_xyz) || index_search_range < 0 ||
        initial_cl_ref_idx < 0 || !best_angle || !best_idx)
        return set_error(MM_ERR_INVALID, "mm_refine_alignment_hausdorff: bad arguments");
    { const hipError_t he = hipSetDevice(e->device); if (he != hipSuccess) return hip_error(he, "hipSetDevice"); }
    double ba, mh; int64_t bi;
    rc = refine(e, geoms, cl, ncl, initial_cl_ref_idx, initial_rotation, points_xyz, n_points, angle_search_range,
                angle_step, index_search_range, ba, bi, mh, all_costs, cap, n_evals);
    if (rc) return rc;
    *best_angle = ba; *best_idx = bi;
    if (min_hausdorff) *min_hausdorff = mh;
    return MM_OK;
}

int mm_align_three_point(const mm_clpoint* cl, int64_t ncl, mm_cl_geometry** geoms, int n_geoms,
                         uint32_t ref_point_index, const double p_main[3], const double p_ccw[3],
                         const double p_cw[3], double angle_step, int align_wall_anomalous, double* spacing,
                         double* total_rotation)
{
    int rc = check_geoms(geoms, n_geoms);
    if (rc) return rc;
    if (!p_main || !p_ccw || !p_cw) return set_error(MM_ERR_INVALID, "mm_align_three_point: bad arguments");
    std::vector<mm_clpoint> rcl;
    double sp = 0.0;
    if ((rc = preprocess(cl, ncl, geoms[0]->g, rcl, sp))) return rc;                           // align.rs:78-80
    int64_t cl_ref_idx = 0; double rot = 0.0;
    if ((rc = three_point_initial(rcl, geoms, ref_point_index, p_main, p_ccw, p_cw, angle_step, cl_ref_idx, rot))) return rc;
    for (int g = 0; g < n_geoms; ++g) rotate_geometry(geoms[g], rot);                          // :102
    std::vector<FrameTf> tfs;
    frame_transforms(geoms[0], rcl.data(), (int64_t)rcl.size(), p_main, tfs);                  // :103
    for (int g = 0; g < n_geoms; ++g) apply_transforms(geoms[g], tfs);
    align_walls(geoms, n_geoms, align_wall_anomalous != 0);                                    // :105-107 / :147-149
    if (spacing) *spacing = sp;
    if (total_rotation) *total_rotation = rot;
    return MM_OK;
}

int mm_align_manual(const mm_clpoint* cl, int64_t ncl, mm_cl_geometry** geoms, int n_geoms, double rotation_angle_deg,
                    const double ref_pt[3], int align_wall_anomalous, double* spacing, double* total_rotation)
{
    int rc = check_geoms(geoms, n_geoms);
    if (rc) return rc;
    if (!ref_pt) return set_error(MM_ERR_INVALID, "mm_align_manual: bad arguments");
    std::vector<mm_clpoint> rcl;
    double sp = 0.0;
    if ((rc = preprocess(cl, ncl, geoms[0]->g, rcl, sp))) return rc;                           // align.rs:139-141
    const double rot = rotation_angle_deg * (kPiCl / 180.0);                                   // :143
    for (int g = 0; g < n_geoms; ++g) rotate_geometry(geoms[g], rot);                          // :144
    std::vector<FrameTf> tfs;
    frame_transforms(geoms[0], rcl.data(), (int64_t)rcl.size(), ref_pt, tfs);                  // :145
    for (int g = 0; g < n_geoms; ++g) apply_transforms(geoms[g], tfs);
    align_walls(geoms, n_geoms, align_wall_anomalous != 0);                                    // :105-107 / :147-149
    if (spacing) *spacing = sp;
    if (total_rotation) *total_rotation = rot;
    return MM_OK;
}

int mm_align_combined(mm_engine* h, const mm_clpoint* cl, int64_t ncl, mm_cl_geometry** geoms, int n_geoms,
                      uint32_t ref_point_index, const double p_main[3], const double p_ccw[3], const double p_cw[3],
                      const double* points_xyz, int64_t n_points, double angle_step, double refine_angle_range,
                      int64_t refine_index_range, int align_wall_anomalous, double* spacing, double* total_rotation,
                      int64_t* refined_idx, int64_t* n_evals)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    int rc = check_geoms(geoms, n_geoms);
    if (rc) return rc;
    if (!p_main || !p_ccw || !p_cw || n_points < 0 || (n_points > 0 && !points_xyz) || refine_index_range < 0)
        return set_error(MM_ERR_INVALID, "mm_align_combined: bad arguments");
    { const hipError_t he = hipSetDevice(e->device); if (he != hipSuccess) return hip_error(he, "hipSetDevice"); }
    TraceTimer tt_all("combined: total (C ABI)");
    std::vector<mm_clpoint> rcl;
    double sp = 0.0;
    if ((rc = preprocess(cl, ncl, geoms[0]->g, rcl, sp))) return rc;                           // align.rs:191-195
    int64_t initial_idx = 0; double initial_rotation = 0.0;
    {
        TraceTimer tt("combined: three-point sweep");
        if ((rc = three_point_initial(rcl, geoms, ref_point_index, p_main, p_ccw, p_cw, angle_step, initial_idx,
                                      initial_rotation))) return rc;                           // :197-217
    }

    // `aligned` = the primary geometry rotated by the initial rotation and placed (:219-223); only its
    // lumen, frame centroids and lumen centroids are read by the refinement
    const mm_geometry* G = geoms[0]->g;
    const int64_t F = G->n_frames, NL = G->lumen_off[F];
    std::vector<double> a_lumen(G->lumen, G->lumen + 3 * NL), a_centroid(G->centroid, G->centroid + 3 * F);
    std::vector<double> a_lc;
    std::vector<uint8_t> a_hlc;
    mm_geometry ag = *G;
    ag.lumen = a_lumen.data(); ag.centroid = a_centroid.data();
    ag.has_catheter = 0; ag.cath_off = nullptr; ag.cath = nullptr; ag.extra_off = nullptr; ag.extra = nullptr;
    ag.has_ref = nullptr; ag.ref = nullptr;
    mm_cl_geometry acg{};                        // no extras, no flags: only lumen and centroids are read
    acg.g = &ag;
    if (geoms[0]->has_lumen_centroid && geoms[0]->lumen_centroid) {
        a_hlc.assign(geoms[0]->has_lumen_centroid, geoms[0]->has_lumen_centroid + F);
        a_lc.assign(geoms[0]->lumen_centroid, geoms[0]->lumen_centroid + 3 * F);
        acg.has_lumen_centroid = a_hlc.data(); acg.lumen_centroid = a_lc.data();
    }
    std::vector<FrameTf> tfs;
    {
        TraceTimer tt("combined: initial placement");
        rotate_geometry(&acg, initial_rotation);
        frame_transforms(&acg, rcl.data(), (int64_t)rcl.size(), p_main, tfs);
        apply_transforms(&acg, tfs);
    }

    mm_cl_geometry* ap[1] = {&acg};
    double delta = 0.0, mh = 0.0; int64_t ridx = initial_idx;
    if ((rc = refine(e, ap, rcl.data(), (int64_t)rcl.size(), initial_idx, 0.0, points_xyz, n_points, refine_angle_range,
                     angle_step, refine_index_range, delta, ridx, mh, nullptr, 0, n_evals))) return rc;  // :228-237
    TraceTimer tt_final("combined: final placement");
    const double total = initial_rotation + delta;                                             // :239
    const double ref_pt[3] = {rcl[(size_t)ridx].x, rcl[(size_t)ridx].y, rcl[(size_t)ridx].z};  // :248-258
    for (int g = 0; g < n_geoms; ++g) rotate_geometry(geoms[g], total);                        // :260-264
    frame_transforms(geoms[0], rcl.data(), (int64_t)rcl.size(), ref_pt, tfs);
    for (int g = 0; g < n_geoms; ++g) apply_transforms(geoms[g], tfs);
    align_walls(geoms, n_geoms, align_wall_anomalous != 0);                                    // :266-268
    if (spacing) *spacing = sp;
    if (total_rotation) *total_rotation = total;
    if (refined_idx) *refined_idx = ridx;
    return MM_OK;
}

}  // extern "C"
