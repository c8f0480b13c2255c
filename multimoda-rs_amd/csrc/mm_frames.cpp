// mm_frames.cpp -- the bookkeeping around the searches on a frame-list model (include/mm_build.h, `mm_frames`):
//   * the post-steps of align_frames_in_geometry      src/intravascular/processing/align_within.rs:136-160,249-328
//       hole filling                                   align_within.rs:330-653
//       wall synthesis                                 src/intravascular/processing/wall.rs:7-213
//       3-frame smoothing                              src/types/native/geometry.rs:165-239
//       rotate_geometry + sort_frame_points            geometry.rs:241-250, frame.rs:40-63,123-129, contour.rs:368-405
//   * postprocess_geom_pair                            src/intravascular/processing/postprocessing.rs:12-476
// Host f64 in the reference's operation order (built with -ffp-contract=off); nothing here enters a search.
// The value types mirror types/native/{contour,frame,geometry}.rs: a Contour carries an optional centroid and
// optional thicknesses, a Frame its lumen, up to five extras contours and an optional reference point.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/mm_build.h"
#include "mm_engine.h"
#include "mm_pool.h"
#include "mm_trace.h"
#include "mm_sort.h"

namespace mm {
namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kEps = 2.220446049250313e-16;   // f64::EPSILON
enum { K_EEM = 0, K_CALC = 1, K_SIDE = 2, K_CATH = 3, K_WALL = 4, K_N = 5 };

struct FContour {
    uint32_t id = 0, orig = 0;
    std::vector<double> p;                 // n x 3
    bool has_cen = false; double cen[3] = {0, 0, 0};
    bool has_a = false, has_p = false; double a_th = 0.0, p_th = 0.0;
    std::vector<uint8_t> aortic;           // n flags (ContourPoint.aortic)
    int64_t n() const { return (int64_t)(p.size() / 3); }
    void compute_centroid()                // contour.rs:213-224; None for an empty contour
    {
        const int64_t m = n();
        if (m == 0) { has_cen = false; return; }
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (int64_t i = 0; i < m; ++i) { sx += p[3 * i]; sy += p[3 * i + 1]; sz += p[3 * i + 2]; }
        cen[0] = sx / (double)m; cen[1] = sy / (double)m; cen[2] = sz / (double)m; has_cen = true;
    }
    void truncate(int64_t m) { p.resize((size_t)m * 3); aortic.resize((size_t)m); }
};

struct FFrame {
    uint32_t id = 0;
    double c[3] = {0, 0, 0};
    FContour lumen;
    bool has[K_N] = {false, false, false, false, false};
    FContour ext[K_N];
    bool has_ref = false; double ref[3] = {0, 0, 0};
    void set_ids(uint32_t i) { id = i; lumen.id = i; for (int k = 0; k < K_N; ++k) if (has[k]) ext[k].id = i; }
    void set_z(double z)                   // Frame::set_value(None, None, None, Some(z)) (frame.rs:96-116)
    {
        for (int64_t i = 0, m = lumen.n(); i < m; ++i) lumen.p[3 * i + 2] = z;
        if (lumen.has_cen) lumen.cen[2] = z;
        for (int k = 0; k < K_N; ++k) if (has[k]) {
            for (int64_t i = 0, m = ext[k].n(); i < m; ++i) ext[k].p[3 * i + 2] = z;
            if (ext[k].has_cen) ext[k].cen[2] = z;
        }
        if (has_ref) ref[2] = z;
        c[2] = z;
    }
    void translate(double dx, double dy, double dz)   // Frame::translate (frame.rs:18-38)
    {
        auto mv = [&](FContour& ct) {
            for (int64_t i = 0, m = ct.n(); i < m; ++i) { ct.p[3 * i] += dx; ct.p[3 * i + 1] += dy; ct.p[3 * i + 2] += dz; }
            ct.compute_centroid();
        };
        mv(lumen);
        for (int k = 0; k < K_N; ++k) if (has[k]) mv(ext[k]);
        if (has_ref) { ref[0] += dx; ref[1] += dy; ref[2] += dz; }
        c[0] += dx; c[1] += dy; c[2] += dz;
    }
};

}  // namespace

struct Frames { std::vector<FFrame> f; };

namespace {

// ---------------------------------------------------------------------------------------------------------------
// hole filling (align_within.rs:330-653)
// ---------------------------------------------------------------------------------------------------------------
double median_of(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    if (n == 0) return 0.0;
    return n % 2 == 1 ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0;
}

bool detect_holes(const std::vector<FFrame>& fr, double& baseline)   // :345-368
{
    std::vector<double> dz;
    for (size_t i = 1; i < fr.size(); ++i) dz.push_back(std::fabs(fr[i].c[2] - fr[i - 1].c[2]));
    baseline = 0.0;
    if (dz.empty()) return false;
    baseline = median_of(dz);
    if (baseline <= kEps) return false;
    for (double d : dz) if (d >= 1.5 * baseline) return true;
    return false;
}

// Option combinators of :476-498 / :575-601: both -> fn, one -> that one, none -> None
template <class F> void opt_scalar(bool ha, double a, bool hb, double b, F both, bool& ho, double& o)
{
    if (ha && hb) { ho = true; o = both(a, b); }
    else if (ha) { ho = true; o = a; }
    else if (hb) { ho = true; o = b; }
    else ho = false;
}

template <class F> FContour combine_contour(const FContour& c1, const FContour& c2, uint32_t cid, uint32_t orig, F f)
{
    const int64_t n = std::min(c1.n(), c2.n());
    FContour o;
    o.id = cid; o.orig = orig;
    o.p.resize((size_t)n * 3); o.aortic.resize((size_t)n);
    for (int64_t i = 0; i < 3 * n; ++i) o.p[(size_t)i] = f(c1.p[(size_t)i], c2.p[(size_t)i]);
    for (int64_t i = 0; i < n; ++i) o.aortic[(size_t)i] = (uint8_t)(c1.aortic[(size_t)i] | c2.aortic[(size_t)i]);
    if (c1.has_cen && c2.has_cen) { o.has_cen = true; for (int k = 0; k < 3; ++k) o.cen[k] = f(c1.cen[k], c2.cen[k]); }
    else if (c1.has_cen) { o.has_cen = true; std::memcpy(o.cen, c1.cen, 24); }
    else if (c2.has_cen) { o.has_cen = true; std::memcpy(o.cen, c2.cen, 24); }
    opt_scalar(c1.has_a, c1.a_th, c2.has_a, c2.a_th, f, o.has_a, o.a_th);
    opt_scalar(c1.has_p, c1.p_th, c2.has_p, c2.p_th, f, o.has_p, o.p_th);
    return o;
}

template <class F> FFrame combine_frame(const FFrame& f1, const FFrame& f2, F f, bool keep_ref)
{
    FFrame o;
    o.id = f2.id;
    for (int k = 0; k < 3; ++k) o.c[k] = f(f1.c[k], f2.c[k]);
    o.lumen = combine_contour(f1.lumen, f2.lumen, f2.lumen.id, f2.lumen.orig, f);
    for (int k = 0; k < K_N; ++k) {
        if (f1.has[k] && f2.has[k]) { o.has[k] = true; o.ext[k] = combine_contour(f1.ext[k], f2.ext[k], f2.ext[k].id, f2.ext[k].orig, f); }
        else if (f1.has[k]) { o.has[k] = true; o.ext[k] = f1.ext[k]; }
        else if (f2.has[k]) { o.has[k] = true; o.ext[k] = f2.ext[k]; }
    }
    if (keep_ref) {                                                   // :636-646
        if (f1.has_ref && f2.has_ref) { o.has_ref = true; for (int k = 0; k < 3; ++k) o.ref[k] = f(f1.ref[k], f2.ref[k]); }
        else if (f1.has_ref) { o.has_ref = true; std::memcpy(o.ref, f1.ref, 24); }
        else if (f2.has_ref) { o.has_ref = true; std::memcpy(o.ref, f2.ref, 24); }
    }
    return o;
}

void insert_frame(std::vector<FFrame>& fr, FFrame&& f, size_t idx)   // Geometry::insert_frame (geometry.rs:285-323)
{
    fr.insert(fr.begin() + (ptrdiff_t)idx, std::move(f));
    for (size_t i = 0; i < fr.size(); ++i) fr[i].set_ids((uint32_t)i);
}

int fill_holes(std::vector<FFrame>& fr)                               // :376-449
{
    double baseline;
    if (!detect_holes(fr, baseline)) return MM_OK;
    if (baseline <= kEps) return set_error(MM_ERR_INVALID, "Baseline spacing is zero or too small to decide.");
    size_t i = 1;
    while (i < fr.size()) {
        const FFrame prev = fr[i - 1], curr = fr[i];
        const double ratio = std::fabs(curr.c[2] - prev.c[2]) / baseline;
        if (ratio < 1.5) { ++i; }
        else if (ratio < 2.5) {
            insert_frame(fr, combine_frame(prev, curr, [](double a, double b) { return (a + b) / 2.0; }, false), i);   // :500-543
            i += 2;
        } else if (ratio < 3.5) {
            for (int k = 1; k <= 2; ++k) {
                const double t = (double)k / 3.0;
                insert_frame(fr, combine_frame(prev, curr, [t](double a, double b) { return a + (b - a) * t; }, true), i + (size_t)k - 1);
            }
            i += 3;
        } else {
            const long missing = (long)std::max(std::floor(ratio - 1.0), 1.0);
            for (long k = 1; k <= missing; ++k) {
                const double t = (double)k / (double)(missing + 1);
                insert_frame(fr, combine_frame(prev, curr, [t](double a, double b) { return a + (b - a) * t; }, true), i + (size_t)k - 1);
            }
            i += (size_t)missing + 1;
        }
    }
    return MM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// anomalous detection and the rotation that puts the reference point to the right (align_within.rs:249-314)
// ---------------------------------------------------------------------------------------------------------------
inline double dist3(const double* a, const double* b)
{
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return std::sqrt(dx * dx + dy * dy + dz * dz);
}

double farthest_points(const FContour& c, int64_t& bi, int64_t& bj)   // contour.rs:227-242
{
    double best = 0.0;
    bi = 0; bj = 0;
    const int64_t n = c.n();
    const double* p = c.p.data();
    // the scan is O(n^2) (125 k pairs for a 501-point lumen); rows are independent, the strict '>' over (i asc, j asc)
    // is restored by reducing the per-row winners in row order
    std::vector<double> rb((size_t)n, 0.0);
    std::vector<int64_t> rj((size_t)n, 0);
    parallel_for((int)((n + 31) / 32), [&](int blk) {
        for (int64_t i = (int64_t)blk * 32; i < std::min(n, ((int64_t)blk + 1) * 32); ++i) {
            double m = 0.0; int64_t mj = 0;
            for (int64_t j = i + 1; j < n; ++j) { const double d = dist3(p + 3 * i, p + 3 * j); if (d > m) { m = d; mj = j; } }
            rb[(size_t)i] = m; rj[(size_t)i] = mj;
        }
    });
    for (int64_t i = 0; i < n; ++i) if (rb[(size_t)i] > best) { best = rb[(size_t)i]; bi = i; bj = rj[(size_t)i]; }
    return best;
}

double closest_opposite_3d(const FContour& c)                         // contour.rs:313-333
{
    const int64_t n = c.n(), half = n / 2;
    double best = 1.7976931348623157e308;
    for (int64_t i = 0; i < n; ++i) {
        const double d = dist3(c.p.data() + 3 * i, c.p.data() + 3 * ((i + half) % n));
        if (d < best) best = d;
    }
    return best;
}

double elliptic_ratio(const FContour& c)                              // contour.rs:335-343
{
    int64_t i, j;
    const double major = farthest_points(c, i, j), minor = closest_opposite_3d(c);
    return major < minor ? minor / major : major / minor;
}

inline double rem_euclid_2pi(double a)
{
    double r = std::fmod(a, 2.0 * kPi);
    if (r < 0.0) r += 2.0 * kPi;
    return r;
}

int angle_ref_point_to_right(const FFrame& f, bool anomalous, double& rotation)   // :256-314
{
    if (!f.has_ref) return set_error(MM_ERR_INVALID, "No reference point found in frame");
    double p1[3], p2[3];
    if (anomalous) {
        int64_t i, j;
        farthest_points(f.lumen, i, j);
        std::memcpy(p1, f.lumen.p.data() + 3 * i, 24); std::memcpy(p2, f.lumen.p.data() + 3 * j, 24);
    } else {
        std::memcpy(p1, f.c, 24); std::memcpy(p2, f.ref, 24);
    }
    const double dx = p2[0] - p1[0], dy = p2[1] - p1[1];
    const double line_angle = std::atan2(dy, dx);
    const double desired = anomalous ? kPi / 2.0 : 0.0;
    rotation = rem_euclid_2pi(desired - line_angle);
    auto rotate2 = [](double x, double y, double cx, double cy, double angle, double& ox, double& oy) {
        const double ddx = x - cx, ddy = y - cy;
        double s, c;
        ::sincos(angle, &s, &c);
        ox = (ddx * c - ddy * s) + cx; oy = (ddx * s + ddy * c) + cy;
    };
    double rrx, rry;
    rotate2(f.ref[0], f.ref[1], p1[0], p1[1], rotation, rrx, rry);
    bool all_good = true;
    const double ops[2][2] = {{p1[0], p1[1]}, {p2[0], p2[1]}};
    for (int k = 0; k < 2; ++k) {
        if (std::fabs(ops[k][0] - f.ref[0]) <= kEps && std::fabs(ops[k][1] - f.ref[1]) <= kEps) continue;   // abs_diff_eq!
        double ox, oy;
        rotate2(ops[k][0], ops[k][1], p1[0], p1[1], rotation, ox, oy);
        if (rrx <= ox) { all_good = false; break; }
    }
    if (!all_good) rotation = rem_euclid_2pi(rotation + kPi);
    return MM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// rotate_geometry (geometry.rs:241-250): Frame::rotate about the frame's own centroid (frame.rs:40-63,
// ContourPoint::rotate contour_point.rs:38-52), then sort_frame_points (frame.rs:123-129, contour.rs:368-405)
// ---------------------------------------------------------------------------------------------------------------
struct SortScratch { std::vector<double> key, tmp; std::vector<int32_t> perm; std::vector<uint8_t> tf; };

void sort_contour(FContour& c, SortScratch& sc)
{
    const int64_t n = c.n();
    if (n == 0) return;
    double sx = 0.0, sy = 0.0;
    for (int64_t i = 0; i < n; ++i) { sx += c.p[3 * i]; sy += c.p[3 * i + 1]; }
    const double cx = sx / (double)n, cy = sy / (double)n;
    sc.key.resize((size_t)n); sc.perm.resize((size_t)n); sc.tmp.resize((size_t)n * 3); sc.tf.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) sc.key[(size_t)i] = std::atan2(c.p[3 * i + 1] - cy, c.p[3 * i] - cx);
    stable_argsort(sc.key.data(), n, sc.perm.data());
    int64_t start = 0;
    for (int64_t i = 1; i < n; ++i)
        if (!(c.p[3 * sc.perm[(size_t)i] + 1] < c.p[3 * sc.perm[(size_t)start] + 1])) start = i;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t s = sc.perm[(size_t)((i + start) % n)];
        sc.tmp[3 * i] = c.p[3 * s]; sc.tmp[3 * i + 1] = c.p[3 * s + 1]; sc.tmp[3 * i + 2] = c.p[3 * s + 2];
        sc.tf[(size_t)i] = c.aortic[(size_t)s];
    }
    std::memcpy(c.p.data(), sc.tmp.data(), (size_t)n * 24);
    std::memcpy(c.aortic.data(), sc.tf.data(), (size_t)n);
}

void rotate_geometry(std::vector<FFrame>& fr, double angle)
{
    if (angle == 0.0) return;
    double s, c;
    ::sincos(angle, &s, &c);
    const int nf = (int)fr.size();
    parallel_for((nf + 7) / 8, [&](int blk) {
        SortScratch sc;
        for (int i = blk * 8; i < std::min(nf, blk * 8 + 8); ++i) {
            FFrame& f = fr[(size_t)i];
            const double cx = f.c[0], cy = f.c[1];
            auto rot = [&](double* q) { const double x = q[0] - cx, y = q[1] - cy; q[0] = x * c - y * s + cx; q[1] = x * s + y * c + cy; };
            for (int64_t j = 0, m = f.lumen.n(); j < m; ++j) rot(f.lumen.p.data() + 3 * j);
            for (int k = 0; k < K_N; ++k) if (f.has[k]) for (int64_t j = 0, m = f.ext[k].n(); j < m; ++j) rot(f.ext[k].p.data() + 3 * j);
            if (f.has_ref) rot(f.ref);
            rot(f.c);
            sort_contour(f.lumen, sc);
            for (int k = 0; k < K_N; ++k) if (f.has[k]) sort_contour(f.ext[k], sc);
        }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// walls (wall.rs)
// ---------------------------------------------------------------------------------------------------------------
// offset_contour (wall.rs:52-100): points (all, or those with index in [lo, hi]) move `distance` away from the freshly
// computed centroid
FContour offset_contour(const FContour& src, double distance, bool ranged, int64_t lo, int64_t hi)
{
    FContour c = src;
    c.compute_centroid();
    const int64_t n = c.n();
    for (int64_t i = 0; i < n; ++i) {
        if (ranged && (i < lo || i > hi)) continue;
        const double dx = src.p[3 * i] - c.cen[0], dy = src.p[3 * i + 1] - c.cen[1], dz = src.p[3 * i + 2] - c.cen[2];
        const double ln = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (ln > kEps) {
            c.p[3 * i] = src.p[3 * i] + (dx / ln) * distance;
            c.p[3 * i + 1] = src.p[3 * i + 1] + (dy / ln) * distance;
            c.p[3 * i + 2] = src.p[3 * i + 2] + (dz / ln) * distance;
        }
    }
    return c;
}

// create_aortic_wall (wall.rs:109-213)
int aortic_wall(const FContour& ct, FContour& out)
{
    const int64_t n = ct.n();
    const int64_t first_quarter = n / 4, half = n / 2, third_quarter = first_quarter * 3;
    if (!ct.has_a) return set_error(MM_ERR_INVALID, "aortic_thickness must be present for this contour");
    const double* P = ct.p.data();
    const double outer_x = P[3 * third_quarter] + ct.a_th;
    const double z = P[3 * third_quarter + 2];
    const double up_mid[2] = {P[0], P[1] + 1.0}, up_right[2] = {outer_x, up_mid[1]};
    const double low_mid[2] = {P[3 * half], P[3 * half + 1] - 1.0}, low_right[2] = {outer_x, low_mid[1]};
    const double dist_up = std::fabs(up_right[0] - up_mid[0]), dist_right = std::fabs(up_right[1] - low_right[1]);
    const double dist_low = std::fabs(low_right[0] - low_mid[0]);
    const double total = dist_up + dist_right + dist_low;
    const double fu = dist_up / total * (double)half, fm = dist_right / total * (double)half;
    if (!std::isfinite(fu) || !std::isfinite(fm)) return set_error(MM_ERR_INVALID, "aortic wall: degenerate contour");
    const int64_t n_up = (int64_t)std::round(fu), n_mid = (int64_t)std::round(fm), n_low = half - n_up - n_mid;   // f64::round
    if (n_low < 0) return set_error(MM_ERR_INVALID, "attempt to subtract with overflow");
    std::vector<double> right;   // x, y pairs
    for (int64_t i = 0; i < n_low; ++i) {                             // low_mid -> low_right
        const double t = (double)i / (double)(n_low - 1);
        right.push_back(low_mid[0] + t * (low_right[0] - low_mid[0])); right.push_back(low_mid[1]);
    }
    for (int64_t i = 0; i < n_mid; ++i) {                             // low_right -> up_right
        const double t = (double)i / (double)(n_mid - 1);
        right.push_back(low_right[0]); right.push_back(low_right[1] + t * (up_right[1] - low_right[1]));
    }
    for (int64_t i = 0; i < n_up; ++i) {                              // up_right -> up_mid
        const double t = (double)i / (double)(std::max<int64_t>(n_up, 1) - 1);
        right.push_back(up_right[0] - t * (up_right[0] - up_mid[0])); right.push_back(up_right[1]);
    }
    const FContour left = offset_contour(ct, 1.0, true, 0, half);
    int64_t left_len = (n % 2) ? half + 1 : half;                     // wall.rs:171-176
    left_len = std::min(left_len, n);
    const int64_t nr = (int64_t)(right.size() / 2);
    if (left_len + nr > n) return set_error(MM_ERR_INVALID, "Index out of bounds: " + std::to_string(left_len + nr - 1) + " >= " + std::to_string(n));
    out = FContour();
    out.id = ct.id; out.orig = ct.orig;
    out.has_cen = ct.has_cen; std::memcpy(out.cen, ct.cen, 24);
    out.has_a = ct.has_a; out.a_th = ct.a_th; out.has_p = ct.has_p; out.p_th = ct.p_th;
    out.p.resize((size_t)(left_len + nr) * 3); out.aortic.resize((size_t)(left_len + nr));
    std::memcpy(out.p.data(), left.p.data(), (size_t)left_len * 24);
    std::memcpy(out.aortic.data(), left.aortic.data(), (size_t)left_len);
    for (int64_t i = 0; i < nr; ++i) {
        out.p[(size_t)(left_len + i) * 3] = right[2 * (size_t)i]; out.p[(size_t)(left_len + i) * 3 + 1] = right[2 * (size_t)i + 1];
        out.p[(size_t)(left_len + i) * 3 + 2] = z;
        out.aortic[(size_t)(left_len + i)] = ct.aortic[(size_t)(left_len + i)];
    }
    return MM_OK;
}

// create_wall_frames (wall.rs:7-34): a Wall contour per frame, from the lumen (anomalous or no EEM) or the EEM
int create_walls(std::vector<FFrame>& fr, bool anomalous)
{
    const int nf = (int)fr.size();
    std::vector<int> rcs((size_t)nf, 0);
    std::vector<std::string> msgs((size_t)nf);
    parallel_for((nf + 15) / 16, [&](int blk) {
        for (int i = blk * 16; i < std::min(nf, blk * 16 + 16); ++i) {
            FFrame& f = fr[(size_t)i];
            const FContour& src = (anomalous || !f.has[K_EEM]) ? f.lumen : f.ext[K_EEM];
            FContour wall;
            if (!src.has_a) wall = offset_contour(src, 1.0, false, 0, 0);
            else if ((rcs[(size_t)i] = aortic_wall(src, wall))) { msgs[(size_t)i] = g_last_error; continue; }
            f.has[K_WALL] = true; f.ext[K_WALL] = std::move(wall);
        }
    });
    for (int i = 0; i < nf; ++i) if (rcs[(size_t)i]) return set_error(rcs[(size_t)i], msgs[(size_t)i]);
    return MM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// smooth_frames (geometry.rs:165-239)
// ---------------------------------------------------------------------------------------------------------------
int smooth_frames(std::vector<FFrame>& fr)
{
    const int nf = (int)fr.size();
    std::vector<FFrame> out((size_t)nf);
    std::vector<uint8_t> bad((size_t)nf, 0);
    parallel_for((nf + 15) / 16, [&](int blk) {
        for (int i = blk * 16; i < std::min(nf, blk * 16 + 16); ++i) {
            const FFrame &cur = fr[(size_t)i], &prev = fr[(size_t)std::max(i - 1, 0)], &nxt = fr[(size_t)std::min(i + 1, nf - 1)];
            const int64_t m = cur.lumen.n();
            auto smooth = [&](const FContour& c, const FContour& p, const FContour& q, FContour& o) {
                if (c.n() < m || p.n() < m || q.n() < m) return false;   // the reference indexes 0..point_count
                o.truncate(m);                                           // (o is f's copy of c already: f = cur below)
                for (int64_t j = 0; j < m; ++j) {
                    o.p[3 * j] = (p.p[3 * j] + c.p[3 * j] + q.p[3 * j]) / 3.0;
                    o.p[3 * j + 1] = (p.p[3 * j + 1] + c.p[3 * j + 1] + q.p[3 * j + 1]) / 3.0;
                }
                o.compute_centroid();
                return true;
            };
            FFrame f = cur;
            if (!smooth(cur.lumen, prev.lumen, nxt.lumen, f.lumen)) { bad[(size_t)i] = 1; continue; }
            for (int k : {K_EEM, K_WALL})
                if (cur.has[k] && prev.has[k] && nxt.has[k] && !smooth(cur.ext[k], prev.ext[k], nxt.ext[k], f.ext[k])) bad[(size_t)i] = 1;
            out[(size_t)i] = std::move(f);
        }
    });
    for (int i = 0; i < nf; ++i) if (bad[(size_t)i]) return set_error(MM_ERR_INVALID, "index out of bounds");
    fr.swap(out);
    return MM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// pair post-processing (postprocessing.rs)
// ---------------------------------------------------------------------------------------------------------------
double avg_z_diff(const std::vector<FFrame>& fr)                      // :100-113
{
    if (fr.size() < 2) return 0.0;
    double s = 0.0;
    for (size_t i = 1; i < fr.size(); ++i) s += fr[i].c[2] - fr[i - 1].c[2];
    return s / (double)(fr.size() - 1);
}

int find_ref_frame_idx(const std::vector<FFrame>& fr, int64_t& idx)  // geometry.rs:62-69: the Frame.id, used as an index
{
    for (const FFrame& f : fr) if (f.has_ref) { idx = (int64_t)f.id; return MM_OK; }
    return set_error(MM_ERR_INVALID, "No reference point found in any frame");
}

void resample_by_diff(std::vector<FFrame>& fr, double diff)           // :116-140
{
    if (!fr.empty()) {
        size_t k = 0;
        for (size_t i = 1; i < fr.size(); ++i) if (fr[i].c[2] < fr[k].c[2]) k = i;   // min_by: first minimum
        if (k) std::rotate(fr.begin(), fr.begin() + (ptrdiff_t)k, fr.end());
    }
    if (fr.empty()) return;
    const double start = fr[0].c[2];
    const int n = (int)fr.size();
    parallel_for((n + 31) / 32, [&](int blk) {               // frames are independent
        for (int i = std::max(1, blk * 32); i < std::min(n, (blk + 1) * 32); ++i) fr[(size_t)i].set_z(start + (double)i * diff);
    });
}

std::vector<double> predict_z_positions(double ref_z, double start_z, double stop_z, double z_diff)   // :142-195
{
    std::vector<double> out;
    if (!std::isfinite(z_diff) || z_diff == 0.0) return out;
    const double eps = 1e-9;
    if (std::fabs(ref_z - start_z) > eps && std::fabs(ref_z - stop_z) > eps) {
        double cur = ref_z;
        while (cur >= start_z - eps) { out.push_back(cur); cur -= z_diff; if (!std::isfinite(cur)) break; }
        std::sort(out.begin(), out.end());
        cur = ref_z + z_diff;
        while (cur <= stop_z + eps) { out.push_back(cur); cur += z_diff; if (!std::isfinite(cur)) break; }
    } else {
        double cur = start_z;
        if (stop_z >= start_z && z_diff > 0.0) {
            while (cur <= stop_z + eps) { out.push_back(cur); cur += z_diff; if (!std::isfinite(cur)) break; }
        } else if (stop_z <= start_z && z_diff < 0.0) {
            while (cur >= stop_z - eps) { out.push_back(cur); cur += z_diff; if (!std::isfinite(cur)) break; }
        }
    }
    return out;
}

FContour blend_contour(const FContour& c1, const FContour& c2, double t)   // :302-340
{
    const int64_t n = std::min(c1.n(), c2.n());
    FContour o = c1;
    o.truncate(n);
    for (int64_t i = 0; i < n; ++i) {
        o.p[3 * i] = c1.p[3 * i] + t * (c2.p[3 * i] - c1.p[3 * i]);
        o.p[3 * i + 1] = c1.p[3 * i + 1] + t * (c2.p[3 * i + 1] - c1.p[3 * i + 1]);
    }
    o.has_cen = c1.has_cen && c2.has_cen;
    if (o.has_cen) for (int k = 0; k < 3; ++k) o.cen[k] = c1.cen[k] + t * (c2.cen[k] - c1.cen[k]);
    o.has_a = c1.has_a && c2.has_a; if (o.has_a) o.a_th = c1.a_th + t * (c2.a_th - c1.a_th);
    o.has_p = c1.has_p && c2.has_p; if (o.has_p) o.p_th = c1.p_th + t * (c2.p_th - c1.p_th);
    return o;
}

int new_frames_by_sample_rate(const std::vector<FFrame>& fr, std::vector<double> zc, std::vector<FFrame>& out)   // :197-300
{
    std::sort(zc.begin(), zc.end());
    out.clear();
    if (fr.empty()) return MM_OK;
    const double max_z = fr.back().c[2];
    for (double z : zc) {
        if (z > max_z) break;
        const FFrame* hit = nullptr;
        for (const FFrame& f : fr) if (std::fabs(f.c[2] - z) < 1e-9) { hit = &f; break; }
        if (hit) { out.push_back(*hit); continue; }
        const FFrame *lo = nullptr, *up = nullptr;
        for (size_t i = 0; i + 1 < fr.size(); ++i) if (fr[i].c[2] <= z && z <= fr[i + 1].c[2]) { lo = &fr[i]; up = &fr[i + 1]; break; }
        if (!lo) return set_error(MM_ERR_INVALID, "Cannot find frames to interpolate between");
        const double t = (z - lo->c[2]) / (up->c[2] - lo->c[2]);
        FFrame f;
        f.id = lo->id;
        f.c[0] = lo->c[0] + t * (up->c[0] - lo->c[0]); f.c[1] = lo->c[1] + t * (up->c[1] - lo->c[1]); f.c[2] = z;
        f.lumen = blend_contour(lo->lumen, up->lumen, t);
        for (int k = 0; k < K_N; ++k) if (lo->has[k] && up->has[k]) { f.has[k] = true; f.ext[k] = blend_contour(lo->ext[k], up->ext[k], t); }
        out.push_back(std::move(f));
    }
    std::stable_sort(out.begin(), out.end(), [](const FFrame& a, const FFrame& b) { return a.c[2] < b.c[2]; });
    for (size_t i = 0; i < out.size(); ++i) {
        FFrame& f = out[i];
        f.id = (uint32_t)i; f.lumen.id = (uint32_t)i;
        const double z = f.c[2];
        for (int64_t j = 0, m = f.lumen.n(); j < m; ++j) f.lumen.p[3 * j + 2] = z;
        if (f.lumen.has_cen) f.lumen.cen[2] = z;
        for (int k = 0; k < K_N; ++k) if (f.has[k]) { f.ext[k].id = (uint32_t)i; for (int64_t j = 0, m = f.ext[k].n(); j < m; ++j) f.ext[k].p[3 * j + 2] = z; }
        if (f.has_ref) f.ref[2] = z;
    }
    return MM_OK;
}

void trim_one(std::vector<FFrame>& fr, int64_t r, int64_t before, int64_t after)   // trim_geom_pair (:342-409)
{
    const int64_t s = r - before, e = r + after;
    if (s < e && e <= (int64_t)fr.size() && s >= 0) {      // keep [s, e): frames are moved, never copied
        fr.erase(fr.begin() + (ptrdiff_t)e, fr.end());
        fr.erase(fr.begin(), fr.begin() + (ptrdiff_t)s);
    }
    for (size_t i = 0; i < fr.size(); ++i) fr[i].set_ids((uint32_t)i);
}

int postprocess_pair(std::vector<FFrame>& A, std::vector<FFrame>& B, double tol, bool anomalous)   // :12-87
{
    if (A.empty() || B.empty()) return set_error(MM_ERR_NO_FRAMES, "Geometry contains no frames");
    const double da = avg_z_diff(A), db = avg_z_diff(B);
    const bool same = (da - db) < tol;                                // :89-98, signed like the reference
    int64_t ia, ib;
    int rc;
    if ((rc = find_ref_frame_idx(A, ia)) || (rc = find_ref_frame_idx(B, ib))) return rc;
    if (ia >= (int64_t)A.size() || ib >= (int64_t)B.size()) return set_error(MM_ERR_REF_INDEX, "reference frame index out of range");
    const double ref_z_a = A[(size_t)ia].c[2], ref_z_b = B[(size_t)ib].c[2];
    auto span = [](const std::vector<FFrame>& fr, double& lo, double& hi) {
        const double z0 = fr.front().c[2], zn = fr.back().c[2];
        if (z0 < zn) { lo = z0; hi = zn; } else { lo = zn; hi = z0; }
    };
    std::vector<double> orig_za(A.size()), orig_zb(B.size());
    for (size_t i = 0; i < A.size(); ++i) orig_za[i] = A[i].c[2];
    for (size_t i = 0; i < B.size(); ++i) orig_zb[i] = B[i].c[2];
    std::vector<FFrame> ra, rb;
    // (A and B are replaced at the end and nothing below reads them again -- the original z values are saved above --
    // so a geometry that is only re-spaced is MOVED into its result: no deep copy of 512 frames x 3 contours.)
    // Every fallible step comes BEFORE the first move, so that an error leaves the caller's handles as they were: the
    // interpolated geometry is built from a const source, and the reference index of a geometry that is only re-spaced is
    // read off the source in the order resample_by_diff will leave it in (rotated to start at the first minimum of z).
    bool move_a = true, move_b = true;
    double diff_a = 0.0, diff_b = 0.0;
    if (same) {
        diff_a = diff_b = (da + db) / 2.0;
    } else if (da < db) {
        double lo, hi;
        span(B, lo, hi);
        if ((rc = new_frames_by_sample_rate(B, predict_z_positions(ref_z_b, lo, hi, da), rb))) return rc;
        move_b = false; diff_a = da;
    } else {
        double lo, hi;
        span(A, lo, hi);
        if ((rc = new_frames_by_sample_rate(A, predict_z_positions(ref_z_a, lo, hi, db), ra))) return rc;
        move_a = false; diff_b = db;
    }
    auto ref_idx_as_resampled = [](const std::vector<FFrame>& fr, int64_t& idx) -> int {
        size_t k = 0;
        for (size_t i = 1; i < fr.size(); ++i) if (fr[i].c[2] < fr[k].c[2]) k = i;
        for (size_t q = 0; q < fr.size(); ++q) {
            const FFrame& f = fr[(k + q) % fr.size()];
            if (f.has_ref) { idx = (int64_t)f.id; return MM_OK; }
        }
        return set_error(MM_ERR_INVALID, "No reference point found in any frame");
    };
    // :70-76 -- the reference indexes the ORIGINAL pair with the resampled geometries' reference indices
    int64_t ja, jb;
    if ((rc = move_a ? ref_idx_as_resampled(A, ja) : find_ref_frame_idx(ra, ja)) ||
        (rc = move_b ? ref_idx_as_resampled(B, jb) : find_ref_frame_idx(rb, jb)))
        return rc;
    if (ja >= (int64_t)orig_za.size() || jb >= (int64_t)orig_zb.size()) return set_error(MM_ERR_REF_INDEX, "index out of bounds");
    if (move_a) { ra = std::move(A); resample_by_diff(ra, diff_a); }
    if (move_b) { rb = std::move(B); resample_by_diff(rb, diff_b); }
    const double translation = orig_za[(size_t)ja] - orig_zb[(size_t)jb];
    {
        const int n = (int)ra.size();
        parallel_for((n + 31) / 32, [&](int blk) {
            for (int i = blk * 32; i < std::min(n, (blk + 1) * 32); ++i) ra[(size_t)i].translate(0.0, 0.0, translation);
        });
    }
    auto ref_or_zero = [](const std::vector<FFrame>& fr) { for (const FFrame& f : fr) if (f.has_ref) return (int64_t)f.id; return (int64_t)0; };
    const int64_t qa = ref_or_zero(ra), qb = ref_or_zero(rb);
    const int64_t before = std::min(qa, qb), after = std::min((int64_t)ra.size() - qa, (int64_t)rb.size() - qb);
    trim_one(ra, qa, before, after); trim_one(rb, qb, before, after);
    if (anomalous) {                                                  // adjust_walls_anomalous_geom_pair (:411-476)
        const size_t n = std::min(ra.size(), rb.size());              // zip()
        ra.resize(n); rb.resize(n);
        for (size_t i = 0; i < n; ++i) {
            FContour &la = ra[i].lumen, &lb = rb[i].lumen;
            if (la.has_a || lb.has_a) {
                const double th = (la.has_a && lb.has_a) ? (la.a_th + lb.a_th) / 2.0 : (la.has_a ? la.a_th : lb.a_th);
                la.has_a = lb.has_a = true; la.a_th = lb.a_th = th;
            }
        }
        if ((rc = create_walls(ra, true)) || (rc = create_walls(rb, true))) {
            // the one failure after the moves (a degenerate contour in the wall synthesis): hand the frames back rather than
            // leave the caller's handles empty -- they hold the re-spaced, trimmed pair without its new walls
            A.swap(ra); B.swap(rb);
            return rc;
        }
    }
    A.swap(ra); B.swap(rb);
    return MM_OK;
}

}  // namespace
}  // namespace mm

using namespace mm;

extern "C" {

int mm_frames_from_flat(const mm_flat_geometry* in, mm_frames** out)
{
    TraceTimer tt_api("frames: from_flat");
    if (!in || !out) return set_error(MM_ERR_INVALID, "mm_frames_from_flat: NULL");
    *out = nullptr;
    const mm_geometry& g = in->g;
    if (g.n_frames < 0 || (g.n_frames > 0 && (!g.id || !g.lumen_id || !g.orig_frame || !g.centroid || !g.lumen_off || !g.lumen)))
        return set_error(MM_ERR_INVALID, "mm_frames_from_flat: geometry arrays missing");
    Frames* F = new Frames();
    F->f.resize((size_t)g.n_frames);
    // per-frame positions in the per-point flag arrays first (a serial prefix sum), then the frames are filled over the
    // worker pool: 512 frames x 3 contours are ~1500 allocations and 12 MB of copies per geometry
    std::vector<int64_t> lum_pos((size_t)g.n_frames + 1, 0), wall_pos((size_t)g.n_frames + 1, 0);
    for (int32_t i = 0; i < g.n_frames; ++i) {
        lum_pos[(size_t)i + 1] = lum_pos[(size_t)i] + (g.lumen_off[i + 1] - g.lumen_off[i]);
        const int64_t nw = (g.extra_off && in->extra_counts) ? std::max<int64_t>(in->extra_counts[4 * i + 3], 0) : 0;
        wall_pos[(size_t)i + 1] = wall_pos[(size_t)i] + nw;
    }
    const int nfr = g.n_frames;
    parallel_for((nfr + 15) / 16, [&](int blk) {
    for (int32_t i = blk * 16; i < std::min(nfr, (blk + 1) * 16); ++i) {
        int64_t lum_at = lum_pos[(size_t)i], wall_at = wall_pos[(size_t)i];
        FFrame& f = F->f[(size_t)i];
        f.id = g.id[i];
        std::memcpy(f.c, g.centroid + 3 * i, 24);
        const int64_t lo = g.lumen_off[i], hi = g.lumen_off[i + 1];
        f.lumen.id = g.lumen_id[i]; f.lumen.orig = g.orig_frame[i];
        f.lumen.p.assign(g.lumen + 3 * lo, g.lumen + 3 * hi);
        f.lumen.aortic.assign((size_t)(hi - lo), 0);
        if (in->lumen_aortic) std::memcpy(f.lumen.aortic.data(), in->lumen_aortic + lum_at, (size_t)(hi - lo));
        lum_at += hi - lo;
        if (in->lumen_centroid && (!in->has_lumen_centroid || in->has_lumen_centroid[i])) {
            f.lumen.has_cen = true; std::memcpy(f.lumen.cen, in->lumen_centroid + 3 * i, 24);
        }
        if (in->has_aortic && in->has_aortic[i]) { f.lumen.has_a = true; f.lumen.a_th = in->aortic_thickness[i]; }
        if (in->has_pulmonary && in->has_pulmonary[i]) { f.lumen.has_p = true; f.lumen.p_th = in->pulmonary_thickness[i]; }
        if (g.extra_off && in->extra_counts) {
            int64_t e = g.extra_off[i];
            const int kinds[4] = {K_EEM, K_CALC, K_SIDE, K_WALL};
            for (int q = 0; q < 4; ++q) {
                const int64_t n = in->extra_counts[4 * i + q];
                if (n <= 0) continue;
                FContour& c = f.ext[kinds[q]];
                f.has[kinds[q]] = true;
                c.id = g.id[i]; c.orig = g.orig_frame[i];
                c.p.assign(g.extra + 3 * e, g.extra + 3 * (e + n));
                c.aortic.assign((size_t)n, 0);
                if (kinds[q] == K_WALL) {
                    c.has_a = f.lumen.has_a; c.a_th = f.lumen.a_th; c.has_p = f.lumen.has_p; c.p_th = f.lumen.p_th;
                    if (in->wall_aortic) std::memcpy(c.aortic.data(), in->wall_aortic + wall_at, (size_t)n);
                    wall_at += n;
                }
                c.compute_centroid();
                e += n;
            }
        }
        if (g.cath_off && g.cath) {
            FContour& c = f.ext[K_CATH];
            f.has[K_CATH] = true;
            c.id = g.id[i]; c.orig = g.orig_frame[i];
            c.p.assign(g.cath + 3 * g.cath_off[i], g.cath + 3 * g.cath_off[i + 1]);
            c.aortic.assign((size_t)c.n(), 0);
            c.compute_centroid();
        }
        if (g.has_ref && g.has_ref[i]) { f.has_ref = true; std::memcpy(f.ref, g.ref + 3 * i, 24); }
    }
    });
    *out = reinterpret_cast<mm_frames*>(F);
    return MM_OK;
}

int mm_frames_dims(const mm_frames* h, int32_t* n_frames, int64_t* n_lumen, int64_t* n_cath, int64_t* n_extra, int64_t* n_wall)
{
    const Frames* F = reinterpret_cast<const Frames*>(h);
    if (!F) return set_error(MM_ERR_INVALID, "mm_frames_dims: NULL");
    bool all_cath = !F->f.empty();
    for (const FFrame& f : F->f) all_cath = all_cath && f.has[K_CATH];
    int64_t nl = 0, nc = 0, ne = 0, nw = 0;
    for (const FFrame& f : F->f) {
        nl += f.lumen.n();
        if (all_cath) nc += f.ext[K_CATH].n();
        for (int k : {K_EEM, K_CALC, K_SIDE, K_WALL}) if (f.has[k]) ne += f.ext[k].n();
        if (f.has[K_WALL]) nw += f.ext[K_WALL].n();
    }
    if (n_frames) *n_frames = (int32_t)F->f.size();
    if (n_lumen) *n_lumen = nl;
    if (n_cath) *n_cath = nc;
    if (n_extra) *n_extra = ne;
    if (n_wall) *n_wall = nw;
    return MM_OK;
}

int mm_frames_export(const mm_frames* h, mm_flat_geometry* out)
{
    TraceTimer tt_api("frames: export");
    const Frames* F = reinterpret_cast<const Frames*>(h);
    if (!F || !out) return set_error(MM_ERR_INVALID, "mm_frames_export: NULL");
    int32_t nf; int64_t nl, nc, ne, nw;
    mm_frames_dims(h, &nf, &nl, &nc, &ne, &nw);
    mm_geometry& g = out->g;
    if (!g.id || !g.lumen_id || !g.orig_frame || !g.centroid || !g.lumen_off || (nl > 0 && !g.lumen) || !g.has_ref || !g.ref ||
        !out->extra_counts || !out->has_lumen_centroid || !out->lumen_centroid || !out->has_aortic || !out->aortic_thickness ||
        !out->has_pulmonary || !out->pulmonary_thickness || (nl > 0 && !out->lumen_aortic) || (nw > 0 && !out->wall_aortic) ||
        (nc > 0 && (!g.cath_off || !g.cath)) || (ne > 0 && (!g.extra_off || !g.extra)))
        return set_error(MM_ERR_INVALID, "mm_frames_export: destination arrays missing");
    g.n_frames = nf; g.has_catheter = nc > 0 ? 1 : 0;
    // output positions per frame (serial prefix sums), then the copies over the worker pool
    std::vector<int64_t> pl((size_t)nf + 1, 0), pc((size_t)nf + 1, 0), pe((size_t)nf + 1, 0), pw((size_t)nf + 1, 0);
    for (int32_t i = 0; i < nf; ++i) {
        const FFrame& f = F->f[(size_t)i];
        pl[(size_t)i + 1] = pl[(size_t)i] + f.lumen.n();
        pc[(size_t)i + 1] = pc[(size_t)i] + (nc > 0 ? f.ext[K_CATH].n() : 0);
        int64_t e = 0;
        for (int k : {K_EEM, K_CALC, K_SIDE, K_WALL}) if (f.has[k]) e += f.ext[k].n();
        pe[(size_t)i + 1] = pe[(size_t)i] + e;
        pw[(size_t)i + 1] = pw[(size_t)i] + (f.has[K_WALL] ? f.ext[K_WALL].n() : 0);
    }
    parallel_for(((int)nf + 15) / 16, [&](int blk) {
    for (int32_t i = blk * 16; i < std::min<int32_t>(nf, (blk + 1) * 16); ++i) {
        int64_t ol = pl[(size_t)i], oc = pc[(size_t)i], oe = pe[(size_t)i], ow = pw[(size_t)i];
        const FFrame& f = F->f[(size_t)i];
        g.id[i] = f.id; g.lumen_id[i] = f.lumen.id; g.orig_frame[i] = f.lumen.orig;
        std::memcpy(g.centroid + 3 * i, f.c, 24);
        g.lumen_off[i] = ol;
        if (f.lumen.n()) {
            std::memcpy(g.lumen + 3 * ol, f.lumen.p.data(), f.lumen.p.size() * 8);
            std::memcpy(out->lumen_aortic + ol, f.lumen.aortic.data(), (size_t)f.lumen.n());
        }
        ol += f.lumen.n();
        out->has_lumen_centroid[i] = f.lumen.has_cen ? 1 : 0;
        for (int k = 0; k < 3; ++k) out->lumen_centroid[3 * i + k] = f.lumen.has_cen ? f.lumen.cen[k] : 0.0;
        out->has_aortic[i] = f.lumen.has_a ? 1 : 0; out->aortic_thickness[i] = f.lumen.has_a ? f.lumen.a_th : 0.0;
        out->has_pulmonary[i] = f.lumen.has_p ? 1 : 0; out->pulmonary_thickness[i] = f.lumen.has_p ? f.lumen.p_th : 0.0;
        if (nc > 0) { g.cath_off[i] = oc; std::memcpy(g.cath + 3 * oc, f.ext[K_CATH].p.data(), f.ext[K_CATH].p.size() * 8); oc += f.ext[K_CATH].n(); }
        if (ne > 0) g.extra_off[i] = oe;
        const int kinds[4] = {K_EEM, K_CALC, K_SIDE, K_WALL};
        for (int q = 0; q < 4; ++q) {
            const int64_t m = f.has[kinds[q]] ? f.ext[kinds[q]].n() : 0;
            out->extra_counts[4 * i + q] = m;
            if (!m) continue;
            std::memcpy(g.extra + 3 * oe, f.ext[kinds[q]].p.data(), (size_t)m * 24);
            oe += m;
            if (kinds[q] == K_WALL) { std::memcpy(out->wall_aortic + ow, f.ext[K_WALL].aortic.data(), (size_t)m); ow += m; }
        }
        g.has_ref[i] = f.has_ref ? 1 : 0;
        for (int k = 0; k < 3; ++k) g.ref[3 * i + k] = f.has_ref ? f.ref[k] : 0.0;
    }
    });
    g.lumen_off[nf] = pl[(size_t)nf];
    if (nc > 0) g.cath_off[nf] = pc[(size_t)nf];
    if (ne > 0) g.extra_off[nf] = pe[(size_t)nf];
    return MM_OK;
}

void mm_frames_destroy(mm_frames* h) { delete reinterpret_cast<Frames*>(h); }

// align_within.rs:136-160 after the chain: hole filling, anomalous detection on the reference frame, the rotation that
// puts the reference point to the right, aortic flags, wall contours, smoothing
int mm_frames_finish_within(mm_frames* h, int64_t ref_idx, int smooth, int* anomalous_out)
{
    TraceTimer tt_api("frames: finish_within");
    Frames* F = reinterpret_cast<Frames*>(h);
    if (!F) return set_error(MM_ERR_INVALID, "mm_frames_finish_within: NULL");
    std::vector<FFrame>& fr = F->f;
    int rc;
    { TraceTimer t("finish: fill_holes"); if ((rc = fill_holes(fr))) return rc; }  // :136
    if (ref_idx < 0 || ref_idx >= (int64_t)fr.size()) return set_error(MM_ERR_REF_INDEX, "reference frame index out of range");
    const FFrame& rf = fr[(size_t)ref_idx];
    if (rf.lumen.n() <= 2) return set_error(MM_ERR_INVALID, "Need at least 3 points");
    const bool anomalous = elliptic_ratio(rf.lumen) > 2.0 || rf.lumen.has_a || rf.lumen.has_p;   // :249-254
    double rot;
    { TraceTimer t("finish: ref point angle"); if ((rc = angle_ref_point_to_right(rf, anomalous, rot))) return rc; }   // :139
    { TraceTimer t("finish: rotate"); rotate_geometry(fr, rot); }                 // :141 (contour centroids stay as they are)
    if (anomalous)                                                                // :143-147 assign_aortic
        for (FFrame& f : fr) { const int64_t n = f.lumen.n(), half = n / 2; for (int64_t i = 0; i < n; ++i) f.lumen.aortic[(size_t)i] = i >= half; }
    { TraceTimer t("finish: walls"); if ((rc = create_walls(fr, anomalous))) return rc; }      // :149-153
    { TraceTimer t("finish: smooth"); if (smooth && (rc = smooth_frames(fr))) return rc; }     // :155-157
    if (anomalous_out) *anomalous_out = anomalous ? 1 : 0;
    return MM_OK;
}

int mm_frames_postprocess_pair(mm_frames* a, mm_frames* b, double tolerance, int anomalous)
{
    TraceTimer tt_api("frames: postprocess_pair");
    Frames *A = reinterpret_cast<Frames*>(a), *B = reinterpret_cast<Frames*>(b);
    if (!A || !B) return set_error(MM_ERR_INVALID, "mm_frames_postprocess_pair: NULL");
    return postprocess_pair(A->f, B->f, tolerance, anomalous != 0);
}

}  // extern "C"
