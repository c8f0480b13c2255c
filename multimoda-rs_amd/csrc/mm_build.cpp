// mm_build.cpp -- the geometry builder behind include/mm_build.h: build_geometry_from_inputdata
// (src/intravascular/io/build.rs:9-205) and the Geometry / Contour / Frame helpers it calls, host f64 in the
// reference's operation order (built with -ffp-contract=off).  Reference lines are cited per step.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstring>
#include <map>
#include <numeric>
#include <vector>

#include "../../include/mm_build.h"
#include "mm_engine.h"
#include "mm_pool.h"
#include "mm_sort.h"
#include "mm_trace.h"

namespace mm {
namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;

struct BContour {
    std::vector<double> xyz;          // n x 3
    std::vector<uint8_t> aortic;      // n (lumen only; empty = all false)
    int64_t n() const { return (int64_t)(xyz.size() / 3); }
};

struct BFrame {
    uint32_t id = 0, orig = 0;
    double centroid[3] = {0, 0, 0};
    BContour lumen;
    bool has_ext[3] = {false, false, false};   // eem, calcification, sidebranch
    BContour ext[3];
    bool has_cath = false;
    BContour cath;
    bool has_ref = false;
    double ref[3] = {0, 0, 0};
    bool has_a = false, has_p = false;         // Contour.aortic_thickness / pulmonary_thickness of the lumen
    double a_th = 0.0, p_th = 0.0;
};

// Rust's `{}` / `{:?}` of an f64: the shortest digits that round-trip, never an exponent for Display; `{:?}` adds ".0"
// to integral values.  (Only the z / centroid messages of the integrity check print floats.)
std::string rust_f64(double v, bool debug)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[400];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    std::string s(buf, r.ptr);
    if (debug && s.find('.') == std::string::npos) s += ".0";
    return s;
}

// Contour::compute_centroid (contour.rs:213-224): sequential sums / n
void centroid_of(const BContour& c, double out[3])
{
    double sx = 0.0, sy = 0.0, sz = 0.0;
    const int64_t n = c.n();
    for (int64_t i = 0; i < n; ++i) { sx += c.xyz[3 * i]; sy += c.xyz[3 * i + 1]; sz += c.xyz[3 * i + 2]; }
    out[0] = sx / (double)n; out[1] = sy / (double)n; out[2] = sz / (double)n;
}

// Contour::sort_contour_points (contour.rs:368-405): stable sort by atan2 about the xy mean, then the LAST point of
// maximal y (Iterator::max_by keeps the last of equal maxima) rotated to the front; flags follow their points.
void sort_contour(BContour& c, std::vector<double>& key, std::vector<int32_t>& perm, std::vector<double>& tmp,
                  std::vector<uint8_t>& tmpf)
{
    const int64_t n = c.n();
    if (n == 0) return;
    double sx = 0.0, sy = 0.0;
    for (int64_t i = 0; i < n; ++i) { sx += c.xyz[3 * i]; sy += c.xyz[3 * i + 1]; }
    const double cx = sx / (double)n, cy = sy / (double)n;
    key.resize((size_t)n); perm.resize((size_t)n); tmp.resize((size_t)n * 3);
    for (int64_t i = 0; i < n; ++i) key[(size_t)i] = std::atan2(c.xyz[3 * i + 1] - cy, c.xyz[3 * i] - cx);
    stable_argsort(key.data(), n, perm.data());
    int64_t start = 0;
    for (int64_t i = 1; i < n; ++i)
        if (!(c.xyz[3 * perm[(size_t)i] + 1] < c.xyz[3 * perm[(size_t)start] + 1])) start = i;
    const bool fl = !c.aortic.empty();
    if (fl) tmpf.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t src = perm[(size_t)((i + start) % n)];
        tmp[3 * i] = c.xyz[3 * src]; tmp[3 * i + 1] = c.xyz[3 * src + 1]; tmp[3 * i + 2] = c.xyz[3 * src + 2];
        if (fl) tmpf[(size_t)i] = c.aortic[(size_t)src];
    }
    std::memcpy(c.xyz.data(), tmp.data(), (size_t)n * 24);
    if (fl) std::memcpy(c.aortic.data(), tmpf.data(), (size_t)n);
}

inline void set_z(BContour& c, double z)
{
    for (int64_t i = 0, n = c.n(); i < n; ++i) c.xyz[3 * i + 2] = z;
}

void set_frame_z(BFrame& f, double z)   // every z of the frame: points, extras, reference point, frame centroid
{
    set_z(f.lumen, z);
    for (int k = 0; k < 3; ++k) if (f.has_ext[k]) set_z(f.ext[k], z);
    if (f.has_cath) set_z(f.cath, z);
    if (f.has_ref) f.ref[2] = z;
    f.centroid[2] = z;
}

// HashMap<u32, Vec<ContourPoint>> of build_contour_with_mapping (contour.rs:164-167): rows of one frame in input order
void group_rows(const double* rows4, int64_t n, const uint8_t* flags, std::map<uint32_t, BContour>& out)
{
    // rows of a frame are usually adjacent: one pass finds the runs of equal frame indices and counts every frame's rows
    // (one map lookup per run; map nodes do not move), every contour is sized once, and the runs are copied over the
    // worker pool -- each to the offset the runs of its frame before it have left, so a frame's rows keep their input order
    struct Run { BContour* c; int64_t at, start, len; };
    std::vector<Run> runs;
    {
        std::map<uint32_t, int64_t> count;
        int64_t* cc = nullptr;
        uint32_t last = 0;
        for (int64_t i = 0; i < n; ++i) {
            const uint32_t f = (uint32_t)rows4[4 * i];
            if (!cc || f != last) {
                cc = &count[f]; last = f;
                runs.push_back(Run{&out[f], *cc, i, 0});
            }
            ++*cc; ++runs.back().len;
        }
        for (const auto& kv : count) {
            BContour& c = out[kv.first];
            c.xyz.resize((size_t)kv.second * 3);
            if (flags) c.aortic.resize((size_t)kv.second);
        }
    }
    const int nr = (int)runs.size();
    parallel_for((nr + 15) / 16, [&](int blk) {
        for (int r = blk * 16; r < std::min(nr, blk * 16 + 16); ++r) {
            const Run& u = runs[(size_t)r];
            double* d = u.c->xyz.data() + 3 * u.at;
            const double* src = rows4 + 4 * u.start;
            for (int64_t k = 0; k < u.len; ++k) { d[3 * k] = src[4 * k + 1]; d[3 * k + 1] = src[4 * k + 2]; d[3 * k + 2] = src[4 * k + 3]; }
            if (flags) std::memcpy(u.c->aortic.data() + u.at, flags + u.start, (size_t)u.len);
        }
    });
}

}  // namespace

struct Built { std::vector<BFrame> frames; bool any_flags = false; };

namespace {
// check_geometry_integrity (integrity_check.rs:8-33) on a built geometry, the reference's messages.  Empty string =
// Ok.  By construction of this builder the frame ids are the positions (:35-46), every extras contour and the
// reference point carry their frame's original index (:169-200) and Frame.lumen.centroid is the value the frame
// centroid was copied from (:49-81, which therefore only trips on non-finite coordinates); those are still walked
// so that the order of the reported failure is the reference's.
std::string integrity_error(const Built& B)
{
    const std::vector<BFrame>& fr = B.frames;
    const size_t n = fr.size();
    auto tup = [](const double c[3]) {
        return "(" + rust_f64(c[0], true) + ", " + rust_f64(c[1], true) + ", " + rust_f64(c[2], true) + ")";
    };
    auto at = [](size_t i, const BFrame& f) { return std::to_string(i) + " (ID " + std::to_string(f.id) + ")"; };
    for (size_t i = 0; i < n; ++i)                                        // :35-46
        if (fr[i].id != (uint32_t)i)
            return "Frame IDs are not consecutive. Expected ID " + std::to_string(i) + ", found ID " + std::to_string(fr[i].id);
    for (size_t i = 0; i < n; ++i)                                        // :49-81 (EPSILON 1e-6, strict)
        for (int k = 0; k < 3; ++k)
            if (!(std::fabs(fr[i].centroid[k] - fr[i].centroid[k]) < 1e-6))
                return "Frame centroid does not match lumen centroid in frame " + at(i, fr[i]) + ". Frame: " +
                       tup(fr[i].centroid) + ", Lumen: " + tup(fr[i].centroid);
    for (size_t i = 0; i < n; ++i)                                        // :84-104
        if (fr[i].lumen.n() == 0) return "Lumen contour has no points in frame " + at(i, fr[i]);
    size_t n_ref = 0;                                                     // :107-118
    for (const BFrame& f : fr) n_ref += f.has_ref ? 1 : 0;
    if (n_ref != 1) return "Expected exactly one reference point, found " + std::to_string(n_ref);
    static const char* kKind[4] = {"Eem", "Calcification", "Sidebranch", "Catheter"};
    int64_t expect_lumen = -1, expect[4] = {-1, -1, -1, -1};              // :121-166
    for (size_t i = 0; i < n; ++i) {
        const BFrame& f = fr[i];
        if (expect_lumen < 0) expect_lumen = f.lumen.n();
        else if (f.lumen.n() != expect_lumen)
            return "Lumen point count mismatch in frame " + at(i, f) + ". Expected " + std::to_string(expect_lumen) +
                   ", found " + std::to_string(f.lumen.n());
        for (int k = 0; k < 4; ++k) {
            const bool has = k < 3 ? f.has_ext[k] : f.has_cath;
            if (!has) continue;
            const int64_t c = k < 3 ? f.ext[k].n() : f.cath.n();
            if (expect[k] < 0) expect[k] = c;
            else if (c != expect[k])
                return std::string(kKind[k]) + " contour point count mismatch in frame " + at(i, f) + ". Expected " +
                       std::to_string(expect[k]) + ", found " + std::to_string(c);
        }
    }
    // :203-221 find_proximal_end_idx (geometry.rs:42-59; lumen ids are the frame ids after build.rs:194-197)
    const size_t prox = n == 1 ? fr[0].id : (fr[0].orig > fr[n - 1].orig ? fr[0].id : fr[n - 1].id);
    double min_z = INFINITY; size_t min_idx = 0;
    for (size_t i = 0; i < n; ++i) if (fr[i].centroid[2] < min_z) { min_z = fr[i].centroid[2]; min_idx = i; }
    if (prox != min_idx)
        return "Proximal end index is " + std::to_string(prox) + ", but frame with minimum z is " + std::to_string(min_idx) +
               " (z=" + rust_f64(min_z, false) + ").";
    if (fr[0].centroid[2] > fr[n - 1].centroid[2])                        // :224-234
        return "First frame has higher z-coords " + rust_f64(fr[0].centroid[2], false) + " than last frame " +
               rust_f64(fr[n - 1].centroid[2], false);
    return std::string();
}
}  // namespace

}  // namespace mm

using namespace mm;

namespace {
int build_impl(const double* lumen4, int64_t n_lumen, const uint8_t* lumen_aortic, const double* eem4, int64_t n_eem,
               const double* calc4, int64_t n_calc, const double* side4, int64_t n_side, const double ref4[4],
               const mm_record* records, int64_t n_records, int diastole, double icx, double icy, double radius,
               uint32_t n_points, bool check, mm_built** out)
{
    if (!out) return set_error(MM_ERR_INVALID, "mm_build_geometry: out == NULL");
    *out = nullptr;
    if (!lumen4 || n_lumen <= 0 || !ref4) return set_error(MM_ERR_INVALID, "mm_build_geometry: lumen and reference point are required");
    if ((n_eem > 0 && !eem4) || (n_calc > 0 && !calc4) || (n_side > 0 && !side4) || n_eem < 0 || n_calc < 0 || n_side < 0)
        return set_error(MM_ERR_INVALID, "mm_build_geometry: bad extras arrays");
    {   // ContourPoint.frame_index is a u32: anything else in column 0 is a caller error (and a UB cast below)
        auto frames_ok = [](const double* rows, int64_t n) {
            for (int64_t i = 0; i < n; ++i) { const double f = rows[4 * i]; if (!(f >= 0.0 && f <= 4294967295.0)) return false; }
            return true;
        };
        if (!frames_ok(lumen4, n_lumen) || !frames_ok(ref4, 1) || (eem4 && !frames_ok(eem4, n_eem)) ||
            (calc4 && !frames_ok(calc4, n_calc)) || (side4 && !frames_ok(side4, n_side)))
            return set_error(MM_ERR_INVALID, "mm_build_geometry: frame index outside u32");
    }
    const double* ext_rows[3] = {eem4, calc4, side4};
    const int64_t ext_n[3] = {eem4 ? n_eem : -1, calc4 ? n_calc : -1, side4 ? n_side : -1};   // -1: None

    TraceTimer tt_map("build: id mapping");
    // build.rs:36-71: shared original-frame -> sequential-id mapping over ALL contour kinds and the reference point
    std::vector<uint32_t> originals;
    originals.reserve((size_t)n_lumen / 64 + 16);
    {
        uint32_t last = 0xffffffffu; bool have = false;
        auto add_rows = [&](const double* rows, int64_t n) {
            for (int64_t i = 0; i < n; ++i) {
                const uint32_t f = (uint32_t)rows[4 * i];
                if (!have || f != last) { originals.push_back(f); last = f; have = true; }
            }
        };
        add_rows(lumen4, n_lumen);
        for (int k = 0; k < 3; ++k) if (ext_n[k] > 0) add_rows(ext_rows[k], ext_n[k]);
        originals.push_back((uint32_t)ref4[0]);
        std::sort(originals.begin(), originals.end());
        originals.erase(std::unique(originals.begin(), originals.end()), originals.end());
    }
    auto map_id = [&](uint32_t orig) { return (uint32_t)(std::lower_bound(originals.begin(), originals.end(), orig) - originals.begin()); };

    // contour.rs:171-181: measurements by original frame, later records override earlier ones
    std::map<uint32_t, const mm_record*> meas;
    if (records) for (int64_t i = 0; i < n_records; ++i) meas[records[i].frame] = &records[i];

    tt_map.stop();
    TraceTimer tt_group("build: group rows + frames");
    // build.rs:74-129: one frame per lumen contour, in ascending original frame (= ascending id)
    Built* B = new Built();
    std::map<uint32_t, BContour> groups;
    group_rows(lumen4, n_lumen, lumen_aortic, groups);
    B->any_flags = lumen_aortic != nullptr;
    std::map<uint32_t, size_t> frame_of_id;
    const uint32_t ref_id = map_id((uint32_t)ref4[0]);
    for (auto& kv : groups) {
        BFrame f;
        f.orig = kv.first; f.id = map_id(kv.first);
        f.lumen = std::move(kv.second);
        centroid_of(f.lumen, f.centroid);
        auto m = meas.find(f.orig);
        if (m != meas.end()) { f.has_a = m->second->has_m1 != 0; f.a_th = m->second->m1; f.has_p = m->second->has_m2 != 0; f.p_th = m->second->m2; }
        if (ref_id == f.id) { f.has_ref = true; f.ref[0] = ref4[1]; f.ref[1] = ref4[2]; f.ref[2] = ref4[3]; }   // :121-125
        frame_of_id[f.id] = B->frames.size();
        B->frames.push_back(std::move(f));
    }
    for (int k = 0; k < 3; ++k) {                                     // build.rs:131-150: extras join the frame of their id
        if (ext_n[k] < 0) continue;
        std::map<uint32_t, BContour> eg;
        group_rows(ext_rows[k], ext_n[k], nullptr, eg);
        for (auto& kv : eg) {
            auto it = frame_of_id.find(map_id(kv.first));
            if (it == frame_of_id.end()) continue;
            BFrame& f = B->frames[it->second];
            f.has_ext[k] = true; f.ext[k] = std::move(kv.second);
        }
    }
    tt_group.stop();
    TraceTimer tt_cath("build: catheter + reorder");
    if (n_points > 0) {                                               // build.rs:152-174, frame.rs:163-204
        for (BFrame& f : B->frames) {
            const double z = f.lumen.xyz[2];                          // the first encountered z of the frame's points
            f.has_cath = true;
            f.cath.xyz.resize((size_t)n_points * 3);
            for (uint32_t i = 0; i < n_points; ++i) {
                const double angle = 2.0 * kPi * (double)i / (double)n_points;
                double s, c;
                ::sincos(angle, &s, &c);                              // frame.rs:192-193: cos and sin of one value
                f.cath.xyz[3 * i] = icx + radius * c;
                f.cath.xyz[3 * i + 1] = icy + radius * s;
                f.cath.xyz[3 * i + 2] = z;
            }
        }
    }
    // frames are already in ascending id order (build.rs:176-177)

    if (records) {                                                    // build.rs:184-186 -> geometry.rs:72-155
        const uint8_t want = diastole ? 0 : 1;
        std::map<uint32_t, size_t> by_orig;
        for (size_t i = 0; i < B->frames.size(); ++i) by_orig[B->frames[i].orig] = i;
        std::vector<uint8_t> used(B->frames.size(), 0);
        std::vector<size_t> order;
        for (int64_t i = 0; i < n_records; ++i) {
            if (records[i].phase != want) continue;
            auto it = by_orig.find(records[i].frame);
            if (it == by_orig.end() || used[it->second]) continue;   // frame_map.remove: a frame is taken once
            used[it->second] = 1; order.push_back(it->second);
        }
        for (size_t i = 0; i < B->frames.size(); ++i) if (!used[i]) order.push_back(i);   // the rest, ascending original frame
        std::vector<BFrame> nf;
        nf.reserve(order.size());
        for (size_t k : order) nf.push_back(std::move(B->frames[k]));
        B->frames.swap(nf);
        for (size_t i = 0; i < B->frames.size(); ++i) {               // :106-141
            BFrame& f = B->frames[i];
            f.id = (uint32_t)i;
            set_frame_z(f, f.lumen.xyz[2]);                           // the z of the frame's first lumen point, everywhere
        }
    }

    tt_cath.stop();
    TraceTimer tt_sort("build: sort contours");
    {                                                                 // build.rs:188-190 sort_frame_points, frames in parallel
        const int nf = (int)B->frames.size();
        parallel_for((nf + 15) / 16, [&](int blk) {
            std::vector<double> key, tmp; std::vector<int32_t> perm; std::vector<uint8_t> tmpf;
            for (int i = blk * 16; i < std::min(nf, blk * 16 + 16); ++i) {
                BFrame& f = B->frames[(size_t)i];
                sort_contour(f.lumen, key, perm, tmp, tmpf);
                for (int k = 0; k < 3; ++k) if (f.has_ext[k]) sort_contour(f.ext[k], key, perm, tmp, tmpf);
                if (f.has_cath) sort_contour(f.cath, key, perm, tmp, tmpf);
            }
        });
    }

    tt_sort.stop();
    TraceTimer tt_tail("build: proximal end + integrity");
    const size_t n = B->frames.size();                                // build.rs:192 -> geometry.rs:325-381
    if (n) {
        size_t prox = n == 1 ? B->frames[0].id
                             : (B->frames[0].orig > B->frames[n - 1].orig ? B->frames[0].id : B->frames[n - 1].id);   // :42-59
        prox = std::min(prox, n - 1);
        if (prox != 0) std::reverse(B->frames.begin(), B->frames.end());
        std::vector<double> zs(n);
        for (size_t i = 0; i < n; ++i) zs[i] = B->frames[i].centroid[2];
        std::sort(zs.begin(), zs.end());
        for (size_t i = 0; i < n; ++i) {
            B->frames[i].id = (uint32_t)i;                            // + build.rs:194-197 set_value(Some(id))
            set_frame_z(B->frames[i], zs[i]);
        }
    }
    if (n == 0) { delete B; return set_error(MM_ERR_NO_FRAMES, "Geometry has no frames"); }   // integrity_check.rs:9-11
    if (check) {
        const std::string why = integrity_error(*B);                  // build.rs:199
        if (!why.empty()) { delete B; return set_error(MM_ERR_INTEGRITY, why); }
    }
    *out = reinterpret_cast<mm_built*>(B);
    return MM_OK;
}
}  // namespace

extern "C" {

int mm_build_geometry(const double* lumen4, int64_t n_lumen, const uint8_t* lumen_aortic, const double* eem4, int64_t n_eem,
                      const double* calc4, int64_t n_calc, const double* side4, int64_t n_side, const double ref4[4],
                      const mm_record* records, int64_t n_records, int diastole, double icx, double icy, double radius,
                      uint32_t n_points, mm_built** out)
{
    return build_impl(lumen4, n_lumen, lumen_aortic, eem4, n_eem, calc4, n_calc, side4, n_side, ref4, records, n_records,
                      diastole, icx, icy, radius, n_points, true, out);
}

int mm_build_geometry_lenient(const double* lumen4, int64_t n_lumen, const uint8_t* lumen_aortic, const double* eem4,
                              int64_t n_eem, const double* calc4, int64_t n_calc, const double* side4, int64_t n_side,
                              const double ref4[4], const mm_record* records, int64_t n_records, int diastole, double icx,
                              double icy, double radius, uint32_t n_points, mm_built** out)
{
    return build_impl(lumen4, n_lumen, lumen_aortic, eem4, n_eem, calc4, n_calc, side4, n_side, ref4, records, n_records,
                      diastole, icx, icy, radius, n_points, false, out);
}

int mm_built_dims(const mm_built* h, int32_t* n_frames, int64_t* n_lumen, int64_t* n_cath, int64_t* n_extra)
{
    const Built* B = reinterpret_cast<const Built*>(h);
    if (!B) return set_error(MM_ERR_INVALID, "mm_built_dims: NULL");
    int64_t nl = 0, nc = 0, ne = 0;
    bool all_cath = !B->frames.empty();
    for (const BFrame& f : B->frames) all_cath = all_cath && f.has_cath;
    for (const BFrame& f : B->frames) {
        nl += f.lumen.n();
        if (all_cath) nc += f.cath.n();
        for (int k = 0; k < 3; ++k) if (f.has_ext[k]) ne += f.ext[k].n();
    }
    if (n_frames) *n_frames = (int32_t)B->frames.size();
    if (n_lumen) *n_lumen = nl;
    if (n_cath) *n_cath = nc;
    if (n_extra) *n_extra = ne;
    return MM_OK;
}

int mm_built_export(const mm_built* h, mm_geometry* dst, int64_t* extra_counts, double* a_th, uint8_t* has_a,
                    double* p_th, uint8_t* has_p, uint8_t* lumen_aortic_out)
{
    const Built* B = reinterpret_cast<const Built*>(h);
    if (!B || !dst || !dst->id || !dst->lumen_id || !dst->orig_frame || !dst->centroid || !dst->lumen_off || !dst->lumen ||
        !dst->has_ref || !dst->ref)
        return set_error(MM_ERR_INVALID, "mm_built_export: destination arrays missing");
    int32_t F; int64_t nl, nc, ne;
    mm_built_dims(h, &F, &nl, &nc, &ne);
    if ((nc > 0 && (!dst->cath_off || !dst->cath)) || (ne > 0 && (!dst->extra_off || !dst->extra)))
        return set_error(MM_ERR_INVALID, "mm_built_export: catheter / extras arrays missing");
    dst->n_frames = F;
    dst->has_catheter = nc > 0 ? 1 : 0;
    int64_t ol = 0, oc = 0, oe = 0;
    for (int32_t i = 0; i < F; ++i) {
        const BFrame& f = B->frames[(size_t)i];
        dst->id[i] = f.id; dst->lumen_id[i] = f.id; dst->orig_frame[i] = f.orig;
        std::memcpy(dst->centroid + 3 * i, f.centroid, 24);
        dst->lumen_off[i] = ol;
        std::memcpy(dst->lumen + 3 * ol, f.lumen.xyz.data(), f.lumen.xyz.size() * 8);
        if (lumen_aortic_out) {
            if (!f.lumen.aortic.empty()) std::memcpy(lumen_aortic_out + ol, f.lumen.aortic.data(), f.lumen.aortic.size());
            else std::memset(lumen_aortic_out + ol, 0, (size_t)f.lumen.n());
        }
        ol += f.lumen.n();
        if (nc > 0) {
            dst->cath_off[i] = oc;
            std::memcpy(dst->cath + 3 * oc, f.cath.xyz.data(), f.cath.xyz.size() * 8);
            oc += f.cath.n();
        }
        if (ne > 0) dst->extra_off[i] = oe;
        for (int k = 0; k < 3; ++k) {
            const int64_t m = f.has_ext[k] ? f.ext[k].n() : 0;
            if (extra_counts) extra_counts[3 * i + k] = m;
            if (m) { std::memcpy(dst->extra + 3 * oe, f.ext[k].xyz.data(), (size_t)m * 24); oe += m; }
        }
        dst->has_ref[i] = f.has_ref ? 1 : 0;
        dst->ref[3 * i] = f.has_ref ? f.ref[0] : 0.0; dst->ref[3 * i + 1] = f.has_ref ? f.ref[1] : 0.0; dst->ref[3 * i + 2] = f.has_ref ? f.ref[2] : 0.0;
        if (a_th) a_th[i] = f.has_a ? f.a_th : 0.0;
        if (has_a) has_a[i] = f.has_a ? 1 : 0;
        if (p_th) p_th[i] = f.has_p ? f.p_th : 0.0;
        if (has_p) has_p[i] = f.has_p ? 1 : 0;
    }
    dst->lumen_off[F] = ol;
    if (nc > 0) dst->cath_off[F] = oc;
    if (ne > 0) dst->extra_off[F] = oe;
    return MM_OK;
}

void mm_built_destroy(mm_built* h) { delete reinterpret_cast<Built*>(h); }

int mm_contour_centroids(const double* xyz, const int64_t* off, int64_t n, double* out)
{
    if (n < 0 || (n > 0 && (!xyz || !off || !out))) return set_error(MM_ERR_INVALID, "mm_contour_centroids: bad arguments");
    constexpr int64_t kBlock = 64;
    parallel_for((int)((n + kBlock - 1) / kBlock), [&](int blk) {
        for (int64_t c = (int64_t)blk * kBlock; c < std::min(n, ((int64_t)blk + 1) * kBlock); ++c) {
            double sx = 0.0, sy = 0.0, sz = 0.0;
            const int64_t lo = off[c], hi = off[c + 1];
            for (int64_t i = lo; i < hi; ++i) { sx += xyz[3 * i]; sy += xyz[3 * i + 1]; sz += xyz[3 * i + 2]; }
            const double m = (double)(hi - lo);
            out[3 * c] = hi > lo ? sx / m : 0.0; out[3 * c + 1] = hi > lo ? sy / m : 0.0; out[3 * c + 2] = hi > lo ? sz / m : 0.0;
        }
    });
    return MM_OK;
}

}  // extern "C"
