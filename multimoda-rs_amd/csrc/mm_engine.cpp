// mm_engine.cpp -- host side of the C ABI (include/mm_hausdorff.h): engine, batch staging,
// device-resident plans.  Compiled with hipcc -ffp-contract=off.
//
// Data layout in HBM.  Point pool (staged once per plan; a set may serve several pairs):
//   [p32x | p32y]   f32 SoA, coordinates relative to the set's centre (screening kernel)
//   [p64x | p64y]   f64 SoA, coordinates as given                     (exact kernel)
// Level blob (re-stageable: a coarse->fine search re-uses the resident points):
//   [PairDesc x P][WorkItem x W][cos32 | sin32 | cos64 | sin64]   <- one H2D copy
//   [sq32 | sq64 | flag | items | n_items]                         per-candidate scratch
//   [best_cost | best_idx | n_rescored | near_cnt | near_idx]      <- one D2H copy
//   [all_costs]                                                    optional
// cos/sin tables are computed on the host with glibc sin/cos (what Rust's f64::sin/cos
// call on linux-gnu) and shared by pairs with identical candidate lists.
#include "mm_engine.h"
#include "mm_pool.h"
#include "mm_trace.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <array>
#include <vector>

namespace mm {

thread_local std::string g_last_error;

int set_error(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

int hip_error(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? MM_ERR_NO_DEVICE : MM_ERR_HIP;
}

#define MM_HIP(call)                                         \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return hip_error(e__, #call); \
    } while (0)

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// -------------------------------------------------------------------------------------
// engine
// -------------------------------------------------------------------------------------
int Engine::ensure(Buf& b, size_t bytes, bool host)
{
    if (bytes <= b.cap) return MM_OK;
    if (b.p) {
        MM_HIP(hipStreamSynchronize(stream));  // a copy may still read the old buffer
        if (aux && aux != stream) MM_HIP(hipStreamSynchronize(aux));
        if (host) (void)hipHostFree(b.p); else (void)hipFree(b.p);
    }
    b.p = nullptr; b.cap = 0;
    size_t cap = std::max(bytes, host ? (size_t)1 << 20 : (size_t)4 << 20);
    cap = align_up(cap + cap / 2, 4096);
    if (host) MM_HIP(hipHostMalloc(&b.p, cap, hipHostMallocDefault));
    else MM_HIP(hipMalloc(&b.p, cap));
    b.cap = cap;
    return MM_OK;
}

int Engine::blob_alloc(void** p, size_t bytes, size_t* cap)
{
    bytes = std::max<size_t>(align_up(bytes, 4096), 4096);
    int best = -1;
    for (int i = 0; i < (int)blob_cache.size(); ++i)
        if (blob_cache[i].cap >= bytes && blob_cache[i].cap <= 2 * bytes + (1 << 20) && (best < 0 || blob_cache[i].cap < blob_cache[best].cap)) best = i;
    if (best >= 0) {
        *p = blob_cache[best].p; *cap = blob_cache[best].cap;
        blob_cache.erase(blob_cache.begin() + best);
        return MM_OK;
    }
    MM_HIP(hipMalloc(p, bytes));
    *cap = bytes;
    return MM_OK;
}

void Engine::blob_release(void* p, size_t cap)
{
    if (!p) return;
    if ((int)blob_cache.size() >= kBlobCache) {   // drop the smallest cached block (hipFree synchronises the device)
        int small = 0;
        for (int i = 1; i < (int)blob_cache.size(); ++i) if (blob_cache[i].cap < blob_cache[small].cap) small = i;
        if (blob_cache[small].cap < cap) { (void)hipFree(blob_cache[small].p); blob_cache[small] = Buf{p, cap}; }
        else (void)hipFree(p);
        return;
    }
    blob_cache.push_back(Buf{p, cap});
}

int Engine::sync_all()
{
    MM_HIP(hipStreamSynchronize(stream));
    if (aux && aux != stream) MM_HIP(hipStreamSynchronize(aux));
    return MM_OK;
}

int Engine::profile_begin(hipStream_t st)
{
    if (!profile) return MM_OK;
    while (events.size() < 2 * (launches + 1)) {
        hipEvent_t ev;
        MM_HIP(hipEventCreate(&ev));
        events.push_back(ev);
    }
    MM_HIP(hipEventRecord(events[2 * launches], st));
    return MM_OK;
}

int Engine::profile_end(hipStream_t st, double pair_evals, int64_t candidates)
{
    if (!profile) return MM_OK;
    MM_HIP(hipEventRecord(events[2 * launches + 1], st));
    ++launches;
    prof_pair_evals += pair_evals;
    prof_candidates += candidates;
    launch_pair_evals.push_back(pair_evals);
    return MM_OK;
}

// -------------------------------------------------------------------------------------
// plan: point pool
// -------------------------------------------------------------------------------------
// Pinned host -> HBM.  Small uploads of transient batches often follow a kernel on the side stream (the second level
// of a between search behind the first level's kernels); the runtime executes such a copy as a 512-thread blit
// kernel, which is not dispatched beside another engine's screen launch (see k_copy_small): those go through the
// 256-thread copy kernel (pinned host memory is device-accessible).  Everything else is a plain async copy (SDMA).
static hipError_t upload(void* dst, const void* src, size_t bytes, bool transient, hipStream_t st)
{
    if (transient && bytes <= ((size_t)1 << 20)) return launch_copy_small(dst, src, bytes, st);
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
}

// Layout and allocation of the point pool for sets of the given sizes; no data yet.
int Plan::alloc_pool(Engine* e, const std::vector<int32_t>& lens, bool transient_)
{
    eng = e;
    transient = transient_;
    stream = transient ? e->aux : e->stream;
    const size_t S = lens.size();
    set_off.assign(S, 0); set_len.assign(S, 0);
    set_rho.assign(S, 0.0);
    n_points = 0;
    for (size_t s = 0; s < S; ++s) {
        if (lens[s] < 0) return set_error(MM_ERR_INVALID, "negative set size");
        set_off[s] = (int32_t)n_points; set_len[s] = lens[s];
        n_points += lens[s];
        if (n_points > (int64_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "batch exceeds 2^30 points");
    }
    const size_t n = (size_t)n_points;
    o32x = 0; o32y = align_up(o32x + n * 4); o64x = align_up(o32y + n * 4); o64y = align_up(o64x + n * 8);
    pts_bytes = align_up(o64y + n * 8);
    if (transient) {
        int rc = e->ensure(e->dev_pts, pts_bytes, false);
        if (rc) return rc;
        pts_blob = (unsigned char*)e->dev_pts.p; own_pts = false;
    } else {
        int rc = e->blob_alloc((void**)&pts_blob, pts_bytes, &pts_cap);
        if (rc) return rc;
        own_pts = true;
    }
    dev.p32x = (const float*)(pts_blob + o32x); dev.p32y = (const float*)(pts_blob + o32y);
    dev.p64x = (const double*)(pts_blob + o64x); dev.p64y = (const double*)(pts_blob + o64y);
    return MM_OK;
}

int Plan::stage_sets(Engine* e, const std::vector<SetRef>& sets, bool transient_, hipStream_t st)
{
    const size_t S = sets.size();
    std::vector<int32_t> lens(S);
    for (size_t s = 0; s < S; ++s) lens[s] = sets[s].n;
    int rc;
    if ((rc = alloc_pool(e, lens, transient_))) return rc;
    if (!st) st = stream;
    if ((rc = e->ensure(e->host_pts, pts_bytes, true))) return rc;
    unsigned char* h = (unsigned char*)e->host_pts.p;
    float *x32 = (float*)(h + o32x), *y32 = (float*)(h + o32y);
    double *x64 = (double*)(h + o64x), *y64 = (double*)(h + o64y);
    for (size_t s = 0; s < S; ++s) {
        const SetRef& r = sets[s];
        const int32_t o = set_off[s];
        double rho2 = 0.0;
        bool bad = false;   // NaN / inf / overflowing coordinates: no f32 screen can bound them (rho = inf -> stage_level)
        for (int32_t i = 0; i < r.n; ++i) {
            const double dx = r.x[i] - r.cx, dy = r.y[i] - r.cy;
            x64[o + i] = r.x[i]; y64[o + i] = r.y[i];
            x32[o + i] = (float)dx; y32[o + i] = (float)dy;
            const double r2 = dx * dx + dy * dy;
            if (r2 <= 1.0e300) rho2 = std::max(rho2, r2); else bad = true;
        }
        set_rho[s] = bad ? INFINITY : std::sqrt(rho2) * (1.0 + 1e-12);
    }
    if (pts_bytes) MM_HIP(upload(pts_blob, h, pts_bytes, transient, st));
    if (!transient) MM_HIP(hipStreamSynchronize(st));
    return MM_OK;
}

// -------------------------------------------------------------------------------------
// plan: level (descriptors, candidate tables, outputs)
// -------------------------------------------------------------------------------------
// f32 screening error bound (DESIGN.md "screen-then-exact"): with u = 2^-24 and rho_r, rho_t
// the largest distance of a reference / target point from the rotation centre,
// |H_f32 - H_true| <= u (3.9 rho_r + 10 rho_t); we use 24 u (rho_r + rho_t).
// The screen works on coordinates relative to the rotation centre; the exact re-score is the REFERENCE's
// arithmetic on the coordinates as given, whose rotation ends with "+ cx" (contour_point.rs:45-48): its result
// is rounded at the magnitude of the coordinates, |H_f64 - H_true| <= 2^-53 (1.5 |c| + 10 rho_t + 3 rho_r).
// That term only matters for point sets a billion times smaller than their coordinates (all points of a set
// equal: every exact cost is 0.0, the screen sees 1e-17 -- found by the randomised search test), but the
// shortlist has to hold every candidate whose EXACT cost can be minimal, so it is part of the bound:
// 2^-49 (|cx| + |cy| + rho_r + rho_t), sixteen roundings at the coordinates' magnitude.
static inline double screen_delta(double rho_r, double rho_t, double cx, double cy)
{
    const double u = 5.9604644775390625e-08;  // 2^-24
    return 24.0 * u * (rho_r + rho_t) + std::ldexp(std::fabs(cx) + std::fabs(cy) + rho_r + rho_t, -49) + 1e-300;
}

// Error bound of the matrix-pipe screen's squared distances: |S~ - S| <= mx_e2(rho_a, rho_b), with rho_a (rho_b) the largest
// distance of a reference (target) point from the rotation centre, R = rho_a + rho_b, u = 2^-24.  In the kernel's scaled
// units (larger radius in [256, 512)) the screened value of a point pair is
//     S~ = fl_acc( |a~|^2~ + n2(b) - 2 a~ . b~' ),   a~ = a1 + a2, b~' = b1' + b2' the f16 hi + lo splits of a and of R32(b),
// the products exact in fp32 (11 x 11 bits), and the budget is (every term in units of u):
//   * split of both points -- |a - a~|, |b' - b~'| <= 2^-22 |.| per coordinate (two roundings of relative size 2^-11), or
//     2^-14 absolute where a lo piece falls into f16's subnormal range and the matrix pipe flushes it; the distance moves by
//     2 |a - b| (|da| + |db|) <= 2 R sqrt(2) 2^-22 R = 11.3 R^2, with the absolute form 2 R 2 sqrt(2) 2^-14 = 22.6 R^2 (256 / R)
//     <= 22.6 R^2 because the larger radius is at least 256:                                                        23 R^2
//   * the row norm |a~|^2: two f32 roundings (2 rho_a^2), its own split n2 = 256 nh + nl (nh to 0.25, nl to 2^-6 absolute
//     = 4 u rho^2 at rho = 256, less above):                                                                        6 rho_a^2
//   * the column norm is taken from the UNROTATED target point, once per work item: the same 6, the f32 rotation (cos / sin
//     tables rounded to f32: |c^2 + s^2 - 1| <= 1.5 u; two fma per coordinate: |b' - R b| <= 2 sqrt(2) u rho_b) keeps
//     |b|^2 to 9, and the split of the rotated point moves |b'|^2 by 2 rho_b sqrt(2) 2^-22 rho_b = 11.3:               27 rho_b^2
//   * twelve fp32 accumulations inside the MFMA, rounding mode and order unspecified: 2 u each on partial sums that stay
//     below (|a| + |b|)^2 <= R^2:                                                                                   24 R^2
// Sum: u (47 R^2 + 6 rho_a^2 + 27 rho_b^2) -- 55 u R^2 for equal radii.  (Rounds 3 shipped a flat 128 u R^2; the directed
// search of tests/test_gpu_mx_error_bound.py -- coordinates on f16 ties, radii at both edges of the scale, far outliers,
// subnormal lo pieces, tile-edge set sizes, the angles whose f32 (cos, sin) is furthest from unit norm -- found at most
// 2.2 % of that, 5 % of this.)  min and max are 1-Lipschitz, so the bound carries over to the screened Hausdorff value.
static double mx_e2(double rho_a, double rho_b)
{
    const double u = 5.9604644775390625e-08, R = rho_a + rho_b;
    return u * (47.0 * R * R + 6.0 * rho_a * rho_a + 27.0 * rho_b * rho_b);
}

int Plan::stage_level(const std::vector<PairSpec>& pairs, int precision_, int32_t angle_begin, int32_t angle_end,
                      bool want_costs_, hipStream_t st)
{
    Engine* e = eng;
    if (!st) st = stream;
    precision = precision_;
    want_costs = want_costs_;
    slice_end = angle_end;
    P = (int)pairs.size();
    if (precision != MM_PRECISION_F64 && precision != MM_PRECISION_F32 && precision != MM_PRECISION_F32_FAST &&
        precision != MM_PRECISION_F32_BOUNDED && precision != MM_PRECISION_F32_MATRIX)
        return set_error(MM_ERR_INVALID, "precision must be MM_PRECISION_F64, MM_PRECISION_F32, MM_PRECISION_F32_FAST, "
                                         "MM_PRECISION_F32_BOUNDED or MM_PRECISION_F32_MATRIX");
    // Non-finite (or overflowing) coordinates: the reference's metric skips the points whose distances are not finite
    // (process_utils.rs:112-114) and carries on.  The exact kernels do the same; a screen cannot bound such values, so
    // every candidate of this level is scored exactly.  (rho = inf marks such a set, see stage_sets / k_build_sets.)
    // The same for finite coordinates an f32 screen cannot hold: squared distances reach (rho_r + rho_t)^2, which must
    // stay below FLT_MAX (rho < 1e18 leaves a factor 1e2), and the error bounds of the screens (relative to rho) assume
    // that u * rho^2 is far above the f32 denormals (rho > 1e-15 leaves a factor 1e8); rho == 0 (all points at the
    // centre) is exact in any format.
    auto f32_safe = [](double rho) { return rho < 1.0e18 && (rho == 0.0 || rho > 1.0e-15); };
    if (precision != MM_PRECISION_F64)
        for (const PairSpec& sp : pairs)
            if (sp.ref_set >= 0 && sp.tgt_set >= 0 && (size_t)sp.ref_set < set_rho.size() && (size_t)sp.tgt_set < set_rho.size() &&
                (!f32_safe(set_rho[sp.ref_set]) || !f32_safe(set_rho[sp.tgt_set]) || !std::isfinite(sp.cx) || !std::isfinite(sp.cy))) {
                precision = MM_PRECISION_F64;
                break;
            }
    const bool expanded = precision == MM_PRECISION_F32_FAST || precision == MM_PRECISION_F32_BOUNDED ||
                          precision == MM_PRECISION_F32_MATRIX;
    if (angle_begin < 0) angle_begin = 0;

    TraceTimer tt_desc("stage_level: descriptors");
    host_pairs.assign(P, PairDesc{});
    trivial.assign(P, 0);
    pair_slice_end.assign(P, 0);
    host_tables.clear();
    A = 0; T = 0;
    max_na = 1; max_nbp = 16; max_nt = 1;
    pair_evals = 0.0;
    // table sharing: same list as the previous distinct table (pointer or content) and same slice
    const double* last_ptr = nullptr; int32_t last_n = -1, last_b = -1, last_tab = 0, last_len = -1;
    for (int p = 0; p < P; ++p) {
        const PairSpec& sp = pairs[p];
        if (sp.ref_set < 0 || sp.tgt_set < 0 || (size_t)sp.ref_set >= set_len.size() || (size_t)sp.tgt_set >= set_len.size())
            return set_error(MM_ERR_INVALID, "pair references a set that was not staged");
        if (sp.n_angles < 0) return set_error(MM_ERR_INVALID, "negative candidate count");
        const int32_t nr = set_len[sp.ref_set], nt = set_len[sp.tgt_set];
        const int32_t b = std::min(std::max(angle_begin, sp.slice_begin), sp.n_angles);
        const int32_t en = std::min(std::min(angle_end, sp.slice_end), sp.n_angles);
        const int32_t na = std::max(en - b, 0);
        pair_slice_end[p] = std::max(en, b);
        PairDesc& d = host_pairs[p];
        d.ref_off = set_off[sp.ref_set]; d.n_ref = nr;
        d.tgt_off = set_off[sp.tgt_set]; d.n_tgt = nt;
        d.out_off = (int32_t)A; d.n_ang = na;
        d.ang_full = sp.n_angles; d.ang_begin = b;
        d.n_slice = na; d.pad0 = 0;
        d.flags = sp.flags;
        d.cx = sp.cx; d.cy = sp.cy;
        d.tol2 = sp.tie_tol;
        d.delta = (precision != MM_PRECISION_F64)
                      ? screen_delta(set_rho[sp.ref_set], set_rho[sp.tgt_set], sp.cx, sp.cy) + sp.delta_extra : 0.0;
        {   // expanded-form screening: |d2_f32 - d2| <= 5 u (rho_a + rho_b)^2; we use 8 u (..)^2
            const double rs = set_rho[sp.ref_set] + set_rho[sp.tgt_set];
            d.e2 = expanded ? 8.0 * 5.9604644775390625e-08 * rs * rs : 0.0;
            d.rho_t = set_rho[sp.tgt_set];
        }
        if (nr == 0 || nt == 0) {
            // process_utils.rs:86-88: an empty set makes every cost 0.0 -> the first candidate wins;
            // nothing to launch for this pair (one table entry keeps its first angle).
            trivial[p] = 1;
            d.n_ang = 0;
            d.tab_off = (int32_t)host_tables.size();
            if (na > 0) host_tables.push_back(sp.angles[b]);
            last_ptr = nullptr; last_len = -1;
            continue;
        }
        const bool share = (na == last_len && b == last_b && sp.n_angles == last_n) &&
                           (sp.angles == last_ptr ||
                            (na > 0 && std::memcmp(sp.angles + b, host_tables.data() + last_tab, (size_t)na * 8) == 0));
        if (!share) {
            last_tab = (int32_t)host_tables.size();
            host_tables.insert(host_tables.end(), sp.angles + b, sp.angles + b + na);
            last_len = na; last_b = b; last_n = sp.n_angles;
        }
        last_ptr = sp.angles;
        d.tab_off = last_tab;
        A += na;
        max_na = std::max<int>(max_na, nr);
        max_nbp = std::max<int>(max_nbp, (nt + 15) & ~15);
        max_nt = std::max<int>(max_nt, nt);
        pair_evals += 2.0 * (double)nr * (double)nt * (double)na;
        if (A > (int64_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "batch exceeds 2^30 candidates");
    }
    T = (int64_t)host_tables.size();
    // the fast screening kernel keeps one row block in registers; bigger sets use the direct form
    use_fast = expanded && max_na <= max_rows_fast() && max_nbp <= max_target_points_fast();
    if (expanded && !use_fast)
        for (PairDesc& d : host_pairs) d.e2 = 0.0;
    // Matrix-pipe screen, chosen PER PAIR: sets of mx_min_points() .. mx_max_points() points (the kernel is instantiated
    // per column-tile count, its row-tile count is a run-time operand, larger target sets are cut into column blocks), a
    // scale exponent that puts the larger radius into [256, 512) (f16 pieces, their doubles and |x|^2 / 256 stay in range),
    // and the error bound of its squared values (mx_e2 below).  A pair outside that range keeps the packed-FMA screen (or,
    // in a batch whose largest sets exceed that kernel's registers, the direct form) with that kernel's own e2; its work
    // items form a group of their own (Plan::groups).
    // the bound rounds pay for themselves on sets of a few dozen points or more and on batches that keep
    // the device busy for several rounds of workgroups (a dozen dependent launches cost more than
    // screening a small batch outright: the between stage's 2 x 722 candidates are 40 % faster without);
    // per-candidate costs need every candidate evaluated
    use_lb = precision == MM_PRECISION_F32_BOUNDED && use_fast && !want_costs && A > 0 && A >= e->bound_min_candidates &&
             std::min(max_na, max_nt) >= 64 && std::max(max_na, max_nt) <= lb_max_points();
    // The bound rounds on the matrix pipe (k_bound_mx) and the survivors through k_screen_mx: every non-trivial pair with
    // sets of 64 .. lb_mx_max_points() points and a radius f16 can be scaled to; the pairs then carry the scale exponent and
    // the matrix kernels' error bound (the larger of the two directions': pass 1 rotates reference queries against target
    // rows, pass 2 target queries against reference rows).  The two per-pair picks stay on the packed-FMA screen (it emits
    // the row / column minima the third round's query choice needs); its e2 is the smaller one.
    lb_mx_tiles = 0; kept_nct = 0; kept_acap = 0;
    if (use_lb && e->bound_matrix) {
        bool ok = true;
        int tiles = 1, nct0 = -1, acap = 1;
        for (int p = 0; p < P && ok; ++p) {
            const PairDesc& d = host_pairs[p];
            if (trivial[p] || d.n_ang == 0) continue;
            const double rmax = std::max(set_rho[pairs[p].ref_set], set_rho[pairs[p].tgt_set]);
            ok = d.n_ref >= mx_min_points() && d.n_tgt >= mx_min_points() && d.n_ref <= lb_mx_max_points() &&
                 d.n_tgt <= lb_mx_max_points() && rmax > 1.0e-30 && rmax < 1.0e30;
            tiles = std::max(tiles, (std::max(d.n_ref, d.n_tgt) + 31) / 32);
            int nct = 0, multi = 0;
            mx_variant(d.n_tgt, &nct, &multi);
            if (multi) nct = 0;
            nct0 = nct0 < 0 ? nct : (nct0 == nct ? nct0 : 0);     // 0: the pairs do not share one variant
            acap = std::max(acap, (d.n_ref + 31) / 32);
        }
        // The picks and the survivors go through k_screen_mx from device queues, which takes ONE variant per launch: every
        // pair must have the same column-tile count (and <= 17 row tiles, two workgroups per CU).  (k_screen_mx gives a
        // candidate to one wave, ~10 us, where the packed-FMA screen spends 2 - 3 us of a whole workgroup -- 222 against
        // 156 us per step for config3's survivors -- but the packed-FMA kernels are not to run beside MFMA kernels:
        // profiles/README.md, "packed-FMA kernels beside MFMA kernels".)  A batch of mixed shapes, or one with sets beyond
        // the bound kernel's range, is screened outright on the matrix pipe instead.
        ok = ok && nct0 > 0 && acap <= 17;
        if (!ok) use_lb = false;
        if (ok) {
            lb_mx_tiles = tiles;
            kept_nct = nct0; kept_acap = acap;
            for (int p = 0; p < P; ++p) {
                PairDesc& d = host_pairs[p];
                if (trivial[p] || d.n_ang == 0) continue;
                const double ra = set_rho[pairs[p].ref_set], rb = set_rho[pairs[p].tgt_set];
                int k = 0;
                (void)std::frexp(std::max(ra, rb) * (1.0 + 1e-6), &k);
                d.pad0 = 9 - k;
                d.e2 = std::max(mx_e2(ra, rb), mx_e2(rb, ra));
            }
        }
    }
    // A bounded search that does not run its bound rounds (small batch, per-candidate costs asked for, sets outside the
    // bound kernel's range) screens every candidate: on the matrix pipe, like MM_PRECISION_F32_MATRIX.
    const bool mx_wanted = precision == MM_PRECISION_F32_MATRIX || (precision == MM_PRECISION_F32_BOUNDED && !use_lb);
    use_mx = false;
    std::vector<int> pair_key((size_t)P, 0);       // 0: direct form, 1: packed FMA, else 2 + the matrix kernel's variant
    if (mx_wanted && A > 0) {
        for (int p = 0; p < P; ++p) {
            PairDesc& d = host_pairs[p];
            if (trivial[p] || d.n_ang == 0) continue;
            pair_key[(size_t)p] = use_fast ? 1 : 0;
            // a set of fewer than 64 points: no f32 screen, every candidate is scored exactly (cheap at that size, and no
            // packed-FMA kernel runs under this precision unless a set exceeds mx_max_points())
            if (std::min(d.n_ref, d.n_tgt) < mx_min_points() && std::max(d.n_ref, d.n_tgt) <= mx_max_points()) { pair_key[(size_t)p] = -1; use_mx = true; continue; }
            const double ra = set_rho[pairs[p].ref_set], rb = set_rho[pairs[p].tgt_set], rmax = std::max(ra, rb);
            if (d.n_ref < mx_min_points() || d.n_ref > mx_max_points() || d.n_tgt < mx_min_points() || d.n_tgt > mx_max_points() ||
                !(rmax > 1.0e-30) || !(rmax < 1.0e30))
                continue;
            int nct = 0, multi = 0, k = 0;
            mx_variant(d.n_tgt, &nct, &multi);
            // LDS is sized per launch by the largest reference set of the group: up to 17 row tiles leave room for two
            // workgroups per CU, more than that for one -- two classes, so that one long contour does not halve the
            // occupancy of every other pair's launch
            const int nrt = (d.n_ref + 31) / 32, cls = nrt <= 17 ? 0 : 1;
            (void)std::frexp(rmax * (1.0 + 1e-6), &k);   // radius < 2^k
            d.pad0 = 9 - k;
            d.e2 = mx_e2(ra, rb);
            pair_key[(size_t)p] = 2 + ((((multi << 8) | nct) << 1 | cls) << 2);
            use_mx = true;
        }
    }
    if (max_nbp > max_target_points_f64() || (precision != MM_PRECISION_F64 && max_nbp > max_target_points_f32()))
        return set_error(MM_ERR_TOO_LARGE, "target set does not fit the kernel's LDS budget (" +
                                               std::to_string(max_target_points_f64()) + " points)");

    tt_desc.stop();
    TraceTimer tt_work("stage_level: work list");
    // ---- work decomposition: one workgroup = `apb` consecutive candidates of one pair ----
    const int64_t target_wgs = 256 * 24;
    // the packed-FMA and exact kernels: at most 8 candidates per workgroup, measured on config3 (apb 2/4/8/16/64/128: 140.6/145.0/146.3/
    // 144.6/138.2/122.3 TFLOP/s) -- many short workgroups keep the co-resident ones out of phase
    // (rotation / epilogue of one overlaps the micro-tile loop of the others) and balance the tail
    int apb = (int)std::min<int64_t>(8, std::max<int64_t>(1, (A + target_wgs - 1) / target_wgs));
    // matrix-pipe screen: one WAVE per candidate, four waves per workgroup -- a workgroup of fewer than four candidates
    // leaves waves idle for the whole item (small batches: a single search of 361 candidates)
    if (use_mx) apb = std::max(apb, 4);
    // ... and with the matrix-pipe screen more of them: a work item stages the pair's rows once (global loads, two barriers:
    // ~3 us, with one other workgroup on the CU to hide it) and a 17 x 17-tile candidate takes a wave 10 us, a 7 x 7 one 1.8.
    // A wave should see ~2300 tiles per item (17 x 17 tiles: 8 candidates, 32 per workgroup; small sets up to 16 per wave) as
    // far as the batch has candidates for 24 workgroups per CU: config3's launch 16.19 ms at 578 tiles, 16.00 at 1156,
    // 15.83 at 2312, 15.77 at 4624 (tools/exp_apb.sh).
    const int64_t apb_fill = std::max<int64_t>(1, (A + target_wgs - 1) / target_wgs);
    auto apb_of = [&](int p) {
        if (!use_mx || pair_key[(size_t)p] < 2) return apb;
        const PairDesc& d = host_pairs[p];
        int nct = 0, multi = 0;
        mx_variant(d.n_tgt, &nct, &multi);
        const int64_t tiles = (int64_t)((d.n_ref + 31) / 32) * ((d.n_tgt + 31) / 32);
        const int64_t per_wave = std::min<int64_t>(16, std::max<int64_t>(2, (2312 + tiles - 1) / tiles));
        return (int)std::max<int64_t>(apb, std::min<int64_t>(4 * per_wave, apb_fill));
    };
    groups.clear();
    {
        // balanced chunks: ceil(n / apb) workgroups whose sizes differ by at most one (a 90-candidate
        // slice is 12 x 7-8 candidates, not 11 x 8 + 2: the short tail would cost a full staging).
        // 186 k items for config3: sized by a prefix sum and filled over the worker pool (one push_back at a time
        // this was half of a case's staging time).  With the matrix-pipe screen the pairs are laid out group by group
        // (stable: pair-major inside a group, as the XCD-aware work order wants it).
        std::vector<int> order((size_t)P);
        for (int p = 0; p < P; ++p) order[(size_t)p] = p;
        if (use_mx) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return pair_key[(size_t)x] < pair_key[(size_t)y]; });
        std::vector<int64_t> wstart((size_t)P + 1, 0);
        for (int q = 0; q < P; ++q) {
            const int ap = apb_of(order[(size_t)q]);
            wstart[(size_t)q + 1] = wstart[(size_t)q] + (host_pairs[order[(size_t)q]].n_ang + ap - 1) / ap;
        }
        if (wstart[(size_t)P] > INT32_MAX) return set_error(MM_ERR_TOO_LARGE, "too many work items");
        host_work.resize((size_t)wstart[(size_t)P]);
        WorkItem* hw = host_work.data();
        constexpr int kBlk = 64;
        parallel_for((P + kBlk - 1) / kBlk, [&](int blk) {
            for (int q = blk * kBlk; q < std::min(P, (blk + 1) * kBlk); ++q) {
                const int p = order[(size_t)q];
                const PairDesc& d = host_pairs[p];
                const int nw = (int)(wstart[(size_t)q + 1] - wstart[(size_t)q]);
                WorkItem* o = hw + wstart[(size_t)q];
                for (int k = 0; k < nw; ++k) {
                    const int a0 = (int)((int64_t)d.n_ang * k / nw), a1 = (int)((int64_t)d.n_ang * (k + 1) / nw);
                    o[k] = WorkItem{p, a0, a1 - a0, 0};
                }
            }
        });
        if (use_mx)
            for (int q = 0; q < P;) {
                const int key = pair_key[(size_t)order[(size_t)q]];
                int q1 = q, a_cap = 1;
                while (q1 < P && pair_key[(size_t)order[(size_t)q1]] == key) {
                    const PairDesc& d = host_pairs[order[(size_t)q1]];
                    if (d.n_ang > 0 && !trivial[order[(size_t)q1]]) a_cap = std::max(a_cap, (d.n_ref + 31) / 32);
                    ++q1;
                }
                const int wb = (int)wstart[(size_t)q], wc = (int)(wstart[(size_t)q1] - wstart[(size_t)q]);
                if (wc > 0)
                    groups.push_back(key < 0 ? ScreenGroup{3, 0, 0, 0, wb, wc} : key < 2 ? ScreenGroup{key, 0, 0, 0, wb, wc}
                                             : ScreenGroup{2, ((key - 2) >> 3) & 0xff, ((key - 2) >> 11) & 1, a_cap, wb, wc});
                q = q1;
            }
    }
    W = (int)host_work.size();
    host_work_lb.clear();
    W_lb = 0; lb_runs_cap = 0; lb_pair_evals = 0.0; lb_sparse_total = 0;
    if (use_lb) {
        // every lb_stride-th point of either set is a query; the subset has to fit the kernel's registers
        const int qmax = lb_mx_tiles > 0 ? 32 * e->bound_matrix_qt : lb_max_query_points();
        lb_stride = std::max(lb_mx_tiles > 0 ? 1 : 8, (std::max(max_na, max_nt) + qmax - 1) / qmax);
        // first round: every lb_candidate_step()-th candidate and the last one, 32 of them per workgroup
        // (4 waves x 8: amortises staging the reference set)
        const int apb_lb = 32, cstep = lb_candidate_step();
        for (int p = 0; p < P; ++p) {
            const PairDesc& d = host_pairs[p];
            const int ne = lb_sparse_candidates(d.n_ang);
            for (int i0 = 0; i0 < ne; i0 += apb_lb)
                host_work_lb.push_back(WorkItem{p, i0 * cstep, std::min(apb_lb, ne - i0), cstep});
            lb_runs_cap += (d.n_ang + 7) / 8;
            const double qa = (d.n_ref + lb_stride - 1) / lb_stride, qb = (d.n_tgt + lb_stride - 1) / lb_stride;
            lb_pair_evals += (qa * d.n_tgt + qb * d.n_ref) * (double)ne;
            lb_sparse_total += ne;
        }
        W_lb = (int)host_work_lb.size();
    }

    tt_work.stop();
    TraceTimer tt_copy("stage_level: layout + tables + copies");
    // ---- layout ---------------------------------------------------------------------------
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = align_up(o + std::max<size_t>(bytes, 16)); return at; };
    const size_t o_pairs = take((size_t)P * sizeof(PairDesc));
    const size_t o_work = take((size_t)W * sizeof(WorkItem));
    const size_t o_work_lb = take((size_t)W_lb * sizeof(WorkItem));
    const size_t o_c32 = take((size_t)T * 4), o_s32 = take((size_t)T * 4);
    const size_t o_c64 = take((size_t)T * 8), o_s64 = take((size_t)T * 8), o_a64 = take((size_t)T * 8);
    lvl_in_bytes = o;
    const size_t o_sq32 = take((size_t)A * 4), o_sq64 = take((size_t)A * 8), o_flag = take((size_t)A);
    const size_t o_items = take((size_t)A * sizeof(WorkItem)), o_nitems = take(32);
    const int emit_rows = (max_na + 3) & ~3, emit_cols = (max_nt + 3) & ~3;
    const size_t o_lb32 = take(use_lb ? (size_t)A * 4 : 0), o_pick = take(use_lb ? (size_t)P * 8 : 0);
    const size_t o_items_pick = take(use_lb ? (size_t)P * 2 * sizeof(WorkItem) : 0);
    const size_t o_items_lb = take(use_lb ? (size_t)lb_runs_cap * 3 * sizeof(WorkItem) : 0);
    const size_t o_klist = take(use_lb && lb_mx_tiles > 0 ? (size_t)A * 4 : 0);
    const size_t o_emit = take(use_lb ? (size_t)P * (size_t)(emit_rows + emit_cols) * 4 : 0);
    const size_t o_qlist = take(use_lb ? (size_t)P * 2 * (size_t)lb_list_queries() * 4 : 0);
    const size_t o_bc = take((size_t)P * 8), o_bi = take((size_t)P * 4), o_nr = take((size_t)P * 4);
    const size_t o_nc = take((size_t)P * 4), o_ni = take((size_t)P * 4 * kMaxNear);
    off_best_cost = o_bc; res_bytes = o - o_bc;
    off_all_costs = want_costs ? take((size_t)A * 8) : 0;
    lvl_bytes = o;
    r_best_idx = o_bi - o_bc; r_n_rescored = o_nr - o_bc; r_near_cnt = o_nc - o_bc; r_near_idx = o_ni - o_bc;

    // ---- stage inputs -----------------------------------------------------------------------
    int rc = e->ensure(e->host_lvl, std::max(lvl_in_bytes, res_bytes), true);
    if (rc) return rc;
    unsigned char* h = (unsigned char*)e->host_lvl.p;
    float *c32 = (float*)(h + o_c32), *s32 = (float*)(h + o_s32);
    double *c64 = (double*)(h + o_c64), *s64 = (double*)(h + o_s64);
    if (T) std::memcpy(h + o_a64, host_tables.data(), (size_t)T * 8);
    for (int64_t t = 0; t < T; ++t) {
        double co, si;
        ::sincos(host_tables[(size_t)t], &si, &co);   // one glibc sincos, like the reference's per-point cos()/sin() pair
        c64[t] = co; s64[t] = si; c32[t] = (float)co; s32[t] = (float)si;
    }
    if (P) std::memcpy(h + o_pairs, host_pairs.data(), (size_t)P * sizeof(PairDesc));
    if (W) std::memcpy(h + o_work, host_work.data(), (size_t)W * sizeof(WorkItem));
    if (W_lb) std::memcpy(h + o_work_lb, host_work_lb.data(), (size_t)W_lb * sizeof(WorkItem));

    if (transient) {
        rc = e->ensure(e->dev_lvl, lvl_bytes, false);
        if (rc) return rc;
        lvl_blob = (unsigned char*)e->dev_lvl.p; own_lvl = false;
    } else if (lvl_bytes > lvl_cap) {
        if (lvl_blob) { MM_HIP(hipStreamSynchronize(stream)); MM_HIP(hipStreamSynchronize(st)); e->blob_release(lvl_blob, lvl_cap); lvl_blob = nullptr; }
        if ((rc = e->blob_alloc((void**)&lvl_blob, lvl_bytes, &lvl_cap))) return rc;
        own_lvl = true;
    }
    MM_HIP(upload(lvl_blob, h, lvl_in_bytes, transient, st));
    if (!transient) MM_HIP(hipStreamSynchronize(st));  // host staging buffer is reused

    unsigned char* B = lvl_blob;
    dev.pairs = (const PairDesc*)(B + o_pairs); dev.work = (const WorkItem*)(B + o_work);
    dev.n_pairs = P; dev.n_work = W;
    dev.cos32 = (const float*)(B + o_c32); dev.sin32 = (const float*)(B + o_s32);
    dev.cos64 = (const double*)(B + o_c64); dev.sin64 = (const double*)(B + o_s64);
    dev.ang64 = (const double*)(B + o_a64);
    dev.sq32 = (float*)(B + o_sq32); dev.sq64 = (double*)(B + o_sq64); dev.flag = (uint8_t*)(B + o_flag); dev.n_cand = A;
    dev.items = (WorkItem*)(B + o_items); dev.n_items = (int32_t*)(B + o_nitems);
    dev.work_lb = (const WorkItem*)(B + o_work_lb); dev.n_work_lb = W_lb; dev.lb_stride = lb_stride;
    dev.lb_mx = lb_mx_tiles; dev.kept_mx_nct = kept_nct; dev.kept_mx_acap = kept_acap;
    dev.lb_mx_qt = e->bound_matrix_qt; dev.lb_mx_nc = e->bound_matrix_nc;
    dev.lb32 = (float*)(B + o_lb32); dev.pick_idx = (int32_t*)(B + o_pick); dev.items_pick = (WorkItem*)(B + o_items_pick);
    dev.items_lb = (WorkItem*)(B + o_items_lb); dev.emit = (float*)(B + o_emit); dev.emit_rows = emit_rows; dev.emit_cols = emit_cols;
    dev.qlist = (int32_t*)(B + o_qlist); dev.klist = (int32_t*)(B + o_klist);
    dev.best_cost = (double*)(B + o_bc); dev.best_idx = (int32_t*)(B + o_bi); dev.n_rescored = (int32_t*)(B + o_nr);
    dev.near_cnt = (int32_t*)(B + o_nc); dev.near_idx = (int32_t*)(B + o_ni);
    dev.all_costs = want_costs ? (double*)(B + off_all_costs) : nullptr;
    return MM_OK;
}

int Plan::run(bool screen_only)
{
    hipStream_t s = stream;
    if (W == 0) {
        if (!screen_only && P > 0) { hipError_t e = launch_finalize(dev, 0, s); if (e != hipSuccess) return hip_error(e, "finalize"); }
        return MM_OK;
    }
    hipError_t e;
    int prc;
    if (precision != MM_PRECISION_F64) {
        if (use_lb) {
            // bound a sparse subset of the candidates, fully screen one per pair (upper bound), spread the
            // bounds to the candidates in between, bound those still possible, screen the survivors
            MM_HIP(hipMemsetAsync(dev.n_items, 0, 32, s));
            dev.stats = eng->profile ? eng->dev_stats : nullptr;
            const int nap = (max_na + 31) & ~31, nbp = (max_nt + 31) & ~31, cap = lb_runs_cap;
            if ((prc = eng->profile_begin(s))) return prc;
            if ((e = launch_screen_lb(dev, nap, nbp, s)) != hipSuccess) return hip_error(e, "bound kernel launch");
            if ((prc = eng->profile_end(s, lb_pair_evals, A))) return prc;
            if ((prc = eng->profile_begin(s))) return prc;
            if ((e = launch_lb_pick(dev, 0, s)) != hipSuccess) return hip_error(e, "pick kernel launch");
            if ((e = launch_screen_picks(dev, 0, max_na, max_nbp, s)) != hipSuccess) return hip_error(e, "screen kernel launch (picks)");
            if ((e = launch_lb_spread(dev, s)) != hipSuccess) return hip_error(e, "spread kernel launch");
            if ((e = launch_screen_lb_queued(dev, nap, nbp, cap, s)) != hipSuccess) return hip_error(e, "bound kernel launch (round 2)");
            if ((e = launch_lb_keep(dev, 0, cap, s)) != hipSuccess) return hip_error(e, "keep kernel launch");
            // round 3: the pick's decisive points as queries, then a second pick on the sharpened bounds
            if ((e = launch_lb_topk(dev, std::max(max_na, max_nt), s)) != hipSuccess) return hip_error(e, "top-k kernel launch");
            if ((e = launch_screen_lb_list(dev, nap, nbp, cap, s)) != hipSuccess) return hip_error(e, "bound kernel launch (round 3)");
            if ((e = launch_lb_pick(dev, 1, s)) != hipSuccess) return hip_error(e, "pick kernel launch (2)");
            if ((e = launch_screen_picks(dev, 1, max_na, max_nbp, s)) != hipSuccess) return hip_error(e, "screen kernel launch (picks 2)");
            if ((e = launch_lb_keep(dev, 1, cap, s)) != hipSuccess) return hip_error(e, "keep kernel launch (final)");
            if ((e = launch_screen_kept(dev, max_na, max_nbp, cap, s)) != hipSuccess)
                return hip_error(e, "screen kernel launch (survivors)");
            if ((prc = eng->profile_end(s, 0.0, 0))) return prc;
            if (eng->profile) { eng->bound_offered += A; eng->bound_round1 += lb_sparse_total; }
        } else {
            if ((prc = eng->profile_begin(s))) return prc;
            if (use_mx) {
                e = hipSuccess;
                for (const ScreenGroup& g : groups) {
                    int64_t cand = 0;
                    for (int k = 0; k < g.work_count; ++k) cand += host_work[(size_t)(g.work_begin + k)].cnt;
                    eng->screened[g.kind == 2 ? 2 + g.multi : (g.kind == 3 ? 4 : g.kind)] += cand;
                    if (g.kind == 2) {
                        e = launch_screen_mx(dev, g.work_begin, g.work_count, g.nct, g.multi, g.a_cap, s);
                    } else if (g.kind == 3) {
                        e = launch_screen_none(dev, g.work_begin, g.work_count, s);
                    } else {
                        BatchDev sub = dev;              // the pairs outside the matrix kernel's range: their own work items
                        sub.work = dev.work + g.work_begin; sub.n_work = g.work_count;
                        e = g.kind == 1 ? launch_screen_fast(sub, max_na, max_nbp, s) : launch_screen_f32(sub, max_na, max_nbp, s);
                    }
                    if (e != hipSuccess) break;
                }
            } else {
                eng->screened[use_fast ? 1 : 0] += A;
                e = use_fast ? launch_screen_fast(dev, max_na, max_nbp, s) : launch_screen_f32(dev, max_na, max_nbp, s);
            }
            if (e != hipSuccess) return hip_error(e, "screen kernel launch");
            if ((prc = eng->profile_end(s, pair_evals, A))) return prc;
        }
        if ((prc = mark_search_done())) return prc;
        if (screen_only) return MM_OK;
        MM_HIP(hipMemsetAsync(dev.n_items, 0, 16, s));
        e = launch_shortlist(dev, s);
        if (e != hipSuccess) return hip_error(e, "shortlist kernel launch");
        e = launch_rescore(dev, max_na, max_nbp, (int)std::min<int64_t>(A, INT32_MAX), s);
        if (e != hipSuccess) return hip_error(e, "rescore kernel launch");
        e = launch_finalize(dev, 1, s);
        if (e != hipSuccess) return hip_error(e, "finalize kernel launch");
    } else {
        if ((prc = eng->profile_begin(s))) return prc;
        eng->screened[4] += A;
        e = launch_exact_all(dev, max_na, max_nbp, s);
        if (e != hipSuccess) return hip_error(e, "exact kernel launch");
        if ((prc = eng->profile_end(s, pair_evals, A))) return prc;
        if ((prc = mark_search_done())) return prc;
        if (screen_only) return MM_OK;
        e = launch_finalize(dev, 0, s);
        if (e != hipSuccess) return hip_error(e, "finalize kernel launch");
    }
    return MM_OK;
}

int Plan::mark_search_done()
{
    if (transient) return MM_OK;
    if (!eng->search_done) MM_HIP(hipEventCreateWithFlags(&eng->search_done, hipEventDisableTiming));
    MM_HIP(hipEventRecord(eng->search_done, stream));
    eng->search_done_recorded = true;
    return MM_OK;
}

int Plan::fetch(BatchResult& out, double* all_costs_plan_order)
{
    const size_t cost_bytes = (all_costs_plan_order && dev.all_costs) ? (size_t)A * 8 : 0;
    unsigned char* h = (unsigned char*)eng->host_lvl.p;  // sized in stage_level
    if (P > 0) MM_HIP(launch_copy_small(h, lvl_blob + off_best_cost, res_bytes, stream));   // not hipMemcpyAsync: k_copy_small
    if (cost_bytes)
        MM_HIP(hipMemcpyAsync(all_costs_plan_order, dev.all_costs, cost_bytes, hipMemcpyDeviceToHost, stream));
    MM_HIP(hipStreamSynchronize(stream));
    const double* hc = (const double*)h;
    const int32_t* hi = (const int32_t*)(h + r_best_idx);
    const int32_t* hn = (const int32_t*)(h + r_n_rescored);
    const int32_t* hnc = (const int32_t*)(h + r_near_cnt);
    const int32_t* hni = (const int32_t*)(h + r_near_idx);
    out.best_idx.assign(P, -1); out.n_rescored.assign(P, 0); out.near_cnt.assign(P, 0);
    out.near_idx.assign((size_t)P * kMaxNear, -1); out.best_cost.assign(P, INFINITY);
    for (int p = 0; p < P; ++p) {
        const PairDesc& d = host_pairs[p];
        if (trivial[p]) {
            // every candidate costs 0.0; the ordered first minimum is the first candidate of the slice
            if (d.ang_begin < slice_hi(p)) {
                out.best_idx[p] = d.ang_begin; out.best_cost[p] = 0.0;
                out.near_cnt[p] = 1; out.near_idx[(size_t)p * kMaxNear] = d.ang_begin;
            }
            continue;
        }
        out.best_idx[p] = hi[p]; out.best_cost[p] = hc[p]; out.n_rescored[p] = hn[p]; out.near_cnt[p] = hnc[p];
        for (int k = 0; k < kMaxNear; ++k) out.near_idx[(size_t)p * kMaxNear + k] = hni[(size_t)p * kMaxNear + k];
    }
    return MM_OK;
}

double Plan::angle_of(int p, int32_t idx) const
{
    const PairDesc& d = host_pairs[p];
    if (idx < d.ang_begin) return NAN;
    const size_t k = (size_t)d.tab_off + (size_t)(idx - d.ang_begin);
    return k < host_tables.size() ? host_tables[k] : NAN;
}

Plan::~Plan()
{
    // callers have synchronised the plan's streams (mm_plan_destroy, mm_within_plan_destroy, run_batch's fetch)
    if (pts_blob && own_pts) eng->blob_release(pts_blob, pts_cap);
    if (lvl_blob && own_lvl) eng->blob_release(lvl_blob, lvl_cap);
}

int run_batch(Engine* e, const std::vector<SetRef>& sets, const std::vector<PairSpec>& pairs, int precision,
              BatchResult& out)
{
    Plan plan;
    int rc = plan.stage_sets(e, sets, /*transient=*/true);
    if (rc) return rc;
    rc = plan.stage_level(pairs, precision, 0, INT32_MAX, false);
    if (rc) return rc;
    rc = plan.run(false);
    if (rc) return rc;
    return plan.fetch(out, nullptr);
}

// Translate the public CSR batch description into sets + pairs.
static int make_specs(int n_pairs, const int64_t* ref_off, const double* ref_x, const double* ref_y,
                      const int64_t* tgt_off, const double* tgt_x, const double* tgt_y, const int64_t* ang_off,
                      const double* angles, const double* cx, const double* cy, const int32_t* flags,
                      std::vector<SetRef>& sets, std::vector<PairSpec>& pairs)
{
    if (n_pairs < 0) return set_error(MM_ERR_INVALID, "n_pairs < 0");
    if (n_pairs > 0 && (!ref_off || !tgt_off || !ang_off || !cx || !cy))
        return set_error(MM_ERR_INVALID, "batch offsets / rotation centres == NULL");
    sets.clear(); pairs.clear();
    sets.reserve(2 * (size_t)n_pairs); pairs.reserve((size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        const int64_t nr = ref_off[p + 1] - ref_off[p], nt = tgt_off[p + 1] - tgt_off[p], na = ang_off[p + 1] - ang_off[p];
        if (nr < 0 || nt < 0 || na < 0 || nr > INT32_MAX || nt > INT32_MAX || na > INT32_MAX)
            return set_error(MM_ERR_INVALID, "bad extent in batch offsets");
        if ((nr > 0 && (!ref_x || !ref_y)) || (nt > 0 && (!tgt_x || !tgt_y)) || (na > 0 && !angles))
            return set_error(MM_ERR_INVALID, "point / candidate arrays == NULL for a non-empty pair");
        sets.push_back(SetRef{ref_x + ref_off[p], ref_y + ref_off[p], (int32_t)nr, cx[p], cy[p]});
        sets.push_back(SetRef{tgt_x + tgt_off[p], tgt_y + tgt_off[p], (int32_t)nt, cx[p], cy[p]});
        pairs.push_back(PairSpec{2 * p, 2 * p + 1, cx[p], cy[p], flags ? flags[p] : 0, angles + ang_off[p], (int32_t)na, 0.0, 0.0});
    }
    return MM_OK;
}

// Scatter plan-order costs into the caller's ang_off indexing (slices, empty-set pairs).
static void scatter_costs(const Plan& plan, const double* plan_costs, const int64_t* ang_off, double* all_costs)
{
    for (int p = 0; p < plan.P; ++p) {
        const PairDesc& d = plan.host_pairs[p];
        double* dst = all_costs + ang_off[p] + d.ang_begin;
        if (plan.trivial[p]) {
            const int32_t n = std::max(0, plan.slice_hi(p) - d.ang_begin);
            for (int32_t a = 0; a < n; ++a) dst[a] = 0.0;
        } else if (d.n_ang > 0) {
            std::memcpy(dst, plan_costs + d.out_off, (size_t)d.n_ang * 8);
        }
    }
}

// Pairs whose sets both exceed the search kernel's LDS budget: streaming kernel, sets uploaded
// as given (f64), row blocks of one pair spread over the device.
struct LargePairH { int32_t a_off, na, b_off, nb, col_off, pad; };
struct LargeWorkH { int32_t pair, row0; };

// Points of all referenced sets staged once in dev_pts ([px | py]); descriptors, scratch and results of
// each launch go through host_lvl / dev_lvl, so several launches (bounds, then exact values of a few
// pairs) share one upload.
struct LargeBatch {
    Engine* e = nullptr;
    const std::vector<SetRef>* sets = nullptr;
    std::vector<int64_t> set_at;      // offset of every staged set in the point pool (-1: not staged)
    int64_t npts = 0;
    size_t o_py = 0;

    int stage(Engine* e_, const std::vector<SetRef>& sets_, const std::vector<std::array<int32_t, 2>>& pr,
              const std::vector<int>& idx)
    {
        e = e_; sets = &sets_;
        set_at.assign(sets_.size(), -1);
        std::vector<int32_t> order;
        npts = 0;
        for (int k : idx)
            for (int32_t sidx : {pr[(size_t)k][0], pr[(size_t)k][1]})
                if (set_at[sidx] < 0) { set_at[sidx] = npts; npts += sets_[sidx].n; order.push_back(sidx); }
        if (npts > (int64_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "batch exceeds 2^30 points");
        o_py = align_up((size_t)npts * 8);
        const size_t bytes = align_up(o_py + (size_t)npts * 8);
        int rc = e->ensure(e->host_pts, bytes, true);
        if (rc) return rc;
        if ((rc = e->ensure(e->dev_pts, bytes, false))) return rc;
        unsigned char* h = (unsigned char*)e->host_pts.p;
        double *hx = (double*)h, *hy = (double*)(h + o_py);
        parallel_for((int)order.size(), [&](int k) {   // tens of MB for a refinement grid: spread the copy
            const int32_t sidx = order[(size_t)k];
            std::memcpy(hx + set_at[sidx], sets_[sidx].x, (size_t)sets_[sidx].n * 8);
            std::memcpy(hy + set_at[sidx], sets_[sidx].y, (size_t)sets_[sidx].n * 8);
        });
        MM_HIP(hipMemcpyAsync(e->dev_pts.p, h, bytes, hipMemcpyHostToDevice, e->stream));
        return MM_OK;
    }

    // exact == true: hausdorff_distance of the pairs `which` (entries of pr).  exact == false: a lower
    // bound of each from every stride-th point of either set against all points of the other.
    int run(const std::vector<std::array<int32_t, 2>>& pr, const std::vector<int>& which, bool exact, int stride, double* out)
    {
        const int P = (int)which.size();
        if (P == 0) return MM_OK;
        std::vector<LargePairH> hp;
        std::vector<LargeWorkH> hw;
        int64_t ncol = 0;
        const int rpb = large_rows_per_block();
        for (int k = 0; k < P; ++k) {
            const int32_t ia = pr[(size_t)which[(size_t)k]][0], ib = pr[(size_t)which[(size_t)k]][1];
            const int64_t na = (*sets)[ia].n, nb = (*sets)[ib].n;
            const int32_t oa = (int32_t)set_at[ia], ob = (int32_t)set_at[ib];
            if (exact) {
                if (ncol + nb > (int64_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "batch exceeds 2^30 points");
                hp.push_back(LargePairH{oa, (int32_t)na, ob, (int32_t)nb, (int32_t)ncol, 0});
                ncol += nb;
                for (int64_t r0 = 0; r0 < na; r0 += rpb) hw.push_back(LargeWorkH{k, (int32_t)r0});
            } else {
                for (int dir = 0; dir < 2; ++dir) {
                    const int64_t nq = dir ? nb : na;
                    hp.push_back(dir ? LargePairH{ob, (int32_t)nb, oa, (int32_t)na, k, stride}
                                     : LargePairH{oa, (int32_t)na, ob, (int32_t)nb, k, stride});
                    for (int64_t r0 = 0; r0 < (nq + stride - 1) / stride; r0 += rpb)
                        hw.push_back(LargeWorkH{(int32_t)hp.size() - 1, (int32_t)r0});
                }
            }
        }
        const size_t o_pairs = 0, o_work = align_up(hp.size() * sizeof(LargePairH));
        const size_t in_bytes = align_up(o_work + hw.size() * sizeof(LargeWorkH));
        const size_t o_col = in_bytes, o_row = align_up(o_col + (size_t)ncol * 8), o_out = align_up(o_row + (size_t)P * 8);
        const size_t total = align_up(o_out + (size_t)P * 8);
        int rc = e->ensure(e->host_lvl, std::max(in_bytes, (size_t)P * 8), true);
        if (rc) return rc;
        if ((rc = e->ensure(e->dev_lvl, total, false))) return rc;
        unsigned char* h = (unsigned char*)e->host_lvl.p;
        std::memcpy(h + o_pairs, hp.data(), hp.size() * sizeof(LargePairH));
        std::memcpy(h + o_work, hw.data(), hw.size() * sizeof(LargeWorkH));
        unsigned char* d = (unsigned char*)e->dev_lvl.p;
        const double *dx = (const double*)e->dev_pts.p, *dy = (const double*)((unsigned char*)e->dev_pts.p + o_py);
        MM_HIP(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, e->stream));
        hipError_t he = exact ? launch_hausdorff_large(d + o_pairs, d + o_work, P, (int)hw.size(), dx, dy, d + o_col, ncol,
                                                       d + o_row, (double*)(d + o_out), e->stream)
                              : launch_hausdorff_large_bound(d + o_pairs, d + o_work, P, (int)hw.size(), dx, dy, d + o_row,
                                                             (double*)(d + o_out), e->stream);
        if (he != hipSuccess) return hip_error(he, "large-set hausdorff launch");
        MM_HIP(hipMemcpyAsync(out, d + o_out, (size_t)P * 8, hipMemcpyDeviceToHost, e->stream));
        MM_HIP(hipStreamSynchronize(e->stream));   // also: the pinned descriptor buffer is free again
        return MM_OK;
    }
};

static int hausdorff_large(Engine* e, const std::vector<SetRef>& sets, const std::vector<std::array<int32_t, 2>>& pr,
                           const std::vector<int>& idx, double* out)
{
    LargeBatch lb;
    int rc = lb.stage(e, sets, pr, idx);
    if (rc) return rc;
    std::vector<double> res(idx.size());
    if ((rc = lb.run(pr, idx, true, 1, res.data()))) return rc;
    for (size_t k = 0; k < idx.size(); ++k) out[idx[k]] = res[k];
    return MM_OK;
}

// First index of minimal hausdorff_distance over the pairs (strict '<' scan in pair order, the
// reference's refinement loop, align_algorithms.rs:433-437), without evaluating every pair: a lower
// bound of each (every 16th point of either set against all of the other: the same squared distances
// as the exact kernel, so bound <= exact holds bit for bit), the exact value of the pair with the
// smallest bound as an upper bound, and exact values only for the pairs whose bound does not exceed it.
// A pair that is skipped costs strictly more than the upper bound, so it is neither the minimum nor tied.
// Needs every pair on the streaming kernel (both sets beyond the LDS kernel's budget) and no empty set;
// otherwise, and for short lists, everything is evaluated.
int hausdorff_sets_first_min(Engine* e, const std::vector<SetRef>& sets, const std::vector<std::array<int32_t, 2>>& pr,
                             int32_t* best, double* best_cost, int64_t* n_exact)
{
    const int P = (int)pr.size();
    *best = -1; *best_cost = INFINITY;
    if (n_exact) *n_exact = 0;
    if (P == 0) return MM_OK;
    const int cap = max_target_points_f64();
    bool all_large = P >= 8;
    for (int p = 0; p < P && all_large; ++p) {
        const int32_t ia = pr[(size_t)p][0], ib = pr[(size_t)p][1];
        if (ia < 0 || ib < 0 || (size_t)ia >= sets.size() || (size_t)ib >= sets.size())
            return set_error(MM_ERR_INVALID, "hausdorff_sets_first_min: set index out of range");
        all_large = sets[ia].n > cap && sets[ib].n > cap;
    }
    std::vector<double> cost((size_t)P, INFINITY);
    if (!all_large) {
        int rc = hausdorff_sets(e, sets, pr, cost.data());
        if (rc) return rc;
        if (n_exact) *n_exact = P;
    } else {
        std::vector<int> all((size_t)P);
        for (int p = 0; p < P; ++p) all[(size_t)p] = p;
        LargeBatch lb;
        int rc = lb.stage(e, sets, pr, all);
        if (rc) return rc;
        std::vector<double> bound((size_t)P);
        if ((rc = lb.run(pr, all, false, 16, bound.data()))) return rc;
        int pick = 0;
        for (int p = 1; p < P; ++p) if (bound[(size_t)p] < bound[(size_t)pick]) pick = p;
        std::vector<int> one{pick};
        double ub = 0.0;
        if ((rc = lb.run(pr, one, true, 1, &ub))) return rc;
        cost[(size_t)pick] = ub;
        std::vector<int> rest;
        for (int p = 0; p < P; ++p) if (p != pick && bound[(size_t)p] <= ub) rest.push_back(p);
        std::vector<double> res(rest.size());
        if ((rc = lb.run(pr, rest, true, 1, res.data()))) return rc;
        for (size_t k = 0; k < rest.size(); ++k) cost[(size_t)rest[k]] = res[k];
        if (n_exact) *n_exact = 1 + (int64_t)rest.size();
    }
    for (int p = 0; p < P; ++p)
        if (cost[(size_t)p] < *best_cost) { *best_cost = cost[(size_t)p]; *best = p; }
    return MM_OK;
}

// hausdorff_distance (process_utils.rs:78-121) of set pairs, exact f64 on the device.  Pairs name
// their sets by index, so a set shared by many pairs (the CCTA points of the refine grid) is
// staged once.
int hausdorff_sets(Engine* e, const std::vector<SetRef>& sets, const std::vector<std::array<int32_t, 2>>& pr, double* out)
{
    static const double zero = 0.0;
    const int cap = max_target_points_f64();
    std::vector<SetRef> ssets; std::vector<PairSpec> pairs;
    std::vector<int32_t> remap(sets.size(), -1);
    std::vector<int> small_idx, large_idx;
    auto use = [&](int32_t sidx) {
        if (remap[sidx] < 0) { remap[sidx] = (int32_t)ssets.size(); ssets.push_back(sets[sidx]); }
        return remap[sidx];
    };
    for (size_t p = 0; p < pr.size(); ++p) {
        const int32_t ia = pr[p][0], ib = pr[p][1];
        if (ia < 0 || ib < 0 || (size_t)ia >= sets.size() || (size_t)ib >= sets.size())
            return set_error(MM_ERR_INVALID, "hausdorff_sets: set index out of range");
        const int32_t na = sets[ia].n, nb = sets[ib].n;
        if (na == 0 || nb == 0) { out[p] = 0.0; continue; }               // process_utils.rs:86-88
        if (na > cap && nb > cap) { large_idx.push_back((int)p); continue; }  // neither side fits LDS: streaming kernel
        // hausdorff_distance is symmetric bit for bit (max of the two directed terms over the same
        // squared distances): put the smaller set on the LDS-staged (target) side if the other one
        // would not fit
        const bool swap = nb > cap && na <= cap;
        const int32_t ra = use(swap ? ib : ia), rb = use(swap ? ia : ib);
        // angle 0 with the rotate() shortcut leaves the target untouched
        pairs.push_back(PairSpec{ra, rb, 0.0, 0.0, MM_SEARCH_SKIP_ZERO, &zero, 1, 0.0, 0.0});
        small_idx.push_back((int)p);
    }
    if (!pairs.empty()) {
        BatchResult res;
        int rc = run_batch(e, ssets, pairs, MM_PRECISION_F64, res);
        if (rc) return rc;
        for (size_t k = 0; k < small_idx.size(); ++k) out[small_idx[k]] = res.best_cost[k];
    }
    if (!large_idx.empty()) {
        int rc = hausdorff_large(e, sets, pr, large_idx, out);
        if (rc) return rc;
    }
    return MM_OK;
}

}  // namespace mm

// =====================================================================================
// C ABI
// =====================================================================================
using namespace mm;

extern "C" {

const char* mm_last_error(void) { return g_last_error.c_str(); }
const char* mm_version(void) { return "multimoda-rs_amd 0.1.0 (gfx950)"; }

int mm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return 0; }
    return n;
}

int mm_engine_create(int device, void* stream, mm_engine** out)
{
    if (!out) return set_error(MM_ERR_INVALID, "out == NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(MM_ERR_NO_DEVICE, "no HIP device available: this engine has no CPU fallback");
    if (device < 0) { MM_HIP(hipGetDevice(&device)); }
    if (device >= n) return set_error(MM_ERR_INVALID, "device index out of range");
    MM_HIP(hipSetDevice(device));
    Engine* en = new Engine();
    en->device = device;
    if (stream) { en->stream = (hipStream_t)stream; en->own_stream = false; en->aux = en->stream; }
    else {
        // main stream at the lowest priority, side stream at the highest (numerically: least >= greatest)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        hipError_t e2 = hipStreamCreateWithPriority(&en->stream, hipStreamNonBlocking, least);
        if (e2 != hipSuccess) { delete en; return hip_error(e2, "hipStreamCreateWithPriority"); }
        en->own_stream = true;
        e2 = hipStreamCreateWithPriority(&en->aux, hipStreamNonBlocking, greatest);
        if (e2 != hipSuccess) { (void)hipStreamDestroy(en->stream); delete en; return hip_error(e2, "hipStreamCreateWithPriority"); }
        en->own_aux = true;
    }
    *out = reinterpret_cast<mm_engine*>(en);
    return MM_OK;
}

void mm_engine_destroy(mm_engine* h)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)e->sync_all();
    for (Engine::Buf* b : {&e->host_pts, &e->host_lvl, &e->host_pof}) if (b->p) (void)hipHostFree(b->p);
    for (Engine::Buf* b : {&e->dev_pts, &e->dev_lvl, &e->dev_raw}) if (b->p) (void)hipFree(b->p);
    for (Engine::Buf& b : e->blob_cache) (void)hipFree(b.p);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
    if (e->search_done) (void)hipEventDestroy(e->search_done);
    if (e->pof_done) (void)hipEventDestroy(e->pof_done);
    if (e->tail_done) (void)hipEventDestroy(e->tail_done);
    if (e->dev_stats) (void)hipFree(e->dev_stats);
    if (e->own_aux) (void)hipStreamDestroy(e->aux);
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int mm_engine_synchronize(mm_engine* h)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    MM_HIP(hipSetDevice(e->device));
    if (int src = e->sync_all()) return src;
    return MM_OK;
}

int mm_engine_wait_search(mm_engine* waiter, mm_engine* other)
{
    Engine *w = reinterpret_cast<Engine*>(waiter), *o = reinterpret_cast<Engine*>(other);
    if (!w || !o) return set_error(MM_ERR_INVALID, "engine == NULL");
    if (w->device != o->device) return set_error(MM_ERR_INVALID, "mm_engine_wait_search: engines on different devices");
    MM_HIP(hipSetDevice(w->device));
    if (o->search_done_recorded) MM_HIP(hipStreamWaitEvent(w->stream, o->search_done, 0));
    return MM_OK;
}

int mm_engine_wait_exchange(mm_engine* waiter, mm_engine* other)
{
    Engine *w = reinterpret_cast<Engine*>(waiter), *o = reinterpret_cast<Engine*>(other);
    if (!w || !o) return set_error(MM_ERR_INVALID, "engine == NULL");
    if (w->device != o->device) return set_error(MM_ERR_INVALID, "mm_engine_wait_exchange: engines on different devices");
    MM_HIP(hipSetDevice(w->device));
    if (o->tail_done_recorded) MM_HIP(hipStreamWaitEvent(w->stream, o->tail_done, 0));
    return MM_OK;
}

int mm_engine_profile(mm_engine* h, int enable)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    MM_HIP(hipSetDevice(e->device));
    if (int src = e->sync_all()) return src;
    e->profile = enable != 0;
    e->launches = 0; e->prof_pair_evals = 0.0; e->prof_candidates = 0; e->launch_pair_evals.clear();
    e->bound_offered = 0; e->bound_round1 = 0;
    if (e->profile) {
        if (!e->dev_stats) MM_HIP(hipMalloc((void**)&e->dev_stats, 64));
        MM_HIP(hipMemsetAsync(e->dev_stats, 0, 64, e->stream));
    }
    // hipEventCreate is slow (~0.5 ms): build the pool now, outside any timed region
    while (e->profile && e->events.size() < 2 * 2048) {
        hipEvent_t ev;
        MM_HIP(hipEventCreate(&ev));
        e->events.push_back(ev);
    }
    return MM_OK;
}

int mm_engine_profile_read(mm_engine* h, int64_t* n_launches, double* ms_total, double* pair_evals,
                           int64_t* candidates)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    MM_HIP(hipSetDevice(e->device));
    if (int src = e->sync_all()) return src;
    double ms = 0.0;
    for (size_t k = 0; k < e->launches; ++k) {
        float t = 0.f;
        MM_HIP(hipEventElapsedTime(&t, e->events[2 * k], e->events[2 * k + 1]));
        ms += (double)t;
    }
    if (n_launches) *n_launches = (int64_t)e->launches;
    if (ms_total) *ms_total = ms;
    if (pair_evals) *pair_evals = e->prof_pair_evals;
    if (candidates) *candidates = e->prof_candidates;
    e->launches = 0; e->prof_pair_evals = 0.0; e->prof_candidates = 0; e->launch_pair_evals.clear();
    return MM_OK;
}

int mm_engine_profile_launches(mm_engine* h, int64_t cap, float* ms, double* pair_evals, int64_t* n_launches)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    MM_HIP(hipSetDevice(e->device));
    if (int src = e->sync_all()) return src;
    for (size_t k = 0; k < e->launches && (int64_t)k < cap; ++k) {
        float t = 0.f;
        MM_HIP(hipEventElapsedTime(&t, e->events[2 * k], e->events[2 * k + 1]));
        if (ms) ms[k] = t;
        if (pair_evals) pair_evals[k] = e->launch_pair_evals[k];
    }
    if (n_launches) *n_launches = (int64_t)e->launches;
    return MM_OK;
}

int mm_engine_set_bound_min_candidates(mm_engine* h, int64_t n)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e || n < 0) return set_error(MM_ERR_INVALID, "engine == NULL or n < 0");
    e->bound_min_candidates = n;
    return MM_OK;
}

int mm_engine_set_bound_matrix(mm_engine* h, int on)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    e->bound_matrix = on != 0;
    if (on > 1) {      // experiments: 10 * (query tiles per side) + (candidates per wave), e.g. 21
        const int qt = on / 10, nc = on % 10;
        if ((qt != 1 && qt != 2) || (nc != 1 && nc != 2)) return set_error(MM_ERR_INVALID, "mm_engine_set_bound_matrix: variant must be 11, 12, 21 or 22");
        e->bound_matrix_qt = qt; e->bound_matrix_nc = nc;
    }
    return MM_OK;
}

int mm_engine_bound_stats(mm_engine* h, int64_t out[5])
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e || !out) return set_error(MM_ERR_INVALID, "engine or out == NULL");
    MM_HIP(hipSetDevice(e->device));
    unsigned long long d[8] = {0};
    if (int src = e->sync_all()) return src;
    if (e->dev_stats) MM_HIP(hipMemcpy(d, e->dev_stats, 64, hipMemcpyDeviceToHost));
    out[0] = e->bound_offered; out[1] = e->bound_round1; out[2] = (int64_t)d[1]; out[3] = (int64_t)d[3];
    out[4] = (int64_t)d[2];
    return MM_OK;
}

int mm_engine_screen_stats(mm_engine* h, int64_t out[5])
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e || !out) return set_error(MM_ERR_INVALID, "engine or out == NULL");
    for (int k = 0; k < 5; ++k) out[k] = e->screened[k].load();
    return MM_OK;
}

void* mm_engine_stream(mm_engine* h)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    return e ? (void*)e->stream : nullptr;
}

int mm_best_rotation_batch(mm_engine* h, int n_pairs,
                           const int64_t* ref_off, const double* ref_x, const double* ref_y,
                           const int64_t* tgt_off, const double* tgt_x, const double* tgt_y,
                           const int64_t* ang_off, const double* angles,
                           const double* cx, const double* cy, const int32_t* flags, int precision,
                           int32_t* best_idx, double* best_angle, double* best_cost,
                           int32_t* n_rescored, double* all_costs)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    if (n_pairs == 0) return MM_OK;
    MM_HIP(hipSetDevice(e->device));
    std::vector<SetRef> sets; std::vector<PairSpec> pairs;
    int rc = make_specs(n_pairs, ref_off, ref_x, ref_y, tgt_off, tgt_x, tgt_y, ang_off, angles, cx, cy, flags, sets, pairs);
    if (rc) return rc;
    Plan plan;
    if ((rc = plan.stage_sets(e, sets, true))) return rc;
    if ((rc = plan.stage_level(pairs, precision, 0, INT32_MAX, all_costs != nullptr))) return rc;
    if ((rc = plan.run(false))) return rc;
    BatchResult res;
    std::vector<double> costs(all_costs ? (size_t)plan.A : 0);
    if ((rc = plan.fetch(res, all_costs ? costs.data() : nullptr))) return rc;
    for (int p = 0; p < n_pairs; ++p) {
        if (best_idx) best_idx[p] = res.best_idx[p];
        if (best_cost) best_cost[p] = res.best_cost[p];
        if (n_rescored) n_rescored[p] = res.n_rescored[p];
        if (best_angle) best_angle[p] = res.best_idx[p] >= 0 ? angles[ang_off[p] + res.best_idx[p]] : NAN;
    }
    if (all_costs) scatter_costs(plan, costs.data(), ang_off, all_costs);
    return MM_OK;
}

// The lower bound MM_PRECISION_F32_BOUNDED's first round would give EVERY candidate of one search (it scores every 8th):
// out_lb2[i] <= (exact cost of candidate i)^2 up to the error bound *e2 of the kernel's squared values and *delta of the
// distance.  A test hook for the bound kernels themselves (packed-FMA: matrix == 0, matrix pipe: matrix != 0); nothing in
// the product calls it.
int mm_lower_bounds(mm_engine* h, const double* rx, const double* ry, int nr, const double* tx, const double* ty, int nt,
                    double cx, double cy, const double* angles, int n_angles, int flags, int matrix, float* out_lb2,
                    double* e2, double* delta, int* stride)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e || !out_lb2 || nr <= 0 || nt <= 0 || n_angles <= 0 || !angles) return set_error(MM_ERR_INVALID, "mm_lower_bounds: bad arguments");
    MM_HIP(hipSetDevice(e->device));
    std::vector<SetRef> sets{SetRef{rx, ry, nr, cx, cy}, SetRef{tx, ty, nt, cx, cy}};
    std::vector<PairSpec> pairs{PairSpec{0, 1, cx, cy, flags, angles, n_angles, 0.0, 0.0}};
    const int64_t keep_min = e->bound_min_candidates;
    const bool keep_mx = e->bound_matrix;
    e->bound_min_candidates = 0; e->bound_matrix = matrix != 0;
    Plan plan;
    int rc = plan.stage_sets(e, sets, true);
    if (!rc) rc = plan.stage_level(pairs, MM_PRECISION_F32_BOUNDED, 0, INT32_MAX, false);
    e->bound_min_candidates = keep_min; e->bound_matrix = keep_mx;
    if (rc) return rc;
    if (!plan.use_lb || (matrix != 0) != (plan.lb_mx_tiles > 0)) return set_error(MM_ERR_INVALID, "mm_lower_bounds: the bound kernel asked for does not take these sets");
    BatchDev sub = plan.dev;
    sub.work_lb = plan.dev.work; sub.n_work_lb = plan.W;      // every candidate, step 1 (WorkItem::pad == 0)
    const int nap = (plan.max_na + 31) & ~31, nbp = (plan.max_nt + 31) & ~31;
    hipError_t he = launch_screen_lb(sub, nap, nbp, plan.stream);
    if (he != hipSuccess) return hip_error(he, "bound kernel launch");
    MM_HIP(hipMemcpyAsync(out_lb2, plan.dev.lb32, (size_t)n_angles * 4, hipMemcpyDeviceToHost, plan.stream));
    MM_HIP(hipStreamSynchronize(plan.stream));
    if (e2) *e2 = plan.host_pairs[0].e2;
    if (delta) *delta = plan.host_pairs[0].delta;
    if (stride) *stride = plan.lb_stride;
    return MM_OK;
}

int mm_pick_minima(mm_engine* h, const double* rx, const double* ry, int nr, const double* tx, const double* ty, int nt,
                   double cx, double cy, double angle, int flags, float* row_min2, float* col_min2, float* value2, double* e2)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e || !row_min2 || !col_min2 || nr <= 0 || nt <= 0) return set_error(MM_ERR_INVALID, "mm_pick_minima: bad arguments");
    MM_HIP(hipSetDevice(e->device));
    std::vector<SetRef> sets{SetRef{rx, ry, nr, cx, cy}, SetRef{tx, ty, nt, cx, cy}};
    std::vector<PairSpec> pairs{PairSpec{0, 1, cx, cy, flags, &angle, 1, 0.0, 0.0}};
    const int64_t keep_min = e->bound_min_candidates;
    const bool keep_mx = e->bound_matrix;
    e->bound_min_candidates = 0; e->bound_matrix = true;
    Plan plan;
    int rc = plan.stage_sets(e, sets, true);
    if (!rc) rc = plan.stage_level(pairs, MM_PRECISION_F32_BOUNDED, 0, INT32_MAX, false);
    e->bound_min_candidates = keep_min; e->bound_matrix = keep_mx;
    if (rc) return rc;
    if (!plan.use_lb || plan.lb_mx_tiles <= 0) return set_error(MM_ERR_INVALID, "mm_pick_minima: the matrix-pipe bounded search does not take these sets");
    const WorkItem item{0, 0, 1, 0};
    const int32_t counters[8] = {0, 1, 0, 0, 0, 0, 0, 0};
    MM_HIP(hipMemcpyAsync(plan.dev.items_pick, &item, sizeof item, hipMemcpyHostToDevice, plan.stream));
    MM_HIP(hipMemcpyAsync(plan.dev.n_items, counters, sizeof counters, hipMemcpyHostToDevice, plan.stream));
    MM_HIP(hipStreamSynchronize(plan.stream));                     // (the sources are on this stack frame)
    hipError_t he = launch_screen_picks(plan.dev, 0, plan.max_na, plan.max_nbp, plan.stream);
    if (he != hipSuccess) return hip_error(he, "pick kernel launch");
    MM_HIP(hipMemcpyAsync(row_min2, plan.dev.emit, (size_t)nr * 4, hipMemcpyDeviceToHost, plan.stream));
    MM_HIP(hipMemcpyAsync(col_min2, plan.dev.emit + plan.dev.emit_rows, (size_t)nt * 4, hipMemcpyDeviceToHost, plan.stream));
    if (value2) MM_HIP(hipMemcpyAsync(value2, plan.dev.sq32, 4, hipMemcpyDeviceToHost, plan.stream));
    MM_HIP(hipStreamSynchronize(plan.stream));
    if (e2) *e2 = plan.host_pairs[0].e2;
    return MM_OK;
}

int mm_best_rotation(mm_engine* h, const double* rx, const double* ry, int nr,
                     const double* tx, const double* ty, int nt, double cx, double cy,
                     const double* angles, int n_angles, int flags, int precision,
                     double* best_angle, double* best_cost, int* best_idx, double* all_costs)
{
    if (nr < 0 || nt < 0 || n_angles <= 0) return set_error(MM_ERR_INVALID, "mm_best_rotation: bad sizes");
    const int64_t ro[2] = {0, nr}, to[2] = {0, nt}, ao[2] = {0, n_angles};
    int32_t bi = -1; double ba = NAN, bc = NAN;
    int rc = mm_best_rotation_batch(h, 1, ro, rx, ry, to, tx, ty, ao, angles, &cx, &cy, &flags, precision,
                                    &bi, &ba, &bc, nullptr, all_costs);
    if (rc) return rc;
    if (best_idx) *best_idx = bi;
    if (best_angle) *best_angle = ba;
    if (best_cost) *best_cost = bc;
    return MM_OK;
}

int mm_hausdorff_batch(mm_engine* h, int n_pairs, const int64_t* a_off, const double* ax, const double* ay,
                       const int64_t* b_off, const double* bx, const double* by, double* out, int32_t* first_min);

int mm_hausdorff_2d(mm_engine* h, const double* ax, const double* ay, int na,
                    const double* bx, const double* by, int nb, double* out)
{
    if (!out) return set_error(MM_ERR_INVALID, "out == NULL");
    if (!h) return set_error(MM_ERR_INVALID, "engine == NULL");
    if (na < 0 || nb < 0) return set_error(MM_ERR_INVALID, "negative set size");
    if (na == 0 || nb == 0) { *out = 0.0; return MM_OK; }  // process_utils.rs:86-88
    const int64_t ao[2] = {0, na}, bo[2] = {0, nb};
    return mm_hausdorff_batch(h, 1, ao, ax, ay, bo, bx, by, out, nullptr);
}

int mm_hausdorff_batch(mm_engine* h, int n_pairs, const int64_t* a_off, const double* ax, const double* ay,
                       const int64_t* b_off, const double* bx, const double* by, double* out, int32_t* first_min)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    if (n_pairs < 0 || (n_pairs > 0 && !out)) return set_error(MM_ERR_INVALID, "mm_hausdorff_batch: bad arguments");
    if (first_min) *first_min = -1;
    if (n_pairs == 0) return MM_OK;
    MM_HIP(hipSetDevice(e->device));
    std::vector<SetRef> sets;
    std::vector<std::array<int32_t, 2>> pr;
    sets.reserve(2 * (size_t)n_pairs); pr.reserve((size_t)n_pairs);
    if (!a_off || !b_off) return set_error(MM_ERR_INVALID, "mm_hausdorff_batch: offsets == NULL");
    for (int p = 0; p < n_pairs; ++p) {
        const int64_t na = a_off[p + 1] - a_off[p], nb = b_off[p + 1] - b_off[p];
        if (na < 0 || nb < 0 || na > INT32_MAX || nb > INT32_MAX) return set_error(MM_ERR_INVALID, "bad set extent");
        if ((na > 0 && (!ax || !ay)) || (nb > 0 && (!bx || !by)))
            return set_error(MM_ERR_INVALID, "mm_hausdorff_batch: point arrays == NULL for a non-empty set");
        sets.push_back(SetRef{ax + a_off[p], ay + a_off[p], (int32_t)na, 0.0, 0.0});
        sets.push_back(SetRef{bx + b_off[p], by + b_off[p], (int32_t)nb, 0.0, 0.0});
        pr.push_back({2 * p, 2 * p + 1});
    }
    int rc = hausdorff_sets(e, sets, pr, out);
    if (rc) return rc;
    int32_t best = -1;
    double best_cost = INFINITY;   // f64::MAX in the reference; costs are finite
    for (int p = 0; p < n_pairs; ++p)
        if (out[p] < best_cost) { best_cost = out[p]; best = p; }
    if (first_min) *first_min = best;
    return MM_OK;
}

// ---- persistent plans ------------------------------------------------------------------
struct PlanHandle {
    Plan plan;
    std::vector<int64_t> ang_off;  // caller's candidate offsets
};

int mm_plan_create(mm_engine* h, int n_pairs,
                   const int64_t* ref_off, const double* ref_x, const double* ref_y,
                   const int64_t* tgt_off, const double* tgt_x, const double* tgt_y,
                   const int64_t* ang_off, const double* angles,
                   const double* cx, const double* cy, const int32_t* flags,
                   int precision, int32_t angle_begin, int32_t angle_end, mm_plan** out)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e || !out) return set_error(MM_ERR_INVALID, "engine/out == NULL");
    *out = nullptr;
    MM_HIP(hipSetDevice(e->device));
    std::vector<SetRef> sets; std::vector<PairSpec> pairs;
    int rc = make_specs(n_pairs, ref_off, ref_x, ref_y, tgt_off, tgt_x, tgt_y, ang_off, angles, cx, cy, flags, sets, pairs);
    if (rc) return rc;
    PlanHandle* ph = new PlanHandle();
    ph->ang_off.assign(ang_off, ang_off + n_pairs + 1);
    if ((rc = ph->plan.stage_sets(e, sets, false)) || (rc = ph->plan.stage_level(pairs, precision, angle_begin, angle_end, true))) {
        delete ph;
        return rc;
    }
    *out = reinterpret_cast<mm_plan*>(ph);
    return MM_OK;
}

int mm_plan_create_indexed(mm_engine* h, int n_sets, const int64_t* set_off, const double* x, const double* y,
                           const double* set_cx, const double* set_cy, int n_pairs, const int32_t* ref_set,
                           const int32_t* tgt_set, const int64_t* ang_off, const double* angles, int shared_angles,
                           const double* cx, const double* cy, const int32_t* flags, int precision, int want_costs,
                           mm_plan** out)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e || !out) return set_error(MM_ERR_INVALID, "engine/out == NULL");
    *out = nullptr;
    if (n_sets < 0 || n_pairs < 0) return set_error(MM_ERR_INVALID, "negative counts");
    if ((n_sets > 0 && (!set_off || !set_cx || !set_cy)) || (n_pairs > 0 && (!ref_set || !tgt_set || !ang_off || !cx || !cy)))
        return set_error(MM_ERR_INVALID, "mm_plan_create_indexed: offsets / centres / pair lists == NULL");
    if (n_pairs > 0 && !angles && (shared_angles ? ang_off[1] > ang_off[0] : ang_off[n_pairs] > ang_off[0]))
        return set_error(MM_ERR_INVALID, "mm_plan_create_indexed: candidate list == NULL");
    MM_HIP(hipSetDevice(e->device));
    std::vector<SetRef> sets((size_t)n_sets);
    for (int s = 0; s < n_sets; ++s) {
        const int64_t n = set_off[s + 1] - set_off[s];
        if (n < 0 || n > INT32_MAX) return set_error(MM_ERR_INVALID, "bad set extent");
        if (n > 0 && (!x || !y)) return set_error(MM_ERR_INVALID, "mm_plan_create_indexed: point arrays == NULL");
        sets[s] = SetRef{x + set_off[s], y + set_off[s], (int32_t)n, set_cx[s], set_cy[s]};
    }
    std::vector<PairSpec> pairs((size_t)n_pairs);
    PlanHandle* ph = new PlanHandle();
    ph->ang_off.resize((size_t)n_pairs + 1);
    for (int p = 0; p < n_pairs; ++p) {
        const int64_t a0 = shared_angles ? ang_off[0] : ang_off[p], a1 = shared_angles ? ang_off[1] : ang_off[p + 1];
        if (a1 < a0 || a1 - a0 > INT32_MAX) { delete ph; return set_error(MM_ERR_INVALID, "bad candidate extent"); }
        if (ref_set[p] < 0 || ref_set[p] >= n_sets || tgt_set[p] < 0 || tgt_set[p] >= n_sets) {
            delete ph; return set_error(MM_ERR_INVALID, "pair references a set that does not exist");
        }
        pairs[p] = PairSpec{ref_set[p], tgt_set[p], cx[p], cy[p], flags ? flags[p] : 0, angles + a0, (int32_t)(a1 - a0), 0.0, 0.0};
        ph->ang_off[p] = shared_angles ? (int64_t)p * (a1 - a0) : a0;   // all_costs layout: pair-major
    }
    if (n_pairs > 0) {
        const int64_t a0 = shared_angles ? ang_off[0] : ang_off[n_pairs - 1], a1 = shared_angles ? ang_off[1] : ang_off[n_pairs];
        ph->ang_off[n_pairs] = shared_angles ? (int64_t)n_pairs * (a1 - a0) : a1;
    }
    int rc;
    if ((rc = ph->plan.stage_sets(e, sets, false)) || (rc = ph->plan.stage_level(pairs, precision, 0, INT32_MAX, want_costs != 0))) {
        delete ph;
        return rc;
    }
    *out = reinterpret_cast<mm_plan*>(ph);
    return MM_OK;
}

void mm_plan_destroy(mm_plan* h)
{
    PlanHandle* p = reinterpret_cast<PlanHandle*>(h);
    if (!p) return;
    (void)hipSetDevice(p->plan.eng->device);
    (void)p->plan.eng->sync_all();
    delete p;
}

int mm_plan_run(mm_plan* h)
{
    PlanHandle* p = reinterpret_cast<PlanHandle*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    MM_HIP(hipSetDevice(p->plan.eng->device));
    return p->plan.run(false);
}

int mm_plan_run_screen_only(mm_plan* h)
{
    PlanHandle* p = reinterpret_cast<PlanHandle*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    MM_HIP(hipSetDevice(p->plan.eng->device));
    return p->plan.run(true);
}

int mm_plan_fetch(mm_plan* h, int32_t* best_idx, double* best_angle, double* best_cost,
                  int32_t* n_rescored, double* all_costs)
{
    PlanHandle* p = reinterpret_cast<PlanHandle*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    Plan& plan = p->plan;
    MM_HIP(hipSetDevice(plan.eng->device));
    BatchResult res;
    std::vector<double> costs(all_costs ? (size_t)plan.A : 0);
    int rc = plan.fetch(res, all_costs ? costs.data() : nullptr);
    if (rc) return rc;
    for (int q = 0; q < plan.P; ++q) {
        if (best_idx) best_idx[q] = res.best_idx[q];
        if (best_cost) best_cost[q] = res.best_cost[q];
        if (n_rescored) n_rescored[q] = res.n_rescored[q];
        if (best_angle) best_angle[q] = res.best_idx[q] >= 0 ? plan.angle_of(q, res.best_idx[q]) : NAN;
    }
    if (all_costs) scatter_costs(plan, costs.data(), p->ang_off.data(), all_costs);
    return MM_OK;
}

int mm_plan_result_dev(mm_plan* h, void** best_cost_dev, void** best_idx_dev)
{
    PlanHandle* p = reinterpret_cast<PlanHandle*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    if (best_cost_dev) *best_cost_dev = p->plan.dev.best_cost;
    if (best_idx_dev) *best_idx_dev = p->plan.dev.best_idx;
    return MM_OK;
}

int mm_plan_time(mm_plan* h, int iters, int screen_only, float* ms_avg)
{
    PlanHandle* p = reinterpret_cast<PlanHandle*>(h);
    if (!p || !ms_avg || iters <= 0) return set_error(MM_ERR_INVALID, "mm_plan_time: bad arguments");
    Plan& plan = p->plan;
    MM_HIP(hipSetDevice(plan.eng->device));
    hipEvent_t t0, t1;
    MM_HIP(hipEventCreate(&t0));
    MM_HIP(hipEventCreate(&t1));
    int rc = plan.run(screen_only != 0);  // warm-up
    if (rc) return rc;
    MM_HIP(hipEventRecord(t0, plan.stream));
    for (int i = 0; i < iters; ++i)
        if ((rc = plan.run(screen_only != 0))) return rc;
    MM_HIP(hipEventRecord(t1, plan.stream));
    MM_HIP(hipEventSynchronize(t1));
    float ms = 0.f;
    MM_HIP(hipEventElapsedTime(&ms, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    *ms_avg = ms / (float)iters;
    return MM_OK;
}

int mm_plan_stats(mm_plan* h, int64_t* n_candidates, double* pair_evals, int64_t* hbm_bytes)
{
    PlanHandle* p = reinterpret_cast<PlanHandle*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    if (n_candidates) *n_candidates = p->plan.A;
    if (pair_evals) *pair_evals = p->plan.pair_evals;
    if (hbm_bytes) *hbm_bytes = (int64_t)p->plan.hbm_bytes();
    return MM_OK;
}

}  // extern "C"
