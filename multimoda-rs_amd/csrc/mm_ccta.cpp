// mm_ccta.cpp -- CCTA diameter search (include/mm_ccta.h): 41 radial scalings x symmetric RMS
// nearest-neighbour distance in 3-D.  Every nearest-neighbour minimum is computed on the device in
// exact f64 (mm_nn_kernels.hip, one batch per search); morphing, sums and selection are host f64
// in the reference's operation order (-ffp-contract=off).  Reference:
// src/ccta/adjust_mesh/scale_coronary.rs (lines cited per function).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/mm_ccta.h"
#include "mm_engine.h"
#include "mm_pool.h"
#include "mm_trace.h"

namespace mm {
namespace {

struct NnPairH { int32_t q_off, nq, p_off, np, out_off, qperm_off; };
struct NnWorkH { int32_t pair, q0, c0, n_chunks; double lb2; };
// AoS triples.  A derived set (unit != nullptr) is xyz moved by adj along per-point unit vectors where has[i]
// (centerline_based_diameter_morphing, scale_coronary.rs:236-239): it is never materialised on the host --
// the device computes it from the base points and the unit vectors staged once for all scalings.
struct Set3 {
    const double* xyz; int64_t n;
    const double* unit = nullptr; const uint8_t* has = nullptr; double adj = 0.0;
    double at(int64_t i, int a) const {
        return (unit && has[i]) ? xyz[3 * i + a] + unit[3 * i + a] * adj : xyz[3 * i + a];   // :236 p + unit * x
    }
};
struct NnMorphH { int32_t dst_off, n, aux_off, pad; double adj; };   // device: pool[dst_off + j] = aux point j moved by adj

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

#define MM_TRY_HIP(call)                                          \
    do {                                                          \
        const hipError_t e__ = (call);                            \
        if (e__ != hipSuccess) return hip_error(e__, #call);      \
    } while (0)

// Slab order of a point set: points sorted by their coordinate along the longest axis of the set's bounding
// box (quantised to 20 bits, stable LSD radix sort), so that groups of consecutive points are slabs.
// The sets here are vessel surfaces -- thin shells around a centerline -- and their nearest neighbours are
// radial, i.e. in the same or the next slab; boxes of compact 3-D patches (k-d leaves, Morton runs) overlap
// their neighbours and the opposite wall and prune far less (measured on the bench case: pass B 6.1 ms
// with Morton runs, 3.5 ms with k-d leaves, 2.5 ms with slabs).  Any permutation gives the same minima.
void slab_order(const Set3& st, std::vector<int32_t>& perm)
{
    const int64_t n = st.n;
    perm.resize((size_t)n);
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { const double v = st.at(i, a); lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
    int ax = 0;
    for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
    const double sc = hi[ax] > lo[ax] ? 1048575.0 / (hi[ax] - lo[ax]) : 0.0;
    std::vector<uint32_t> key((size_t)n), key2((size_t)n);
    std::vector<int32_t> idx2((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const double t = (st.at(i, ax) - lo[ax]) * sc;
        key[(size_t)i] = t > 0.0 ? (t < 1048575.0 ? (uint32_t)t : 1048575u) : 0u;   // NaN-safe clamp
        perm[(size_t)i] = (int32_t)i;
    }
    for (int pass = 0; pass < 2; ++pass) {
        const int sh = 10 * pass;
        uint32_t cnt[1025] = {0};
        for (int64_t i = 0; i < n; ++i) ++cnt[((key[(size_t)i] >> sh) & 1023u) + 1];
        for (int b = 0; b < 1024; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) {
            const uint32_t d = cnt[(key[(size_t)i] >> sh) & 1023u]++;
            key2[d] = key[(size_t)i]; idx2[d] = perm[(size_t)i];
        }
        key.swap(key2); perm.swap(idx2);
    }
}

// Per-query minima of every pair (sets[q] against sets[p]); one upload, two launches, one download.
// view[k] = {pointer, count}: pair k's minima, in the query set's ORIGINAL order, inside the engine's pinned
// staging buffer (valid until the next call on this engine); pointer == nullptr means "all +inf" (an empty
// point set) or no queries.
// order_like (nullable, one entry per set): the set whose spatial order this one shares -- a morphed copy of
// a set moves every point by at most a few mm, so the 41 scalings of a search reuse one sort.  Large sets
// are staged in slab order and every (query block, chunk) item carries the squared distance between the
// two bounding boxes; the kernel skips items that cannot lower any of their queries' minima.  Minima are
// exact and order-independent: pruned or not, sorted or not, the results are the same bits.
struct MinView { const double* p; int64_t n; };

// sums (nullable): if given, only the per-pair sums of the minima (sequential, index order) come back
// -- sums[k] for pair k, NaN where the pair has no minima -- and view[k].p stays null.
int nn_batch_view(Engine* e, const std::vector<Set3>& sets, const std::vector<std::array<int32_t, 2>>& pr,
                  std::vector<MinView>& view, const std::vector<int32_t>* order_like = nullptr,
                  std::vector<double>* sums = nullptr)
{
    if (sums) sums->assign(pr.size(), NAN);
    view.assign(pr.size(), MinView{nullptr, 0});
    const size_t S = sets.size();
    std::vector<int64_t> soff(S + 1, 0);
    for (size_t s = 0; s < S; ++s) soff[s + 1] = soff[s] + sets[s].n;
    const int64_t npts = soff.back();
    const int qpb = nn_queries_per_block(), ch = nn_chunk_points(), span = nn_span_chunks();
    constexpr int64_t kSortMin = 4096;   // smaller sets: original order, every chunk scanned

    // ---- spatial orders: one sort per distinct base set ---------------------------------------
    std::vector<int32_t> base(S, -1), perm_of(S, -1);
    std::vector<int32_t> bases;   // distinct sets that get sorted
    for (size_t s = 0; s < S; ++s) {
        if (sets[s].n < kSortMin) continue;
        int32_t b = order_like ? (*order_like)[s] : (int32_t)s;
        if (b < 0 || (size_t)b >= S || sets[(size_t)b].n != sets[s].n) b = (int32_t)s;
        base[s] = b;
        if (std::find(bases.begin(), bases.end(), b) == bases.end()) bases.push_back(b);
    }
    std::vector<std::vector<int32_t>> perms(bases.size());
    TraceTimer tt_all("nn: batch total");
    { TraceTimer tt("nn: slab order");
    parallel_for((int)bases.size(), [&](int k) { slab_order(sets[(size_t)bases[(size_t)k]], perms[(size_t)k]); });
    }
    std::vector<int64_t> perm_off(bases.size() + 1, 0);
    for (size_t k = 0; k < bases.size(); ++k) perm_off[k + 1] = perm_off[k] + (int64_t)perms[k].size();
    for (size_t s = 0; s < S; ++s)
        if (base[s] >= 0) perm_of[s] = (int32_t)(std::find(bases.begin(), bases.end(), base[s]) - bases.begin());

    // ---- pairs, and the groups (half chunks = one query block) whose boxes are needed -------------
    std::vector<NnPairH> hp;
    std::vector<int> owner;   // device pair -> caller's pair
    int64_t nout = 0;
    for (size_t k = 0; k < pr.size(); ++k) {
        const int32_t q = pr[k][0], p = pr[k][1];
        if (q < 0 || p < 0 || (size_t)q >= S || (size_t)p >= S)
            return set_error(MM_ERR_INVALID, "nn batch: set index out of range");
        const int64_t nq = sets[(size_t)q].n, np = sets[(size_t)p].n;
        view[k].n = nq;                       // fold(INFINITY, min) over an empty set: all +inf
        if (nq == 0 || np == 0) continue;
        hp.push_back(NnPairH{(int32_t)soff[(size_t)q], (int32_t)nq, (int32_t)soff[(size_t)p], (int32_t)np, (int32_t)nout,
                             perm_of[(size_t)q] >= 0 ? (int32_t)perm_off[(size_t)perm_of[(size_t)q]] : -1});
        owner.push_back((int)k);
        nout += nq;
    }
    if (hp.empty()) return MM_OK;
    if (npts > (int64_t)1 << 30 || nout > (int64_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "nn batch exceeds 2^30 points");

    // ---- stage the points (permuted where sorted) and the bounding box of every group of qpb points ---
    // derived sets: only their boxes are computed here; their coordinates are produced on the device from
    // an auxiliary pool (base point, unit vector, flag per point; one entry per distinct base + order)
    struct Aux { const double* xyz; const double* unit; const uint8_t* has; int32_t perm; int64_t n, off; };
    std::vector<Aux> aux;
    std::vector<int32_t> aux_of(S, -1);
    int64_t naux = 0;
    for (size_t s = 0; s < S; ++s) {
        const Set3& st = sets[s];
        if (!st.unit || st.n == 0) continue;
        size_t k = 0;
        for (; k < aux.size(); ++k)
            if (aux[k].xyz == st.xyz && aux[k].unit == st.unit && aux[k].has == st.has && aux[k].perm == perm_of[s] && aux[k].n == st.n) break;
        if (k == aux.size()) { aux.push_back(Aux{st.xyz, st.unit, st.has, perm_of[s], st.n, naux}); naux += st.n; }
        aux_of[s] = (int32_t)k;
    }
    std::vector<NnMorphH> morphs;
    for (size_t s = 0; s < S; ++s)
        if (aux_of[s] >= 0) morphs.push_back(NnMorphH{(int32_t)soff[s], (int32_t)sets[s].n, (int32_t)aux[(size_t)aux_of[s]].off, 0, sets[s].adj});
    const size_t o_x = 0, o_y = up256((size_t)npts * 8), o_z = up256(o_y + (size_t)npts * 8);
    const size_t o_perm = up256(o_z + (size_t)npts * 8), o_aux = up256(o_perm + (size_t)perm_off.back() * 4);
    const size_t o_morph = up256(o_aux + (size_t)naux * 7 * 8), o_pairs = up256(o_morph + morphs.size() * sizeof(NnMorphH));
    const size_t pts_bytes = o_pairs;   // the work lists follow once they are known
    int rc = e->ensure(e->host_pts, pts_bytes, true);
    if (rc) return rc;
    unsigned char* h = (unsigned char*)e->host_pts.p;
    double *hx = (double*)(h + o_x), *hy = (double*)(h + o_y), *hz = (double*)(h + o_z);
    double* haux = (double*)(h + o_aux);   // 7 planes of naux: bx by bz ux uy uz flag
    std::vector<int64_t> goff(S + 1, 0);
    for (size_t s = 0; s < S; ++s) goff[s + 1] = goff[s] + (sets[s].n + qpb - 1) / qpb;
    std::vector<double> box((size_t)goff.back() * 6);   // lo xyz, hi xyz
    { TraceTimer tt("nn: stage points + boxes");
    parallel_for((int)(S + aux.size()), [&](int job) {
        if ((size_t)job >= S) {   // one auxiliary pool entry
            const Aux& ax = aux[(size_t)job - S];
            const int32_t* pm = ax.perm >= 0 ? perms[(size_t)ax.perm].data() : nullptr;
            for (int64_t j = 0; j < ax.n; ++j) {
                const int64_t i = pm ? (int64_t)pm[j] : j;
                for (int a = 0; a < 3; ++a) {
                    haux[(size_t)a * (size_t)naux + (size_t)(ax.off + j)] = ax.xyz[3 * i + a];
                    haux[(size_t)(3 + a) * (size_t)naux + (size_t)(ax.off + j)] = ax.unit[3 * i + a];
                }
                haux[(size_t)6 * (size_t)naux + (size_t)(ax.off + j)] = ax.has[i] ? 1.0 : 0.0;
            }
            return;
        }
        const int si = job;
        const Set3& st = sets[(size_t)si];
        const bool derived = aux_of[(size_t)si] >= 0;
        double *dx = hx + soff[(size_t)si], *dy = hy + soff[(size_t)si], *dz = hz + soff[(size_t)si];
        const int32_t* pm = perm_of[(size_t)si] >= 0 ? perms[(size_t)perm_of[(size_t)si]].data() : nullptr;
        for (int64_t g0 = 0, g = goff[(size_t)si]; g0 < st.n; g0 += qpb, ++g) {
            double* b = box.data() + (size_t)g * 6;
            b[0] = b[1] = b[2] = DBL_MAX; b[3] = b[4] = b[5] = -DBL_MAX;
            for (int64_t j = g0; j < std::min(st.n, g0 + qpb); ++j) {
                const int64_t i = pm ? (int64_t)pm[j] : j;
                const double v[3] = {st.at(i, 0), st.at(i, 1), st.at(i, 2)};
                if (!derived) { dx[j] = v[0]; dy[j] = v[1]; dz[j] = v[2]; }
                for (int a = 0; a < 3; ++a) { b[a] = std::min(b[a], v[a]); b[3 + a] = std::max(b[3 + a], v[a]); }
            }
        }
    });
    }
    for (size_t k = 0; k < perms.size(); ++k)
        std::memcpy(h + o_perm + (size_t)perm_off[k] * 4, perms[k].data(), perms[k].size() * 4);
    if (!morphs.empty()) std::memcpy(h + o_morph, morphs.data(), morphs.size() * sizeof(NnMorphH));

    // ---- work lists -------------------------------------------------------------------------------
    // squared distance between two boxes, shaved so that rounding can never overstate it
    auto box_lb2 = [](const double* a, const double* b) {
        double s = 0.0;
        for (int ax = 0; ax < 3; ++ax) {
            const double gap = std::max(0.0, std::max(a[ax] - b[3 + ax], b[ax] - a[3 + ax]));
            s += gap * gap;
        }
        return s * (1.0 - 1e-12);
    };
    TraceTimer tt_wl("nn: work lists");
    const int gpc = ch / qpb;   // groups per chunk
    std::vector<std::vector<NnWorkH>> la(hp.size()), lb(hp.size());   // per pair, built over the worker pool
    parallel_for((int)hp.size(), [&](int ii) {
        const size_t i = (size_t)ii;
        std::vector<NnWorkH>&wa = la[i], &wb = lb[i];
        const int32_t q = pr[(size_t)owner[i]][0], p = pr[(size_t)owner[i]][1];
        const int64_t nq = hp[i].nq, np = hp[i].np;
        const int64_t n_chunks = (np + ch - 1) / ch;
        const bool prune = perm_of[(size_t)q] >= 0 && perm_of[(size_t)p] >= 0 && n_chunks > 2 && gpc >= 1 && ch % qpb == 0;
        if (!prune) {
            for (int64_t q0 = 0; q0 < nq; q0 += qpb)
                for (int64_t c0 = 0; c0 < np; c0 += (int64_t)span * ch)
                    wa.push_back(NnWorkH{(int32_t)i, (int32_t)q0, (int32_t)c0, span, 0.0});
            return;
        }
        std::vector<std::pair<double, int32_t>> cand((size_t)n_chunks);
        for (int64_t q0 = 0, qb = 0; q0 < nq; q0 += qpb, ++qb) {
            const double* bq = box.data() + (size_t)(goff[(size_t)q] + qb) * 6;
            for (int64_t c = 0; c < n_chunks; ++c) {
                double lb2 = DBL_MAX;   // a chunk's box is the union of its groups': the smallest of their distances
                for (int64_t g = c * gpc; g < std::min<int64_t>((c + 1) * gpc, goff[(size_t)p + 1] - goff[(size_t)p]); ++g)
                    lb2 = std::min(lb2, box_lb2(bq, box.data() + (size_t)(goff[(size_t)p] + g) * 6));
                cand[(size_t)c] = {lb2, (int32_t)c};
            }
            std::sort(cand.begin(), cand.end());   // nearest chunks first: they tighten the minima the others check
            wa.push_back(NnWorkH{(int32_t)i, (int32_t)q0, cand[0].second * ch, 1, 0.0});
            for (size_t c = 1; c < cand.size(); ++c)
                wb.push_back(NnWorkH{(int32_t)i, (int32_t)q0, cand[c].second * ch, 1, cand[c].first});
        }
    });
    std::vector<NnWorkH> wa, wb;
    {
        size_t na_ = 0, nb_ = 0;
        for (size_t i = 0; i < hp.size(); ++i) { na_ += la[i].size(); nb_ += lb[i].size(); }
        wa.reserve(na_); wb.reserve(nb_);
        for (size_t i = 0; i < hp.size(); ++i) { wa.insert(wa.end(), la[i].begin(), la[i].end()); wb.insert(wb.end(), lb[i].begin(), lb[i].end()); }
    }
    if (wa.size() + wb.size() > (size_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "nn batch exceeds 2^30 work items");
    tt_wl.stop();
    TraceTimer tt_dev("nn: copies + kernels");

    const size_t o_wa = up256(o_pairs + hp.size() * sizeof(NnPairH)), o_wb = up256(o_wa + wa.size() * sizeof(NnWorkH));
    const size_t in_bytes = up256(o_wb + wb.size() * sizeof(NnWorkH));
    const size_t o_out = in_bytes, o_sums = up256(o_out + (size_t)nout * 8), total = up256(o_sums + hp.size() * 8);
    // descriptors and results go through the level buffers (the point staging above is still in flight-free
    // pinned memory of its own), so nothing staged so far moves
    if ((rc = e->ensure(e->host_lvl, std::max(in_bytes - o_pairs, (size_t)nout * 8), true))) return rc;
    if ((rc = e->ensure(e->dev_pts, total, false))) return rc;
    unsigned char* hl = (unsigned char*)e->host_lvl.p;
    std::memcpy(hl, hp.data(), hp.size() * sizeof(NnPairH));
    std::memcpy(hl + (o_wa - o_pairs), wa.data(), wa.size() * sizeof(NnWorkH));
    std::memcpy(hl + (o_wb - o_pairs), wb.data(), wb.size() * sizeof(NnWorkH));
    unsigned char* d = (unsigned char*)e->dev_pts.p;
    if (morphs.empty()) {
        MM_TRY_HIP(hipMemcpyAsync(d, h, pts_bytes, hipMemcpyHostToDevice, e->stream));
    } else {
        // only the sets that exist on the host travel; the derived ones are written by the device
        for (size_t s2 = 0; s2 < S; ++s2) {
            if (aux_of[s2] >= 0 || sets[s2].n == 0) continue;
            for (size_t o : {o_x, o_y, o_z})
                MM_TRY_HIP(hipMemcpyAsync(d + o + (size_t)soff[s2] * 8, h + o + (size_t)soff[s2] * 8, (size_t)sets[s2].n * 8,
                                          hipMemcpyHostToDevice, e->stream));
        }
        MM_TRY_HIP(hipMemcpyAsync(d + o_perm, h + o_perm, o_pairs - o_perm, hipMemcpyHostToDevice, e->stream));
        const hipError_t hm = launch_nn3_morph(d + o_morph, (int)morphs.size(), (const double*)(d + o_aux), naux,
                                               (double*)(d + o_x), (double*)(d + o_y), (double*)(d + o_z), e->stream);
        if (hm != hipSuccess) return hip_error(hm, "morph launch");
    }
    MM_TRY_HIP(hipMemcpyAsync(d + o_pairs, hl, in_bytes - o_pairs, hipMemcpyHostToDevice, e->stream));
    const hipError_t he = launch_nn3_min(d + o_pairs, d + o_wa, (int)wa.size(), d + o_wb, (int)wb.size(),
                                         (const double*)(d + o_x), (const double*)(d + o_y), (const double*)(d + o_z),
                                         (const int32_t*)(d + o_perm), (double*)(d + o_out), nout, e->stream);
    if (he != hipSuccess) return hip_error(he, "nearest-neighbour launch");
    if (sums) {
        const hipError_t hs = launch_nn3_sums(d + o_pairs, (int)hp.size(), (const double*)(d + o_out), (double*)(d + o_sums), e->stream);
        if (hs != hipSuccess) return hip_error(hs, "sum launch");
        MM_TRY_HIP(hipMemcpyAsync(hl, d + o_sums, hp.size() * 8, hipMemcpyDeviceToHost, e->stream));
        MM_TRY_HIP(hipStreamSynchronize(e->stream));
        for (size_t i = 0; i < hp.size(); ++i) (*sums)[(size_t)owner[i]] = ((const double*)hl)[i];
        return MM_OK;
    }
    MM_TRY_HIP(hipMemcpyAsync(hl, d + o_out, (size_t)nout * 8, hipMemcpyDeviceToHost, e->stream));
    MM_TRY_HIP(hipStreamSynchronize(e->stream));
    const double* res = (const double*)hl;
    for (size_t i = 0; i < hp.size(); ++i) view[(size_t)owner[i]].p = res + hp[i].out_off;
    return MM_OK;
}

int nn_batch(Engine* e, const std::vector<Set3>& sets, const std::vector<std::array<int32_t, 2>>& pr,
             std::vector<std::vector<double>>& out)
{
    std::vector<MinView> view;
    int rc = nn_batch_view(e, sets, pr, view);
    if (rc) return rc;
    out.assign(pr.size(), {});
    for (size_t k = 0; k < pr.size(); ++k) {
        if (view[k].p) out[k].assign(view[k].p, view[k].p + view[k].n);
        else out[k].assign((size_t)view[k].n, INFINITY);
    }
    return MM_OK;
}

// symmetric_nn_distance (:188-216) from the two vectors of minima
double symmetric_from_minima(const MinView& a_to_b, const MinView& b_to_a)
{
    if (a_to_b.n == 0 || b_to_a.n == 0) return INFINITY;                      // :189-191
    if (!a_to_b.p || !b_to_a.p) return INFINITY;                              // unreachable: both sets non-empty
    double sa = 0.0;
    for (int64_t i = 0; i < a_to_b.n; ++i) sa += a_to_b.p[i];                 // :193-200, index order
    const double avg_a = sa / (double)a_to_b.n;                              // :202
    double sb = 0.0;
    for (int64_t i = 0; i < b_to_a.n; ++i) sb += b_to_a.p[i];                 // :204-211
    const double avg_b = sb / (double)b_to_a.n;                              // :213
    return std::sqrt((avg_a + avg_b) / 2.0);                                  // :215
}

// the same from the two sums of minima (sequential folds done on the device)
double symmetric_from_sums(double sa, int64_t na, double sb, int64_t nb)
{
    if (na == 0 || nb == 0) return INFINITY;                                  // :189-191
    return std::sqrt((sa / (double)na + sb / (double)nb) / 2.0);              // :202, :213, :215
}

// unit vector from the closest centerline point to each point (:226-235); has[i] = 0 when the point sits
// on its centerline point (try_normalize(0.0) fails -> the point does not move)
void radial_units(const mm_clpoint* cl, int64_t ncl, const double* pts, int64_t n, std::vector<double>& unit,
                  std::vector<uint8_t>& has)
{
    unit.assign((size_t)n * 3, 0.0);
    has.assign((size_t)n, 0);
    constexpr int64_t kBlock = 256;
    parallel_for((int)((n + kBlock - 1) / kBlock), [&](int blk) {
        const int64_t i0 = (int64_t)blk * kBlock;
        for (int64_t i = i0; i < std::min(n, i0 + kBlock); ++i) {
            const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
            double best = DBL_MAX;                                            // :249-258
            int64_t kb = 0;
            for (int64_t k = 0; k < ncl; ++k) {
                const double dx = x - cl[k].x, dy = y - cl[k].y, dz = z - cl[k].z;
                const double d = dx * dx + dy * dy + dz * dz;
                if (d < best) { best = d; kb = k; }
            }
            const double vx = x - cl[kb].x, vy = y - cl[kb].y, vz = z - cl[kb].z;
            const double nn = std::sqrt(vx * vx + vy * vy + vz * vz);
            if (nn > 0.0) { unit[3 * i] = vx / nn; unit[3 * i + 1] = vy / nn; unit[3 * i + 2] = vz / nn; has[i] = 1; }
        }
    });
}

inline void morph(const double* pts, const std::vector<double>& unit, const std::vector<uint8_t>& has, int64_t n,
                  double adj, double* out)
{
    for (int64_t i = 0; i < n; ++i) {
        if (has[i]) {
            out[3 * i] = pts[3 * i] + unit[3 * i] * adj;                      // :236 p + unit * x
            out[3 * i + 1] = pts[3 * i + 1] + unit[3 * i + 1] * adj;
            out[3 * i + 2] = pts[3 * i + 2] + unit[3 * i + 2] * adj;
        } else { out[3 * i] = pts[3 * i]; out[3 * i + 1] = pts[3 * i + 1]; out[3 * i + 2] = pts[3 * i + 2]; }  // :239
    }
}

// the 41-step loop of :75-87 / :107-129: all scalings scored in one device batch
int scaling_search(Engine* e, const double* pts, int64_t n, const double* ref, int64_t nr, const mm_clpoint* cl,
                   int64_t ncl, double& best, double* all_dist)
{
    const double start = -2.0, end = 2.0, step = 0.1;
    const int steps = (int)std::round((end - start) / step);                  // :70-73
    best = DBL_MAX;
    double min_dist = DBL_MAX;
    std::vector<double> dist((size_t)steps + 1, INFINITY);
    if (n > 0 && nr > 0) {
        if (ncl <= 0) return set_error(MM_ERR_INVALID, "diameter search: empty centerline");
        std::vector<double> unit; std::vector<uint8_t> has;
        { TraceTimer tt("ccta: radial units"); radial_units(cl, ncl, pts, n, unit, has); }
        std::vector<Set3> sets;
        std::vector<std::array<int32_t, 2>> pr;
        sets.push_back(Set3{ref, nr});
        for (int i = 0; i <= steps; ++i) {
            const double x = start + (double)i * step;                        // :79
            sets.push_back(Set3{pts, n, unit.data(), has.data(), x});         // :80, computed where it is used
            pr.push_back({0, i + 1});                                         // reference -> moved
            pr.push_back({i + 1, 0});                                         // moved -> reference
        }
        // every morphed copy shares the spatial order of the first (points move by at most 2 mm)
        std::vector<int32_t> order_like(sets.size(), 1);
        order_like[0] = 0;
        std::vector<MinView> mins;
        std::vector<double> sums;   // the 82 sequential sums are taken on the device: 82 numbers come back, not 13 MB
        int rc = nn_batch_view(e, sets, pr, mins, &order_like, &sums);
        if (rc) return rc;
        for (int i = 0; i <= steps; ++i)
            dist[(size_t)i] = symmetric_from_sums(sums[2 * (size_t)i], mins[2 * (size_t)i].n, sums[2 * (size_t)i + 1], mins[2 * (size_t)i + 1].n);  // :81
    }
    for (int i = 0; i <= steps; ++i) {
        if (all_dist) all_dist[i] = dist[(size_t)i];
        if (dist[(size_t)i] < min_dist) { min_dist = dist[(size_t)i]; best = start + (double)i * step; }  // :82-85
    }
    return MM_OK;
}

// find_region_points (:133-183)
int region_points(Engine* e, const double* an, int64_t n, const double* ref, int64_t nr, int64_t n_points,
                  std::vector<double>& selected, std::vector<double>& remaining)
{
    selected.clear();
    if (n == 0 || nr == 0 || n_points == 0) { remaining.assign(an, an + 3 * n); return MM_OK; }  // :138-140
    std::vector<std::vector<double>> mins;
    int rc = nn_batch(e, {Set3{an, n}, Set3{ref, nr}}, {{0, 1}}, mins);      // :142-152
    if (rc) return rc;
    const std::vector<double>& d = mins[0];
    std::vector<int64_t> order((size_t)n);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::sort(order.begin(), order.end(), [&d](int64_t a, int64_t b) {       // :154-158
        if (d[(size_t)a] < d[(size_t)b]) return true;
        if (d[(size_t)a] > d[(size_t)b]) return false;
        return a < b;
    });
    const int64_t take = std::min(n_points, n);                               // :160
    std::vector<uint8_t> sel((size_t)n, 0);
    selected.reserve((size_t)take * 3);
    for (int64_t k = 0; k < take; ++k) {
        const int64_t i = order[(size_t)k];
        sel[(size_t)i] = 1;
        selected.insert(selected.end(), an + 3 * i, an + 3 * i + 3);          // :165-168
    }
    remaining.clear();
    remaining.reserve((size_t)(n - take) * 3);
    for (int64_t i = 0; i < n; ++i) if (!sel[(size_t)i]) remaining.insert(remaining.end(), an + 3 * i, an + 3 * i + 3);  // :170-180
    return MM_OK;
}

// ---- neighbour counts within a radius (clean_up_non_section_points, scale_coronary.rs:342-409) ------------
// counts[k][i] = #{p in sets[pr[k][1]] : |q_i - p|^2 <= r2} for every query q_i of sets[pr[k][0]], exact f64 on
// the device (k_nn3_count).  Large sets are staged in slab order and only the (query block, chunk) combinations
// whose bounding boxes come within the radius are launched; a count does not depend on the order.
int radius_counts(Engine* e, const std::vector<Set3>& sets, const std::vector<std::array<int32_t, 2>>& pr, double r2,
                  std::vector<std::vector<uint32_t>>& counts)
{
    counts.assign(pr.size(), {});
    const size_t S = sets.size();
    std::vector<int64_t> soff(S + 1, 0);
    for (size_t s = 0; s < S; ++s) soff[s + 1] = soff[s] + sets[s].n;
    const int64_t npts = soff.back();
    const int qpb = nn_queries_per_block(), ch = nn_chunk_points();
    constexpr int64_t kSortMin = 4096;
    std::vector<std::vector<int32_t>> perms(S);
    parallel_for((int)S, [&](int s) { if (sets[(size_t)s].n >= kSortMin) slab_order(sets[(size_t)s], perms[(size_t)s]); });
    std::vector<int64_t> perm_off(S + 1, 0);
    for (size_t s = 0; s < S; ++s) perm_off[s + 1] = perm_off[s] + (int64_t)perms[s].size();

    std::vector<NnPairH> hp;
    std::vector<int> owner;
    int64_t nout = 0;
    for (size_t k = 0; k < pr.size(); ++k) {
        const int32_t q = pr[k][0], p = pr[k][1];
        if (q < 0 || p < 0 || (size_t)q >= S || (size_t)p >= S) return set_error(MM_ERR_INVALID, "radius counts: set index out of range");
        counts[k].assign((size_t)sets[(size_t)q].n, 0u);
        if (sets[(size_t)q].n == 0 || sets[(size_t)p].n == 0) continue;
        hp.push_back(NnPairH{(int32_t)soff[(size_t)q], (int32_t)sets[(size_t)q].n, (int32_t)soff[(size_t)p], (int32_t)sets[(size_t)p].n,
                             (int32_t)nout, perms[(size_t)q].empty() ? -1 : (int32_t)perm_off[(size_t)q]});
        owner.push_back((int)k);
        nout += sets[(size_t)q].n;
    }
    if (hp.empty()) return MM_OK;
    if (npts > (int64_t)1 << 30 || nout > (int64_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "radius counts: batch exceeds 2^30 points");

    // staged points (slab order where sorted) + bounding boxes of groups of ch points (ch is a multiple of qpb)
    const size_t o_x = 0, o_y = up256((size_t)npts * 8), o_z = up256(o_y + (size_t)npts * 8);
    const size_t o_perm = up256(o_z + (size_t)npts * 8), o_pairs = up256(o_perm + (size_t)perm_off.back() * 4);
    int rc = e->ensure(e->host_pts, o_pairs, true);
    if (rc) return rc;
    unsigned char* h = (unsigned char*)e->host_pts.p;
    double *hx = (double*)(h + o_x), *hy = (double*)(h + o_y), *hz = (double*)(h + o_z);
    const int g = qpb;   // box granularity: one query block; a chunk's box is the union of its ch / qpb groups
    std::vector<int64_t> goff(S + 1, 0);
    for (size_t s = 0; s < S; ++s) goff[s + 1] = goff[s] + (sets[s].n + g - 1) / g;
    std::vector<double> box((size_t)goff.back() * 6);
    parallel_for((int)S, [&](int si) {
        const Set3& st = sets[(size_t)si];
        const int32_t* pm = perms[(size_t)si].empty() ? nullptr : perms[(size_t)si].data();
        double *dx = hx + soff[(size_t)si], *dy = hy + soff[(size_t)si], *dz = hz + soff[(size_t)si];
        for (int64_t g0 = 0, gi = goff[(size_t)si]; g0 < st.n; g0 += g, ++gi) {
            double* b = box.data() + (size_t)gi * 6;
            b[0] = b[1] = b[2] = DBL_MAX; b[3] = b[4] = b[5] = -DBL_MAX;
            for (int64_t j = g0; j < std::min<int64_t>(st.n, g0 + g); ++j) {
                const int64_t i = pm ? (int64_t)pm[j] : j;
                const double v[3] = {st.xyz[3 * i], st.xyz[3 * i + 1], st.xyz[3 * i + 2]};
                dx[j] = v[0]; dy[j] = v[1]; dz[j] = v[2];
                for (int a = 0; a < 3; ++a) { b[a] = std::min(b[a], v[a]); b[3 + a] = std::max(b[3 + a], v[a]); }
            }
        }
    });
    for (size_t s = 0; s < S; ++s)
        if (!perms[s].empty()) std::memcpy(h + o_perm + (size_t)perm_off[s] * 4, perms[s].data(), perms[s].size() * 4);

    auto box_lb2 = [](const double* a, const double* b) {   // squared distance between two boxes, never overstated
        double s2 = 0.0;
        for (int ax = 0; ax < 3; ++ax) {
            const double gap = std::max(0.0, std::max(a[ax] - b[3 + ax], b[ax] - a[3 + ax]));
            s2 += gap * gap;
        }
        return s2 * (1.0 - 1e-12);
    };
    const int gpc = std::max(1, ch / g);
    std::vector<std::vector<NnWorkH>> lw(hp.size());
    parallel_for((int)hp.size(), [&](int ii) {
        const size_t i = (size_t)ii;
        const int32_t q = pr[(size_t)owner[i]][0], p = pr[(size_t)owner[i]][1];
        const int64_t nq = hp[i].nq, np = hp[i].np, n_chunks = (np + ch - 1) / ch;
        for (int64_t q0 = 0, qb = 0; q0 < nq; q0 += qpb, ++qb) {
            const double* bq = box.data() + (size_t)(goff[(size_t)q] + qb) * 6;
            for (int64_t c = 0; c < n_chunks; ++c) {
                double lb2 = DBL_MAX;
                for (int64_t gi = c * gpc; gi < std::min<int64_t>((c + 1) * gpc, goff[(size_t)p + 1] - goff[(size_t)p]); ++gi)
                    lb2 = std::min(lb2, box_lb2(bq, box.data() + (size_t)(goff[(size_t)p] + gi) * 6));
                if (lb2 <= r2) lw[i].push_back(NnWorkH{(int32_t)i, (int32_t)q0, (int32_t)(c * ch), 1, lb2});
            }
        }
    });
    std::vector<NnWorkH> work;
    for (auto& v : lw) work.insert(work.end(), v.begin(), v.end());
    if (work.size() > (size_t)1 << 30) return set_error(MM_ERR_TOO_LARGE, "radius counts: too many work items");

    const size_t o_work = up256(o_pairs + hp.size() * sizeof(NnPairH)), in_bytes = up256(o_work + work.size() * sizeof(NnWorkH));
    const size_t o_out = in_bytes, total = up256(o_out + (size_t)nout * 4);
    if ((rc = e->ensure(e->host_lvl, std::max(in_bytes - o_pairs, (size_t)nout * 4), true))) return rc;
    if ((rc = e->ensure(e->dev_pts, total, false))) return rc;
    unsigned char* hl = (unsigned char*)e->host_lvl.p;
    std::memcpy(hl, hp.data(), hp.size() * sizeof(NnPairH));
    if (!work.empty()) std::memcpy(hl + (o_work - o_pairs), work.data(), work.size() * sizeof(NnWorkH));
    unsigned char* d = (unsigned char*)e->dev_pts.p;
    MM_TRY_HIP(hipMemcpyAsync(d, h, o_pairs, hipMemcpyHostToDevice, e->stream));
    MM_TRY_HIP(hipMemcpyAsync(d + o_pairs, hl, in_bytes - o_pairs, hipMemcpyHostToDevice, e->stream));
    const hipError_t he = launch_nn3_count(d + o_pairs, d + o_work, (int)work.size(), (const double*)(d + o_x), (const double*)(d + o_y),
                                           (const double*)(d + o_z), (const int32_t*)(d + o_perm), r2, (unsigned int*)(d + o_out), nout,
                                           e->stream);
    if (he != hipSuccess) return hip_error(he, "radius-count launch");
    MM_TRY_HIP(hipMemcpyAsync(hl, d + o_out, (size_t)nout * 4, hipMemcpyDeviceToHost, e->stream));
    MM_TRY_HIP(hipStreamSynchronize(e->stream));
    const uint32_t* res = (const uint32_t*)hl;
    for (size_t i = 0; i < hp.size(); ++i)
        std::memcpy(counts[(size_t)owner[i]].data(), res + hp[i].out_off, (size_t)hp[i].nq * 4);
    return MM_OK;
}

// clean_up_non_section_points (:342-409).  to_ref[i] = 1: point i of `cleanup` joins the reference set (appended
// in input order), 0: it stays.
int clean_up_points(Engine* e, const double* cleanup, int64_t nc, const double* reference, int64_t nr, double radius,
                    double min_ratio, std::vector<uint8_t>& to_ref)
{
    to_ref.assign((size_t)nc, 0);
    if (nc == 0) return MM_OK;                                                        // :353-355
    const double r2 = radius * radius;                                                // :348
    std::vector<std::vector<uint32_t>> cnt;
    int rc = radius_counts(e, {Set3{cleanup, nc}, Set3{reference, nr}}, {{0, 1}, {0, 0}}, r2, cnt);
    if (rc) return rc;
    for (int64_t i = 0; i < nc; ++i) {
        const uint64_t ref_n = cnt[0][(size_t)i];                                     // :375-377
        const uint64_t self_n = cnt[1][(size_t)i] > 0 ? cnt[1][(size_t)i] - 1 : 0;    // :381-384 saturating_sub(1)
        const uint64_t total = ref_n + self_n;
        if (total > 0) {                                                              // :388-400
            const double ratio = (double)ref_n / (double)total;
            to_ref[(size_t)i] = ratio >= min_ratio ? 1 : 0;
        }
    }
    return MM_OK;
}

int engine_of(mm_engine* h, Engine*& e)
{
    e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    const hipError_t he = hipSetDevice(e->device);
    if (he != hipSuccess) return hip_error(he, "hipSetDevice");
    return MM_OK;
}

}  // namespace
}  // namespace mm

using namespace mm;

extern "C" {

int mm_nn_min_sq_batch(mm_engine* h, int n_sets, const int64_t* set_off, const double* xyz, int n_pairs,
                       const int32_t* q_set, const int32_t* p_set, const int64_t* out_off, double* out)
{
    Engine* e;
    int rc = engine_of(h, e);
    if (rc) return rc;
    if (n_sets < 0 || n_pairs < 0 || (n_sets > 0 && !set_off) || (n_pairs > 0 && (!q_set || !p_set || !out_off || !out)))
        return set_error(MM_ERR_INVALID, "mm_nn_min_sq_batch: bad arguments");
    std::vector<Set3> sets((size_t)n_sets);
    for (int s = 0; s < n_sets; ++s) {
        const int64_t n = set_off[s + 1] - set_off[s];
        if (n < 0 || n > INT32_MAX || (n > 0 && !xyz)) return set_error(MM_ERR_INVALID, "bad set extent");
        sets[(size_t)s] = Set3{xyz ? xyz + 3 * set_off[s] : nullptr, n};
    }
    std::vector<std::array<int32_t, 2>> pr((size_t)n_pairs);
    for (int k = 0; k < n_pairs; ++k) {
        pr[(size_t)k] = {q_set[k], p_set[k]};
        if (q_set[k] < 0 || q_set[k] >= n_sets || p_set[k] < 0 || p_set[k] >= n_sets)
            return set_error(MM_ERR_INVALID, "mm_nn_min_sq_batch: set index out of range");
        if (out_off[k + 1] - out_off[k] != sets[(size_t)q_set[k]].n)
            return set_error(MM_ERR_INVALID, "mm_nn_min_sq_batch: out_off does not match the query set sizes");
    }
    std::vector<std::vector<double>> mins;
    if ((rc = nn_batch(e, sets, pr, mins))) return rc;
    for (int k = 0; k < n_pairs; ++k)
        if (!mins[(size_t)k].empty()) std::memcpy(out + out_off[k], mins[(size_t)k].data(), mins[(size_t)k].size() * 8);
    return MM_OK;
}

int mm_symmetric_nn_distance(mm_engine* h, const double* a, int64_t na, const double* b, int64_t nb, double* out)
{
    Engine* e;
    int rc = engine_of(h, e);
    if (rc) return rc;
    if (!out || na < 0 || nb < 0 || (na > 0 && !a) || (nb > 0 && !b)) return set_error(MM_ERR_INVALID, "mm_symmetric_nn_distance: bad arguments");
    if (na == 0 || nb == 0) { *out = INFINITY; return MM_OK; }
    std::vector<MinView> mins;
    if ((rc = nn_batch_view(e, {Set3{a, na}, Set3{b, nb}}, {{0, 1}, {1, 0}}, mins))) return rc;
    *out = symmetric_from_minima(mins[0], mins[1]);
    return MM_OK;
}

int mm_diameter_morphing(const mm_clpoint* cl, int64_t ncl, const double* pts, int64_t n, double adj, double* out)
{
    if (n < 0 || (n > 0 && (!pts || !out))) return set_error(MM_ERR_INVALID, "mm_diameter_morphing: bad arguments");
    if (n > 0 && (ncl <= 0 || !cl)) return set_error(MM_ERR_INVALID, "mm_diameter_morphing: empty centerline");  // points[0] panics
    std::vector<double> unit; std::vector<uint8_t> has;
    radial_units(cl, ncl, pts, n, unit, has);
    morph(pts, unit, has, n, adj, out);
    return MM_OK;
}

int64_t mm_find_region_points(mm_engine* h, const double* an, int64_t n, const double* ref, int64_t nr,
                              int64_t n_points, double* selected, double* remaining)
{
    Engine* e;
    int rc = engine_of(h, e);
    if (rc) return rc;
    if (n < 0 || nr < 0 || n_points < 0 || (n > 0 && (!an || !selected || !remaining)) || (nr > 0 && !ref))
        return set_error(MM_ERR_INVALID, "mm_find_region_points: bad arguments");
    std::vector<double> sel, rem;
    if ((rc = region_points(e, an, n, ref, nr, n_points, sel, rem))) return rc;
    if (!sel.empty()) std::memcpy(selected, sel.data(), sel.size() * 8);
    if (!rem.empty()) std::memcpy(remaining, rem.data(), rem.size() * 8);
    return (int64_t)(sel.size() / 3);
}

int mm_aortic_diameter_optimization(mm_engine* h, const double* intramural, int64_t ni, const double* reference,
                                    int64_t nr, const mm_clpoint* cl, int64_t ncl, double* best, double* all_dist)
{
    Engine* e;
    int rc = engine_of(h, e);
    if (rc) return rc;
    if (!best || ni < 0 || nr < 0 || (ni > 0 && !intramural) || (nr > 0 && !reference))
        return set_error(MM_ERR_INVALID, "mm_aortic_diameter_optimization: bad arguments");
    return scaling_search(e, intramural, ni, reference, nr, cl, ncl, *best, all_dist);
}

int mm_diameter_optimization(mm_engine* h, const double* an, int64_t n, int64_t n_prox, int64_t n_dist,
                             const mm_clpoint* cl, int64_t ncl, const double* pref, int64_t npr, const double* dref,
                             int64_t ndr, double* prox_best, double* dist_best)
{
    Engine* e;
    int rc = engine_of(h, e);
    if (rc) return rc;
    if (!prox_best || !dist_best || n < 0 || npr < 0 || ndr < 0 || n_prox < 0 || n_dist < 0 || (n > 0 && !an) ||
        (npr > 0 && !pref) || (ndr > 0 && !dref))
        return set_error(MM_ERR_INVALID, "mm_diameter_optimization: bad arguments");
    std::vector<double> prox, rest, dist, rest2;
    if ((rc = region_points(e, an, n, pref, npr, n_prox, prox, rest))) return rc;                    // :98-99
    if ((rc = region_points(e, rest.data(), (int64_t)(rest.size() / 3), dref, ndr, n_dist, dist, rest2))) return rc;  // :100
    if ((rc = scaling_search(e, prox.data(), (int64_t)(prox.size() / 3), pref, npr, cl, ncl, *prox_best, nullptr))) return rc;  // :112-120
    return scaling_search(e, dist.data(), (int64_t)(dist.size() / 3), dref, ndr, cl, ncl, *dist_best, nullptr);                // :121-129
}

int mm_wall_diameter_optimization(const mm_clpoint* cl, int64_t ncl, const double ref[3], const double* aortic,
                                  int64_t na, double* out)
{
    if (!out || !ref || ncl < 0 || na < 0 || (ncl > 0 && !cl) || (na > 0 && !aortic))
        return set_error(MM_ERR_INVALID, "mm_wall_diameter_optimization: bad arguments");
    *out = 0.0;
    if (ncl == 0 || na == 0) return MM_OK;                                                           // :13-15
    int64_t kc = 0, ka = 0;
    double bc = INFINITY, ba = INFINITY;                                                             // min_by: first minimum (:17-37)
    for (int64_t k = 0; k < ncl; ++k) {
        const double dx = cl[k].x - ref[0], dy = cl[k].y - ref[1], dz = cl[k].z - ref[2];
        const double d = dx * dx + dy * dy + dz * dz;
        if (d < bc) { bc = d; kc = k; }
    }
    for (int64_t k = 0; k < na; ++k) {
        const double dx = aortic[3 * k] - ref[0], dy = aortic[3 * k + 1] - ref[1], dz = aortic[3 * k + 2] - ref[2];
        const double d = dx * dx + dy * dy + dz * dz;
        if (d < ba) { ba = d; ka = k; }
    }
    const double vx = ref[0] - cl[kc].x, vy = ref[1] - cl[kc].y, vz = ref[2] - cl[kc].z;             // :52
    const double nn = std::sqrt(vx * vx + vy * vy + vz * vz);
    if (!(nn > 0.0)) return MM_OK;                                                                   // :53-55
    const double ux = vx / nn, uy = vy / nn, uz = vz / nn;
    const double tx = ref[0] - aortic[3 * ka], ty = ref[1] - aortic[3 * ka + 1], tz = ref[2] - aortic[3 * ka + 2];  // :59
    const double t = tx * ux + ty * uy + tz * uz;                                                    // :60
    *out = t > 0.0 ? t : 0.0;                                                                        // :62
    return MM_OK;
}

int mm_clean_outlier_points(mm_engine* h, const double* cleanup, int64_t nc, const double* reference, int64_t nr,
                            double neighborhood_radius, double min_neighbor_ratio, uint8_t* to_reference)
{
    Engine* e;
    int rc = engine_of(h, e);
    if (rc) return rc;
    if (nc < 0 || nr < 0 || (nc > 0 && (!cleanup || !to_reference)) || (nr > 0 && !reference))
        return set_error(MM_ERR_INVALID, "mm_clean_outlier_points: bad arguments");
    std::vector<uint8_t> f;
    if ((rc = clean_up_points(e, cleanup, nc, reference, nr, neighborhood_radius, min_neighbor_ratio, f))) return rc;
    if (nc > 0) std::memcpy(to_reference, f.data(), (size_t)nc);
    return MM_OK;
}

// find_points_by_cl_region_rs (:263-312)
int mm_find_points_by_cl_region(mm_engine* h, const mm_clpoint* cl, const uint32_t* cl_frame_index, int64_t ncl,
                                const double* frame_centroids, int64_t n_frames, const double* pts, int64_t n,
                                uint8_t* label)
{
    Engine* e;
    int rc = engine_of(h, e);
    if (rc) return rc;
    // (no frame at all: the reference's `frames.len() - 1` underflows and panics.  ONE frame is not an error there: the
    // mean spacing is 0.0 / 0 = NaN, no centerline point is "in range" of it, every point is proximal or distal)
    if (n < 0 || ncl <= 0 || !cl || n_frames < 1 || !frame_centroids || (n > 0 && (!pts || !label)))
        return set_error(MM_ERR_INVALID, "mm_find_points_by_cl_region: bad arguments (needs a centerline and >= 1 frame)");
    double mean_dz = 0.0;                                                              // :268-272
    for (int64_t i = 1; i < n_frames; ++i) mean_dz += std::fabs(frame_centroids[3 * i + 2] - frame_centroids[3 * (i - 1) + 2]);
    mean_dz /= (double)(n_frames - 1);
    // find_cl_points_in_range (:314-338): frame indices of the centerline points within mean_dz of a frame centroid
    auto fidx = [&](int64_t k) { return cl_frame_index ? cl_frame_index[k] : (uint32_t)k; };
    std::vector<uint32_t> in_range;
    const double rr = mean_dz * mean_dz;
    for (int64_t f = 0; f < n_frames; ++f)
        for (int64_t k = 0; k < ncl; ++k) {
            const double dx = frame_centroids[3 * f] - cl[k].x, dy = frame_centroids[3 * f + 1] - cl[k].y,
                         dz = frame_centroids[3 * f + 2] - cl[k].z;
            if (dx * dx + dy * dy + dz * dz <= rr) in_range.push_back(fidx(k));
        }
    std::sort(in_range.begin(), in_range.end());
    in_range.erase(std::unique(in_range.begin(), in_range.end()), in_range.end());
    const double* dref = frame_centroids + 3 * (n_frames - 1);                        // :279
    // first pass (:289-300): between = the closest centerline point (first minimum, :245-260) is one of those;
    // second pass (:303-309): the rest is proximal if it exceeds the last centroid in all three coordinates
    std::vector<uint8_t> cls((size_t)n, 0);   // 0 proximal, 1 distal, 2 between
    constexpr int64_t kBlock = 256;
    parallel_for((int)((n + kBlock - 1) / kBlock), [&](int blk) {
        const int64_t i0 = (int64_t)blk * kBlock;
        for (int64_t i = i0; i < std::min(n, i0 + kBlock); ++i) {
            const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
            double best = DBL_MAX;
            int64_t kb = 0;
            for (int64_t k = 0; k < ncl; ++k) {
                const double dx = x - cl[k].x, dy = y - cl[k].y, dz = z - cl[k].z;
                const double d = dx * dx + dy * dy + dz * dz;
                if (d < best) { best = d; kb = k; }
            }
            if (std::binary_search(in_range.begin(), in_range.end(), fidx(kb))) cls[(size_t)i] = 2;
            else cls[(size_t)i] = (x > dref[0] && y > dref[1] && z > dref[2]) ? 0 : 1;
        }
    });
    std::vector<double> prox, dist, betw;
    std::vector<int64_t> iprox, idist;
    for (int64_t i = 0; i < n; ++i) {
        std::vector<double>& dst = cls[(size_t)i] == 0 ? prox : (cls[(size_t)i] == 1 ? dist : betw);
        dst.insert(dst.end(), pts + 3 * i, pts + 3 * i + 3);
        if (cls[(size_t)i] == 0) iprox.push_back(i); else if (cls[(size_t)i] == 1) idist.push_back(i);
        label[i] = cls[(size_t)i];
    }
    // :310-313 the two clean-ups; the second one sees the points the first one moved into `between`
    std::vector<uint8_t> mv;
    if ((rc = clean_up_points(e, prox.data(), (int64_t)iprox.size(), betw.data(), (int64_t)(betw.size() / 3), 1.0, 0.6, mv))) return rc;
    for (size_t k = 0; k < iprox.size(); ++k)
        if (mv[k]) { label[iprox[k]] = 3; betw.insert(betw.end(), pts + 3 * iprox[k], pts + 3 * iprox[k] + 3); }
    if ((rc = clean_up_points(e, dist.data(), (int64_t)idist.size(), betw.data(), (int64_t)(betw.size() / 3), 1.0, 0.6, mv))) return rc;
    for (size_t k = 0; k < idist.size(); ++k)
        if (mv[k]) label[idist[k]] = 4;
    return MM_OK;
}

}  // extern "C"
