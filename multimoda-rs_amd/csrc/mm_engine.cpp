// mm_engine.cpp -- host side of the C ABI (include/mm_hausdorff.h): engine, batch staging,
// device-resident plans.  Compiled with hipcc -ffp-contract=off.
//
// Data layout in HBM (one contiguous blob per plan, 256-B aligned sections):
//   [PairDesc x P][WorkItem x W]
//   [cos32 | sin32 | cos64 | sin64]                    per candidate (angle tables are
//                                                      computed on the host with glibc
//                                                      sin/cos = what Rust's f64::sin/cos
//                                                      call on linux-gnu)
//   [ref32x | ref32y | tgt32x | tgt32y]                f32 SoA, relative to the centre
//   [ref64x | ref64y | tgt64x | tgt64y]                f64 SoA, absolute
//   ---- outputs ----
//   [sq32 | sq64 | flag | items | n_items | best_cost | best_idx | n_rescored | all_costs]
#include "mm_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
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
int Engine::ensure_host(size_t bytes)
{
    if (bytes <= host_cap) return MM_OK;
    if (host_buf) (void)hipHostFree(host_buf);
    host_buf = nullptr;
    host_cap = 0;
    size_t cap = std::max(bytes, (size_t)1 << 20);
    cap = align_up(cap + cap / 2, 4096);
    MM_HIP(hipHostMalloc(&host_buf, cap, hipHostMallocDefault));
    host_cap = cap;
    return MM_OK;
}

int Engine::ensure_dev(size_t bytes)
{
    if (bytes <= dev_cap) return MM_OK;
    if (dev_buf) (void)hipFree(dev_buf);
    dev_buf = nullptr;
    dev_cap = 0;
    size_t cap = std::max(bytes, (size_t)4 << 20);
    cap = align_up(cap + cap / 2, 4096);
    MM_HIP(hipMalloc(&dev_buf, cap));
    dev_cap = cap;
    return MM_OK;
}

int Engine::profile_begin()
{
    if (!profile) return MM_OK;
    while (events.size() < 2 * (launches + 1)) {
        hipEvent_t ev;
        MM_HIP(hipEventCreate(&ev));
        events.push_back(ev);
    }
    MM_HIP(hipEventRecord(events[2 * launches], stream));
    return MM_OK;
}

int Engine::profile_end(double pair_evals, int64_t candidates)
{
    if (!profile) return MM_OK;
    MM_HIP(hipEventRecord(events[2 * launches + 1], stream));
    ++launches;
    prof_pair_evals += pair_evals;
    prof_candidates += candidates;
    return MM_OK;
}

// -------------------------------------------------------------------------------------
// plan construction
// -------------------------------------------------------------------------------------
struct Layout {
    size_t pairs, work, cos32, sin32, cos64, sin64;
    size_t r32x, r32y, t32x, t32y, r64x, r64y, t64x, t64y;
    size_t in_bytes;  // everything above (one H2D copy)
    size_t sq32, sq64, flag, items, n_items, best_cost, best_idx, n_rescored, all_costs;
    size_t total;
};

static Layout make_layout(int P, int W, int64_t A, int64_t NR, int64_t NT, bool want_costs)
{
    Layout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + std::max<size_t>(bytes, 16)); return at; };
    L.pairs = take((size_t)P * sizeof(PairDesc));
    L.work = take((size_t)W * sizeof(WorkItem));
    L.cos32 = take((size_t)A * 4); L.sin32 = take((size_t)A * 4);
    L.cos64 = take((size_t)A * 8); L.sin64 = take((size_t)A * 8);
    L.r32x = take((size_t)NR * 4); L.r32y = take((size_t)NR * 4);
    L.t32x = take((size_t)NT * 4); L.t32y = take((size_t)NT * 4);
    L.r64x = take((size_t)NR * 8); L.r64y = take((size_t)NR * 8);
    L.t64x = take((size_t)NT * 8); L.t64y = take((size_t)NT * 8);
    L.in_bytes = o;
    L.sq32 = take((size_t)A * 4); L.sq64 = take((size_t)A * 8); L.flag = take((size_t)A);
    L.items = take((size_t)A * sizeof(WorkItem)); L.n_items = take(16);
    L.best_cost = take((size_t)P * 8); L.best_idx = take((size_t)P * 4); L.n_rescored = take((size_t)P * 4);
    L.all_costs = want_costs ? take((size_t)A * 8) : 0;
    L.total = o;
    return L;
}

// f32 screening error bound for one pair (see DESIGN.md "screen-then-exact"): with
// u = 2^-24, rho_r / rho_t the largest distance of a reference / target point from the
// rotation centre, |H_f32 - H_f64| <= u * (3.9 rho_r + 10 rho_t); we use 24 u (rho_r+rho_t).
static double screen_delta(const double* rx, const double* ry, int64_t nr, const double* tx,
                           const double* ty, int64_t nt, double cx, double cy)
{
    double rr = 0.0, rt = 0.0;
    for (int64_t i = 0; i < nr; ++i) rr = std::max(rr, std::hypot(rx[i] - cx, ry[i] - cy));
    for (int64_t i = 0; i < nt; ++i) rt = std::max(rt, std::hypot(tx[i] - cx, ty[i] - cy));
    const double u = 5.9604644775390625e-08;  // 2^-24
    return 24.0 * u * (rr + rt) + 1e-300;
}

int Plan::build(Engine* e, int n_pairs, const int64_t* ref_off, const double* ref_x, const double* ref_y,
                const int64_t* tgt_off, const double* tgt_x, const double* tgt_y,
                const int64_t* ang_off, const double* angles, const double* cx, const double* cy,
                const int32_t* flags, int precision_, int32_t angle_begin, int32_t angle_end,
                bool want_costs, bool transient_)
{
    eng = e;
    P = n_pairs;
    precision = precision_;
    transient = transient_;
    if (n_pairs < 0) return set_error(MM_ERR_INVALID, "n_pairs < 0");
    if (precision != MM_PRECISION_F64 && precision != MM_PRECISION_F32)
        return set_error(MM_ERR_INVALID, "precision must be MM_PRECISION_F64 or MM_PRECISION_F32");
    if (angle_begin < 0) angle_begin = 0;
    slice_end = angle_end;

    // ---- sizes -----------------------------------------------------------------------
    host_pairs.assign(P, PairDesc{});
    trivial.assign(P, 0);
    first_angle.assign(P, NAN);
    user_ang_off.assign(ang_off, ang_off + P + 1);
    A = 0;
    int64_t NR = 0, NT = 0;
    max_na = 1; max_nbp = 16;
    pair_evals = 0.0;
    for (int p = 0; p < P; ++p) {
        const int64_t nr = ref_off[p + 1] - ref_off[p], nt = tgt_off[p + 1] - tgt_off[p];
        const int64_t na_full = ang_off[p + 1] - ang_off[p];
        if (nr < 0 || nt < 0 || na_full < 0) return set_error(MM_ERR_INVALID, "negative extent in batch offsets");
        const int64_t b = std::min<int64_t>(angle_begin, na_full), en = std::min<int64_t>(angle_end, na_full);
        const int64_t na = std::max<int64_t>(en - b, 0);
        PairDesc& d = host_pairs[p];
        d.ref_off = (int32_t)NR; d.n_ref = (int32_t)nr;
        d.tgt_off = (int32_t)NT; d.n_tgt = (int32_t)nt;
        d.ang_off = (int32_t)A;  d.n_ang = (int32_t)na;
        d.ang_full = (int32_t)na_full; d.ang_begin = (int32_t)b;
        d.flags = flags ? flags[p] : 0;
        d.cx = cx[p]; d.cy = cy[p];
        if (na > 0) first_angle[p] = angles[ang_off[p] + b];
        if (nr == 0 || nt == 0) {
            // process_utils.rs:86-88: an empty set makes every cost 0.0 -> first candidate wins;
            // nothing to launch for this pair.
            trivial[p] = 1;
            d.n_ang = 0;
            continue;
        }
        NR += nr; NT += nt; A += na;
        max_na = std::max<int>(max_na, (int)nr);
        max_nbp = std::max<int>(max_nbp, (int)((nt + 15) & ~(int64_t)15));
        pair_evals += 2.0 * (double)nr * (double)nt * (double)na;
    }
    if (A > (int64_t)1 << 30 || NR > (int64_t)1 << 30 || NT > (int64_t)1 << 30)
        return set_error(MM_ERR_TOO_LARGE, "batch exceeds 2^30 candidates or points");
    if (max_nbp > max_target_points_f64() || (precision == MM_PRECISION_F32 && max_nbp > max_target_points_f32()))
        return set_error(MM_ERR_TOO_LARGE, "target set does not fit the kernel's LDS budget (" +
                                               std::to_string(max_target_points_f64()) + " points)");

    // ---- work decomposition: one workgroup = `apb` consecutive candidates of one pair ----
    const int64_t target_wgs = 256 * 24;
    int apb = (int)std::min<int64_t>(64, std::max<int64_t>(1, (A + target_wgs - 1) / target_wgs));
    host_work.clear();
    for (int p = 0; p < P; ++p) {
        const PairDesc& d = host_pairs[p];
        for (int a0 = 0; a0 < d.n_ang; a0 += apb) {
            WorkItem w{p, a0, std::min(apb, d.n_ang - a0), 0};
            host_work.push_back(w);
        }
    }
    W = (int)host_work.size();

    const Layout L = make_layout(P, W, A, NR, NT, want_costs);
    in_bytes = L.in_bytes;
    total_bytes = L.total;

    // ---- stage inputs in pinned host memory -----------------------------------------------
    int rc = e->ensure_host(L.in_bytes);
    if (rc) return rc;
    unsigned char* h = (unsigned char*)e->host_buf;
    float *c32 = (float*)(h + L.cos32), *s32 = (float*)(h + L.sin32);
    double *c64 = (double*)(h + L.cos64), *s64 = (double*)(h + L.sin64);
    float *r32x = (float*)(h + L.r32x), *r32y = (float*)(h + L.r32y);
    float *t32x = (float*)(h + L.t32x), *t32y = (float*)(h + L.t32y);
    double *r64x = (double*)(h + L.r64x), *r64y = (double*)(h + L.r64y);
    double *t64x = (double*)(h + L.t64x), *t64y = (double*)(h + L.t64y);
    host_angles.assign((size_t)A, 0.0);
    for (int p = 0; p < P; ++p) {
        PairDesc& d = host_pairs[p];
        if (trivial[p]) continue;
        const double* rx = ref_x + ref_off[p]; const double* ry = ref_y + ref_off[p];
        const double* tx = tgt_x + tgt_off[p]; const double* ty = tgt_y + tgt_off[p];
        for (int i = 0; i < d.n_ref; ++i) {
            r64x[d.ref_off + i] = rx[i]; r64y[d.ref_off + i] = ry[i];
            r32x[d.ref_off + i] = (float)(rx[i] - d.cx); r32y[d.ref_off + i] = (float)(ry[i] - d.cy);
        }
        for (int i = 0; i < d.n_tgt; ++i) {
            t64x[d.tgt_off + i] = tx[i]; t64y[d.tgt_off + i] = ty[i];
            t32x[d.tgt_off + i] = (float)(tx[i] - d.cx); t32y[d.tgt_off + i] = (float)(ty[i] - d.cy);
        }
        const double* ang = angles + ang_off[p] + d.ang_begin;
        for (int a = 0; a < d.n_ang; ++a) {
            const double co = std::cos(ang[a]), si = std::sin(ang[a]);
            c64[d.ang_off + a] = co; s64[d.ang_off + a] = si;
            c32[d.ang_off + a] = (float)co; s32[d.ang_off + a] = (float)si;
            host_angles[(size_t)d.ang_off + a] = ang[a];
        }
        d.delta = (precision == MM_PRECISION_F32)
                      ? screen_delta(rx, ry, d.n_ref, tx, ty, d.n_tgt, d.cx, d.cy) : 0.0;
    }
    std::memcpy(h + L.pairs, host_pairs.data(), (size_t)P * sizeof(PairDesc));
    if (W) std::memcpy(h + L.work, host_work.data(), (size_t)W * sizeof(WorkItem));

    // ---- device blob ------------------------------------------------------------------------
    if (transient) {
        rc = e->ensure_dev(L.total);
        if (rc) return rc;
        blob = (unsigned char*)e->dev_buf;
    } else {
        MM_HIP(hipMalloc((void**)&blob, L.total));
    }
    MM_HIP(hipMemcpyAsync(blob, h, L.in_bytes, hipMemcpyHostToDevice, e->stream));
    if (!transient) MM_HIP(hipStreamSynchronize(e->stream));  // host staging buffer is reused

    dev.pairs = (const PairDesc*)(blob + L.pairs);
    dev.work = (const WorkItem*)(blob + L.work);
    dev.n_pairs = P; dev.n_work = W;
    dev.cos32 = (const float*)(blob + L.cos32); dev.sin32 = (const float*)(blob + L.sin32);
    dev.cos64 = (const double*)(blob + L.cos64); dev.sin64 = (const double*)(blob + L.sin64);
    dev.ref32x = (const float*)(blob + L.r32x); dev.ref32y = (const float*)(blob + L.r32y);
    dev.tgt32x = (const float*)(blob + L.t32x); dev.tgt32y = (const float*)(blob + L.t32y);
    dev.ref64x = (const double*)(blob + L.r64x); dev.ref64y = (const double*)(blob + L.r64y);
    dev.tgt64x = (const double*)(blob + L.t64x); dev.tgt64y = (const double*)(blob + L.t64y);
    dev.sq32 = (float*)(blob + L.sq32); dev.sq64 = (double*)(blob + L.sq64);
    dev.flag = (uint8_t*)(blob + L.flag);
    dev.items = (WorkItem*)(blob + L.items); dev.n_items = (int32_t*)(blob + L.n_items);
    dev.best_cost = (double*)(blob + L.best_cost); dev.best_idx = (int32_t*)(blob + L.best_idx);
    dev.n_rescored = (int32_t*)(blob + L.n_rescored);
    dev.all_costs = want_costs ? (double*)(blob + L.all_costs) : nullptr;
    off_best_cost = L.best_cost; off_best_idx = L.best_idx; off_n_rescored = L.n_rescored;
    off_all_costs = L.all_costs;
    return MM_OK;
}

int Plan::run(bool screen_only)
{
    hipStream_t s = eng->stream;
    if (W == 0) {
        if (!screen_only && P > 0) { hipError_t e = launch_finalize(dev, 0, s); if (e != hipSuccess) return hip_error(e, "finalize"); }
        return MM_OK;
    }
    hipError_t e;
    int prc;
    if (precision == MM_PRECISION_F32) {
        if ((prc = eng->profile_begin())) return prc;
        e = launch_screen_f32(dev, max_na, max_nbp, s);
        if (e != hipSuccess) return hip_error(e, "screen kernel launch");
        if ((prc = eng->profile_end(pair_evals, A))) return prc;
        if (screen_only) return MM_OK;
        MM_HIP(hipMemsetAsync(dev.n_items, 0, 16, s));
        e = launch_shortlist(dev, s);
        if (e != hipSuccess) return hip_error(e, "shortlist kernel launch");
        e = launch_rescore(dev, max_na, max_nbp, (int)A, s);
        if (e != hipSuccess) return hip_error(e, "rescore kernel launch");
        e = launch_finalize(dev, 1, s);
        if (e != hipSuccess) return hip_error(e, "finalize kernel launch");
    } else {
        if ((prc = eng->profile_begin())) return prc;
        e = launch_exact_all(dev, max_na, max_nbp, s);
        if (e != hipSuccess) return hip_error(e, "exact kernel launch");
        if ((prc = eng->profile_end(pair_evals, A))) return prc;
        if (screen_only) return MM_OK;
        e = launch_finalize(dev, 0, s);
        if (e != hipSuccess) return hip_error(e, "finalize kernel launch");
    }
    return MM_OK;
}

int Plan::fetch(int32_t* best_idx, double* best_angle, double* best_cost, int32_t* n_rescored, double* all_costs)
{
    // results are contiguous: best_cost | best_idx | n_rescored (+ all_costs)
    const size_t res_bytes = (off_n_rescored + align_up((size_t)P * 4)) - off_best_cost;
    const size_t cost_bytes = (all_costs && dev.all_costs) ? (size_t)A * 8 : 0;
    int rc = eng->ensure_host(std::max(in_bytes, res_bytes + cost_bytes + 256));
    if (rc) return rc;
    unsigned char* h = (unsigned char*)eng->host_buf;
    if (P > 0) MM_HIP(hipMemcpyAsync(h, blob + off_best_cost, res_bytes, hipMemcpyDeviceToHost, eng->stream));
    if (cost_bytes) MM_HIP(hipMemcpyAsync(h + align_up(res_bytes), dev.all_costs, cost_bytes, hipMemcpyDeviceToHost, eng->stream));
    MM_HIP(hipStreamSynchronize(eng->stream));
    const double* hc = (const double*)h;
    const int32_t* hi = (const int32_t*)(h + (off_best_idx - off_best_cost));
    const int32_t* hn = (const int32_t*)(h + (off_n_rescored - off_best_cost));
    const double* hall = (const double*)(h + align_up(res_bytes));
    for (int p = 0; p < P; ++p) {
        const PairDesc& d = host_pairs[p];
        int32_t idx; double cost; int32_t nres;
        if (trivial[p]) {
            // every candidate costs 0.0; the ordered first-minimum is candidate ang_begin
            // (if this plan's slice is non-empty)
            const bool any = !std::isnan(first_angle[p]);
            idx = any ? d.ang_begin : -1; cost = any ? 0.0 : INFINITY; nres = 0;
        } else {
            idx = hi[p]; cost = hc[p]; nres = hn[p];
        }
        if (best_idx) best_idx[p] = idx;
        if (best_cost) best_cost[p] = cost;
        if (n_rescored) n_rescored[p] = nres;
        if (best_angle) {
            if (idx < 0) best_angle[p] = NAN;
            else if (trivial[p]) best_angle[p] = first_angle[p];
            else best_angle[p] = host_angles[(size_t)d.ang_off + (idx - d.ang_begin)];
        }
    }
    if (all_costs && dev.all_costs) {
        // scatter from the plan's candidate order to the caller's ang_off indexing
        for (int p = 0; p < P; ++p) {
            const PairDesc& d = host_pairs[p];
            double* dst = all_costs + user_ang_off[p] + d.ang_begin;
            if (trivial[p]) {
                const int64_t n = std::max<int64_t>(0, std::min<int64_t>(slice_end, d.ang_full) - d.ang_begin);
                for (int64_t a = 0; a < n; ++a) dst[a] = 0.0;
            } else if (d.n_ang > 0) {
                std::memcpy(dst, hall + d.ang_off, (size_t)d.n_ang * 8);
            }
        }
    }
    return MM_OK;
}

Plan::~Plan()
{
    if (blob && !transient) (void)hipFree(blob);
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
    if (stream) { en->stream = (hipStream_t)stream; en->own_stream = false; }
    else {
        hipError_t e2 = hipStreamCreateWithFlags(&en->stream, hipStreamNonBlocking);
        if (e2 != hipSuccess) { delete en; return hip_error(e2, "hipStreamCreate"); }
        en->own_stream = true;
    }
    *out = reinterpret_cast<mm_engine*>(en);
    return MM_OK;
}

void mm_engine_destroy(mm_engine* h)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (e->host_buf) (void)hipHostFree(e->host_buf);
    if (e->dev_buf) (void)hipFree(e->dev_buf);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int mm_engine_synchronize(mm_engine* h)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    MM_HIP(hipStreamSynchronize(e->stream));
    return MM_OK;
}

int mm_engine_profile(mm_engine* h, int enable)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    MM_HIP(hipStreamSynchronize(e->stream));
    e->profile = enable != 0;
    e->launches = 0; e->prof_pair_evals = 0.0; e->prof_candidates = 0;
    return MM_OK;
}

int mm_engine_profile_read(mm_engine* h, int64_t* n_launches, double* ms_total, double* pair_evals,
                           int64_t* candidates)
{
    Engine* e = reinterpret_cast<Engine*>(h);
    if (!e) return set_error(MM_ERR_INVALID, "engine == NULL");
    MM_HIP(hipStreamSynchronize(e->stream));
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
    e->launches = 0; e->prof_pair_evals = 0.0; e->prof_candidates = 0;
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
    Plan plan;
    int rc = plan.build(e, n_pairs, ref_off, ref_x, ref_y, tgt_off, tgt_x, tgt_y, ang_off, angles, cx, cy, flags,
                        precision, 0, INT32_MAX, all_costs != nullptr, /*transient=*/true);
    if (rc) return rc;
    rc = plan.run(false);
    if (rc) return rc;
    rc = plan.fetch(best_idx, best_angle, best_cost, n_rescored, all_costs);
    if (rc) return rc;
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

int mm_hausdorff_2d(mm_engine* h, const double* ax, const double* ay, int na,
                    const double* bx, const double* by, int nb, double* out)
{
    if (!out) return set_error(MM_ERR_INVALID, "out == NULL");
    if (na < 0 || nb < 0) return set_error(MM_ERR_INVALID, "negative set size");
    if (na == 0 || nb == 0) {  // process_utils.rs:86-88
        if (!h) return set_error(MM_ERR_INVALID, "engine == NULL");
        *out = 0.0;
        return MM_OK;
    }
    const double zero = 0.0;
    double cost = NAN;
    // angle 0 with the rotate() shortcut leaves the target untouched -> plain hausdorff_distance
    int rc = mm_best_rotation(h, ax, ay, na, bx, by, nb, 0.0, 0.0, &zero, 1, MM_SEARCH_SKIP_ZERO,
                              MM_PRECISION_F64, nullptr, &cost, nullptr, nullptr);
    if (rc) return rc;
    *out = cost;
    return MM_OK;
}

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
    Plan* p = new Plan();
    int rc = p->build(e, n_pairs, ref_off, ref_x, ref_y, tgt_off, tgt_x, tgt_y, ang_off, angles, cx, cy, flags,
                      precision, angle_begin, angle_end, /*want_costs=*/true, /*transient=*/false);
    if (rc) { delete p; return rc; }
    *out = reinterpret_cast<mm_plan*>(p);
    return MM_OK;
}

void mm_plan_destroy(mm_plan* h)
{
    Plan* p = reinterpret_cast<Plan*>(h);
    if (!p) return;
    (void)hipSetDevice(p->eng->device);
    (void)hipStreamSynchronize(p->eng->stream);
    delete p;
}

int mm_plan_run(mm_plan* h)
{
    Plan* p = reinterpret_cast<Plan*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    return p->run(false);
}

int mm_plan_run_screen_only(mm_plan* h)
{
    Plan* p = reinterpret_cast<Plan*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    return p->run(true);
}

int mm_plan_fetch(mm_plan* h, int32_t* best_idx, double* best_angle, double* best_cost,
                  int32_t* n_rescored, double* all_costs)
{
    Plan* p = reinterpret_cast<Plan*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    return p->fetch(best_idx, best_angle, best_cost, n_rescored, all_costs);
}

int mm_plan_result_dev(mm_plan* h, void** best_cost_dev, void** best_idx_dev)
{
    Plan* p = reinterpret_cast<Plan*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    if (best_cost_dev) *best_cost_dev = p->dev.best_cost;
    if (best_idx_dev) *best_idx_dev = p->dev.best_idx;
    return MM_OK;
}

int mm_plan_time(mm_plan* h, int iters, int screen_only, float* ms_avg)
{
    Plan* p = reinterpret_cast<Plan*>(h);
    if (!p || !ms_avg || iters <= 0) return set_error(MM_ERR_INVALID, "mm_plan_time: bad arguments");
    hipEvent_t t0, t1;
    MM_HIP(hipEventCreate(&t0));
    MM_HIP(hipEventCreate(&t1));
    int rc = p->run(screen_only != 0);  // warm-up
    if (rc) return rc;
    MM_HIP(hipEventRecord(t0, p->eng->stream));
    for (int i = 0; i < iters; ++i) {
        rc = p->run(screen_only != 0);
        if (rc) return rc;
    }
    MM_HIP(hipEventRecord(t1, p->eng->stream));
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
    Plan* p = reinterpret_cast<Plan*>(h);
    if (!p) return set_error(MM_ERR_INVALID, "plan == NULL");
    if (n_candidates) *n_candidates = p->A;
    if (pair_evals) *pair_evals = p->pair_evals;
    if (hbm_bytes) *hbm_bytes = (int64_t)p->total_bytes;
    return MM_OK;
}

}  // extern "C"
