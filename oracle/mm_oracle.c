/*
 * mm_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see mm_oracle.h).
 * f64 restatement of the reference's Hausdorff pose search.  Every function cites the
 * reference lines it follows (paths relative to the reference checkout).
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (see oracle/Makefile).
 */
#define _GNU_SOURCE /* sincos */
#include "mm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846264338327950288 /* std::f64::consts::PI */

/* f64::to_radians / to_degrees: one multiply by a pre-folded constant. */
static double to_radians(double deg) { return deg * (ORC_PI / 180.0); }
static double to_degrees(double rad) { return rad * (180.0 / ORC_PI); }

/* Wherever the reference evaluates `angle.cos()` and `angle.sin()` of the same value in one
 * function (contour_point.rs:45-46, frame.rs:58-59, align_between.rs:96-97,197-198), LLVM on
 * x86_64-unknown-linux-gnu lowers the pair to ONE `sincos` libcall.  glibc's sincos is not
 * bit-identical to separate sin()/cos() for every argument (1 ulp apart for some |x| > 2.4, seen
 * at x = -23.17, -32.35, -67.66 ...), so the pair is requested explicitly here instead of being
 * left to whatever this compiler's optimiser does with two calls. */
static void sin_cos(double x, double* s, double* c) { sincos(x, s, c); }

/* f64::rem_euclid */
static double rem_euclid(double a, double m)
{
    double r = fmod(a, m);
    return (r < 0.0) ? r + fabs(m) : r;
}

/* ------------------------------------------------------------------------------------
 * process_utils.rs:84-121 directed_hausdorff.  The rayon chunking (lines 90-98) only
 * partitions the outer loop; max over chunks is exact, so a sequential loop is
 * bit-identical.
 * ---------------------------------------------------------------------------------- */
double orc_directed_hausdorff(const orc_point* a, size_t na, const orc_point* b, size_t nb)
{
    if (na == 0 || nb == 0) return 0.0;                       /* :86-88 */
    double max_sq = 0.0;                                      /* :100, :118 */
    for (size_t i = 0; i < na; ++i) {
        double min_sq = INFINITY;                             /* :103 */
        const double pax = a[i].x, pay = a[i].y;
        for (size_t j = 0; j < nb; ++j) {
            double dx = pax - b[j].x;                         /* :105 */
            double dy = pay - b[j].y;                         /* :106 */
            double d2 = dx * dx + dy * dy;                    /* :107 (no fma) */
            if (d2 < min_sq) min_sq = d2;                     /* :108-110 */
        }
        if (isfinite(min_sq) && min_sq > max_sq) max_sq = min_sq; /* :112-114 */
    }
    return sqrt(max_sq);                                      /* :120 */
}

/* process_utils.rs:78-82 */
double orc_hausdorff(const orc_point* s1, size_t n1, const orc_point* s2, size_t n2)
{
    double forward = orc_directed_hausdorff(s1, n1, s2, n2);
    double backward = orc_directed_hausdorff(s2, n2, s1, n1);
    return (forward > backward) ? forward : backward;        /* f64::max, no NaNs here */
}

double orc_hausdorff_xy(const double* ax, const double* ay, size_t na,
                        const double* bx, const double* by, size_t nb)
{
    orc_point* a = (orc_point*)malloc((na + 1) * sizeof(orc_point));
    orc_point* b = (orc_point*)malloc((nb + 1) * sizeof(orc_point));
    for (size_t i = 0; i < na; ++i) { a[i].x = ax[i]; a[i].y = ay[i]; a[i].z = 0.0; }
    for (size_t i = 0; i < nb; ++i) { b[i].x = bx[i]; b[i].y = by[i]; b[i].z = 0.0; }
    double h = orc_hausdorff(a, na, b, nb);
    free(a); free(b);
    return h;
}

/* ------------------------------------------------------------------------------------
 * process_utils.rs:43-67 candidate enumeration.
 * ---------------------------------------------------------------------------------- */
size_t orc_search_angles(double step_deg, double range_deg, int has_center, double center_in,
                         double limes_deg, double* out, size_t cap,
                         int* degenerate, double* early_value)
{
    double range_rad = to_radians(range_deg);                 /* :43 */
    double step_rad = to_radians(step_deg);                   /* :44 */
    *degenerate = 0;
    *early_value = 0.0;
    if (step_rad <= 0.0) {                                    /* :47-49 */
        *degenerate = 1;
        *early_value = has_center ? center_in : 0.0;
        return 0;
    }
    double center = has_center ? center_in : 0.0;             /* :51 */
    double limes = to_radians(limes_deg);                     /* :52 */
    double start_angle = fmax(center - range_rad, -limes);    /* :54 */
    double stop_angle = fmin(center + range_rad, limes);      /* :55 */
    if (stop_angle <= start_angle) {                          /* :57-59 */
        *degenerate = 1;
        *early_value = center;
        return 0;
    }
    /* :61  (x.ceil() as usize).max(1); `as usize` saturates, values here are small */
    double steps_f = ceil((stop_angle - start_angle) / step_rad);
    size_t steps = (steps_f <= 0.0) ? 0 : (size_t)steps_f;
    if (steps < 1) steps = 1;
    size_t n = 0;
    for (size_t i = 0; i <= steps; ++i) {                     /* :63-67 */
        double a = start_angle + (double)i * step_rad;
        if (!(a <= stop_angle)) break;                        /* take_while */
        double w = rem_euclid(a + ORC_PI, 2.0 * ORC_PI) - ORC_PI;
        if (n < cap) out[n] = w;
        ++n;
    }
    return n;
}

/* process_utils.rs:33-75 */
double orc_search_range(orc_cost_fn f, void* ctx, double step_deg, double range_deg,
                        int has_center, double center, double limes_deg, int n_threads)
{
    int degenerate; double early;
    size_t n = orc_search_angles(step_deg, range_deg, has_center, center, limes_deg,
                                 NULL, 0, &degenerate, &early);
    if (degenerate) return early;
    double* angles = (double*)malloc(n * sizeof(double));
    double* costs = (double*)malloc(n * sizeof(double));
    orc_search_angles(step_deg, range_deg, has_center, center, limes_deg, angles, n,
                      &degenerate, &early);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (long i = 0; i < (long)n; ++i) costs[i] = f(angles[i], ctx);   /* :69-71 */
    /* :72  reduce_with(|a, b| if b.1 < a.1 { b } else { a }) -- ordered, first minimum */
    size_t best = 0;
    for (size_t i = 1; i < n; ++i)
        if (costs[i] < costs[best]) best = i;
    double r = (n > 0) ? angles[best] : (has_center ? center : 0.0);    /* :73-74 */
    free(angles); free(costs);
    return r;
}

/* ------------------------------------------------------------------------------------
 * contour_point.rs:38-52 rotate
 * ---------------------------------------------------------------------------------- */
orc_point orc_rotate_point(orc_point p, double angle, double cx, double cy)
{
    if (angle == 0.0) return p;                               /* :39-41 */
    double x = p.x - cx;                                      /* :43 */
    double y = p.y - cy;                                      /* :44 */
    double cos_a, sin_a;
    sin_cos(angle, &sin_a, &cos_a);                           /* :45-46 */
    orc_point r = p;
    r.x = x * cos_a - y * sin_a + cx;                         /* :48 */
    r.y = x * sin_a + y * cos_a + cy;                         /* :49 */
    return r;
}

/* contour.rs:47-58 downsample_contour_points */
size_t orc_downsample(const orc_point* pts, size_t len, size_t n, orc_point* out)
{
    if (len <= n) {                                           /* :48-50 */
        memcpy(out, pts, len * sizeof(orc_point));
        return len;
    }
    double step = (double)len / (double)n;                    /* :51 */
    for (size_t i = 0; i < n; ++i) {
        size_t index = (size_t)((double)i * step);            /* :54 `as usize` truncates */
        out[i] = pts[index];
    }
    return n;
}

/* align_within.rs:100-104 / 200-206 */
double orc_cost_within(const orc_point* ref, size_t nr, const orc_point* tgt, size_t nt,
                       double angle, double cx, double cy)
{
    orc_point* rot = (orc_point*)malloc((nt + 1) * sizeof(orc_point));
    for (size_t i = 0; i < nt; ++i) rot[i] = orc_rotate_point(tgt[i], angle, cx, cy);
    double h = orc_hausdorff(ref, nr, rot, nt);
    free(rot);
    return h;
}

/* align_between.rs:189-216 (no angle==0 shortcut, sin/cos per point -- same values) */
double orc_cost_between(const orc_point* ref, size_t nr, const orc_point* tgt, size_t nt,
                        double angle, double cx, double cy)
{
    orc_point* rot = (orc_point*)malloc((nt + 1) * sizeof(orc_point));
    double cos_angle, sin_angle;
    sin_cos(angle, &sin_angle, &cos_angle);                   /* :197-198 */
    for (size_t i = 0; i < nt; ++i) {
        double translated_x = tgt[i].x - cx;                  /* :194 */
        double translated_y = tgt[i].y - cy;                  /* :195 */
        double rotated_x = translated_x * cos_angle - translated_y * sin_angle; /* :200 */
        double rotated_y = translated_x * sin_angle + translated_y * cos_angle; /* :201 */
        rot[i] = tgt[i];
        rot[i].x = rotated_x + cx;                            /* :205 */
        rot[i].y = rotated_y + cy;                            /* :206 */
    }
    double h = orc_hausdorff(ref, nr, rot, nt);               /* :215 */
    free(rot);
    return h;
}

typedef struct {
    const orc_point* ref; size_t nr;
    const orc_point* tgt; size_t nt;
    double cx, cy;
    int between;
} cost_ctx;

static double cost_trampoline(double angle, void* p)
{
    const cost_ctx* c = (const cost_ctx*)p;
    return c->between ? orc_cost_between(c->ref, c->nr, c->tgt, c->nt, angle, c->cx, c->cy)
                      : orc_cost_within(c->ref, c->nr, c->tgt, c->nt, angle, c->cx, c->cy);
}

/* align_within.rs:208-246 / align_between.rs:219-257 -- identical match ladders.
 * Rust range patterns: 1.0..=INF, 0.1..1.0, 0.01..0.1 are half-open on the right. */
static double hierarchical(orc_cost_fn f, void* ctx, double step_deg, double range_deg, int nt)
{
    if (step_deg >= 1.0 && step_deg <= INFINITY) {
        return orc_search_range(f, ctx, step_deg, range_deg, 0, 0.0, range_deg, nt);
    } else if (step_deg >= 0.1 && step_deg < 1.0) {
        double coarse = orc_search_range(f, ctx, 1.0, range_deg, 0, 0.0, range_deg, nt);
        double range = (range_deg > 5.0) ? 5.0 : range_deg;
        return orc_search_range(f, ctx, step_deg, range, 1, coarse, range_deg, nt);
    } else if (step_deg >= 0.01 && step_deg < 0.1) {
        double coarse = orc_search_range(f, ctx, 1.0, range_deg, 0, 0.0, range_deg, nt);
        double range = (range_deg > 5.0) ? 5.0 : range_deg;
        double medium = orc_search_range(f, ctx, 0.1, range, 1, coarse, range_deg, nt);
        double range_small = (range_deg > 10.0 * step_deg) ? 10.0 * step_deg : range_deg;
        return orc_search_range(f, ctx, step_deg, range_small, 1, medium, range_deg, nt);
    } else {
        double coarse = orc_search_range(f, ctx, 1.0, range_deg, 0, 0.0, range_deg, nt);
        double range = (range_deg > 5.0) ? 5.0 : range_deg;
        double medium = orc_search_range(f, ctx, 0.1, range, 1, coarse, range_deg, nt);
        double range_small = (range_deg > 0.1) ? 0.1 : range_deg;
        double fine = orc_search_range(f, ctx, 0.01, range_small, 1, medium, range_deg, nt);
        double range_fine = (range_deg > 10.0 * step_deg) ? 10.0 * step_deg : range_deg;
        return orc_search_range(f, ctx, step_deg, range_fine, 1, fine, range_deg, nt);
    }
}

double orc_find_best_rotation(const orc_point* ref, size_t nr, const orc_point* tgt, size_t nt,
                              double step_deg, double range_deg, double cx, double cy,
                              int between, int n_threads)
{
    cost_ctx c = { ref, nr, tgt, nt, cx, cy, between };
    return hierarchical(cost_trampoline, &c, step_deg, range_deg, n_threads);
}

/* align_within.rs:97-110 */
double orc_bruteforce_rotation(const orc_point* ref, size_t nr, const orc_point* tgt, size_t nt,
                               double step_deg, double range_deg, double cx, double cy,
                               int n_threads)
{
    cost_ctx c = { ref, nr, tgt, nt, cx, cy, 0 };
    return orc_search_range(cost_trampoline, &c, step_deg, range_deg, 0, 0.0, range_deg,
                            n_threads);
}

static double count_cost(double a, void* p) { (void)a; ++*(size_t*)p; return 0.0; }

size_t orc_count_evals(double step_deg, double range_deg, int bruteforce)
{
    size_t n = 0;
    if (bruteforce)
        orc_search_range(count_cost, &n, step_deg, range_deg, 0, 0.0, range_deg, 1);
    else
        hierarchical(count_cost, &n, step_deg, range_deg, 1);
    return n;
}

/* ------------------------------------------------------------------------------------
 * frame.rs:17-64
 * ---------------------------------------------------------------------------------- */
static void translate_span(orc_point* p, int64_t lo, int64_t hi, double dx, double dy, double dz)
{
    for (int64_t k = lo; k < hi; ++k) { p[k].x += dx; p[k].y += dy; p[k].z += dz; } /* contour_point.rs:29-36 */
}

static void rotate_span(orc_point* p, int64_t lo, int64_t hi, double angle, double cx, double cy)
{
    for (int64_t k = lo; k < hi; ++k) p[k] = orc_rotate_point(p[k], angle, cx, cy);
}

void orc_frame_translate(orc_geometry* g, int32_t i, double dx, double dy, double dz)
{
    translate_span(g->lumen, g->lumen_off[i], g->lumen_off[i + 1], dx, dy, dz);     /* :19 */
    if (g->cath_off) translate_span(g->cath, g->cath_off[i], g->cath_off[i + 1], dx, dy, dz);
    if (g->extra_off) translate_span(g->extra, g->extra_off[i], g->extra_off[i + 1], dx, dy, dz);
    if (g->has_ref && g->has_ref[i]) {                                              /* :33 */
        g->ref[i].x += dx; g->ref[i].y += dy; g->ref[i].z += dz;
    }
    g->centroid[3 * i + 0] += dx;                                                   /* :35-37 */
    g->centroid[3 * i + 1] += dy;
    g->centroid[3 * i + 2] += dz;
    if (g->lumen_centroid && g->lumen_off[i + 1] > g->lumen_off[i]) {               /* :20 self.lumen.compute_centroid() */
        double sx = 0.0, sy = 0.0, sz = 0.0;                                        /* contour.rs:219-223: sequential fold */
        for (int64_t k = g->lumen_off[i]; k < g->lumen_off[i + 1]; ++k) { sx += g->lumen[k].x; sy += g->lumen[k].y; sz += g->lumen[k].z; }
        const double n = (double)(g->lumen_off[i + 1] - g->lumen_off[i]);
        g->lumen_centroid[3 * i] = sx / n; g->lumen_centroid[3 * i + 1] = sy / n; g->lumen_centroid[3 * i + 2] = sz / n;
    }
}

void orc_frame_rotate(orc_geometry* g, int32_t i, double angle, double cx, double cy)
{
    if (angle == 0.0) return;                                                       /* :41-43 */
    rotate_span(g->lumen, g->lumen_off[i], g->lumen_off[i + 1], angle, cx, cy);     /* :45 */
    if (g->cath_off) rotate_span(g->cath, g->cath_off[i], g->cath_off[i + 1], angle, cx, cy);
    if (g->extra_off) rotate_span(g->extra, g->extra_off[i], g->extra_off[i + 1], angle, cx, cy);
    if (g->has_ref && g->has_ref[i]) g->ref[i] = orc_rotate_point(g->ref[i], angle, cx, cy); /* :53 */
    double x = g->centroid[3 * i + 0] - cx;                                         /* :56-57 */
    double y = g->centroid[3 * i + 1] - cy;
    double cos_a, sin_a;
    sin_cos(angle, &sin_a, &cos_a);                                                 /* :58-59 */
    g->centroid[3 * i + 0] = x * cos_a - y * sin_a + cx;                            /* :60 */
    g->centroid[3 * i + 1] = x * sin_a + y * cos_a + cy;                            /* :61 */
}

/* align_within.rs:173-191 */
size_t orc_catheter_lumen_vec(const orc_geometry* g, int32_t i, size_t sample_size_lumen,
                              int has_sc, size_t sample_size_catheter, orc_point* out)
{
    size_t n = orc_downsample(g->lumen + g->lumen_off[i],
                              (size_t)(g->lumen_off[i + 1] - g->lumen_off[i]),
                              sample_size_lumen, out);
    if (has_sc && g->cath_off) {
        size_t len = (size_t)(g->cath_off[i + 1] - g->cath_off[i]);
        n += orc_downsample(g->cath + g->cath_off[i], len, sample_size_catheter, out + n);
    }
    return n;
}

/* geometry.rs:42-60 */
size_t orc_find_proximal_end_idx(const orc_geometry* g)
{
    int32_t n = g->n_frames;
    if (n == 0) return 0;
    if (n == 1) return (size_t)g->lumen_id[0];
    uint32_t idx = (g->orig_frame[0] > g->orig_frame[n - 1]) ? g->lumen_id[0] : g->lumen_id[n - 1];
    return (size_t)idx;
}

/* geometry.rs:62-69 : returns Frame.id of the first frame with a reference point */
int orc_find_ref_frame_idx(const orc_geometry* g, size_t* out)
{
    for (int32_t i = 0; i < g->n_frames; ++i)
        if (g->has_ref && g->has_ref[i]) { *out = (size_t)g->id[i]; return 0; }
    return -1;
}

/* align_within.rs:24-134 */
int orc_align_within_chain(orc_geometry* g, double step_deg, double range_deg,
                           int bruteforce, size_t sample_size, orc_alignlog* logs,
                           int n_threads)
{
    if (g->n_frames <= 0) return -1;                                                 /* :32-34 */
    size_t len0 = (size_t)(g->lumen_off[1] - g->lumen_off[0]);
    if (len0 == 0) return -2;                                                        /* :35-37 */
    if (sample_size == 0) return -3;                                                 /* :38-40 */

    double sample_ratio = (double)sample_size / (double)len0;                        /* :45 */
    int has_sc = 0; size_t sample_size_catheter = 0;
    if (g->has_catheter && g->cath_off) {                                            /* :46-59 */
        size_t clen0 = (size_t)(g->cath_off[1] - g->cath_off[0]);
        has_sc = 1;
        sample_size_catheter = (size_t)ceil((double)clen0 * sample_ratio);
    }

    size_t max_pts = 0;
    for (int32_t i = 0; i < g->n_frames; ++i) {
        size_t l = (size_t)(g->lumen_off[i + 1] - g->lumen_off[i]);
        size_t c = g->cath_off ? (size_t)(g->cath_off[i + 1] - g->cath_off[i]) : 0;
        if (l + c > max_pts) max_pts = l + c;
    }
    orc_point* testing = (orc_point*)malloc((max_pts + 1) * sizeof(orc_point));
    orc_point* reference = (orc_point*)malloc((max_pts + 1) * sizeof(orc_point));

    double cumulative_rotation = 0.0;                                                /* :70 */
    for (int32_t i = 1; i < g->n_frames; ++i) {                                      /* :72 */
        /* prev_frame = frames[i-1].clone(): frame i-1 is not modified below */
        const double pcx = g->centroid[3 * (i - 1) + 0];
        const double pcy = g->centroid[3 * (i - 1) + 1];

        if (cumulative_rotation != 0.0) {                                            /* :79-82 */
            orc_frame_rotate(g, i, cumulative_rotation,
                             g->centroid[3 * i + 0], g->centroid[3 * i + 1]);
        }
        double tx = pcx - g->centroid[3 * i + 0];                                    /* :84-88 */
        double ty = pcy - g->centroid[3 * i + 1];
        orc_frame_translate(g, i, tx, ty, 0.0);                                      /* :90 */

        size_t nt = orc_catheter_lumen_vec(g, i, sample_size, has_sc, sample_size_catheter, testing);       /* :92-93 */
        size_t nr = orc_catheter_lumen_vec(g, i - 1, sample_size, has_sc, sample_size_catheter, reference); /* :94-95 */

        double ccx = g->centroid[3 * i + 0], ccy = g->centroid[3 * i + 1];
        double best_rotation;
        if (bruteforce)                                                              /* :97-110 */
            best_rotation = orc_bruteforce_rotation(reference, nr, testing, nt, step_deg,
                                                    range_deg, ccx, ccy, n_threads);
        else                                                                         /* :112-118 */
            best_rotation = orc_find_best_rotation(reference, nr, testing, nt, step_deg,
                                                   range_deg, ccx, ccy, 0, n_threads);

        orc_frame_rotate(g, i, best_rotation, ccx, ccy);                             /* :121-122 */
        cumulative_rotation += best_rotation;                                        /* :123 */

        if (logs) {                                                                  /* :125-133 */
            orc_alignlog* l = &logs[i - 1];
            l->contour_id = g->id[i];
            l->matched_to = g->id[i - 1];
            l->rot_deg = to_degrees(best_rotation);
            l->tx = tx; l->ty = ty;
            l->cx = g->centroid[3 * i + 0];
            l->cy = g->centroid[3 * i + 1];
        }
    }
    free(testing); free(reference);
    return 0;
}

/* align_between.rs:154-178 */
size_t orc_extract_between_points(const orc_geometry* g, size_t sample_size,
                                  orc_point* out, size_t cap)
{
    size_t total_points = (size_t)(g->lumen_off[g->n_frames] - g->lumen_off[0]);     /* :158 */
    double sample_ratio = (double)sample_size / (double)total_points;                /* :160 */
    size_t n = 0;
    for (int32_t i = 0; i < g->n_frames; ++i) {
        size_t len = (size_t)(g->lumen_off[i + 1] - g->lumen_off[i]);
        size_t fs = (size_t)ceil((double)len * sample_ratio);                        /* :164 */
        if (fs < 1) fs = 1;                                                          /* :166 */
        size_t take = (len <= fs) ? len : fs;
        if (out && n + take <= cap) orc_downsample(g->lumen + g->lumen_off[i], len, fs, out + n);
        n += take;
    }
    return n;
}

/* align_between.rs:95-145 rotate_geometry_around_point */
static void rotate_geometry_around_point(orc_geometry* g, double angle_rad, double cx, double cy)
{
    double cos_angle, sin_angle;
    sin_cos(angle_rad, &sin_angle, &cos_angle);               /* :96-97 */
#define ROT_PT(X, Y) do { \
        double tx_ = (X) - cx, ty_ = (Y) - cy; \
        double rx_ = tx_ * cos_angle - ty_ * sin_angle; \
        double ry_ = tx_ * sin_angle + ty_ * cos_angle; \
        (X) = rx_ + cx; (Y) = ry_ + cy; } while (0)
    for (int32_t i = 0; i < g->n_frames; ++i) {
        for (int64_t k = g->lumen_off[i]; k < g->lumen_off[i + 1]; ++k) ROT_PT(g->lumen[k].x, g->lumen[k].y);
        ROT_PT(g->centroid[3 * i + 0], g->centroid[3 * i + 1]);
        if (g->cath_off)
            for (int64_t k = g->cath_off[i]; k < g->cath_off[i + 1]; ++k) ROT_PT(g->cath[k].x, g->cath[k].y);
        if (g->extra_off)
            for (int64_t k = g->extra_off[i]; k < g->extra_off[i + 1]; ++k) ROT_PT(g->extra[k].x, g->extra[k].y);
        if (g->has_ref && g->has_ref[i]) ROT_PT(g->ref[i].x, g->ref[i].y);
    }
#undef ROT_PT
}

static size_t ref_or_proximal(const orc_geometry* g)
{
    size_t idx;
    if (orc_find_ref_frame_idx(g, &idx) == 0) return idx;      /* unwrap_or(find_proximal_end_idx()) */
    return orc_find_proximal_end_idx(g);
}

/* align_between.rs:11-68 */
int orc_align_between(orc_geometry* a, orc_geometry* b, double rot_deg, double step_rot_deg,
                      size_t sample_size, double* best_rotation_out, int n_threads)
{
    size_t ia = ref_or_proximal(a), ib = ref_or_proximal(b);                         /* :19-24 */
    if (ia >= (size_t)a->n_frames || ib >= (size_t)b->n_frames) return -1;           /* Rust would panic */
    double ac[3] = { a->centroid[3 * ia], a->centroid[3 * ia + 1], a->centroid[3 * ia + 2] };
    double bc[3] = { b->centroid[3 * ib], b->centroid[3 * ib + 1], b->centroid[3 * ib + 2] };
    double it[3] = { ac[0] - bc[0], ac[1] - bc[1], ac[2] - bc[2] };                  /* :33-37 */
    for (int32_t i = 0; i < b->n_frames; ++i) orc_frame_translate(b, i, it[0], it[1], it[2]); /* :40 */

    size_t s = (sample_size > 500) ? sample_size : 500;                              /* :43-44 */
    size_t na = orc_extract_between_points(a, s, NULL, 0);
    size_t nb = orc_extract_between_points(b, s, NULL, 0);
    orc_point* pa = (orc_point*)malloc((na + 1) * sizeof(orc_point));
    orc_point* pb = (orc_point*)malloc((nb + 1) * sizeof(orc_point));
    orc_extract_between_points(a, s, pa, na);
    orc_extract_between_points(b, s, pb, nb);

    /* :260-271 calculate_global_centroid(reference) */
    double gx = 0.0, gy = 0.0;
    if (na > 0) {
        double sx = 0.0, sy = 0.0;
        for (size_t k = 0; k < na; ++k) sx += pa[k].x;
        for (size_t k = 0; k < na; ++k) sy += pa[k].y;
        gx = sx / (double)na; gy = sy / (double)na;
    }
    double best = orc_find_best_rotation(pa, na, pb, nb, step_rot_deg, rot_deg, gx, gy, 1, n_threads); /* :46-47 */
    free(pa); free(pb);

    rotate_geometry_around_point(b, best, ac[0], ac[1]);                             /* :50 */

    ia = ref_or_proximal(a); ib = ref_or_proximal(b);                                /* :53-58 */
    double ft[3] = { a->centroid[3 * ia] - b->centroid[3 * ib],                      /* :60-66 */
                     a->centroid[3 * ia + 1] - b->centroid[3 * ib + 1],
                     a->centroid[3 * ia + 2] - b->centroid[3 * ib + 2] };
    for (int32_t i = 0; i < b->n_frames; ++i) orc_frame_translate(b, i, ft[0], ft[1], ft[2]); /* :68 */
    if (best_rotation_out) *best_rotation_out = best;
    return 0;
}

/* ------------------------------------------------------------------------------------
 * align_algorithms.rs:339-451 helpers
 * ---------------------------------------------------------------------------------- */
size_t orc_refine_angles(double initial, double range, double step, double* out, size_t cap)
{
    size_t n = 0;
    if (!(step > 0.0)) return 0;
    double angle = initial - range;                           /* :386 */
    while (angle <= initial + range) {                        /* :387 */
        if (out && n < cap) out[n] = angle;
        ++n;
        angle += step;                                        /* :439 */
    }
    return n;
}

size_t orc_filter_points_in_region(const orc_point* pts, size_t n, const orc_point* s,
                                   const orc_point* e, int64_t* out_idx, size_t cap)
{
    const double margin = 5.0;                                /* :460 */
    const double min_x = fmin(s->x, e->x) - margin, max_x = fmax(s->x, e->x) + margin;
    const double min_y = fmin(s->y, e->y) - margin, max_y = fmax(s->y, e->y) + margin;
    const double min_z = fmin(s->z, e->z) - margin, max_z = fmax(s->z, e->z) + margin;
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {                          /* :493-504 */
        const orc_point* p = &pts[i];
        if (p->x >= min_x && p->x <= max_x && p->y >= min_y && p->y <= max_y && p->z >= min_z && p->z <= max_z) {
            if (out_idx && m < cap) out_idx[m] = (int64_t)i;
            ++m;
        }
    }
    return m;
}

size_t orc_refine_downsample_count(size_t n_filtered, size_t n_points_per_frame, size_t n_frames)
{
    double ratio = (double)n_filtered / ((double)n_points_per_frame * (double)n_frames); /* :415-416 */
    double nd = ceil(ratio * (double)n_points_per_frame);                                /* :417 */
    size_t n = !(nd > 0.0) ? 0 : (nd >= 1.8e19 ? (size_t)-1 : (size_t)nd);                /* `as usize` saturates (NaN -> 0) */
    if (n < 1) n = 1;                                                                    /* :418 clamp */
    if (n > n_points_per_frame) n = n_points_per_frame;
    return n;
}
