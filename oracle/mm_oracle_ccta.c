/*
 * mm_oracle_ccta.c -- see mm_oracle_ccta.h.  TEST INFRASTRUCTURE ONLY.
 * Compile with -ffp-contract=off (Rust never fuses a*b+c).
 */
#include "mm_oracle_ccta.h"

#include <float.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

static double sq_dist(double ax, double ay, double az, double bx, double by, double bz)
{
    double dx = ax - bx, dy = ay - by, dz = az - bz;            /* adjust_mesh.rs:7-12 */
    return dx * dx + dy * dy + dz * dz;
}

void orc_nn_min_sq(const orc_point* a, size_t na, const orc_point* b, size_t nb, double* out)
{
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
    if (nt > 16) nt = 16;   /* a GPU box's CPU share is 16 cores whatever nproc says */
#endif
#pragma omp parallel for schedule(static) num_threads(nt)
    for (long long i = 0; i < (long long)na; ++i) {
        double m = INFINITY;
        for (size_t j = 0; j < nb; ++j) {
            double v = sq_dist(a[i].x, a[i].y, a[i].z, b[j].x, b[j].y, b[j].z);
            if (v < m) m = v;                                    /* fold(INF, |m, v| if v < m {v} else {m}) */
        }
        out[i] = m;
    }
}

double orc_symmetric_nn_distance(const orc_point* a, size_t na, const orc_point* b, size_t nb)
{
    if (na == 0 || nb == 0) return INFINITY;                     /* :189-191 */
    double* m = (double*)malloc((na > nb ? na : nb) * sizeof(double));
    orc_nn_min_sq(a, na, b, nb, m);
    double sum_ab = 0.0;
    for (size_t i = 0; i < na; ++i) sum_ab += m[i];              /* :193-200 (sequential order) */
    double avg_ab = sum_ab / (double)na;                         /* :202 */
    orc_nn_min_sq(b, nb, a, na, m);
    double sum_ba = 0.0;
    for (size_t i = 0; i < nb; ++i) sum_ba += m[i];              /* :204-211 */
    double avg_ba = sum_ba / (double)nb;                         /* :213 */
    free(m);
    return sqrt((avg_ab + avg_ba) / 2.0);                        /* :215 */
}

static size_t closest_cl(const orc_clpoint* cl, size_t ncl, double x, double y, double z)
{
    double best = DBL_MAX;                                       /* :249-258 */
    size_t idx = 0;
    for (size_t k = 0; k < ncl; ++k) {
        double d = sq_dist(x, y, z, cl[k].x, cl[k].y, cl[k].z);
        if (d < best) { best = d; idx = k; }
    }
    return idx;
}

void orc_diameter_morphing(const orc_clpoint* cl, size_t ncl, const orc_point* pts, size_t n,
                           double adj, orc_point* out)
{
    for (size_t i = 0; i < n; ++i) {
        size_t k = closest_cl(cl, ncl, pts[i].x, pts[i].y, pts[i].z);                /* :226 */
        double vx = pts[i].x - cl[k].x, vy = pts[i].y - cl[k].y, vz = pts[i].z - cl[k].z;
        double nn = sqrt(vx * vx + vy * vy + vz * vz);           /* try_normalize(0.0): norm > 0 -> v / norm */
        if (nn > 0.0) {
            out[i].x = pts[i].x + (vx / nn) * adj;               /* :236 p + unit * x */
            out[i].y = pts[i].y + (vy / nn) * adj;
            out[i].z = pts[i].z + (vz / nn) * adj;
        } else out[i] = pts[i];                                  /* :239 */
    }
}

typedef struct { size_t i; double d; } idx_dist;
static int idx_dist_cmp(const void* a, const void* b)
{
    const idx_dist* x = (const idx_dist*)a;
    const idx_dist* y = (const idx_dist*)b;
    if (x->d < y->d) return -1;                                  /* :154-158 */
    if (x->d > y->d) return 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

size_t orc_find_region_points(const orc_point* an, size_t n, const orc_point* ref, size_t nr, size_t n_points,
                              orc_point* selected, orc_point* remaining)
{
    if (n == 0 || nr == 0 || n_points == 0) {                    /* :138-140 */
        memcpy(remaining, an, n * sizeof(orc_point));
        return 0;
    }
    double* m = (double*)malloc(n * sizeof(double));
    orc_nn_min_sq(an, n, ref, nr, m);                            /* :142-152 */
    idx_dist* e = (idx_dist*)malloc(n * sizeof(idx_dist));
    for (size_t i = 0; i < n; ++i) { e[i].i = i; e[i].d = m[i]; }
    qsort(e, n, sizeof(idx_dist), idx_dist_cmp);
    size_t take = n_points < n ? n_points : n;                   /* :160 */
    unsigned char* sel = (unsigned char*)calloc(n, 1);
    for (size_t k = 0; k < take; ++k) { selected[k] = an[e[k].i]; sel[e[k].i] = 1; }  /* :165-168 */
    size_t r = 0;
    for (size_t i = 0; i < n; ++i) if (!sel[i]) remaining[r++] = an[i];               /* :170-180 */
    free(sel); free(e); free(m);
    return take;
}

static double scaling_search(const orc_point* pts, size_t n, const orc_point* ref, size_t nr,
                             const orc_clpoint* cl, size_t ncl, double* all_dist)
{
    const double start = -2.0, end = 2.0, step = 0.1;
    const int steps = (int)round((end - start) / step);          /* :70-73 -> 40 */
    double min_dist = DBL_MAX, best = DBL_MAX;
    orc_point* tmp = (orc_point*)malloc((n ? n : 1) * sizeof(orc_point));
    for (int i = 0; i <= steps; ++i) {
        double x = start + (double)i * step;                     /* :79 */
        orc_diameter_morphing(cl, ncl, pts, n, x, tmp);
        double d = orc_symmetric_nn_distance(ref, nr, tmp, n);   /* :81 (reference set first) */
        if (all_dist) all_dist[i] = d;
        if (d < min_dist) { min_dist = d; best = x; }            /* :82-85 */
    }
    free(tmp);
    return best;
}

double orc_aortic_diameter_optimization(const orc_point* intramural, size_t ni, const orc_point* reference,
                                        size_t nr, const orc_clpoint* cl, size_t ncl, double* all_dist)
{
    return scaling_search(intramural, ni, reference, nr, cl, ncl, all_dist);
}

void orc_diameter_optimization(const orc_point* anomalous, size_t n, size_t n_proximal, size_t n_distal,
                               const orc_clpoint* cl, size_t ncl, const orc_point* prox_ref, size_t npr,
                               const orc_point* dist_ref, size_t ndr, double* prox_best, double* dist_best)
{
    orc_point* prox = (orc_point*)malloc((n ? n : 1) * sizeof(orc_point));
    orc_point* rest = (orc_point*)malloc((n ? n : 1) * sizeof(orc_point));
    orc_point* dist = (orc_point*)malloc((n ? n : 1) * sizeof(orc_point));
    orc_point* rest2 = (orc_point*)malloc((n ? n : 1) * sizeof(orc_point));
    size_t np = orc_find_region_points(anomalous, n, prox_ref, npr, n_proximal, prox, rest);   /* :98-99 */
    size_t nd = orc_find_region_points(rest, n - np, dist_ref, ndr, n_distal, dist, rest2);    /* :100 */
    *prox_best = scaling_search(prox, np, prox_ref, npr, cl, ncl, NULL);                       /* :112-120 */
    *dist_best = scaling_search(dist, nd, dist_ref, ndr, cl, ncl, NULL);                       /* :121-129 */
    free(prox); free(rest); free(dist); free(rest2);
}

double orc_wall_diameter_optimization(const orc_clpoint* cl, size_t ncl, const double ref[3],
                                      const orc_point* aortic, size_t na)
{
    if (ncl == 0 || na == 0) return 0.0;                         /* :13-15 */
    /* Iterator::min_by keeps the FIRST of equal minima (:17-37) */
    size_t kc = 0; double bc = INFINITY;
    for (size_t k = 0; k < ncl; ++k) {
        double d = sq_dist(cl[k].x, cl[k].y, cl[k].z, ref[0], ref[1], ref[2]);
        if (d < bc) { bc = d; kc = k; }
    }
    size_t ka = 0; double ba = INFINITY;
    for (size_t k = 0; k < na; ++k) {
        double d = sq_dist(aortic[k].x, aortic[k].y, aortic[k].z, ref[0], ref[1], ref[2]);
        if (d < ba) { ba = d; ka = k; }
    }
    double vx = ref[0] - cl[kc].x, vy = ref[1] - cl[kc].y, vz = ref[2] - cl[kc].z;   /* :52 */
    double nn = sqrt(vx * vx + vy * vy + vz * vz);
    if (!(nn > 0.0)) return 0.0;                                 /* :53-55 */
    double ux = vx / nn, uy = vy / nn, uz = vz / nn;
    double tx = ref[0] - aortic[ka].x, ty = ref[1] - aortic[ka].y, tz = ref[2] - aortic[ka].z; /* :59 */
    double t = tx * ux + ty * uy + tz * uz;                      /* :60 */
    return t > 0.0 ? t : 0.0;                                    /* :62 f64::max(0.0) */
}

/* clean_up_non_section_points (:342-409).  The reference counts neighbours with two rstar R-trees
 * (locate_within_distance: distance_2 <= squared radius, distance_2 = the sum of the squared coordinate
 * differences folded x, y, z); an exhaustive count gives the same numbers.  PARITY UNPINNED: the reference
 * holds no test for this function nor for find_points_by_cl_region_rs. */
void orc_clean_outlier_points(const orc_point* cleanup, size_t nc, const orc_point* reference, size_t nr,
                              double radius, double min_ratio, uint8_t* to_reference)
{
    const double r2 = radius * radius;                            /* :348 */
    for (size_t i = 0; i < nc; ++i) {
        size_t ref_n = 0, self_n = 0;
        for (size_t k = 0; k < nr; ++k)
            if (sq_dist(cleanup[i].x, cleanup[i].y, cleanup[i].z, reference[k].x, reference[k].y, reference[k].z) <= r2) ++ref_n;
        for (size_t k = 0; k < nc; ++k)
            if (sq_dist(cleanup[i].x, cleanup[i].y, cleanup[i].z, cleanup[k].x, cleanup[k].y, cleanup[k].z) <= r2) ++self_n;
        self_n = self_n > 0 ? self_n - 1 : 0;                     /* :381-384 saturating_sub(1): the point itself */
        const size_t total = ref_n + self_n;
        to_reference[i] = 0;
        if (total > 0) {                                          /* :388-400 */
            const double ratio = (double)ref_n / (double)total;
            if (ratio >= min_ratio) to_reference[i] = 1;
        }
    }
}

/* find_points_by_cl_region_rs (:263-312) with find_cl_points_in_range (:314-338) and
 * find_closest_centerline_point_optimized (:245-260).  label: 0 proximal, 1 distal, 2 between, 3 / 4 = moved
 * into `between` by the first / second clean-up (the order in which the reference appends them). */
void orc_find_points_by_cl_region(const orc_clpoint* cl, const uint32_t* cl_frame_index, size_t ncl,
                                  const double* centroids, size_t n_frames, const orc_point* pts, size_t n,
                                  uint8_t* label)
{
    double mean_dz = 0.0;
    for (size_t i = 1; i < n_frames; ++i) mean_dz += fabs(centroids[3 * i + 2] - centroids[3 * (i - 1) + 2]);   /* :268-272 */
    mean_dz /= (double)(n_frames - 1);
    uint8_t* sel = (uint8_t*)calloc(ncl ? ncl : 1, 1);             /* centerline point k is within range of a centroid */
    for (size_t f = 0; f < n_frames; ++f)
        for (size_t k = 0; k < ncl; ++k)
            if (sq_dist(centroids[3 * f], centroids[3 * f + 1], centroids[3 * f + 2], cl[k].x, cl[k].y, cl[k].z) <= mean_dz * mean_dz)
                sel[k] = 1;
    const double* dref = centroids + 3 * (n_frames - 1);           /* :279 */
    orc_point* prox = (orc_point*)calloc(n ? n : 1, sizeof(orc_point));
    orc_point* dist = (orc_point*)calloc(n ? n : 1, sizeof(orc_point));
    orc_point* betw = (orc_point*)calloc(n ? n : 1, sizeof(orc_point));
    size_t* ip = (size_t*)malloc((n ? n : 1) * sizeof(size_t));
    size_t* id = (size_t*)malloc((n ? n : 1) * sizeof(size_t));
    size_t np = 0, nd = 0, nb = 0;
    for (size_t i = 0; i < n; ++i) {
        double best = DBL_MAX; size_t kb = 0;                      /* :249-258 strict <: first minimum */
        for (size_t k = 0; k < ncl; ++k) {
            const double d = sq_dist(pts[i].x, pts[i].y, pts[i].z, cl[k].x, cl[k].y, cl[k].z);
            if (d < best) { best = d; kb = k; }
        }
        /* cl_points_indices.contains(&frame_index): some selected centerline point carries this frame index */
        const uint32_t fi = cl_frame_index ? cl_frame_index[kb] : (uint32_t)kb;
        int between = 0;
        for (size_t k = 0; k < ncl && !between; ++k)
            if (sel[k] && (cl_frame_index ? cl_frame_index[k] : (uint32_t)k) == fi) between = 1;
        if (between) { label[i] = 2; betw[nb++] = pts[i]; }
        else if (pts[i].x > dref[0] && pts[i].y > dref[1] && pts[i].z > dref[2]) { label[i] = 0; ip[np] = i; prox[np++] = pts[i]; }  /* :303-309 */
        else { label[i] = 1; id[nd] = i; dist[nd++] = pts[i]; }
    }
    uint8_t* mv = (uint8_t*)malloc(n ? n : 1);
    orc_clean_outlier_points(prox, np, betw, nb, 1.0, 0.6, mv);    /* :310-311 */
    for (size_t k = 0; k < np; ++k) if (mv[k]) { label[ip[k]] = 3; betw[nb++] = prox[k]; }
    orc_clean_outlier_points(dist, nd, betw, nb, 1.0, 0.6, mv);    /* :312-313: sees the points moved above */
    for (size_t k = 0; k < nd; ++k) if (mv[k]) label[id[k]] = 4;
    free(sel); free(prox); free(dist); free(betw); free(ip); free(id); free(mv);
}
