/*
 * mm_oracle_cl.c -- see mm_oracle_cl.h.  TEST INFRASTRUCTURE ONLY.
 * Compile with -ffp-contract=off (Rust never fuses a*b+c).
 */
#define _GNU_SOURCE
#include "mm_oracle_cl.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_TAU 6.283185307179586476925286766559

/* linux-gnu lowers a sin()/cos() pair on one operand (f64::sin_cos) to glibc sincos */
static void sin_cos(double x, double* s, double* c) { sincos(x, s, c); }

/* ---- nalgebra 0.35 pieces ------------------------------------------------------------- */
static double v3_dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double v3_norm(const double a[3]) { return sqrt(v3_dot(a, a)); }
static void v3_cross(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
/* Matrix::angle (base/matrix.rs) */
static double v3_angle(const double a[3], const double b[3])
{
    double prod = v3_dot(a, b);
    double n1 = v3_norm(a), n2 = v3_norm(b);
    if (n1 == 0.0 || n2 == 0.0) return 0.0;
    double cang = prod / (n1 * n2);
    if (cang < -1.0) cang = -1.0;
    if (cang > 1.0) cang = 1.0;
    return acos(cang);
}
static void m3_identity(double r[9])
{
    memset(r, 0, 9 * sizeof(double));
    r[0] = r[4] = r[8] = 1.0;
}
/* Rotation3::from_axis_angle(&Unit::new_normalize(axis), angle) */
static void m3_axis_angle(const double axis[3], double angle, double r[9])
{
    if (angle == 0.0) { m3_identity(r); return; }
    double n = v3_norm(axis);
    double ux = axis[0] / n, uy = axis[1] / n, uz = axis[2] / n;   /* Unit::new_normalize */
    double sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    double s, c;
    sin_cos(angle, &s, &c);
    double omc = 1.0 - c;
    r[0] = sqx + (1.0 - sqx) * c;
    r[1] = ux * uy * omc - uz * s;
    r[2] = ux * uz * omc + uy * s;
    r[3] = ux * uy * omc + uz * s;
    r[4] = sqy + (1.0 - sqy) * c;
    r[5] = uy * uz * omc - ux * s;
    r[6] = ux * uz * omc - uy * s;
    r[7] = uy * uz * omc + ux * s;
    r[8] = sqz + (1.0 - sqz) * c;
}
/* Rotation3 * Vector3 (blas.rs gemv: column 0 first, then += column j) */
static void m3_mul(const double r[9], const double v[3], double o[3])
{
    for (int i = 0; i < 3; ++i) {
        double y = r[3 * i + 0] * v[0];
        y = r[3 * i + 1] * v[1] + y;
        y = r[3 * i + 2] * v[2] + y;
        o[i] = y;
    }
}

/* ---- centerline.rs ---------------------------------------------------------------------- */
int orc_centerline_from_points(const orc_point* pts, size_t n, orc_clpoint* out)
{
    for (size_t i = 0; i < n; ++i) {
        double t[3];
        if (i + 1 < n) {                                                             /* :20-22 */
            double d[3] = { pts[i + 1].x - pts[i].x, pts[i + 1].y - pts[i].y, pts[i + 1].z - pts[i].z };
            double nn = v3_norm(d);
            t[0] = d[0] / nn; t[1] = d[1] / nn; t[2] = d[2] / nn;                    /* normalize() */
        } else {                                                                     /* :23-24 */
            if (i == 0) return -1;  /* points[i - 1] underflows: the reference panics */
            t[0] = out[i - 1].tx; t[1] = out[i - 1].ty; t[2] = out[i - 1].tz;
        }
        out[i].x = pts[i].x; out[i].y = pts[i].y; out[i].z = pts[i].z;
        out[i].tx = t[0]; out[i].ty = t[1]; out[i].tz = t[2];
        out[i].radius = 0.0; out[i].branch_id = 0; out[i].pad_ = 0;
    }
    return 0;
}

size_t orc_cl_find_ref_idx(const orc_clpoint* cl, size_t n, const double ref[3])
{
    size_t best_idx = 0;
    double best_dist = INFINITY;
    for (size_t i = 0; i < n; ++i) {
        double dx = cl[i].x - ref[0], dy = cl[i].y - ref[1], dz = cl[i].z - ref[2];  /* native.rs:27-32 */
        double dist = sqrt(dx * dx + dy * dy + dz * dz);
        if (dist < best_dist) { best_dist = dist; best_idx = i; }
    }
    return best_idx;
}

/* ---- preprocessing.rs ------------------------------------------------------------------- */
static orc_clpoint interpolate_at_s(const orc_clpoint* cl, size_t n, const double* cum, double target_s)
{
    /* :169-173 binary search: Ok(i) -> i, Err(0) -> 0, Err(pos) -> pos-1  ==  last index whose
     * cumulative length is <= target (strictly increasing cum; with repeated values the std
     * binary search may return any of the equal entries) */
    size_t idx = 0;
    {
        size_t lo = 0, hi = n;  /* first index with cum > target */
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if (cum[mid] <= target_s) lo = mid + 1; else hi = mid;
        }
        idx = lo == 0 ? 0 : lo - 1;
    }
    orc_clpoint o;
    memset(&o, 0, sizeof o);
    if (idx >= (n == 0 ? 0 : n - 1)) {                                               /* :176-193 */
        o = cl[n - 1];
        o.branch_id = 0;
        return o;
    }
    const orc_clpoint* p0 = &cl[idx];
    const orc_clpoint* p1 = &cl[idx + 1];
    double s0 = cum[idx], s1 = cum[idx + 1];
    double denom = s1 - s0;
    double t = fabs(denom) < 1e-12 ? 0.0 : (target_s - s0) / denom;                  /* :201-205 */
    o.x = p0->x + t * (p1->x - p0->x);
    o.y = p0->y + t * (p1->y - p0->y);
    o.z = p0->z + t * (p1->z - p0->z);
    double t0[3] = { p0->tx, p0->ty, p0->tz }, t1[3] = { p1->tx, p1->ty, p1->tz };
    double tg[3] = { 0.0, 0.0, 0.0 };
    if (v3_norm(t0) > 0.0 || v3_norm(t1) > 0.0) {                                    /* :215-223 */
        for (int k = 0; k < 3; ++k) tg[k] = t0[k] * (1.0 - t) + t1[k] * t;
        double tn = v3_norm(tg);
        if (tn > 1e-12) { tg[0] /= tn; tg[1] /= tn; tg[2] /= tn; }
        else tg[0] = tg[1] = tg[2] = 0.0;
    }
    o.tx = tg[0]; o.ty = tg[1]; o.tz = tg[2];
    o.radius = p0->radius * (1.0 - t) + p1->radius * t;                              /* :225-227 */
    o.branch_id = 0;
    return o;
}

int64_t orc_preprocess_centerline(const orc_clpoint* cl_in, size_t n_in, const orc_geometry* ref_mesh,
                                  orc_clpoint* out, size_t cap, double* spacing_out)
{
    /* :23-30 keep branch 0 only */
    orc_clpoint* cl = (orc_clpoint*)malloc((n_in ? n_in : 1) * sizeof(orc_clpoint));
    size_t n = 0;
    for (size_t i = 0; i < n_in; ++i) if (cl_in[i].branch_id == 0) cl[n++] = cl_in[i];
    if (n == 0) { free(cl); return -1; }
    /* :39-47 ensure_descending_z */
    if (cl[0].z < cl[n - 1].z)
        for (size_t i = 0, j = n - 1; i < j; ++i, --j) { orc_clpoint t = cl[i]; cl[i] = cl[j]; cl[j] = t; }
    if (ref_mesh->n_frames <= 0) { free(cl); return -3; }                            /* :56-58 */

    /* :244-280 mean spacing of consecutive frame centroids */
    int has_mean = 0;
    double mean = 0.0;
    if (ref_mesh->n_frames >= 2) {
        double sum = 0.0;
        for (int32_t i = 0; i + 1 < ref_mesh->n_frames; ++i) {
            const double* a = &ref_mesh->centroid[3 * i];
            const double* b = &ref_mesh->centroid[3 * (i + 1)];
            double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
            sum += sqrt(dx * dx + dy * dy + dz * dz);
        }
        mean = sum / (double)(ref_mesh->n_frames - 1);
        has_mean = isfinite(mean) && mean > 1e-12;
    }
    /* :110-126 cumulative arc length */
    double* cum = (double*)malloc(n * sizeof(double));
    cum[0] = 0.0;
    for (size_t i = 1; i < n; ++i) {
        double dx = cl[i].x - cl[i - 1].x, dy = cl[i].y - cl[i - 1].y, dz = cl[i].z - cl[i - 1].z;
        cum[i] = cum[i - 1] + sqrt(dx * dx + dy * dy + dz * dz);
    }
    double total = cum[n - 1];
    /* :128-143 decide_spacing */
    double spacing = 0.0;
    int ok = 0;
    if (has_mean) { spacing = mean; ok = 1; }
    else if (n - 1 >= 1) {
        double fb = total / (double)(n - 1);
        if (isfinite(fb) && fb > 1e-12) { spacing = fb; ok = 1; }
    }
    int64_t count = 0;
    if (!ok) {                                                                       /* :70-73 */
        for (size_t i = 0; i < n; ++i) { if ((size_t)count < cap) out[count] = cl[i]; ++count; }
        *spacing_out = 0.0;
        free(cum); free(cl);
        return count;
    }
    /* :145-160 build_samples, :86-89 interpolate */
    size_t ns = 0, scap = 64;
    double* s_new = (double*)malloc(scap * sizeof(double));
    for (double s = 0.0; s <= total + 1e-9; s += spacing) {
        if (ns == scap) { scap *= 2; s_new = (double*)realloc(s_new, scap * sizeof(double)); }
        s_new[ns++] = s;
    }
    if (ns && s_new[ns - 1] > total + 1e-6) s_new[ns - 1] = total;
    for (size_t k = 0; k < ns; ++k) {
        if ((size_t)count < cap) out[count] = interpolate_at_s(cl, n, cum, s_new[k]);
        ++count;
    }
    *spacing_out = spacing;
    free(s_new); free(cum); free(cl);
    return count;
}

/* ---- contour.rs:368-405 ----------------------------------------------------------------- */
typedef struct { double key; size_t idx; } sort_ent;
static int sort_cmp(const void* a, const void* b)
{
    const sort_ent* x = (const sort_ent*)a;
    const sort_ent* y = (const sort_ent*)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);   /* sort_by is stable */
}

void orc_sort_contour_points(orc_point* pts, size_t n) { orc_sort_contour_points_flags(pts, NULL, n); }

void orc_sort_contour_points_flags(orc_point* pts, uint8_t* flags, size_t n)
{
    if (n == 0) return;
    double sx = 0.0, sy = 0.0;
    for (size_t i = 0; i < n; ++i) { sx += pts[i].x; sy += pts[i].y; }               /* :377-381 */
    double cx = sx / (double)n, cy = sy / (double)n;
    sort_ent* e = (sort_ent*)malloc(n * sizeof(sort_ent));
    for (size_t i = 0; i < n; ++i) { e[i].key = atan2(pts[i].y - cy, pts[i].x - cx); e[i].idx = i; }
    qsort(e, n, sizeof(sort_ent), sort_cmp);                                         /* :385-390 */
    orc_point* tmp = (orc_point*)malloc(n * sizeof(orc_point));
    for (size_t i = 0; i < n; ++i) tmp[i] = pts[e[i].idx];
    size_t start = 0;                                                                /* :393-401: max_by keeps the LAST maximum */
    for (size_t i = 1; i < n; ++i) if (!(tmp[i].y < tmp[start].y)) start = i;
    for (size_t i = 0; i < n; ++i) pts[i] = tmp[(i + start) % n];                    /* rotate_left */
    if (flags) {                                                                     /* the flag is a field of the point */
        uint8_t* tf = (uint8_t*)malloc(n);
        for (size_t i = 0; i < n; ++i) tf[i] = flags[e[(i + start) % n].idx];
        memcpy(flags, tf, n);
        free(tf);
    }
    free(tmp); free(e);
}

/* wall points of the frames before `frame` = the frame's offset into wall_aortic */
static int64_t wall_points_before(const orc_clgeom* cg, int32_t frame)
{
    if (!cg->wall_kind1 || !cg->g->extra_off) return 0;
    const int32_t K = cg->extra_kind_off ? cg->n_extra_kinds : 1, w = cg->wall_kind1 - 1;
    int64_t n = 0;
    for (int32_t i = 0; i < frame; ++i)
        n += cg->extra_kind_off ? cg->extra_kind_off[(int64_t)i * K + w + 1] - cg->extra_kind_off[(int64_t)i * K + w]
                                : cg->g->extra_off[i + 1] - cg->g->extra_off[i];
    return n;
}

/* ---- geometry.rs:241-250 ---------------------------------------------------------------- */
void orc_rotate_geometry(orc_clgeom* cg, double angle)
{
    if (angle == 0.0) return;
    orc_geometry* g = cg->g;
    for (int32_t i = 0; i < g->n_frames; ++i) {
        orc_frame_rotate(g, i, angle, g->centroid[3 * i], g->centroid[3 * i + 1]);   /* :246-247 */
        /* frame.rs:123-129 sort_frame_points: lumen and every extras contour */
        orc_sort_contour_points_flags(g->lumen + g->lumen_off[i], cg->lumen_aortic ? cg->lumen_aortic + g->lumen_off[i] : NULL,
                                      (size_t)(g->lumen_off[i + 1] - g->lumen_off[i]));
        if (g->cath_off)
            orc_sort_contour_points(g->cath + g->cath_off[i], (size_t)(g->cath_off[i + 1] - g->cath_off[i]));
        if (g->extra_off) {
            uint8_t* wf = cg->wall_aortic && cg->wall_kind1 ? cg->wall_aortic + wall_points_before(cg, i) : NULL;
            if (cg->extra_kind_off && cg->n_extra_kinds > 0) {
                for (int32_t k = 0; k < cg->n_extra_kinds; ++k) {
                    int64_t lo = cg->extra_kind_off[(int64_t)i * cg->n_extra_kinds + k];
                    int64_t hi = cg->extra_kind_off[(int64_t)i * cg->n_extra_kinds + k + 1];
                    orc_sort_contour_points_flags(g->extra + lo, k == cg->wall_kind1 - 1 ? wf : NULL, (size_t)(hi - lo));
                }
            } else {
                orc_sort_contour_points_flags(g->extra + g->extra_off[i], cg->wall_kind1 == 1 ? wf : NULL,
                                              (size_t)(g->extra_off[i + 1] - g->extra_off[i]));
            }
        }
    }
}

/* ---- align_algorithms.rs ---------------------------------------------------------------- */
void orc_newell_normal(const orc_point* p, size_t n, const double c[3], double out[3])
{
    if (n < 3) { out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; return; }                 /* :207-209 */
    double nx = 0.0, ny = 0.0, nz = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const orc_point* cur = &p[i];
        const orc_point* nxt = &p[(i + 1) % n];
        nx += (cur->y - c[1]) * (nxt->z - c[2]) - (cur->z - c[2]) * (nxt->y - c[1]); /* :218-223 */
        ny += (cur->z - c[2]) * (nxt->x - c[0]) - (cur->x - c[0]) * (nxt->z - c[2]);
        nz += (cur->x - c[0]) * (nxt->y - c[1]) - (cur->y - c[1]) * (nxt->x - c[0]);
    }
    double v[3] = { nx, ny, nz };
    double norm = v3_norm(v);
    if (norm > 1e-12) { out[0] = nx / norm; out[1] = ny / norm; out[2] = nz / norm; }
    else { out[0] = 0.0; out[1] = 0.0; out[2] = 1.0; }
}

static void mean_point(const orc_point* p, size_t n, double c[3])
{
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (size_t i = 0; i < n; ++i) sx += p[i].x;
    for (size_t i = 0; i < n; ++i) sy += p[i].y;
    for (size_t i = 0; i < n; ++i) sz += p[i].z;
    c[0] = sx / (double)n; c[1] = sy / (double)n; c[2] = sz / (double)n;
}

void orc_align_frame(const orc_point* pts, size_t n, int has_centroid, const double centroid_in[3],
                     const orc_clpoint* clp, orc_frame_tf* tf)
{
    double c[3];
    if (has_centroid) memcpy(c, centroid_in, sizeof c); else mean_point(pts, n, c);  /* :130-135 */
    tf->t[0] = clp->x - c[0]; tf->t[1] = clp->y - c[1]; tf->t[2] = clp->z - c[2];    /* :138-142 */
    double cur[3], des[3] = { clp->tx, clp->ty, clp->tz };
    orc_newell_normal(pts, n, c, cur);                                               /* :145 */
    double angle = v3_angle(cur, des);                                               /* :147 */
    if (fabs(angle) < 1e-6) m3_identity(tf->r);
    else {
        double axis[3];
        v3_cross(cur, des, axis);
        if (v3_norm(axis) < 1e-6) m3_identity(tf->r);
        else m3_axis_angle(axis, angle, tf->r);                                      /* :155-156 */
    }
    tf->pivot[0] = clp->x; tf->pivot[1] = clp->y; tf->pivot[2] = clp->z;             /* :161-165 */
}

orc_point orc_tf_apply(const orc_frame_tf* tf, orc_point p)
{
    double rel[3] = { (p.x + tf->t[0]) - tf->pivot[0], (p.y + tf->t[1]) - tf->pivot[1],
                      (p.z + tf->t[2]) - tf->pivot[2] };                             /* :75-80 */
    double rot[3];
    m3_mul(tf->r, rel, rot);
    orc_point o = { tf->pivot[0] + rot[0], tf->pivot[1] + rot[1], tf->pivot[2] + rot[2] };
    return o;
}

static void tf_span(const orc_frame_tf* tf, orc_point* p, int64_t lo, int64_t hi)
{
    for (int64_t i = lo; i < hi; ++i) p[i] = orc_tf_apply(tf, p[i]);
}

size_t orc_apply_transformations(orc_clgeom** geoms, int n_geoms, const orc_clpoint* cl, size_t ncl,
                                 const double ref_pt[3])
{
    const orc_clgeom* prim = geoms[0];
    const orc_geometry* pg = prim->g;
    size_t ref_idx_cl = orc_cl_find_ref_idx(cl, ncl, ref_pt);                        /* :104 */
    size_t n_tf = 0;
    orc_frame_tf* tfs = (orc_frame_tf*)malloc((size_t)(pg->n_frames > 0 ? pg->n_frames : 1) * sizeof(orc_frame_tf));
    for (int32_t i = 0; i < pg->n_frames; ++i) {
        size_t cl_index = ref_idx_cl + (size_t)i;                                    /* :112 */
        if (cl_index < ncl) {
            int hc = prim->has_lumen_centroid && prim->has_lumen_centroid[i];
            orc_align_frame(pg->lumen + pg->lumen_off[i], (size_t)(pg->lumen_off[i + 1] - pg->lumen_off[i]),
                            hc, hc ? &prim->lumen_centroid[3 * i] : NULL, &cl[cl_index], &tfs[n_tf++]);
        }                                                                            /* else: skipped with a warning */
    }
    for (int gi = 0; gi < n_geoms; ++gi) {                                           /* :521-535 */
        orc_clgeom* cg = geoms[gi];
        orc_geometry* g = cg->g;
        for (int32_t i = 0; i < g->n_frames && (size_t)i < n_tf; ++i) {
            const orc_frame_tf* tf = &tfs[i];
            tf_span(tf, g->lumen, g->lumen_off[i], g->lumen_off[i + 1]);
            int hc = cg->has_lumen_centroid && cg->has_lumen_centroid[i];
            if (hc) {                                                                /* :186-201 */
                orc_point c = { cg->lumen_centroid[3 * i], cg->lumen_centroid[3 * i + 1], cg->lumen_centroid[3 * i + 2] };
                c = orc_tf_apply(tf, c);
                cg->lumen_centroid[3 * i] = c.x; cg->lumen_centroid[3 * i + 1] = c.y; cg->lumen_centroid[3 * i + 2] = c.z;
            }
            if (g->cath_off) tf_span(tf, g->cath, g->cath_off[i], g->cath_off[i + 1]);
            if (g->extra_off) tf_span(tf, g->extra, g->extra_off[i], g->extra_off[i + 1]);
            if (g->has_ref && g->has_ref[i]) g->ref[i] = orc_tf_apply(tf, g->ref[i]); /* :529-531 */
            if (hc) memcpy(&g->centroid[3 * i], &cg->lumen_centroid[3 * i], 3 * sizeof(double)); /* :532 */
            else g->centroid[3 * i] = g->centroid[3 * i + 1] = g->centroid[3 * i + 2] = 0.0;
        }
    }
    free(tfs);
    return n_tf;
}

void orc_rotation_from_axis_angle(const double axis[3], double angle, double r[9])
{
    m3_axis_angle(axis, angle, r);
}

/* :238-259 */
void orc_rotate_contour_around_centroid(orc_point* pts, size_t n, int has_centroid,
                                        const double centroid_in[3], double angle)
{
    double c[3];
    if (has_centroid) memcpy(c, centroid_in, sizeof c); else mean_point(pts, n, c);
    double axis[3], r[9];
    orc_newell_normal(pts, n, c, axis);
    m3_axis_angle(axis, angle, r);
    for (size_t i = 0; i < n; ++i) {
        double rel[3] = { pts[i].x - c[0], pts[i].y - c[1], pts[i].z - c[2] }, rot[3];
        m3_mul(r, rel, rot);
        pts[i].x = c[0] + rot[0]; pts[i].y = c[1] + rot[1]; pts[i].z = c[2] + rot[2];
    }
}

double orc_best_rotation_three_point(const orc_point* pts, size_t n, int has_centroid,
                                     const double centroid_in[3], uint32_t index_reference,
                                     const double p_main[3], const double p_ccw[3],
                                     const double p_cw[3], double angle_step, const orc_clpoint* clp)
{
    double best_angle = 0.0, min_total_error = DBL_MAX;
    orc_point* tmp = (orc_point*)malloc((n ? n : 1) * sizeof(orc_point));
    size_t i_ccw = 0, i_cw = n / 2;                                                  /* :306-311 (point_index == position) */
    for (double angle = 0.0; angle < ORC_TAU; angle += angle_step) {                 /* :286, 332 */
        memcpy(tmp, pts, n * sizeof(orc_point));
        orc_rotate_contour_around_centroid(tmp, n, has_centroid, centroid_in, angle);  /* :291 */
        /* :294-295 align to the centerline point (contour centroid is not moved by the rotation) */
        orc_frame_tf tf;
        orc_align_frame(tmp, n, has_centroid, centroid_in, clp, &tf);
        orc_point pm = orc_tf_apply(&tf, tmp[index_reference]);
        orc_point pc = orc_tf_apply(&tf, tmp[i_ccw]);
        orc_point pw = orc_tf_apply(&tf, tmp[i_cw]);
        double d0[3] = { p_main[0] - pm.x, p_main[1] - pm.y, p_main[2] - pm.z };     /* distance(p, q) = |q - p| */
        double d1[3] = { p_ccw[0] - pc.x, p_ccw[1] - pc.y, p_ccw[2] - pc.z };
        double d2[3] = { p_cw[0] - pw.x, p_cw[1] - pw.y, p_cw[2] - pw.z };
        double dm = v3_norm(d0), dc = v3_norm(d1), dw = v3_norm(d2);
        double total_error = dm * dm + dc * dc + dw * dw;                            /* :326 */
        if (total_error < min_total_error) { min_total_error = total_error; best_angle = angle; }
    }
    free(tmp);
    return best_angle;
}

/* deep copy helpers for `target.clone()` */
static void* dup_mem(const void* p, size_t bytes)
{
    if (!p) return NULL;
    void* q = malloc(bytes ? bytes : 1);
    memcpy(q, p, bytes);
    return q;
}
static orc_clgeom* clgeom_clone(const orc_clgeom* s)
{
    orc_clgeom* d = (orc_clgeom*)calloc(1, sizeof(orc_clgeom));
    orc_geometry* g = (orc_geometry*)calloc(1, sizeof(orc_geometry));
    const orc_geometry* sg = s->g;
    size_t F = (size_t)sg->n_frames;
    *g = *sg;
    g->id = (uint32_t*)dup_mem(sg->id, F * 4);
    g->lumen_id = (uint32_t*)dup_mem(sg->lumen_id, F * 4);
    g->orig_frame = (uint32_t*)dup_mem(sg->orig_frame, F * 4);
    g->centroid = (double*)dup_mem(sg->centroid, F * 24);
    g->lumen_off = (int64_t*)dup_mem(sg->lumen_off, (F + 1) * 8);
    g->lumen = (orc_point*)dup_mem(sg->lumen, (size_t)sg->lumen_off[F] * sizeof(orc_point));
    g->cath_off = (int64_t*)dup_mem(sg->cath_off, (F + 1) * 8);
    g->cath = sg->cath_off ? (orc_point*)dup_mem(sg->cath, (size_t)sg->cath_off[F] * sizeof(orc_point)) : NULL;
    g->extra_off = (int64_t*)dup_mem(sg->extra_off, (F + 1) * 8);
    g->extra = sg->extra_off ? (orc_point*)dup_mem(sg->extra, (size_t)sg->extra_off[F] * sizeof(orc_point)) : NULL;
    g->has_ref = (uint8_t*)dup_mem(sg->has_ref, F);
    g->ref = (orc_point*)dup_mem(sg->ref, F * sizeof(orc_point));
    d->g = g;
    d->has_lumen_centroid = (uint8_t*)dup_mem(s->has_lumen_centroid, F);
    d->lumen_centroid = (double*)dup_mem(s->lumen_centroid, F * 24);
    d->n_extra_kinds = s->n_extra_kinds;
    d->extra_kind_off = (int64_t*)dup_mem(s->extra_kind_off, (F * (size_t)(s->n_extra_kinds > 0 ? s->n_extra_kinds : 0) + 1) * 8);
    return d;
}
static void clgeom_free(orc_clgeom* d)
{
    orc_geometry* g = d->g;
    free(g->id); free(g->lumen_id); free(g->orig_frame); free(g->centroid); free(g->lumen_off);
    free(g->lumen); free(g->cath_off); free(g->cath); free(g->extra_off); free(g->extra);
    free(g->has_ref); free(g->ref); free(g);
    free(d->has_lumen_centroid); free(d->lumen_centroid); free(d->extra_kind_off); free(d);
}
static void clgeom_assign(orc_clgeom* dst, const orc_clgeom* src)  /* same shapes */
{
    orc_geometry* g = dst->g;
    const orc_geometry* sg = src->g;
    size_t F = (size_t)sg->n_frames;
    memcpy(g->centroid, sg->centroid, F * 24);
    memcpy(g->lumen, sg->lumen, (size_t)sg->lumen_off[F] * sizeof(orc_point));
    if (sg->cath_off) memcpy(g->cath, sg->cath, (size_t)sg->cath_off[F] * sizeof(orc_point));
    if (sg->extra_off) memcpy(g->extra, sg->extra, (size_t)sg->extra_off[F] * sizeof(orc_point));
    if (sg->ref) memcpy(g->ref, sg->ref, F * sizeof(orc_point));
    if (src->lumen_centroid) memcpy(dst->lumen_centroid, src->lumen_centroid, F * 24);
}

int orc_refine_alignment_hausdorff(orc_clgeom** geoms, int n_geoms, const orc_clpoint* cl, size_t ncl,
                                   size_t initial_cl_ref_idx, double initial_rotation,
                                   const orc_point* points, size_t n_points,
                                   double angle_search_range, double angle_step, size_t index_search_range,
                                   double* best_angle_out, size_t* best_idx_out, double* min_h_out,
                                   double* all_costs, size_t cap, size_t* n_evals_out)
{
    const size_t len_frames = (size_t)geoms[0]->g->n_frames;                         /* :349 */
    double best_angle = initial_rotation;
    size_t best_idx = initial_cl_ref_idx;
    double min_h = DBL_MAX;
    size_t n_evals = 0;
    orc_clgeom* work[2] = { NULL, NULL };
    if (n_geoms > 2) return -7;
    for (int gi = 0; gi < n_geoms; ++gi) work[gi] = clgeom_clone(geoms[gi]);
    orc_point* filtered = (orc_point*)malloc((n_points ? n_points : 1) * sizeof(orc_point));
    int64_t* fidx = (int64_t*)malloc((n_points ? n_points : 1) * sizeof(int64_t));
    size_t total_lumen = (size_t)geoms[0]->g->lumen_off[len_frames];
    orc_point* flat = (orc_point*)malloc((total_lumen ? total_lumen : 1) * sizeof(orc_point));

    long long lo = index_search_range == 0 ? 0 : -(long long)index_search_range;     /* :363-367 */
    long long hi = index_search_range == 0 ? 0 : (long long)index_search_range;
    for (long long delta = lo; delta <= hi; ++delta) {
        long long sgn = (long long)initial_cl_ref_idx + delta;
        if (sgn < 0) continue;                                                       /* :371-373 */
        size_t cur = (size_t)sgn;
        if (cur + len_frames >= ncl) continue;                                       /* :376-378 */
        size_t cl_end = cur + len_frames;
        const orc_clpoint* seg = cl + cur;                                           /* :381-384 */
        double ref_pt[3] = { cl[cur].x, cl[cur].y, cl[cur].z };
        orc_point ps = { cl[cur].x, cl[cur].y, cl[cur].z };
        orc_point pe = { cl[cl_end - 1].x, cl[cl_end - 1].y, cl[cl_end - 1].z };
        size_t nf = orc_filter_points_in_region(points, n_points, &ps, &pe, fidx, n_points); /* :400-404 */
        for (size_t i = 0; i < nf; ++i) filtered[i] = points[fidx[i]];

        for (double angle = initial_rotation - angle_search_range;
             angle <= initial_rotation + angle_search_range; angle += angle_step) { /* :386-387, 439 */
            if (nf == 0) continue;                                                   /* :406-409 */
            for (int gi = 0; gi < n_geoms; ++gi) clgeom_assign(work[gi], geoms[gi]); /* target.clone() */
            for (int gi = 0; gi < n_geoms; ++gi) orc_rotate_geometry(work[gi], angle); /* :395 */
            orc_apply_transformations(work, n_geoms, seg, len_frames, ref_pt);       /* :394-398 */
            const orc_geometry* g = work[0]->g;
            size_t m = (size_t)(g->lumen_off[1] - g->lumen_off[0]);                  /* :412 */
            size_t n_down = orc_refine_downsample_count(nf, m, len_frames);          /* :415-418 */
            size_t nflat = 0;
            for (size_t f = 0; f < len_frames; ++f) {                                /* :420-428 */
                const orc_point* fp = g->lumen + g->lumen_off[f];
                size_t flen = (size_t)(g->lumen_off[f + 1] - g->lumen_off[f]);
                if (n_down < m) nflat += orc_downsample(fp, flen, n_down, flat + nflat);
                else { memcpy(flat + nflat, fp, flen * sizeof(orc_point)); nflat += flen; }
            }
            double h = orc_hausdorff(filtered, nf, flat, nflat);                     /* :431 */
            if (all_costs && n_evals < cap) all_costs[n_evals] = h;
            ++n_evals;
            if (h < min_h) { min_h = h; best_angle = angle; best_idx = cur; }        /* :433-437 */
        }
    }
    free(flat); free(fidx); free(filtered);
    for (int gi = 0; gi < n_geoms; ++gi) clgeom_free(work[gi]);
    *best_angle_out = best_angle; *best_idx_out = best_idx;
    if (min_h_out) *min_h_out = min_h;
    if (n_evals_out) *n_evals_out = n_evals;
    return 0;
}

/* ---- align.rs --------------------------------------------------------------------------- */
static int cl_prepare(const orc_clpoint* cl, size_t ncl, const orc_geometry* g, orc_clpoint** out,
                      size_t* nout, double* spacing)
{
    double sp = 0.0;
    int64_t need = orc_preprocess_centerline(cl, ncl, g, NULL, 0, &sp);
    if (need < 0) return (int)need;
    *out = (orc_clpoint*)malloc((size_t)(need ? need : 1) * sizeof(orc_clpoint));
    orc_preprocess_centerline(cl, ncl, g, *out, (size_t)need, &sp);
    *nout = (size_t)need; *spacing = sp;
    return 0;
}

static int three_point_initial(const orc_clpoint* rcl, size_t nrcl, orc_clgeom** geoms,
                               uint32_t ref_point_index, const double p_main[3], const double p_ccw[3],
                               const double p_cw[3], double angle_step, size_t* cl_ref_idx, double* rot)
{
    const orc_clgeom* prim = geoms[0];
    size_t ref_idx;
    if (orc_find_ref_frame_idx(prim->g, &ref_idx) != 0) return -4;                   /* align.rs:82-85 */
    if (ref_idx >= (size_t)prim->g->n_frames) return -4;  /* frames[ref_idx]: Frame.id used as an index, Rust would panic */
    if (!(prim->g->has_ref && prim->g->has_ref[ref_idx])) return -5;                 /* :86-89 */
    *cl_ref_idx = orc_cl_find_ref_idx(rcl, nrcl, p_main);                            /* :90 */
    int hc = prim->has_lumen_centroid && prim->has_lumen_centroid[ref_idx];
    *rot = orc_best_rotation_three_point(prim->g->lumen + prim->g->lumen_off[ref_idx],
                                         (size_t)(prim->g->lumen_off[ref_idx + 1] - prim->g->lumen_off[ref_idx]),
                                         hc, hc ? &prim->lumen_centroid[3 * ref_idx] : NULL, ref_point_index,
                                         p_main, p_ccw, p_cw, angle_step, &rcl[*cl_ref_idx]); /* :92-100 */
    return 0;
}

/* ---- align.rs:381-595 wall twist compensation -------------------------------------------- */
typedef struct { const orc_point* p; const uint8_t* aortic; size_t n; int present; } wall_view;

static wall_view wall_of(const orc_clgeom* cg, int32_t i)
{
    wall_view w = { NULL, NULL, 0, 0 };
    const orc_geometry* g = cg->g;
    if (!cg->wall_kind1 || !g->extra_off) return w;
    const int32_t K = cg->extra_kind_off ? cg->n_extra_kinds : 1, k = cg->wall_kind1 - 1;
    int64_t lo = cg->extra_kind_off ? cg->extra_kind_off[(int64_t)i * K + k] : g->extra_off[i];
    int64_t hi = cg->extra_kind_off ? cg->extra_kind_off[(int64_t)i * K + k + 1] : g->extra_off[i + 1];
    w.p = g->extra + lo; w.n = (size_t)(hi - lo); w.present = hi > lo;
    w.aortic = cg->wall_aortic ? cg->wall_aortic + wall_points_before(cg, i) : NULL;
    return w;
}
/* :385-407 aortic_centroid_direction */
static int aortic_direction(wall_view w, const double c[3], double out[3])
{
    size_t m = 0;
    if (w.aortic) for (size_t i = 0; i < w.n; ++i) m += w.aortic[i] ? 1 : 0;
    if (m == 0) return 0;
    double n = (double)m, sx = 0.0, sy = 0.0, sz = 0.0;
    for (size_t i = 0; i < w.n; ++i) if (w.aortic[i]) sx += w.p[i].x;
    for (size_t i = 0; i < w.n; ++i) if (w.aortic[i]) sy += w.p[i].y;
    for (size_t i = 0; i < w.n; ++i) if (w.aortic[i]) sz += w.p[i].z;
    out[0] = sx / n - c[0]; out[1] = sy / n - c[1]; out[2] = sz / n - c[2];
    return !(v3_norm(out) < 1e-9);
}
/* :410-437 wall_major_axis */
static int major_axis(wall_view w, double out[3])
{
    if (w.n < 2) return 0;
    double best = 0.0; size_t fa = 0, fb = 0;
    for (size_t i = 0; i < w.n; ++i)
        for (size_t j = i + 1; j < w.n; ++j) {
            double dx = w.p[i].x - w.p[j].x, dy = w.p[i].y - w.p[j].y, dz = w.p[i].z - w.p[j].z;
            double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 > best) { best = d2; fa = i; fb = j; }
        }
    out[0] = w.p[fb].x - w.p[fa].x; out[1] = w.p[fb].y - w.p[fa].y; out[2] = w.p[fb].z - w.p[fa].z;
    return !(v3_norm(out) < 1e-9);
}
/* :440-461 lumen_normal: Newell about the FRAME centroid */
static void lumen_normal(const orc_geometry* g, int32_t i, double out[3])
{
    orc_newell_normal(g->lumen + g->lumen_off[i], (size_t)(g->lumen_off[i + 1] - g->lumen_off[i]), &g->centroid[3 * i], out);
}
/* :465-472 */
static int project_normalized(const double v[3], const double t[3], double out[3])
{
    double k = v3_dot(v, t);
    double p[3] = { v[0] - t[0] * k, v[1] - t[1] * k, v[2] - t[2] * k };
    double n = v3_norm(p);
    if (n < 1e-9) return 0;
    out[0] = p[0] / n; out[1] = p[1] / n; out[2] = p[2] / n;
    return 1;
}
/* :476-492 */
static void parallel_transport(const double v[3], const double tf[3], const double tt[3], double out[3])
{
    double ang = v3_angle(tf, tt);
    if (ang < 1e-9) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; return; }
    double axis[3], r[9];
    v3_cross(tf, tt, axis);
    if (v3_norm(axis) < 1e-9) {
        double perp[3];
        if (fabs(tf[0]) < 0.9) { perp[0] = 1.0 - tf[0] * tf[0]; perp[1] = 0.0 - tf[1] * tf[0]; perp[2] = 0.0 - tf[2] * tf[0]; }
        else                   { perp[0] = 0.0 - tf[0] * tf[1]; perp[1] = 1.0 - tf[1] * tf[1]; perp[2] = 0.0 - tf[2] * tf[1]; }
        double n = v3_norm(perp);
        perp[0] /= n; perp[1] /= n; perp[2] /= n;
        m3_axis_angle(perp, 3.14159265358979323846, r);
    } else {
        m3_axis_angle(axis, ang, r);
    }
    double o[3];
    m3_mul(r, v, o);
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
}
/* :495-497 */
static double signed_angle(const double from[3], const double to[3], const double axis[3])
{
    double c[3];
    v3_cross(from, to, c);
    return atan2(v3_dot(c, axis), v3_dot(from, to));
}
/* :507-584 align_walls_on_geometry */
static void align_walls_on_geometry(orc_clgeom* cg)
{
    orc_geometry* g = cg->g;
    if (g->n_frames < 1) return;
    double t0[3], d0[3], u[3];
    lumen_normal(g, 0, t0);
    wall_view w0 = wall_of(cg, 0);
    if (!w0.present) return;
    if (!aortic_direction(w0, &g->centroid[0], d0) && !major_axis(w0, d0)) return;
    if (!project_normalized(d0, t0, u)) return;
    for (int32_t i = 1; i < g->n_frames; ++i) {
        double tp[3], tc[3], tr[3], pu[3];
        lumen_normal(g, i - 1, tp);
        lumen_normal(g, i, tc);
        parallel_transport(u, tp, tc, tr);
        u[0] = tr[0]; u[1] = tr[1]; u[2] = tr[2];                  /* `u = parallel_transport(..)` is kept even when ... */
        if (!project_normalized(u, tc, pu)) continue;              /* ... the projection fails (:537-540) */
        u[0] = pu[0]; u[1] = pu[1]; u[2] = pu[2];
        const double* c = &g->centroid[3 * i];
        wall_view w = wall_of(cg, i);
        if (!w.present) continue;
        double d[3], v[3];
        int has_aortic = aortic_direction(w, c, d);
        if (!has_aortic && !major_axis(w, d)) continue;
        if (!project_normalized(d, tc, v)) continue;
        double ang;
        if (has_aortic) ang = signed_angle(v, u, tc);
        else {
            double nv[3] = { -v[0], -v[1], -v[2] };
            double a1 = signed_angle(v, u, tc), a2 = signed_angle(nv, u, tc);
            ang = fabs(a1) <= fabs(a2) ? a1 : a2;
        }
        if (fabs(ang) < 1e-6) continue;
        double r[9];
        m3_axis_angle(tc, ang, r);
        orc_point* wp = (orc_point*)w.p;
        for (size_t k = 0; k < w.n; ++k) {
            double rel[3] = { wp[k].x - c[0], wp[k].y - c[1], wp[k].z - c[2] }, o[3];
            m3_mul(r, rel, o);
            wp[k].x = c[0] + o[0]; wp[k].y = c[1] + o[1]; wp[k].z = c[2] + o[2];
        }
    }
}
/* :589-595 */
void orc_align_walls(orc_clgeom** geoms, int n_geoms, int anomalous)
{
    if (!anomalous || geoms[0]->g->n_frames < 2) return;
    for (int gi = 0; gi < n_geoms; ++gi) align_walls_on_geometry(geoms[gi]);
}

int orc_align_three_point(const orc_clpoint* cl, size_t ncl, orc_clgeom** geoms, int n_geoms,
                          uint32_t ref_point_index, const double p_main[3], const double p_ccw[3],
                          const double p_cw[3], double angle_step, int align_wall_anomalous,
                          double* spacing, double* total_rotation)
{
    orc_clpoint* rcl = NULL; size_t nrcl = 0;
    int rc = cl_prepare(cl, ncl, geoms[0]->g, &rcl, &nrcl, spacing);                 /* :78-80 */
    if (rc) return rc;
    size_t cl_ref_idx; double rot;
    rc = three_point_initial(rcl, nrcl, geoms, ref_point_index, p_main, p_ccw, p_cw, angle_step, &cl_ref_idx, &rot);
    if (rc) { free(rcl); return rc; }
    for (int gi = 0; gi < n_geoms; ++gi) orc_rotate_geometry(geoms[gi], rot);        /* :102 */
    orc_apply_transformations(geoms, n_geoms, rcl, nrcl, p_main);                    /* :103 */
    orc_align_walls(geoms, n_geoms, align_wall_anomalous);                           /* :105-107 */
    *total_rotation = rot;
    free(rcl);
    return 0;
}

int orc_align_manual(const orc_clpoint* cl, size_t ncl, orc_clgeom** geoms, int n_geoms,
                     double rotation_angle_deg, const double ref_pt[3], int align_wall_anomalous,
                     double* spacing, double* total_rotation)
{
    orc_clpoint* rcl = NULL; size_t nrcl = 0;
    int rc = cl_prepare(cl, ncl, geoms[0]->g, &rcl, &nrcl, spacing);                 /* :139-141 */
    if (rc) return rc;
    double rot = rotation_angle_deg * (3.14159265358979323846 / 180.0);              /* :143 to_radians */
    for (int gi = 0; gi < n_geoms; ++gi) orc_rotate_geometry(geoms[gi], rot);        /* :144 */
    orc_apply_transformations(geoms, n_geoms, rcl, nrcl, ref_pt);                    /* :145 */
    orc_align_walls(geoms, n_geoms, align_wall_anomalous);                           /* :147-149 */
    *total_rotation = rot;
    free(rcl);
    return 0;
}

int orc_align_combined(const orc_clpoint* cl, size_t ncl, orc_clgeom** geoms, int n_geoms,
                       uint32_t ref_point_index, const double p_main[3], const double p_ccw[3],
                       const double p_cw[3], const orc_point* points, size_t n_points,
                       double angle_step, double refine_angle_range, size_t refine_index_range,
                       int align_wall_anomalous, double* spacing, double* total_rotation,
                       size_t* refined_idx_out)
{
    if (n_geoms > 2) return -7;
    orc_clpoint* rcl = NULL; size_t nrcl = 0;
    int rc = cl_prepare(cl, ncl, geoms[0]->g, &rcl, &nrcl, spacing);                 /* :191-195 */
    if (rc) return rc;
    size_t initial_idx; double initial_rotation;
    rc = three_point_initial(rcl, nrcl, geoms, ref_point_index, p_main, p_ccw, p_cw, angle_step,
                             &initial_idx, &initial_rotation);                       /* :197-217 */
    if (rc) { free(rcl); return rc; }
    orc_clgeom* aligned[2] = { NULL, NULL };
    for (int gi = 0; gi < n_geoms; ++gi) {
        aligned[gi] = clgeom_clone(geoms[gi]);
        orc_rotate_geometry(aligned[gi], initial_rotation);                          /* :219-223 */
    }
    orc_apply_transformations(aligned, n_geoms, rcl, nrcl, p_main);
    double delta; size_t refined_idx;
    orc_refine_alignment_hausdorff(aligned, n_geoms, rcl, nrcl, initial_idx, 0.0, points, n_points,
                                   refine_angle_range, angle_step, refine_index_range,
                                   &delta, &refined_idx, NULL, NULL, 0, NULL);       /* :228-237 */
    for (int gi = 0; gi < n_geoms; ++gi) clgeom_free(aligned[gi]);
    double total = initial_rotation + delta;                                         /* :239 */
    double refined_ref_pt[3] = { rcl[refined_idx].x, rcl[refined_idx].y, rcl[refined_idx].z }; /* :248-258 */
    for (int gi = 0; gi < n_geoms; ++gi) orc_rotate_geometry(geoms[gi], total);      /* :260-264 */
    orc_apply_transformations(geoms, n_geoms, rcl, nrcl, refined_ref_pt);
    orc_align_walls(geoms, n_geoms, align_wall_anomalous);                           /* :266-268 */
    *total_rotation = total;
    if (refined_idx_out) *refined_idx_out = refined_idx;
    free(rcl);
    return 0;
}
