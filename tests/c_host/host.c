/*
 * A host written in plain C against the drop-in boundary (include/mm_hausdorff.h): no Python, no torch, nothing but
 * the shared library -- what a Rust / C host of the reference would link.  It runs the three things the boundary is
 * for -- the metric, one rotation search at every precision, a batch of searches -- on seeded data and checks every
 * result against the CPU oracle (oracle/mm_oracle.h; test infrastructure, linked here as the checker only).
 * Built and run by tests/test_gpu_c_host.py; exit code 0 = all checks passed.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mm_hausdorff.h"
#include "mm_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void)            /* splitmix64 -> [0, 1) */
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

static void blob(int n, double rot, double* x, double* y)     /* a noisy closed contour, IVUS-lumen sized */
{
    for (int i = 0; i < n; ++i) {
        double t = 6.283185307179586 * (double)i / (double)n;
        double r = 2.5 * (1.0 + 0.15 * cos(2.0 * t + 0.7) + 0.07 * sin(3.0 * t)) + 0.05 * (urand() - 0.5);
        double px = r * cos(t), py = 0.8 * r * sin(t);
        x[i] = 4.5 + px * cos(rot) - py * sin(rot);
        y[i] = 4.5 + px * sin(rot) + py * cos(rot);
    }
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
    fprintf(stderr, __VA_ARGS__); fprintf(stderr, " [%s]\n", mm_last_error()); return 1; } } while (0)

int main(void)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    CHECK(mm_device_count() >= 1, "no HIP device");
    mm_engine* e = NULL;
    CHECK(mm_engine_create(-1, NULL, &e) == MM_OK && e, "mm_engine_create");
    printf("library %s, %d device(s)\n", mm_version(), mm_device_count());

    /* 1. the metric: the reference's known answers (process_utils.rs:214-547) and a seeded pair vs the oracle */
    {
        const double ax[] = {0, 3}, ay[] = {0, 0}, bx[] = {1, 2, 4}, by[] = {0, 0, 0};
        double h = -1.0;
        CHECK(mm_hausdorff_2d(e, ax, ay, 2, bx, by, 3, &h) == MM_OK && h == 1.0, "hausdorff KAT: %g", h);
        CHECK(mm_hausdorff_2d(e, ax, ay, 0, bx, by, 3, &h) == MM_OK && h == 0.0, "empty set: %g", h);
    }
    enum { N = 521 };
    static double rx[N], ry[N], tx[N], ty[N];
    static orc_point ro[N], to[N];
    blob(N, 0.0, rx, ry);
    blob(N, 0.31, tx, ty);
    for (int i = 0; i < N; ++i) { ro[i].x = rx[i]; ro[i].y = ry[i]; ro[i].z = 0; to[i].x = tx[i]; to[i].y = ty[i]; to[i].z = 0; }
    {
        double h = -1.0;
        CHECK(mm_hausdorff_2d(e, rx, ry, N, tx, ty, N, &h) == MM_OK, "mm_hausdorff_2d");
        CHECK(h == orc_hausdorff(ro, N, to, N), "metric differs from the oracle: %.17g", h);
    }

    /* 2. one search: search_range(|a| hausdorff(ref, rotate(tgt, a))) at 0.5 deg x +-180 deg, every precision */
    static double angles[2048], costs[2048];
    int degenerate = 0; double early = 0.0;
    int64_t na = mm_search_angles(0.5, 180.0, 0, 0.0, 180.0, angles, 2048, &degenerate, &early);
    CHECK(na == 721 && !degenerate, "mm_search_angles: %lld", (long long)na);
    double cx = 0.0, cy = 0.0;
    for (int i = 0; i < N; ++i) { cx += tx[i]; cy += ty[i]; }
    cx /= N; cy /= N;
    const double o_angle = orc_bruteforce_rotation(ro, N, to, N, 0.5, 180.0, cx, cy, 8);
    const double o_cost = orc_cost_within(ro, N, to, N, o_angle, cx, cy);
    for (int prec = MM_PRECISION_F64; prec <= MM_PRECISION_F32_BOUNDED; ++prec) {
        double ba = 0, bc = 0; int bi = -1;
        CHECK(mm_best_rotation(e, rx, ry, N, tx, ty, N, cx, cy, angles, (int)na, MM_SEARCH_SKIP_ZERO, prec, &ba, &bc, &bi,
                               prec == MM_PRECISION_F64 ? costs : NULL) == MM_OK, "mm_best_rotation(prec %d)", prec);
        CHECK(ba == o_angle && bc == o_cost && angles[bi] == ba, "precision %d: angle %.17g cost %.17g, oracle %.17g %.17g",
              prec, ba, bc, o_angle, o_cost);
        if (prec == MM_PRECISION_F64)
            for (int a = 0; a < (int)na; a += 97)
                CHECK(costs[a] == orc_cost_within(ro, N, to, N, angles[a], cx, cy), "cost of candidate %d", a);
    }
    printf("search: best angle %.6f rad, cost %.6f mm -- identical to the oracle at all four precisions\n", o_angle, o_cost);

    /* 3. a ragged batch: 6 pairs of different sizes (one with an empty target), one launch sequence */
    enum { P = 6 };
    const int sizes[P][2] = {{6, 6}, {120, 0}, {521, 521}, {100, 333}, {17, 1}, {300, 299}};
    static double bxr[4096], byr[4096], bxt[4096], byt[4096], bang[P * 721], bcx[P], bcy[P];
    int64_t roff[P + 1] = {0}, toff[P + 1] = {0}, aoff[P + 1] = {0};
    int32_t flags[P], bidx[P];
    double bbest[P], bcost[P];
    for (int p = 0; p < P; ++p) {
        roff[p + 1] = roff[p] + sizes[p][0]; toff[p + 1] = toff[p] + sizes[p][1]; aoff[p + 1] = aoff[p] + na;
        blob(sizes[p][0], 0.05 * p, bxr + roff[p], byr + roff[p]);
        blob(sizes[p][1], 0.4 - 0.1 * p, bxt + toff[p], byt + toff[p]);
        for (int a = 0; a < (int)na; ++a) bang[aoff[p] + a] = angles[a];
        bcx[p] = 4.5 + 0.01 * p; bcy[p] = 4.5 - 0.02 * p; flags[p] = MM_SEARCH_SKIP_ZERO;
    }
    CHECK(mm_best_rotation_batch(e, P, roff, bxr, byr, toff, bxt, byt, aoff, bang, bcx, bcy, flags, MM_PRECISION_F32_FAST,
                                 bidx, bbest, bcost, NULL, NULL) == MM_OK, "mm_best_rotation_batch");
    for (int p = 0; p < P; ++p) {
        static orc_point a_[600], b_[600];
        for (int i = 0; i < sizes[p][0]; ++i) { a_[i].x = bxr[roff[p] + i]; a_[i].y = byr[roff[p] + i]; a_[i].z = 0; }
        for (int i = 0; i < sizes[p][1]; ++i) { b_[i].x = bxt[toff[p] + i]; b_[i].y = byt[toff[p] + i]; b_[i].z = 0; }
        const double oa = orc_bruteforce_rotation(a_, (size_t)sizes[p][0], b_, (size_t)sizes[p][1], 0.5, 180.0, bcx[p], bcy[p], 8);
        const double oc = orc_cost_within(a_, (size_t)sizes[p][0], b_, (size_t)sizes[p][1], oa, bcx[p], bcy[p]);
        CHECK(bbest[p] == oa && bcost[p] == oc, "batch pair %d: %.17g %.17g vs %.17g %.17g", p, bbest[p], bcost[p], oa, oc);
    }
    printf("batch: %d ragged pairs identical to the oracle\n", P);

    /* 4. the path itself: two pullbacks through the decoupled plan (all frame pairs in one launch sequence, then the
     *    exact chain walk) and the between alignment of the second onto the first -- logs, best rotation and every
     *    coordinate against the oracle's sequential chain (align_within.rs:24-134, align_between.rs:11-68) */
    {
        enum { F = 9, M = 120, NC = 20, G = 2 };
        static uint32_t ids[G][F], origs[G][F];
        static double cen[G][F * 3], lum[G][F * M * 3], cath[G][F * NC * 3], refp[G][F * 3], lcen[G][F * 3];
        static double o_cen[G][F * 3], o_lum[G][F * M * 3], o_cath[G][F * NC * 3], o_ref[G][F * 3], o_lcen[G][F * 3];
        static double p_cen[G][F * 3], p_lum[G][F * M * 3], p_cath[G][F * NC * 3], p_lcen[G][F * 3];   /* pristine inputs */
        static int64_t loff[F + 1], coff[F + 1];
        static uint8_t href[G][F];
        mm_geometry g[G]; orc_geometry og[G];
        mm_geometry* gp[G]; mm_alignlog logs[G][F - 1]; mm_alignlog* lp[G]; orc_alignlog ologs[G][F - 1];
        for (int k = 0; k <= F; ++k) { loff[k] = (int64_t)k * M; coff[k] = (int64_t)k * NC; }
        for (int q = 0; q < G; ++q) {
            double twist = 0.0;
            for (int k = 0; k < F; ++k) {
                static double x[M], y[M];
                twist += 0.12 * (urand() - 0.3);                     /* a random-walk torsion along the pullback */
                blob(M, twist + 0.4 * q, x, y);
                double sx = 0, sy = 0;
                for (int i = 0; i < M; ++i) {
                    double* p3 = &lum[q][(k * M + i) * 3];
                    p3[0] = x[i] + 0.03 * k; p3[1] = y[i] - 0.02 * k; p3[2] = 0.5 * k;
                    sx += p3[0]; sy += p3[1];
                }
                cen[q][3 * k] = sx / M; cen[q][3 * k + 1] = sy / M; cen[q][3 * k + 2] = 0.5 * k;
                lcen[q][3 * k] = cen[q][3 * k]; lcen[q][3 * k + 1] = cen[q][3 * k + 1]; lcen[q][3 * k + 2] = cen[q][3 * k + 2];
                for (int i = 0; i < NC; ++i) {
                    double a = 6.283185307179586 * i / NC;
                    double* c3 = &cath[q][(k * NC + i) * 3];
                    c3[0] = 4.5 + 0.5 * cos(a); c3[1] = 4.5 + 0.5 * sin(a); c3[2] = 0.5 * k;
                }
                ids[q][k] = (uint32_t)k; origs[q][k] = (uint32_t)(F - 1 - k); href[q][k] = k == 0;
                refp[q][3 * k] = k == 0 ? 6.9 : 0.0; refp[q][3 * k + 1] = k == 0 ? 4.5 : 0.0; refp[q][3 * k + 2] = 0.0;
            }
            memcpy(o_cen[q], cen[q], sizeof cen[q]); memcpy(o_lum[q], lum[q], sizeof lum[q]);
            memcpy(o_cath[q], cath[q], sizeof cath[q]); memcpy(o_ref[q], refp[q], sizeof refp[q]);
            memcpy(o_lcen[q], lcen[q], sizeof lcen[q]);
            memcpy(p_cen[q], cen[q], sizeof cen[q]); memcpy(p_lum[q], lum[q], sizeof lum[q]);
            memcpy(p_cath[q], cath[q], sizeof cath[q]); memcpy(p_lcen[q], lcen[q], sizeof lcen[q]);
            memset(&g[q], 0, sizeof g[q]); memset(&og[q], 0, sizeof og[q]);
            g[q].n_frames = F; g[q].id = ids[q]; g[q].lumen_id = ids[q]; g[q].orig_frame = origs[q]; g[q].centroid = cen[q];
            g[q].lumen_off = loff; g[q].lumen = lum[q]; g[q].has_catheter = 1; g[q].cath_off = coff; g[q].cath = cath[q];
            g[q].has_ref = href[q]; g[q].ref = refp[q]; g[q].lumen_centroid = lcen[q];
            og[q].n_frames = F; og[q].id = ids[q]; og[q].lumen_id = ids[q]; og[q].orig_frame = origs[q]; og[q].centroid = o_cen[q];
            og[q].lumen_off = loff; og[q].lumen = (orc_point*)o_lum[q]; og[q].has_catheter = 1; og[q].cath_off = coff;
            og[q].cath = (orc_point*)o_cath[q]; og[q].has_ref = href[q]; og[q].ref = (orc_point*)o_ref[q];
            og[q].lumen_centroid = o_lcen[q];
            gp[q] = &g[q]; lp[q] = logs[q];
        }
        mm_within_plan* plan = NULL;
        int64_t evals = 0, unresolved = -1;
        CHECK(mm_within_plan_create(e, G, gp, 1.0, 60.0, 1, 200, MM_PRECISION_F32_FAST, &plan) == MM_OK && plan, "mm_within_plan_create");
        CHECK(mm_within_plan_run(plan, lp, &evals, &unresolved) == MM_OK, "mm_within_plan_run");
        mm_within_plan_destroy(plan);
        for (int q = 0; q < G; ++q) {
            CHECK(orc_align_within_chain(&og[q], 1.0, 60.0, 1, 200, ologs[q], 8) == 0, "oracle chain");
            CHECK(sizeof(mm_alignlog) == sizeof(orc_alignlog) && memcmp(logs[q], ologs[q], sizeof ologs[q]) == 0, "chain logs of pullback %d", q);
            CHECK(memcmp(lum[q], o_lum[q], sizeof lum[q]) == 0 && memcmp(cath[q], o_cath[q], sizeof cath[q]) == 0 &&
                  memcmp(cen[q], o_cen[q], sizeof cen[q]) == 0 && memcmp(lcen[q], o_lcen[q], sizeof lcen[q]) == 0,
                  "chain coordinates of pullback %d", q);
        }
        /* 4b. the same alignment as rank 0 of a one-rank job through the multi-GPU entry points: the library's own RCCL
         *     communicator (mm_comm_*), a 1 x 1 shard grid, export kernels + ncclAllReduce(MIN) x 2 per level on the
         *     engine's stream (mm_within_plan_run_sharded) -- what every rank of an N-GPU host calls */
        {
            unsigned char uid[MM_COMM_ID_BYTES];
            mm_comm* comm = NULL;
            mm_alignlog logs2[G][F - 1]; mm_alignlog* lp2[G];
            int pb = 0, cs = 0;
            CHECK(mm_shard_grid(8, 2044, &pb, &cs) == MM_OK && pb == 8 && cs == 1, "mm_shard_grid");
            CHECK(mm_comm_version() > 0, "RCCL not loadable: %s", mm_last_error());
            CHECK(mm_comm_unique_id(uid) == MM_OK, "mm_comm_unique_id: %s", mm_last_error());
            CHECK(mm_comm_init_rank(uid, 0, 1, -1, &comm) == MM_OK && comm, "mm_comm_init_rank: %s", mm_last_error());
            CHECK(mm_comm_rank(comm) == 0 && mm_comm_world(comm) == 1, "communicator rank / world");
            for (int q = 0; q < G; ++q) {
                memcpy(cen[q], p_cen[q], sizeof cen[q]); memcpy(lum[q], p_lum[q], sizeof lum[q]);
                memcpy(cath[q], p_cath[q], sizeof cath[q]); memcpy(lcen[q], p_lcen[q], sizeof lcen[q]);
                lp2[q] = logs2[q];
            }
            int64_t evals2 = 0, unres2 = -1;
            plan = NULL;
            CHECK(mm_within_plan_create_grid(e, G, gp, 1.0, 60.0, 1, 200, MM_PRECISION_F32_FAST, 0, 1, 1, &plan) == MM_OK && plan,
                  "mm_within_plan_create_grid");
            CHECK(mm_within_plan_run_sharded(plan, comm, lp2, &evals2, &unres2) == MM_OK, "mm_within_plan_run_sharded: %s", mm_last_error());
            mm_within_plan_destroy(plan);
            CHECK(evals2 == evals && unres2 == unresolved, "sharded run: pose-evals / re-searched steps");
            for (int q = 0; q < G; ++q) {
                CHECK(memcmp(logs2[q], ologs[q], sizeof ologs[q]) == 0, "sharded run: chain logs of pullback %d", q);
                CHECK(memcmp(lum[q], o_lum[q], sizeof lum[q]) == 0 && memcmp(cath[q], o_cath[q], sizeof cath[q]) == 0 &&
                      memcmp(cen[q], o_cen[q], sizeof cen[q]) == 0 && memcmp(lcen[q], o_lcen[q], sizeof lcen[q]) == 0,
                      "sharded run: chain coordinates of pullback %d", q);
            }
            /* a plan made for another job size is refused, not run */
            CHECK(mm_within_plan_create_grid(e, G, gp, 1.0, 60.0, 1, 200, MM_PRECISION_F32_FAST, 1, 1, 2, &plan) == MM_OK, "grid plan");
            CHECK(mm_within_plan_search_sharded(plan, comm) == MM_ERR_INVALID, "a plan of another world must be refused");
            mm_within_plan_destroy(plan);
            mm_comm_destroy(comm);
            printf("sharded entry points: world = 1 RCCL communicator, logs and coordinates identical to the oracle\n");
        }
        double best = 0.0, obest = 1.0; int64_t bevals = 0;
        mm_geometry* ga[1] = {&g[0]}; mm_geometry* gb[1] = {&g[1]};
        CHECK(mm_align_between(e, 1, ga, gb, 60.0, 0.5, 200, MM_PRECISION_F32_BOUNDED, &best, &bevals) == MM_OK, "mm_align_between");
        CHECK(orc_align_between(&og[0], &og[1], 60.0, 0.5, 200, &obest, 8) == 0, "oracle between");
        CHECK(best == obest, "between rotation %.17g vs %.17g", best, obest);
        CHECK(memcmp(lum[1], o_lum[1], sizeof lum[1]) == 0 && memcmp(cen[1], o_cen[1], sizeof cen[1]) == 0 &&
              memcmp(refp[1], o_ref[1], sizeof refp[1]) == 0, "between coordinates");
        printf("alignment: %d pullbacks x %d frames, %lld within pose-evals (%lld steps re-searched), between rotation %.6f rad -- "
               "logs and coordinates identical to the oracle\n", G, F, (long long)evals, (long long)unresolved, best);
    }

    /* errors are codes + mm_last_error(), never a crash */
    CHECK(mm_hausdorff_2d(e, NULL, NULL, 3, NULL, NULL, 3, &early) < 0, "NULL sets must be an error");
    CHECK(mm_best_rotation(e, rx, ry, N, tx, ty, N, cx, cy, angles, (int)na, 0, 99, &early, &early, &degenerate, NULL) == MM_ERR_INVALID,
          "unknown precision must be MM_ERR_INVALID");
    CHECK(mm_engine_synchronize(e) == MM_OK, "mm_engine_synchronize");
    mm_engine_destroy(e);
    printf("C_HOST_OK\n");
    return 0;
}
