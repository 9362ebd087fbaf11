/*
 * oracle/poly_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE double, no FMA contraction) of the reference's
 * quadrilateral IoU and greedy polygon NMS.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; nothing under
 * dafne_amd/ may.
 *
 * What it follows (paths relative to /root/reference):
 *   - tools/prepare_dota/polyiou.cpp:10-133   sig / cross / area / lineCross /
 *     polygon_cut / triangle-fan intersectArea / iou_poly  (fp64)
 *   - dafne/utils/ResultMerge_multi_process.py:24-58   py_cpu_nms_poly: sort by
 *     score descending, keep the head, drop everything with IoU > thresh
 *   - dafne/modeling/nms/nms.py:74-91   class merge 5->4, fp32 class offset,
 *     float32 [M,9] array handed to the NMS
 *
 * Pinned against: oracle/_ref/libpolyiou_ref.so (the reference's own polyiou.cpp
 * compiled here, see oracle/Makefile) bit-for-bit on random + adversarial pairs
 * (tests/test_oracle_polyiou.py) and the reference's known answers
 * (polyiou.cpp:137-150, ResultMerge.py:54-63, poly_overlaps_test.py:7-24).
 *
 * Two places where the reference leaves behaviour open, and what this file
 * (and therefore the HIP product) defines:
 *   (1) polygon_cut appends lineCross's output slot even when lineCross returns
 *       0 (|s2-s1| <= 1e-8 while sig(s1) != sig(s2)); the slot then holds
 *       whatever the scratch array held (polyiou.cpp:60, 34-35).  Here the
 *       scratch array is zero-filled at the start of every triangle pair and
 *       persists across that pair's three cuts, so the stale value is defined.
 *   (2) `scores.argsort()[::-1]` (ResultMerge_multi_process.py:34, and the
 *       poly_nms_gpu wrapper) leaves the order of equal scores to numpy's
 *       unstable default sort.  Here the sort kind is pinned to stable:
 *       order == np.argsort(scores, kind="stable")[::-1], i.e. score descending,
 *       then index DESCENDING among equal scores.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS 1E-8
#define ORC_MAXV 16
/* capacities: the reference has Point p[10] (n <= 9 plus the closing vertex);
 * a clipped triangle never gets near either bound, the clamps only keep
 * garbage input from running off the arrays (same clamps in the HIP kernel) */
#define ORC_CAP_P 10
#define ORC_CAP_PP 12

typedef struct { double x, y; } pt_t;

static int sgn_eps(double d) { return (d > ORC_EPS) - (d < -ORC_EPS); }

static int pt_same(pt_t a, pt_t b) {
    return sgn_eps(a.x - b.x) == 0 && sgn_eps(a.y - b.y) == 0;
}

/* polyiou.cpp:22-24 */
static double cross3(pt_t o, pt_t a, pt_t b) {
    return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

/* polyiou.cpp:25-32; writes the closing vertex like the reference does */
static double shoelace(pt_t *ps, int n) {
    double res = 0;
    ps[n] = ps[0];
    for (int i = 0; i < n; i++)
        res += ps[i].x * ps[i + 1].y - ps[i].y * ps[i + 1].x;
    return res / 2.0;
}

/* polyiou.cpp:33-43.  s1/s2 are cross3(a,b,c), cross3(a,b,d) -- the caller has
 * them already (same expression, same operands, so the same bits). */
static int line_cross(double s1, double s2, pt_t c, pt_t d, pt_t *out) {
    if (sgn_eps(s1) == 0 && sgn_eps(s2) == 0) return 2;
    if (sgn_eps(s2 - s1) == 0) return 0;
    out->x = (c.x * s2 - d.x * s1) / (s2 - s1);
    out->y = (c.y * s2 - d.y * s1) / (s2 - s1);
    return 1;
}

/* polyiou.cpp:62-75: keep the part of p left of a->b, in place. */
static void cut_left(pt_t *p, int *pn, pt_t a, pt_t b, pt_t *pp) {
    int n = *pn, m = 0;
    p[n] = p[0];
    for (int i = 0; i < n; i++) {
        double ci = cross3(a, b, p[i]);
        double cj = cross3(a, b, p[i + 1]);
        if (sgn_eps(ci) > 0 && m < ORC_CAP_PP) pp[m++] = p[i];
        if (sgn_eps(ci) != sgn_eps(cj) && m < ORC_CAP_PP) {
            line_cross(ci, cj, p[i], p[i + 1], &pp[m]);
            m++;
        }
    }
    n = 0;
    for (int i = 0; i < m; i++)
        if ((!i || !pt_same(pp[i], pp[i - 1])) && n < ORC_CAP_P - 1) p[n++] = pp[i];
    while (n > 1 && pt_same(p[n - 1], p[0])) n--;
    *pn = n;
}

/* polyiou.cpp:79-93: signed overlap of triangles (o,a,b) and (o,c,d). */
static double tri_overlap(pt_t a, pt_t b, pt_t c, pt_t d) {
    pt_t o = {0.0, 0.0};
    int s1 = sgn_eps(cross3(o, a, b));
    int s2 = sgn_eps(cross3(o, c, d));
    if (s1 == 0 || s2 == 0) return 0.0;
    if (s1 == -1) { pt_t t = a; a = b; b = t; }
    if (s2 == -1) { pt_t t = c; c = d; d = t; }
    pt_t p[ORC_MAXV], pp[ORC_MAXV];
    memset(pp, 0, sizeof pp); /* definition (1) in the header */
    p[0] = o; p[1] = a; p[2] = b;
    int n = 3;
    cut_left(p, &n, o, c, pp);
    cut_left(p, &n, c, d, pp);
    cut_left(p, &n, d, o, pp);
    double res = fabs(shoelace(p, n));
    if (s1 * s2 == -1) res = -res;
    return res;
}

static void reverse4(pt_t *q) {
    pt_t t = q[0]; q[0] = q[3]; q[3] = t;
    t = q[1]; q[1] = q[2]; q[2] = t;
}

/* polyiou.cpp:95-107 + 112-133 */
double orc_iou_poly(const double *p8, const double *q8) {
    pt_t a[ORC_MAXV], b[ORC_MAXV];
    for (int i = 0; i < 4; i++) {
        a[i].x = p8[2 * i]; a[i].y = p8[2 * i + 1];
        b[i].x = q8[2 * i]; b[i].y = q8[2 * i + 1];
    }
    if (shoelace(a, 4) < 0) reverse4(a);
    if (shoelace(b, 4) < 0) reverse4(b);
    a[4] = a[0];
    b[4] = b[0];
    double inter = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            inter += tri_overlap(a[i], a[i + 1], b[j], b[j + 1]);
    double uni = fabs(shoelace(a, 4)) + fabs(shoelace(b, 4)) - inter;
    if (uni == 0) return (inter + 1) / (uni + 1);
    return inter / uni;
}

/* IoU of n pairs; p, q are [n,8] double. */
void orc_iou_poly_pairs(const double *p, const double *q, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_iou_poly(p + 8 * i, q + 8 * i);
}

/* IoU of n pairs given as float32 [n,8] (what the NMS sees: nms.py:86-90). */
void orc_iou_poly_pairs_f32(const float *p, const float *q, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) {
        double a[8], b[8];
        for (int k = 0; k < 8; k++) { a[k] = p[8 * i + k]; b[k] = q[8 * i + k]; }
        out[i] = orc_iou_poly(a, b);
    }
}

/* ---- ordering: np.argsort(scores, kind="stable")[::-1] --------------------- */
typedef struct { float s; int64_t i; } sidx_t;

static int cmp_desc(const void *pa, const void *pb) {
    const sidx_t *a = (const sidx_t *)pa, *b = (const sidx_t *)pb;
    if (a->s > b->s) return -1;
    if (a->s < b->s) return 1;
    if (a->i > b->i) return -1; /* equal scores: larger index first */
    if (a->i < b->i) return 1;
    return 0;
}

void orc_score_order(const float *dets9, int64_t m, int64_t *order) {
    sidx_t *v = (sidx_t *)malloc(sizeof(sidx_t) * (size_t)(m > 0 ? m : 1));
    for (int64_t i = 0; i < m; i++) { v[i].s = dets9[9 * i + 8]; v[i].i = i; }
    qsort(v, (size_t)m, sizeof(sidx_t), cmp_desc);
    for (int64_t i = 0; i < m; i++) order[i] = v[i].i;
    free(v);
}

/*
 * Greedy polygon NMS on a float32 [m,9] array (8 corner coords + score), the
 * array nms.py:90 builds.  ResultMerge_multi_process.py:24-58 loop: take the
 * best remaining box, drop every remaining box whose IoU with it is > thresh
 * (kept iff iou <= thresh).  Returns the number kept; keep[] receives original
 * row indices in descending-score order.  No pre-filter of any kind: every
 * (kept, remaining) pair goes through orc_iou_poly.
 */
int64_t orc_poly_nms(const float *dets9, int64_t m, double thresh, int64_t *keep) {
    if (m <= 0) return 0;
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
    unsigned char *dead = (unsigned char *)calloc((size_t)m, 1);
    double *poly = (double *)malloc(sizeof(double) * 8 * (size_t)m);
    orc_score_order(dets9, m, order);
    for (int64_t r = 0; r < m; r++)
        for (int k = 0; k < 8; k++) poly[8 * r + k] = (double)dets9[9 * order[r] + k];
    int64_t nk = 0;
    for (int64_t r = 0; r < m; r++) {
        if (dead[r]) continue;
        keep[nk++] = order[r];
        for (int64_t s = r + 1; s < m; s++) {
            if (dead[s]) continue;
            double iou = orc_iou_poly(poly + 8 * r, poly + 8 * s);
            if (iou > thresh) dead[s] = 1;
        }
    }
    free(order); free(dead); free(poly);
    return nk;
}

/*
 * Same result as orc_poly_nms, but skips the clip when the two axis-aligned
 * hulls are separated by more than a guard band AND at least one box has a
 * non-negligible area.  Used only to make the cpu_baseline / large-M checks
 * finish; tests/test_oracle_nms.py proves it equal to orc_poly_nms on every
 * fixture.  (Separated hulls => true intersection 0; the fp64 triangle-fan sum
 * then deviates from 0 by rounding only, far below thresh*union unless union
 * itself is ~0, which the area guard excludes.)
 */
int64_t orc_poly_nms_fast(const float *dets9, int64_t m, double thresh, int64_t *keep) {
    if (m <= 0) return 0;
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
    unsigned char *dead = (unsigned char *)calloc((size_t)m, 1);
    double *poly = (double *)malloc(sizeof(double) * 8 * (size_t)m);
    double *hull = (double *)malloc(sizeof(double) * 5 * (size_t)m);
    orc_score_order(dets9, m, order);
    double amax = 0;
    for (int64_t r = 0; r < m; r++) {
        double *q = poly + 8 * r, *h = hull + 5 * r;
        for (int k = 0; k < 8; k++) q[k] = (double)dets9[9 * order[r] + k];
        h[0] = h[2] = q[0]; h[1] = h[3] = q[1];
        for (int k = 1; k < 4; k++) {
            if (q[2 * k] < h[0]) h[0] = q[2 * k];
            if (q[2 * k] > h[2]) h[2] = q[2 * k];
            if (q[2 * k + 1] < h[1]) h[1] = q[2 * k + 1];
            if (q[2 * k + 1] > h[3]) h[3] = q[2 * k + 1];
        }
        pt_t t[ORC_MAXV];
        for (int k = 0; k < 4; k++) { t[k].x = q[2 * k]; t[k].y = q[2 * k + 1]; }
        h[4] = fabs(shoelace(t, 4));
        for (int k = 0; k < 8; k++) if (fabs(q[k]) > amax) amax = fabs(q[k]);
    }
    /* rounding budget of the 16-term fan sum: 16 x (shoelace + clip-vertex
     * roundings ~1.3e-13*amax^2, plus ~1e-7 from the 1e-8 snapping in sig());
     * the guard is > 50x that. */
    double area_guard = 256.0 * (2e-13 * amax * amax + 1e-6) / (thresh > 1e-6 ? thresh : 1e-6);
    int64_t nk = 0;
    for (int64_t r = 0; r < m; r++) {
        if (dead[r]) continue;
        keep[nk++] = order[r];
        const double *hr = hull + 5 * r;
        for (int64_t s = r + 1; s < m; s++) {
            if (dead[s]) continue;
            const double *hs = hull + 5 * s;
            int apart = hs[0] > hr[2] || hr[0] > hs[2] || hs[1] > hr[3] || hr[1] > hs[3];
            if (apart && (hr[4] + hs[4]) > area_guard) continue;
            double iou = orc_iou_poly(poly + 8 * r, poly + 8 * s);
            if (iou > thresh) dead[s] = 1;
        }
    }
    free(order); free(dead); free(poly); free(hull);
    return nk;
}

/* ---- tile ResultMerge NMS on float64 rows ------------------------------------ */
typedef struct { double s; int64_t i; } didx_t;

static int cmp_desc_d(const void *pa, const void *pb) {
    const didx_t *a = (const didx_t *)pa, *b = (const didx_t *)pb;
    if (a->s > b->s) return -1;
    if (a->s < b->s) return 1;
    if (a->i > b->i) return -1; /* equal scores: larger index first (stable argsort, reversed) */
    if (a->i < b->i) return 1;
    return 0;
}

/*
 * dafne/utils/ResultMerge_multi_process.py:61-121 (py_cpu_nms_poly_fast; strict_hbb != 0) and :24-58
 * (py_cpu_nms_poly; strict_hbb == 0) on a float64 [m,9] array, the array nmsbynamedict (:160-176)
 * builds with np.array(nameboxdict[imgname]).  Greedy in descending score; with strict_hbb a
 * remaining row is compared through iou_poly only when the reference's axis-aligned pre-test
 * `hbb_ovr > 0` holds (:80-98: areas with +1, intersection without), otherwise its overlap value
 * stays 0 and it is kept.  Order convention for equal scores: argsort(kind="stable")[::-1].
 */
int64_t orc_poly_nms_f64(const double *dets9, int64_t m, double thresh, int strict_hbb, int64_t *keep) {
    if (m <= 0) return 0;
    didx_t *v = (didx_t *)malloc(sizeof(didx_t) * (size_t)m);
    unsigned char *dead = (unsigned char *)calloc((size_t)m, 1);
    double *hb = (double *)malloc(sizeof(double) * 5 * (size_t)m);
    for (int64_t i = 0; i < m; i++) { v[i].s = dets9[9 * i + 8]; v[i].i = i; }
    qsort(v, (size_t)m, sizeof(didx_t), cmp_desc_d);
    for (int64_t r = 0; r < m; r++) {
        const double *q = dets9 + 9 * v[r].i;
        double *h = hb + 5 * r;
        h[0] = h[2] = q[0]; h[1] = h[3] = q[1];
        for (int k = 1; k < 4; k++) {
            if (q[2 * k] < h[0]) h[0] = q[2 * k];
            if (q[2 * k] > h[2]) h[2] = q[2 * k];
            if (q[2 * k + 1] < h[1]) h[1] = q[2 * k + 1];
            if (q[2 * k + 1] > h[3]) h[3] = q[2 * k + 1];
        }
        h[4] = (h[2] - h[0] + 1) * (h[3] - h[1] + 1);
    }
    int64_t nk = 0;
    for (int64_t r = 0; r < m; r++) {
        if (dead[r]) continue;
        keep[nk++] = v[r].i;
        const double *hr = hb + 5 * r;
        for (int64_t s = r + 1; s < m; s++) {
            if (dead[s]) continue;
            if (strict_hbb) {
                const double *hs = hb + 5 * s;
                double xx1 = hr[0] > hs[0] ? hr[0] : hs[0], yy1 = hr[1] > hs[1] ? hr[1] : hs[1];
                double xx2 = hr[2] < hs[2] ? hr[2] : hs[2], yy2 = hr[3] < hs[3] ? hr[3] : hs[3];
                double w = xx2 - xx1 > 0.0 ? xx2 - xx1 : 0.0, h = yy2 - yy1 > 0.0 ? yy2 - yy1 : 0.0;
                double inter = w * h;
                if (!(inter / (hr[4] + hs[4] - inter) > 0.0)) continue;
            }
            double iou = orc_iou_poly(dets9 + 9 * v[r].i, dets9 + 9 * v[s].i);
            if (iou > thresh) dead[s] = 1;
        }
    }
    free(v); free(dead); free(hb);
    return nk;
}

/*
 * nms.py:74-90: build the float32 [m,9] array from boxes[m,8], scores[m],
 * classes[m]:  class 5 -> 4, offset = float(class) * (max - min + 1) in fp32,
 * added in fp32.
 */
void orc_build_dets9(const float *boxes8, const float *scores, const int64_t *classes,
                     int64_t m, float *dets9) {
    if (m <= 0) return;
    float mx = boxes8[0], mn = boxes8[0];
    for (int64_t i = 0; i < 8 * m; i++) {
        if (boxes8[i] > mx) mx = boxes8[i];
        if (boxes8[i] < mn) mn = boxes8[i];
    }
    volatile float span = mx - mn;
    volatile float span1 = span + 1.0f;
    for (int64_t i = 0; i < m; i++) {
        int64_t c = classes[i] == 5 ? 4 : classes[i];
        volatile float off = (float)c * span1;
        for (int k = 0; k < 8; k++) {
            volatile float v = boxes8[8 * i + k] + off;
            dets9[9 * i + k] = v;
        }
        dets9[9 * i + 8] = scores[i];
    }
}
