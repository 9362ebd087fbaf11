// Rotated (polygon-IoU) NMS for gfx950.
//
// Replaces poly_nms.poly_gpu_nms (dafne/modeling/nms/nms.py:6,91 -> external CUDA
// extension DOTA_devkit/poly_nms_gpu) plus the fp32 class-offset arithmetic of
// batched_nms_poly (nms.py:74-90) and the kthvalue cap of select_over_all_levels
// (dafne/modeling/dafne/dafne_outputs.py:907-925).
//
// Numerics: IoU is the fp64 triangle-fan clip of tools/prepare_dota/polyiou.cpp
// (:10-133) applied to the float32 [M,9] rows, same operation order, compiled
// with -ffp-contract=off, so decisions `iou > thresh` equal the CPU reference bit
// for bit.  Everything else here is integer / ordering work.
//
// Pipeline (all images of a batch in one set of launches, counts read on device):
//   nms_minmax / nms_offset   (select path only) per-image max/min -> fp32 span+1, class offsets (nms.py:74-90), zero-area census
//   nms_sort_prep + nms_gather   rows into tile order: score descending, larger index first on ties, class-major when
//                the classes are independent (in-LDS radix sort; nms_chunk_sort + nms_merge_rank above 16384 rows;
//                nms_prep_f64 rank-by-counting for the fp64 ResultMerge rows), hull boxes, |area|, max|coord|
//   nms_scan     one wave per 64x64 tile of the upper triangle: class test, guarded hull pre-filter and IoU upper bound
//                (wave ballots) -> per-row-block lists of candidate pairs
//   nms_iou      persistent waves over the pair lists: one lane per pair on the convex decision fast path, the
//                undecided pairs pooled and clipped by 16 lanes each in polyiou.cpp's operation order -> 64-bit row
//                words of the tile-major suppression matrix
//   nms_class_reduce   one workgroup per (image, class): block-serial greedy scan over the class's part of the matrix
//   nms_compact  kept bits -> keep list in global score order, kthvalue cap with ties
//
// Wave64 throughout: one ballot == one 64-column tile row.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

typedef unsigned long long u64;

constexpr int kTile = 64;
constexpr int kCapP = 10;   // clipped polygon capacity (reference: Point p[10])
constexpr int kCapPP = 12;  // raw cut output capacity
constexpr int kPairCap = 4096;   // candidate pairs listed per 64-row block (more: clipped in place); class-major
                                 // order concentrates a class's top rows in one block: ~1800 at M = 10 000
constexpr double kEps = 1E-8;

struct P2 {
    double x, y;
};

__device__ __forceinline__ int sgn(double d) { return (d > kEps) - (d < -kEps); }

__device__ __forceinline__ bool same_pt(P2 a, P2 b) {
    return sgn(a.x - b.x) == 0 && sgn(a.y - b.y) == 0;
}

// polyiou.cpp:22-24
__device__ __forceinline__ double cross3(P2 o, P2 a, P2 b) {
    return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

// Per-lane scratch lives in LDS, slot-major / lane-minor so that lanes touching
// the same slot hit distinct banks: element s of lane l is at [s * 64 + l].
struct Scratch {
    P2* p;   // kCapP slots
    P2* pp;  // kCapPP slots
};

// polyiou.cpp:25-32 over the scratch polygon (closing vertex by wrap-around)
__device__ __forceinline__ double shoelace_s(const P2* p, int n) {
    double res = 0;
    if (n <= 0) return res / 2.0;
    P2 first = p[0];
    P2 cur = first;
    for (int i = 0; i < n; i++) {
        P2 nxt = (i + 1 < n) ? p[(i + 1) * kTile] : first;
        res += cur.x * nxt.y - cur.y * nxt.x;
        cur = nxt;
    }
    return res / 2.0;
}

// polyiou.cpp:62-75 (+ lineCross :33-43): keep the part of p left of a->b.
__device__ __forceinline__ void cut_left(Scratch s, int& n, P2 a, P2 b) {
    int m = 0;
    if (n > 0) {
        P2 first = s.p[0];
        double cfirst = cross3(a, b, first);
        P2 cur = first;
        double ci = cfirst;
        for (int i = 0; i < n; i++) {
            P2 nxt;
            double cj;
            if (i + 1 < n) {
                nxt = s.p[(i + 1) * kTile];
                cj = cross3(a, b, nxt);
            } else {
                nxt = first;
                cj = cfirst;
            }
            int si = sgn(ci), sj = sgn(cj);
            if (si > 0 && m < kCapPP) {
                s.pp[m * kTile] = cur;
                m++;
            }
            if (si != sj && m < kCapPP) {
                // lineCross(a,b,cur,nxt): s1 = ci, s2 = cj; they cannot both be ~0 here
                double den = cj - ci;
                if (sgn(den) != 0) {
                    P2 r;
                    r.x = (cur.x * cj - nxt.x * ci) / den;
                    r.y = (cur.y * cj - nxt.y * ci) / den;
                    s.pp[m * kTile] = r;
                }  // else: the slot keeps its stale value (see oracle/poly_oracle.c note 1)
                m++;
            }
            cur = nxt;
            ci = cj;
        }
    }
    int nn = 0;
    P2 prev = {0.0, 0.0};
    for (int i = 0; i < m; i++) {
        P2 v = s.pp[i * kTile];
        if ((i == 0 || !same_pt(v, prev)) && nn < kCapP - 1) {
            s.p[nn * kTile] = v;
            nn++;
        }
        prev = v;
    }
    while (nn > 1 && same_pt(s.p[(nn - 1) * kTile], s.p[0])) nn--;
    n = nn;
}

// polyiou.cpp:79-93: signed overlap of triangles (o,a,b) and (o,c,d).
__device__ __forceinline__ double tri_overlap(Scratch s, P2 a, P2 b, P2 c, P2 d) {
    P2 o = {0.0, 0.0};
    int s1 = sgn(cross3(o, a, b));
    int s2 = sgn(cross3(o, c, d));
    if (s1 == 0 || s2 == 0) return 0.0;
    if (s1 == -1) { P2 t = a; a = b; b = t; }
    if (s2 == -1) { P2 t = c; c = d; d = t; }
#pragma unroll
    for (int k = 0; k < kCapPP; k++) s.pp[k * kTile] = o;
    s.p[0] = o;
    s.p[1 * kTile] = a;
    s.p[2 * kTile] = b;
    int n = 3;
    cut_left(s, n, o, c);
    cut_left(s, n, c, d);
    cut_left(s, n, d, o);
    double res = fabs(shoelace_s(s.p, n));
    if (s1 * s2 == -1) res = -res;
    return res;
}

struct Quad {
    P2 v[4];
};

__device__ __forceinline__ double quad_area(const Quad& q) {
    double res = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const P2& a = q.v[i];
        const P2& b = q.v[(i + 1) & 3];
        res += a.x * b.y - a.y * b.x;
    }
    return res / 2.0;
}

__device__ __forceinline__ void quad_orient(Quad& q) {  // polyiou.cpp:96-97
    if (quad_area(q) < 0) {
        P2 t = q.v[0]; q.v[0] = q.v[3]; q.v[3] = t;
        t = q.v[1]; q.v[1] = q.v[2]; q.v[2] = t;
    }
}

__device__ __forceinline__ P2 quad_vertex(const Quad& q, int i) {
    // register-resident select (no dynamic indexing -> no scratch memory)
    P2 r = q.v[0];
    if (i == 1) r = q.v[1];
    if (i == 2) r = q.v[2];
    if (i == 3) r = q.v[3];
    return r;
}

// IoU of (A,B) computed by a group of 16 consecutive lanes: lane `sub` clips
// triangle pair (i=sub/4, j=sub%4); the 16 partial areas are then summed in the
// reference's loop order (i outer, j inner) by every lane of the group.
__device__ __forceinline__ double iou_group16(Scratch s, Quad A, Quad B, int lane) {
    quad_orient(A);
    quad_orient(B);
    const int sub = lane & 15;
    const int i = sub >> 2, j = sub & 3;
    double t = tri_overlap(s, quad_vertex(A, i), quad_vertex(A, (i + 1) & 3),
                           quad_vertex(B, j), quad_vertex(B, (j + 1) & 3));
    const int gbase = lane & ~15;
    double inter = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) inter += __shfl(t, gbase + k, 64);
    double uni = fabs(quad_area(A)) + fabs(quad_area(B)) - inter;
    if (uni == 0) return (inter + 1) / (uni + 1);
    return inter / uni;
}

__device__ __forceinline__ Quad load_quad_f64(const double* p) {
    Quad q;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        q.v[k].x = p[2 * k];
        q.v[k].y = p[2 * k + 1];
    }
    return q;
}

// strict overlap of the axis-aligned hulls, in the rows' own fp64 arithmetic
// (ResultMerge_multi_process.py:80-98: w = max(0, xx2 - xx1) > 0 and h > 0)
__device__ __forceinline__ bool hulls_overlap_strict(const Quad& a, const Quad& b) {
    const double ax0 = fmin(fmin(a.v[0].x, a.v[1].x), fmin(a.v[2].x, a.v[3].x)), ax1 = fmax(fmax(a.v[0].x, a.v[1].x), fmax(a.v[2].x, a.v[3].x));
    const double ay0 = fmin(fmin(a.v[0].y, a.v[1].y), fmin(a.v[2].y, a.v[3].y)), ay1 = fmax(fmax(a.v[0].y, a.v[1].y), fmax(a.v[2].y, a.v[3].y));
    const double bx0 = fmin(fmin(b.v[0].x, b.v[1].x), fmin(b.v[2].x, b.v[3].x)), bx1 = fmax(fmax(b.v[0].x, b.v[1].x), fmax(b.v[2].x, b.v[3].x));
    const double by0 = fmin(fmin(b.v[0].y, b.v[1].y), fmin(b.v[2].y, b.v[3].y)), by1 = fmax(fmax(b.v[0].y, b.v[1].y), fmax(b.v[2].y, b.v[3].y));
    const double wd = fmax(0.0, fmin(ax1, bx1) - fmax(ax0, bx0)), hd = fmax(0.0, fmin(ay1, by1) - fmax(ay0, by0));
    const double inter = wd * hd;
    const double aa = (ax1 - ax0 + 1) * (ay1 - ay0 + 1), ab = (bx1 - bx0 + 1) * (by1 - by0 + 1);
    return inter / (aa + ab - inter) > 0.0;          // h_inds = np.where(hbb_ovr > 0)
}

__device__ __forceinline__ Quad load_quad_f32(const float* p) {
    Quad q;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        q.v[k].x = (double)p[2 * k];
        q.v[k].y = (double)p[2 * k + 1];
    }
    return q;
}

// ------------------------------------------------------------ pairwise IoU API
__global__ void __launch_bounds__(64) iou_pairs_kernel(const double* __restrict__ p,
                                                       const double* __restrict__ q, long long n,
                                                       double* __restrict__ out) {
    __shared__ P2 lds_p[kCapP * kTile];
    __shared__ P2 lds_pp[kCapPP * kTile];
    const int lane = threadIdx.x;
    Scratch s{lds_p + lane, lds_pp + lane};
    long long pair = (long long)blockIdx.x * 4 + (lane >> 4);
    bool live = pair < n;
    long long idx = live ? pair : 0;
    Quad A, B;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        A.v[k].x = p[idx * 8 + 2 * k];
        A.v[k].y = p[idx * 8 + 2 * k + 1];
        B.v[k].x = q[idx * 8 + 2 * k];
        B.v[k].y = q[idx * 8 + 2 * k + 1];
    }
    double iou = iou_group16(s, A, B, lane);
    if (live && (lane & 15) == 0) out[pair] = iou;
}

// -------------------------------------------------------------- NMS workspace
struct NmsWs {
    int* order;      // [N][Mp]    sorted position -> original row
    float* sbox;     // [N][Mp][8] rows in sorted order
    float* sscore;   // [N][Mp]
    float4* hull;    // [N][Mp]    xmin, ymin, xmax, ymax
    double* area;    // [N][Mp]    |shoelace|
    float* farea;    // [N][Mp]    |shoelace| rounded DOWN to fp32 for strictly convex rows with edges >= 1 px, else -1
                     //            (the IoU upper bound of the tile pre-filter)
    u64* mask;       // [N][ntiles][64]  tile-major
    u64* rowflag;    // [N][nblk]  bit r of word b: row 64b+r suppresses something
    u64* keptw;      // [N][nblk]  kept rows in tile order (nms_class_reduce -> nms_compact)
    unsigned* meta;  // [N][4]     0: max|coord| (float bits) 1: span+1 (float bits)
    unsigned* nzero; // [N]        rows with exactly zero area (census of nms_offset_kernel: select path)
    unsigned* stats; // [N][4]     pairs decided by nms_iou: 0 fast path "suppress", 1 fast path "keep", 2 exact
                     //            (reference-order) path from the pair lists, 3 exact path of overflowed tiles
    float* dets9;    // [N][Mp][9] (select path only)
    unsigned* pair_cnt;        // [N][nblk]  pairs appended per row block (may exceed pair_cap)
    u64* pairs;                // [N][nblk][pair_cap]  (row | col << 32), sorted positions
    unsigned char* tile_flag;  // [N][ntiles]  tile did not fit the pair list
    double* dbox;              // [N][Mp][8] fp64 rows in sorted order (fp64 entry point only, else null)
    int strict;                // ResultMerge predicate: suppress iff hulls overlap strictly AND IoU > thresh
    int fast;                  // convex fast path for the decision, exact path for the rest
    unsigned char* cls;        // [N][Mp]  merged class of every ORIGINAL row (select path), else null
    int* perm;                 // [N][Mp]  score-sorted position -> position in the order the tiles use
    unsigned char* bcls;       // [N][nblk][2]  min / max class of every 64-row block in that order
    int* cbase;                // [N][65]  first row of every class in that order (class-major images)
    u64* ckeys;                // [2][N][Mp]  chunk-sorted keys of the large-M sort (score order / class-major order)
    int* posidx;               // [N][Mp]  original row -> position in tile order (large-M sort)
    int use_perm;              // sort_prep path: sbox/hull/area/order are in class-major order, perm/bcls valid
    int Mp, nblk, pair_cap;    // pair_cap: per row block
    size_t mask_words;         // per image
};

// flags & DAFNE_NMS_EXACT_ONLY (per call): parity runs switch the three analytic shortcuts off -- the guarded hull
// pre-filter and the IoU upper bound of nms_scan, the decision fast paths of nms_iou -- so that every pair of a live
// tile is clipped in polyiou.cpp's own operation order.  The library holds no mutable global state.
size_t carve(NmsWs& w, void* base, int N, int m_cap, bool f64 = false, int flags = 0) {
    int Mp = (m_cap + kTile - 1) / kTile * kTile;
    if (Mp == 0) Mp = kTile;
    int nblk = Mp / kTile;
    dafne::WsCarver c(base);
    size_t n = (size_t)N;
    w.Mp = Mp;
    w.nblk = nblk;
    const size_t ntiles = (size_t)nblk * (nblk + 1) / 2;
    // a 64-row block of a TTA-merge-sized set (27 000 boxes in one tile) has ~10^4 candidate pairs: with the small
    // list nearly every tile overflowed into the slower in-place path
    w.pair_cap = Mp <= 12288 ? 2 * kPairCap : 4 * kPairCap;
    w.meta = c.take<unsigned>(n * 4);
    w.nzero = c.take<unsigned>(n);
    w.stats = c.take<unsigned>(n * 4);
    w.pair_cnt = c.take<unsigned>(n * nblk);
    w.rowflag = c.take<u64>(n * nblk);
    w.keptw = c.take<u64>(n * nblk);
    w.tile_flag = c.take<unsigned char>(n * ntiles);   // meta .. tile_flag are zeroed per call (contiguous)
    w.pairs = c.take<u64>(n * nblk * (size_t)w.pair_cap);
    w.order = c.take<int>(n * Mp);
    w.sbox = c.take<float>(n * Mp * 8);
    w.sscore = c.take<float>(n * Mp);
    w.hull = c.take<float4>(n * Mp);
    w.area = c.take<double>(n * Mp);
    w.farea = c.take<float>(n * Mp);
    w.dets9 = c.take<float>(n * Mp * 9);
    w.dbox = f64 ? c.take<double>(n * Mp * 8) : nullptr;
    w.cls = c.take<unsigned char>(n * Mp);
    w.perm = c.take<int>(n * Mp);
    w.bcls = c.take<unsigned char>(n * nblk * 2);
    w.cbase = c.take<int>(n * 65);
    w.ckeys = c.take<u64>(2 * n * Mp);
    w.posidx = c.take<int>(n * Mp);
    w.use_perm = 0;
    w.strict = 0;
    w.fast = (flags & DAFNE_NMS_EXACT_ONLY) ? 0 : 1;
    w.mask_words = ntiles * kTile;
    w.mask = c.take<u64>(n * w.mask_words);
    return dafne::align_up(c.off, 256);
}

size_t zero_bytes(const NmsWs& w, int N) {
    const size_t ntiles = (size_t)w.nblk * (w.nblk + 1) / 2;
    return (size_t)((char*)(w.tile_flag + (size_t)N * ntiles) - (char*)w.meta);
}

__device__ __forceinline__ int img_count(const int* counts, int img, int m_cap) {
    int m = counts ? counts[img] : m_cap;
    return m < 0 ? 0 : (m > m_cap ? m_cap : m);
}

// --------------------------------------------------- class offsets (nms.py:74-90)
__global__ void __launch_bounds__(1024) nms_minmax_kernel(const float* __restrict__ boxes,
                                                          const int* __restrict__ counts, int m_cap,
                                                          unsigned* __restrict__ meta) {
    const int img = blockIdx.x;
    const int M = img_count(counts, img, m_cap);
    const float* b = boxes + (size_t)img * m_cap * 8;
    float mx = -INFINITY, mn = INFINITY;
    // rows are 32 B: float4 loads, four in flight per thread (one workgroup per image: latency, not bandwidth, is the cost)
    const float4* b4 = reinterpret_cast<const float4*>(b);
    const int n4 = M * 2;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * (int)blockDim.x) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + k * (int)blockDim.x;
            v[k] = i < n4 ? b4[i] : make_float4(mx, mx, mx, mx);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k * (int)blockDim.x < n4) {
                mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
                mn = fminf(mn, fminf(fminf(v[k].x, v[k].y), fminf(v[k].z, v[k].w)));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        mn = fminf(mn, __shfl_xor(mn, o, 64));
    }
    __shared__ float smx[16], smn[16];
    if ((threadIdx.x & 63) == 0) {
        smx[threadIdx.x >> 6] = mx;
        smn[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); k++) {
            mx = fmaxf(mx, smx[k]);
            mn = fminf(mn, smn[k]);
        }
        float span = mx - mn;          // fp32, like torch
        float span1 = span + 1.0f;
        meta[img * 4 + 1] = __float_as_uint(M > 0 ? span1 : 1.0f);
    }
}

__global__ void __launch_bounds__(256) nms_offset_kernel(const float* __restrict__ boxes,
                                                         const float* __restrict__ scores,
                                                         const int* __restrict__ classes,
                                                         const int* __restrict__ counts, int m_cap,
                                                         int Mp, const unsigned* __restrict__ meta,
                                                         float* __restrict__ dets9, unsigned char* __restrict__ cls,
                                                         unsigned* __restrict__ nzero) {
    const int img = blockIdx.y;
    const int M = img_count(counts, img, m_cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float span1 = __uint_as_float(meta[img * 4 + 1]);
    int c = classes[(size_t)img * m_cap + i];
    if (c == 5) c = 4;                                       // nms.py:77-79
    const float off = (float)c * span1;                      // nms.py:81
    const float* b = boxes + ((size_t)img * m_cap + i) * 8;
    float* d = dets9 + ((size_t)img * Mp + i) * 9;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = b[k] + off;           // nms.py:83
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = v[k];
    d[8] = scores[(size_t)img * m_cap + i];
    // census of exactly-zero-area rows (of the OFFSET boxes, as the clip sees them): two of them can suppress each
    // other across classes (union == 0 -> IoU 1), which rules out the class-major tile order for the image
    if (quad_area(load_quad_f32(v)) == 0.0) atomicAdd(&nzero[img], 1u);
    cls[(size_t)img * Mp + i] = (unsigned char)(c < 0 ? 255 : (c > 255 ? 255 : c));
}

// Class layout for the rank-by-counting path (M > 16384: the TTA merge): class histogram, zero-area census,
// first row of every class and the class range of every 64-row block -- what nms_sort_prep computes in place.
__global__ void __launch_bounds__(1024) nms_cls_layout_kernel(const float* __restrict__ dets9, int row_cap,
                                                              const int* __restrict__ counts, int m_cap, NmsWs w) {
    constexpr int kMaxCls = 64;
    const int img = blockIdx.x;
    const int M = img_count(counts, img, m_cap);
    __shared__ int ccnt[kMaxCls], cbase[kMaxCls + 1];
    __shared__ int badcls, ncls_s;
    if (threadIdx.x < kMaxCls) ccnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) { badcls = 0; ncls_s = 1; }
    __syncthreads();
    // class histogram (zero-area census: nms_offset_kernel, w.nzero).  Lanes with the same class find each other with
    // 6 ballots and the lowest one adds their count: one LDS atomic per distinct class and wave round instead of 64
    // colliding ones.
    int cmaxl = 0;
    for (int i0 = 0; i0 < M; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int c = i < M ? (int)w.cls[(size_t)img * w.Mp + i] : -1;
        const bool ok = c >= 0 && c < kMaxCls;
        if (c >= kMaxCls) badcls = 1;
        u64 peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const bool bit = (c >> b) & 1;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        if (ok) {
            if ((int)(threadIdx.x & 63) == __ffsll((long long)peers) - 1) atomicAdd(&ccnt[c], __popcll(peers));
            cmaxl = max(cmaxl, c + 1);
        }
    }
    for (int o = 32; o > 0; o >>= 1) cmaxl = max(cmaxl, __shfl_xor(cmaxl, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&ncls_s, cmaxl);
    __syncthreads();
    const bool cm = M > 0 && w.nzero[img] < 2u && !badcls;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < kMaxCls; c++) { cbase[c] = run; run += cm ? ccnt[c] : 0; }
        cbase[kMaxCls] = run;
        w.meta[img * 4 + 2] = cm ? (unsigned)ncls_s : 0u;
    }
    __syncthreads();
    if (threadIdx.x <= kMaxCls) w.cbase[(size_t)img * 65 + threadIdx.x] = cbase[threadIdx.x];
    for (int b = threadIdx.x; b < w.nblk; b += 1024) {
        int cmin = 0, cmax = 255;
        if (cm && b * kTile < M) {
            const int first = b * kTile, last = min(M, (b + 1) * kTile) - 1;
            while (cmin + 1 < kMaxCls && cbase[cmin + 1] <= first) cmin++;
            cmax = cmin;
            while (cmax + 1 < kMaxCls && cbase[cmax + 1] <= last) cmax++;
        }
        w.bcls[((size_t)img * w.nblk + b) * 2 + 0] = (unsigned char)cmin;
        w.bcls[((size_t)img * w.nblk + b) * 2 + 1] = (unsigned char)cmax;
    }
}

// ------------------------------------------------------------------ nms_prep
struct Quad;
__device__ __forceinline__ int quad_fast_class(Quad& q, double& area);
__device__ __forceinline__ bool quad_fast_ok(Quad& q, double& area);
// fp32 lower bound of the area of a strictly convex quad with edges >= 1 px (quad_fast_ok), else -1
__device__ __forceinline__ float convex_area_lb(Quad q) {
    double a;
    return quad_fast_ok(q, a) ? __double2float_rd(a) : -1.f;
}

__global__ void __launch_bounds__(256) nms_prep_kernel(const float* __restrict__ dets9, int row_cap,
                                                       const int* __restrict__ counts, int m_cap,
                                                       NmsWs w) {
    const int img = blockIdx.y;
    const int M = img_count(counts, img, m_cap);
    if ((int)(blockIdx.x * blockDim.x) >= M) return;
    const float* d = dets9 + (size_t)img * row_cap * 9;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < M;
    const float si = live ? d[(size_t)i * 9 + 8] : 0.f;
    __shared__ __attribute__((aligned(16))) float ss[256];
    __shared__ __attribute__((aligned(4))) unsigned char sc[256];
    // class-major layout (see nms_sort_prep_kernel): the position among the rows of the same class is counted
    // in the same pass (nms_cls_layout_kernel ran before and decided whether the image is class-major)
    const bool cm = w.use_perm && w.cls != nullptr && w.meta[img * 4 + 2] != 0u;
    const int ci = (cm && live) ? w.cls[(size_t)img * w.Mp + i] : -1;
    int rank = 0, crank = 0;
    for (int j0 = 0; j0 < M; j0 += 256) {
        int j = j0 + threadIdx.x;
        ss[threadIdx.x] = j < M ? d[(size_t)j * 9 + 8] : -INFINITY;
        if (cm) sc[threadIdx.x] = j < M ? w.cls[(size_t)img * w.Mp + j] : 255;
        __syncthreads();
        const float4* ss4 = reinterpret_cast<const float4*>(ss);
        const uchar4* sc4 = reinterpret_cast<const uchar4*>(sc);
#pragma unroll 4
        for (int j4 = 0; j4 < 64; j4++) {                 // entries beyond M hold -inf: never counted
            const float4 v = ss4[j4];
            const int jg = j0 + 4 * j4;
            const int a0 = (v.x > si) || (v.x == si && jg > i);  // argsort(kind="stable")[::-1]
            const int a1 = (v.y > si) || (v.y == si && jg + 1 > i);
            const int a2 = (v.z > si) || (v.z == si && jg + 2 > i);
            const int a3 = (v.w > si) || (v.w == si && jg + 3 > i);
            rank += a0 + a1 + a2 + a3;
            if (cm) {
                const uchar4 c4 = sc4[j4];
                crank += (a0 & (c4.x == ci)) + (a1 & (c4.y == ci)) + (a2 & (c4.z == ci)) + (a3 & (c4.w == ci));
            }
        }
        __syncthreads();
    }
    float amax = 0.f;
    if (live) {
        const int pos = cm ? w.cbase[(size_t)img * 65 + ci] + crank : rank;
        const size_t base = (size_t)img * w.Mp + pos;            // tile order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = d[(size_t)i * 9 + k];
        w.order[base] = i;
        w.sscore[(size_t)img * w.Mp + rank] = si;                 // score order
        if (w.use_perm) w.perm[(size_t)img * w.Mp + rank] = pos;
        float4* sb = reinterpret_cast<float4*>(w.sbox + base * 8);
        sb[0] = make_float4(v[0], v[1], v[2], v[3]);
        sb[1] = make_float4(v[4], v[5], v[6], v[7]);
        float xmin = fminf(fminf(v[0], v[2]), fminf(v[4], v[6]));
        float xmax = fmaxf(fmaxf(v[0], v[2]), fmaxf(v[4], v[6]));
        float ymin = fminf(fminf(v[1], v[3]), fminf(v[5], v[7]));
        float ymax = fmaxf(fmaxf(v[1], v[3]), fmaxf(v[5], v[7]));
        w.hull[base] = make_float4(xmin, ymin, xmax, ymax);
        Quad q = load_quad_f32(v);
        w.area[base] = fabs(quad_area(q));
        w.farea[base] = convex_area_lb(q);
#pragma unroll
        for (int k = 0; k < 8; k++) amax = fmaxf(amax, fabsf(v[k]));
    }
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0 && amax > 0.f) atomicMax(&w.meta[img * 4 + 0], __float_as_uint(amax));
}

// One stable ranking pass of the LDS radix sorts below over the digits dg[r] of a wave-striped sequence (wave w owns
// positions [w*64*E, (w+1)*64*E), round r covers w*64*E + r*64 + lane): afterwards lrank[r] = keys with the same digit
// earlier in this wave's range, hist[wave][digit] = first output position of that (digit, wave) group.  Per round
// the lanes holding the same digit find each other with 8 ballots, the lowest one bumps the wave's counter of that
// digit, the others take the old value by readlane; an exclusive scan over (digit major, wave minor) turns the
// per-wave counters into scatter bases.  1024 threads; hist = 16 x 256 counters, wsum = 16 words of LDS.
__device__ __forceinline__ void radix_rank_pass(unsigned* hist, unsigned* wsum, const unsigned (&dg)[16],
                                                unsigned (&lrank)[16], int E, int lane, int wv) {
    const u64 lt = (1ull << lane) - 1ull;
    for (int k = threadIdx.x; k < 16 * 256; k += 1024) hist[k] = 0u;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r < E) {
            u64 peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool bit = (dg[r] >> b) & 1u;
                const u64 m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const int leader = __ffsll((long long)peers) - 1;
            unsigned old = 0u;
            if (lane == leader) {
                old = hist[wv * 256 + dg[r]];
                hist[wv * 256 + dg[r]] = old + (unsigned)__popcll(peers);
            }
            old = __shfl(old, leader, 64);
            lrank[r] = old + (unsigned)__popcll(peers & lt);
        }
    }
    __syncthreads();
    // exclusive scan over (digit major, wave minor): thread t owns digit t>>2, waves 4*(t&3) .. +3
    const int dgt = threadIdx.x >> 2, w0 = (threadIdx.x & 3) * 4;
    unsigned c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) c[q] = hist[(w0 + q) * 256 + dgt];
    const unsigned tsum = c[0] + c[1] + c[2] + c[3];
    unsigned x = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    unsigned base = 0u;
    for (int w2 = 0; w2 < wv; w2++) base += wsum[w2];
    unsigned e = base + x - tsum;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        hist[(w0 + q) * 256 + dgt] = e;
        e += c[q];
    }
    __syncthreads();
}

// Same result as nms_prep_kernel for M <= 16384 rows, without the O(M^2) rank-by-counting: one workgroup per
// image sorts 64-bit keys (inverted order-preserving score bits << 32 | row index) in LDS with a stable LSD radix
// sort (4 passes of 8 bits over the score word), and then gathers the rows into sorted order.  The initial
// sequence is the rows in DESCENDING index order, so equal scores keep that order: score descending, larger
// index first on ties = argsort(kind="stable")[::-1].
//   * sequence positions are wave-striped: wave w owns positions [w*64*E, (w+1)*64*E), round r of the wave covers
//     w*64*E + r*64 + lane (E = ceil(M / 1024) <= 16 rounds); a thread keeps its E keys in registers across a
//     pass, so the scatter goes back into the same LDS array (no ping-pong buffer: 16384 keys = 128 KB);
//   * rank of a key = (keys with a smaller digit) + (keys with the same digit earlier in the sequence): per round
//     the lanes holding the same digit find each other with 8 ballots, the lowest one bumps the wave's counter of
//     that digit, the others take the old value by readlane; an exclusive scan over (digit, wave) turns the
//     per-wave counters into scatter bases.
// (A bitonic network needed 105 barrier-separated LDS stages for 16384 keys: 230 us per image; this: 4 passes.)
constexpr int kSortMax = 16384;
constexpr int kSortHistBytes = 16 * 256 * 4;
__global__ void __launch_bounds__(1024) nms_sort_prep_kernel(const float* __restrict__ dets9, int row_cap,
                                                             const int* __restrict__ counts, int m_cap, NmsWs w) {
    extern __shared__ u64 skey[];
    const int img = blockIdx.x;
    const int M = img_count(counts, img, m_cap);
    if (M == 0) return;
    const float* d = dets9 + (size_t)img * row_cap * 9;
    const int E = (M + 1023) >> 10;                        // rounds per wave (uniform)
    const int n = E << 10;                                 // padded sequence length
    unsigned* hist = reinterpret_cast<unsigned*>(skey + n);      // [16 waves][256 digits]
    __shared__ unsigned wsum[16];
    constexpr int kMaxCls = 64;
    __shared__ int cbase[kMaxCls + 1];
    __shared__ int badcls, ncls_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wbase = wv * 64 * E;
    u64 key[16];
    unsigned dg[16], lrank[16];
    auto rank_pass = [&]() { radix_rank_pass(hist, wsum, dg, lrank, E, lane, wv); };
#pragma unroll
    for (int r = 0; r < 16; r++) {
        key[r] = ~0ull;                                    // padding: largest key, after every real row (stable)
        if (r < E) {
            const int sp = wbase + r * 64 + lane;          // sequence position: row M-1-sp
            if (sp < M) {
                const int i = M - 1 - sp;
                unsigned u = __float_as_uint(d[(size_t)i * 9 + 8]);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // float order -> unsigned order
                key[r] = ((u64)(~u) << 32) | (u64)(unsigned)i;         // ascending ~u = descending score
            }
        }
    }
    for (int pass = 0; pass < 4; pass++) {
        const int sh = 32 + 8 * pass;
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = (unsigned)(key[r] >> sh) & 255u;
        rank_pass();
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (r < E) skey[hist[wv * 256 + dg[r]] + lrank[r]] = key[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (r < E) key[r] = skey[wbase + r * 64 + lane];            // key at sequence position wbase + r*64 + lane
        __syncthreads();
    }
    // ---- class-major tile order -------------------------------------------------------------------
    // Boxes of different classes sit >= 1 px apart after the class offsets (nms.py:81-83), so they never
    // suppress each other -- unless BOTH have exactly zero area (the reference's union == 0 -> IoU 1 quirk).
    // With at most one zero-area box in the image (census taken by nms_offset_kernel) the greedy result is the same
    // whether the rows are walked in global score order or class by class (score order inside a class); laid out
    // class by class, a 64x64 tile of two blocks without a common class has no candidates and nms_scan skips it
    // (1/15 of the tiles remain for 15 classes).  The layout is one more stable ranking pass with the class as the
    // digit; perm maps the global score order to it and the keep list is still emitted in global score order
    // (nms_compact walks perm).
    const bool have_cls = w.cls != nullptr && w.use_perm;
    if (threadIdx.x == 0) { badcls = 0; ncls_s = 1; }
    __syncthreads();
    bool cm = false;
    if (have_cls) {
        int cmaxl = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            dg[r] = 255u;                                  // padding ranks behind every class
            if (r < E && wbase + r * 64 + lane < M) {
                const int c = w.cls[(size_t)img * w.Mp + (int)(unsigned)key[r]];
                dg[r] = (unsigned)c;
                if (c >= kMaxCls) badcls = 1;
                else cmaxl = max(cmaxl, c + 1);
            }
        }
        for (int o = 32; o > 0; o >>= 1) cmaxl = max(cmaxl, __shfl_xor(cmaxl, o, 64));
        if (lane == 0) atomicMax(&ncls_s, cmaxl);
        rank_pass();                                       // (its barriers also publish badcls / ncls_s)
        cm = w.nzero[img] < 2u && !badcls;
    }
    const int ncls = cm ? ncls_s : 1;
    if (threadIdx.x <= kMaxCls)
        cbase[threadIdx.x] = !cm ? 0 : (threadIdx.x < kMaxCls ? (int)hist[threadIdx.x] : M);      // hist[0][c]: first row of class c
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int sp = wbase + r * 64 + lane;
        if (r < E && sp < M) {
            const int pos = cm ? (int)(hist[wv * 256 + dg[r]] + lrank[r]) : sp;
            const unsigned nu = (unsigned)(key[r] >> 32);              // ~u
            const unsigned u = ~nu;
            const unsigned bits = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
            const size_t gbase = (size_t)img * w.Mp + sp;              // score order
            w.perm[gbase] = pos;
            w.sscore[gbase] = __uint_as_float(bits);
            w.order[(size_t)img * w.Mp + pos] = (int)(unsigned)key[r]; // tile order -> original row
        }
    }
    if (threadIdx.x <= kMaxCls) w.cbase[(size_t)img * 65 + threadIdx.x] = cbase[threadIdx.x];
    if (threadIdx.x == 0) w.meta[img * 4 + 2] = cm ? (unsigned)ncls : 0u;      // 0: score order
    // class range of every 64-row block in tile order (classes ascend with the position)
    for (int b = threadIdx.x; b < w.nblk; b += 1024) {
        int cmin = 0, cmax = 255;
        if (cm && b * kTile < M) {
            const int first = b * kTile, last = min(M, (b + 1) * kTile) - 1;
            cmin = 0;
            while (cmin + 1 < kMaxCls && cbase[cmin + 1] <= first) cmin++;
            cmax = cmin;
            while (cmax + 1 < kMaxCls && cbase[cmax + 1] <= last) cmax++;
        }
        w.bcls[((size_t)img * w.nblk + b) * 2 + 0] = (unsigned char)cmin;
        w.bcls[((size_t)img * w.nblk + b) * 2 + 1] = (unsigned char)cmax;
    }
}

// ---- M > 16384 (the TTA merge: 27 000 rows): chunked LDS radix sort + merge by binary search -------------------
// The rows of an image are cut into K = ceil(M / 16384) equal chunks; a workgroup per (chunk, image, order) sorts its
// chunk with the LDS radix sort above and writes the sorted 64-bit keys to the workspace; nms_merge_rank_kernel then
// gives every key its rank in the whole image: rank in its own chunk + (keys of the other chunks below it), the
// latter by binary search (keys are unique: they end in the row index).  Two orders are needed:
//   order 0  score order          key = ~score bits << 32 | (2^32-1 - row)          -> sscore, perm
//   order 1  class-major order    key = class << 56 | ~score bits << 24 | (2^24-1 - row)   -> order, posidx
// (row stored inverted: ascending keys = larger row first on equal scores, as argsort(kind="stable")[::-1]; for an
// image that is not class-major the class field is 0 and the two orders coincide).  Replaces the O(M^2)
// rank-by-counting of nms_prep_kernel: 1.25 ms -> ~0.1 ms at M = 27 000.
__device__ __forceinline__ void chunk_geometry(int M, int& K, int& CH) {
    K = (M + kSortMax - 1) / kSortMax;
    if (K < 1) K = 1;
    CH = (M + K - 1) / K;
}

__global__ void __launch_bounds__(1024) nms_chunk_sort_kernel(const float* __restrict__ dets9, int row_cap,
                                                              const int* __restrict__ counts, int m_cap, NmsWs w, int N) {
    extern __shared__ u64 skey[];
    const int img = blockIdx.y, mode = blockIdx.z;
    const int M = img_count(counts, img, m_cap);
    int K, CH;
    chunk_geometry(M, K, CH);
    const int ck = blockIdx.x;
    if (ck >= K || M == 0) return;
    const int r0 = ck * CH, Mc = min(M, r0 + CH) - r0;        // rows [r0, r0 + Mc)
    const float* d = dets9 + (size_t)img * row_cap * 9;
    const int E = (Mc + 1023) >> 10;
    const int n = E << 10;
    unsigned* hist = reinterpret_cast<unsigned*>(skey + n);
    __shared__ unsigned wsum[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wbase = wv * 64 * E;
    const bool cm = mode == 1 && w.cls != nullptr && w.meta[img * 4 + 2] != 0u;
    u64 key[16];
    unsigned dg[16], lrank[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        key[r] = ~0ull;
        if (r < E) {
            const int sp = wbase + r * 64 + lane;              // initial sequence: descending row
            if (sp < Mc) {
                const int i = r0 + Mc - 1 - sp;
                unsigned u = __float_as_uint(d[(size_t)i * 9 + 8]);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                if (mode == 0) {
                    key[r] = ((u64)(~u) << 32) | (u64)(0xffffffffu - (unsigned)i);
                } else {
                    const u64 c = cm ? (u64)w.cls[(size_t)img * w.Mp + i] : 0ull;
                    key[r] = (c << 56) | ((u64)(~u) << 24) | (u64)(0xffffffu - (unsigned)i);
                }
            }
        }
    }
    const int sh0 = mode == 0 ? 32 : 24;
    const int npass = mode == 0 ? 4 : 5;                       // order 1: the class byte is the last digit
    for (int pass = 0; pass < npass; pass++) {
        const int sh = sh0 + 8 * pass;
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = (unsigned)(key[r] >> sh) & 255u;
        radix_rank_pass(hist, wsum, dg, lrank, E, lane, wv);
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (r < E) skey[hist[wv * 256 + dg[r]] + lrank[r]] = key[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (r < E) key[r] = skey[wbase + r * 64 + lane];
        __syncthreads();
    }
    u64* out = w.ckeys + ((size_t)mode * N + img) * w.Mp + r0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int sp = wbase + r * 64 + lane;
        if (r < E && sp < Mc) out[sp] = key[r];
    }
}

__global__ void __launch_bounds__(256) nms_merge_rank_kernel(const int* __restrict__ counts, int m_cap, NmsWs w, int N, int mode) {
    const int img = blockIdx.y;
    const int M = img_count(counts, img, m_cap);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    int K, CH;
    chunk_geometry(M, K, CH);
    const u64* keys = w.ckeys + ((size_t)mode * N + img) * w.Mp;
    const u64 key = keys[t];
    const int ck = t / CH;
    int rank = t - ck * CH;
    for (int c2 = 0; c2 < K; c2++) {
        if (c2 == ck) continue;
        const u64* kc = keys + c2 * CH;
        int lo = 0, hi = min(M, (c2 + 1) * CH) - c2 * CH;      // count of keys < key in chunk c2
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (kc[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        rank += lo;
    }
    if (mode == 1) {
        const int i = (int)(0xffffffu - (unsigned)(key & 0xffffffull));
        w.order[(size_t)img * w.Mp + rank] = i;
        w.posidx[(size_t)img * w.Mp + i] = rank;
    } else {
        const int i = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
        const unsigned u = ~(unsigned)(key >> 32);
        const unsigned bits = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        w.sscore[(size_t)img * w.Mp + rank] = __uint_as_float(bits);
        w.perm[(size_t)img * w.Mp + rank] = w.posidx[(size_t)img * w.Mp + i];
    }
}

// Second half of the sort path: rows gathered into tile order (order[pos] -> sbox / hull / area), chip-wide.
__global__ void __launch_bounds__(256) nms_gather_kernel(const float* __restrict__ dets9, int row_cap,
                                                         const int* __restrict__ counts, int m_cap, NmsWs w) {
    const int img = blockIdx.y;
    const int M = img_count(counts, img, m_cap);
    if ((int)(blockIdx.x * blockDim.x) >= M) return;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    const float* d = dets9 + (size_t)img * row_cap * 9;
    float amax = 0.f;
    if (pos < M) {
        const size_t base = (size_t)img * w.Mp + pos;
        const int i = w.order[base];
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = d[(size_t)i * 9 + k];
        float4* sb = reinterpret_cast<float4*>(w.sbox + base * 8);
        sb[0] = make_float4(v[0], v[1], v[2], v[3]);
        sb[1] = make_float4(v[4], v[5], v[6], v[7]);
        const float xmin = fminf(fminf(v[0], v[2]), fminf(v[4], v[6])), xmax = fmaxf(fmaxf(v[0], v[2]), fmaxf(v[4], v[6]));
        const float ymin = fminf(fminf(v[1], v[3]), fminf(v[5], v[7])), ymax = fmaxf(fmaxf(v[1], v[3]), fmaxf(v[5], v[7]));
        w.hull[base] = make_float4(xmin, ymin, xmax, ymax);
        Quad q = load_quad_f32(v);
        w.area[base] = fabs(quad_area(q));
        w.farea[base] = convex_area_lb(q);
#pragma unroll
        for (int k = 0; k < 8; k++) amax = fmaxf(amax, fabsf(v[k]));
    }
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0 && amax > 0.f) atomicMax(&w.meta[img * 4 + 0], __float_as_uint(amax));
}

// fp64 rows (tile ResultMerge: coordinates are (tile poly + offset) / rate in float64,
// ResultMerge_multi_process.py:175-187,217-228).  Same rank-by-counting sort on the fp64 scores; the
// rows are kept in fp64 (dbox) for the clip, the fp32 hull is rounded OUTWARD so that the hull
// pre-filter can only over-select.
__global__ void __launch_bounds__(256) nms_prep_f64_kernel(const double* __restrict__ dets9, int row_cap,
                                                           const int* __restrict__ counts, int m_cap,
                                                           NmsWs w) {
    const int img = blockIdx.y;
    const int M = img_count(counts, img, m_cap);
    if ((int)(blockIdx.x * blockDim.x) >= M) return;
    const double* d = dets9 + (size_t)img * row_cap * 9;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < M;
    const double si = live ? d[(size_t)i * 9 + 8] : 0.0;
    __shared__ double ss[256];
    int rank = 0;
    for (int j0 = 0; j0 < M; j0 += 256) {
        const int j = j0 + threadIdx.x;
        ss[threadIdx.x] = j < M ? d[(size_t)j * 9 + 8] : -INFINITY;
        __syncthreads();
#pragma unroll 8
        for (int jj = 0; jj < 256; jj++) {
            const double v = ss[jj];
            rank += (v > si) || (v == si && j0 + jj > i);     // argsort(kind="stable")[::-1]
        }
        __syncthreads();
    }
    float amax = 0.f;
    if (live) {
        const size_t base = (size_t)img * w.Mp + rank;
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = d[(size_t)i * 9 + k];
        w.order[base] = i;
        w.sscore[base] = (float)si;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            w.dbox[base * 8 + k] = v[k];
            w.sbox[base * 8 + k] = (float)v[k];
        }
        const double xmin = fmin(fmin(v[0], v[2]), fmin(v[4], v[6])), xmax = fmax(fmax(v[0], v[2]), fmax(v[4], v[6]));
        const double ymin = fmin(fmin(v[1], v[3]), fmin(v[5], v[7])), ymax = fmax(fmax(v[1], v[3]), fmax(v[5], v[7]));
        w.hull[base] = make_float4(__double2float_rd(xmin), __double2float_rd(ymin), __double2float_ru(xmax),
                                   __double2float_ru(ymax));
        Quad q;
#pragma unroll
        for (int k = 0; k < 4; k++) { q.v[k].x = v[2 * k]; q.v[k].y = v[2 * k + 1]; }
        w.area[base] = fabs(quad_area(q));
        w.farea[base] = convex_area_lb(q);
#pragma unroll
        for (int k = 0; k < 8; k++) amax = fmaxf(amax, __double2float_ru(fabs(v[k])));
    }
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0 && amax > 0.f) atomicMax(&w.meta[img * 4 + 0], __float_as_uint(amax));
}

// ------------------------------------------------------------------ tile pre-filter
__device__ __forceinline__ int kth_set_bit(u64 m, int t) {
    int pos = 0;
#pragma unroll
    for (int wbits = 32; wbits >= 1; wbits >>= 1) {
        u64 low = m & ((1ull << wbits) - 1ull);
        int c = __popcll(low);
        if (t >= c) {
            t -= c;
            m >>= wbits;
            pos += wbits;
        } else {
            m = low;
        }
    }
    return pos;
}

// The suppression matrix is stored tile-major: the 64 row words of tile (rb, cb) are
// contiguous (512 B), tiles in row-major order over the upper triangle.
__device__ __forceinline__ size_t tile_id(int rb, int cb, int nb) {
    return (size_t)rb * nb - (size_t)rb * (rb - 1) / 2 + (size_t)(cb - rb);
}

// tile id -> (rb, cb), cb >= rb, row-major over the upper triangle of nb x nb blocks
__device__ __forceinline__ void tile_rc(long long t, int nb, int& rb, int& cb) {
    // row rb starts at rb*nb - rb*(rb-1)/2
    int r = (int)((2.0 * nb + 1.0 - sqrt((2.0 * nb + 1.0) * (2.0 * nb + 1.0) - 8.0 * (double)t)) * 0.5);
    if (r < 0) r = 0;
    if (r >= nb) r = nb - 1;
    while (r > 0 && (long long)r * nb - (long long)r * (r - 1) / 2 > t) r--;
    while ((long long)(r + 1) * nb - (long long)(r + 1) * r / 2 <= t) r++;
    rb = r;
    cb = r + (int)(t - ((long long)r * nb - (long long)r * (r - 1) / 2));
}

// Hull pre-filter of one 64x64 tile by one wave.  Lane r ends up with the 64-bit set
// of columns row r must be clipped against.  Guard: see oracle/poly_oracle.c
// (orc_poly_nms_fast): separated hulls mean a true intersection of 0; the fp64 fan
// sum then differs from 0 by rounding only, which cannot reach thresh*union unless
// the union is itself negligible.
//
// Second rejection (fp32 select / plain paths, not the strict ResultMerge predicate): for two strictly convex quads the
// intersection lies inside both quads and inside the overlap of their hulls, so the geometric IoU is at most
// ub / (area_r + area_c - ub), ub = min(hull overlap, area_r, area_c).  When that bound is below thresh - 2e-3 and
// the union is >= 16 px^2 the pair cannot suppress: the reference value differs from the geometric one by <= 1e-4
// under exactly these conditions (see the convex fast path below, whose decision for such a pair would be the same
// "no").  Evaluated in fp32 on a lower bound of the areas (farea, rounded down; -1 = not convex) and an upper bound
// of the overlap, with 1e-5 relative slack for the fp32 roundings.  On the synthetic candidate sets this removes
// ~64 % of the hull-overlapping pairs before they reach the pair lists.
__device__ __forceinline__ u64 tile_candidates(const NmsWs& w, int img, int M, int rb, int cb, double thresh,
                                               float4* rhull, float* rarea) {
    const int lane = threadIdx.x & 63;
    const size_t ibase = (size_t)img * w.Mp;
    const int grow = rb * kTile + lane;
    const int gcol = cb * kTile + lane;
    const bool colv = gcol < M;
    const float4 ch = colv ? w.hull[ibase + gcol] : make_float4(0, 0, 0, 0);
    const float R = __uint_as_float(w.meta[img * 4 + 0]);
    const bool prefilter = thresh >= 1e-6 && w.fast;        // exact-only runs clip every pair of a live tile
    const bool strict = w.strict != 0;
    const double guard = 256.0 * (2e-13 * (double)R * (double)R + 1e-6) / (prefilter ? thresh : 1.0);
    // area_r + area_c > guard is implied by either area alone exceeding it (areas >= 0);
    // testing the two flags keeps fp64 out of the 64-step loop (never skips more than
    // the exact test would)
    const bool bigc = colv && w.area[ibase + gcol] > guard;
    const u64 rowbig = __ballot(grow < M && w.area[ibase + grow] > guard);
    // row hulls: one coalesced load per lane, then LDS broadcast reads (whole float4,
    // branch-free tests, so the unrolled loop keeps 8 reads in flight)
    rhull[lane] = grow < M ? w.hull[ibase + grow] : make_float4(0, 0, 0, 0);
    const float tb = (float)thresh - 2e-3f;
    const bool bound_on = !strict && w.dbox == nullptr && tb > 1e-3f && w.fast;
    const float k1 = (1.f + tb) * (1.f + 1e-5f), k2 = tb * (1.f - 1e-5f);
    const float ac = (bound_on && colv) ? w.farea[ibase + gcol] : -1.f;
    rarea[lane] = (bound_on && grow < M) ? w.farea[ibase + grow] : -1.f;
    __builtin_amdgcn_wave_barrier();
    u64 mycand = 0ull;
    const int rlim = min(kTile, M - rb * kTile);
    const bool offdiag = rb != cb;
#pragma unroll 8
    for (int r = 0; r < rlim; r++) {
        const float4 h = rhull[r];
        const bool apart = (ch.x > h.z) | (h.x > ch.z) | (ch.y > h.w) | (h.y > ch.w);
        // strict mode: separated hulls never suppress (py_cpu_nms_poly_fast: hbb_ovr == 0 keeps the box)
        const float ar = rarea[r];
        const float ow = fminf(ch.z, h.z) - fmaxf(ch.x, h.x), oh = fminf(ch.w, h.w) - fmaxf(ch.y, h.y);
        const float ub = fminf(ow * oh, fminf(ac, ar));
        const float sum = ac + ar;
        const bool far = (ac > 0.f) & (ar > 0.f) & (ow > 0.f) & (oh > 0.f) & (sum - ub >= 16.5f) & (ub * k1 < sum * k2);
        const bool skip = strict ? apart : ((prefilter & apart & (bigc | (bool)((rowbig >> r) & 1ull))) | far);
        const bool c = colv & !skip & (offdiag | (lane > r));
        const u64 b = __ballot(c);
        if (lane == r) mycand = b;
    }
    return mycand;
}

// ------------------------------------------------------------------ nms_scan
// One wave per tile: zero the tile's mask words, run the hull pre-filter and append
// the surviving (row, col) pairs to the image's pair list.  A tile whose pairs do not
// fit the list is flagged and clipped later in place (nms_iou, second phase).
__global__ void __launch_bounds__(256) nms_scan_kernel(const int* __restrict__ counts, int m_cap,
                                                       double thresh, NmsWs w, long long ntiles) {
    __shared__ float4 rhull_s[4][kTile];
    __shared__ float rarea_s[4][kTile];
    const int img = blockIdx.y;
    const int M = img_count(counts, img, m_cap);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + wv;     // wave-uniform by construction
    if (t >= ntiles) return;
    int rb, cb;
    tile_rc(t, w.nblk, rb, cb);
    if (rb * kTile >= M || cb * kTile >= M) return;
    const int grow = rb * kTile + lane;
    if (w.use_perm && w.cls) { // class-major layout: two blocks without a common class have no candidate pair --
        // and nobody ever reads such a tile (nms_class_reduce walks tiles inside a class only): not even zeroed
        const unsigned char* bc = w.bcls + (size_t)img * w.nblk * 2;
        if (bc[rb * 2] > bc[cb * 2 + 1] || bc[cb * 2] > bc[rb * 2 + 1]) return;
    }
    w.mask[(size_t)img * w.mask_words + (size_t)t * kTile + lane] = 0ull;
    const u64 mycand = tile_candidates(w, img, M, rb, cb, thresh, rhull_s[wv], rarea_s[wv]);
    const int cnt = __popcll(mycand);
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const int total = __shfl(incl, 63, 64);
    if (total == 0) return;
    unsigned base = 0;
    // one list per (image, row block): the counters of different row blocks sit on
    // different addresses, so the appends do not serialise on one L2 atomic unit
    const size_t lbase = ((size_t)img * w.nblk + rb) * w.pair_cap;
    if (lane == 0) base = atomicAdd(&w.pair_cnt[(size_t)img * w.nblk + rb], (unsigned)total);
    base = __shfl(base, 0, 64);
    if (base + (unsigned)total > (unsigned)w.pair_cap) {
        // the part of the reservation that lies inside the list gets skip markers
        for (unsigned i = (unsigned)lane; i < (unsigned)total && base + i < (unsigned)w.pair_cap; i += 64)
            w.pairs[lbase + base + i] = ~0ull;
        if (lane == 0) {
            w.tile_flag[(size_t)img * ntiles + t] = 1;
            w.meta[img * 4 + 3] = 1u;               // some tile overflowed
        }
        return;
    }
    u64* list = w.pairs + lbase + base + (incl - cnt);
    u64 bits = mycand;
    const u64 rowpart = (u64)(unsigned)grow;
    int k = 0;
    while (bits) {
        const int c = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        list[k++] = rowpart | ((u64)(unsigned)(cb * kTile + c) << 32);
    }
}

// ---------------------------------------------------------------- convex fast path
// Most candidate pairs are two ordinary convex quads whose IoU is far from the threshold.  For those the
// DECISION `iou_poly > thresh` does not need polyiou.cpp's 16 x 3 half-plane cuts in its exact order: one
// lane clips quad A by the 4 edges of quad B (Sutherland-Hodgman, <= 8 vertices) and gets the geometric IoU
// to ~1e-12.  The reference value differs from the geometric one only by fp64 rounding and by its 1e-8
// `sig()` snapping -- slivers of width <= 1e-8/|edge| along each cut, i.e. <= 3e-5 px^2 per cut for edges
// >= 1 px and chords <= 3000 px, <= 1.5e-3 px^2 over the 48 cuts, <= 1e-4 in IoU once the union is >= 16
// px^2.  So with BOTH quads strictly convex, every edge >= 1 px and union >= 16 px^2, a fast IoU outside
// thresh +- 1e-3 fixes the decision; everything else (non-convex, degenerate, tiny, near the threshold)
// goes through the reference-order computation (iou_group16).  Keep lists are unchanged (all fixtures).
__device__ __forceinline__ bool quad_fast_ok(Quad& q, double& area) { return (quad_fast_class(q, area) & 1) != 0; }

// bit 0: strictly convex; bit 1: "sane" -- every edge >= 1 px and every vertex >= 1 px away from the coordinate origin (the
// reference fans both polygons from the ORIGIN, polyiou.cpp:69-93, so its 48 cuts run along origin-vertex lines as well as
// along the edges; the sliver bound above needs every cut line to be >= 1 px long).  Bit 0 implies bit 1.
__device__ __forceinline__ int quad_fast_class(Quad& q, double& area) {
    quad_orient(q);                       // counter-clockwise (polyiou.cpp:96-97)
    area = quad_area(q);
    bool cvx = true, sane = true;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const P2 a = q.v[i], b = q.v[(i + 1) & 3], c = q.v[(i + 2) & 3];
        const double ex = b.x - a.x, ey = b.y - a.y, fx = c.x - b.x, fy = c.y - b.y;
        cvx &= (ex * fy - ey * fx) > 1e-3;               // strictly convex corner
        sane &= (ex * ex + ey * ey) >= 1.0;              // edge >= 1 px
        sane &= (a.x * a.x + a.y * a.y) >= 1.0;          // origin-vertex cut line >= 1 px
    }
    return (cvx && sane ? 1 : 0) | (sane ? 2 : 0);
}

// returns 0: IoU < thresh - margin, 1: IoU > thresh + margin, 2: undecided (use the exact path)
//
// Geometric intersection area of two strictly convex CCW polygons WITHOUT building the intersection polygon: its boundary
// consists of the parts of A's edges that lie inside B and of B's edges that lie inside A, and by Green's theorem the
// area is half the sum of cross(start, end) over those parts.  The part of an edge inside the other polygon is a parameter
// interval [t0, t1] clipped by the other polygon's half-planes (Cyrus-Beck), so everything stays in registers: for two
// quads 32 signed distances g = cross(edge, vertex - edge origin), 32 interval updates, 8 cross products -- no LDS scratch,
// no data-dependent indexing, no divergent loop (a Sutherland-Hodgman clip through LDS took ~14 000 cycles per pair).
// Robustness: a vertex closer than 1e-6 px to the other polygon's edge line could be classified on either side (and two
// coincident edges would then be counted twice or not at all), so any |g| <= 1e-6 * |edge| sends the pair to the exact
// path; otherwise all inside/outside decisions are certain in fp64 and the area is accurate to ~1e-9 px^2 (coordinates
// are taken relative to A's first vertex).  The decision margins are those of the comment above.
template <int N>
struct Poly {
    P2 v[N];
};
template <int NP, int NQ>
__device__ __forceinline__ void boundary_part(const P2 (&Pv)[NP], const P2 (&Qv)[NQ], double& area2, bool& shaky) {
    double g[NP][NQ];                        // g[i][j]: vertex i of P against edge j of Q (>= 0: inside)
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        const P2 c = Qv[j], d = Qv[(j + 1) % NQ];
        const double ex = d.x - c.x, ey = d.y - c.y;
        const double tol = 1e-12 * (ex * ex + ey * ey);
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const double v = ex * (Pv[i].y - c.y) - ey * (Pv[i].x - c.x);
            g[i][j] = v;
            shaky |= v * v <= tol;
        }
    }
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const P2 a = Pv[i], b = Pv[(i + 1) % NP];
        double t0 = 0.0, t1 = 1.0;
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const double ga = g[i][j], gb = g[(i + 1) % NP][j];
            const double tc = ga / (ga - gb);            // only used when the signs differ (ga != gb then)
            const bool na = ga < 0.0, nb = gb < 0.0;
            t0 = (na && !nb) ? fmax(t0, tc) : t0;        // entering the half-plane
            t1 = (!na && nb) ? fmin(t1, tc) : t1;        // leaving it
            t1 = (na && nb) ? -1.0 : t1;                 // wholly outside
        }
        if (t1 > t0) {
            const double dx = b.x - a.x, dy = b.y - a.y;
            const double sx = a.x + t0 * dx, sy = a.y + t0 * dy, ex2 = a.x + t1 * dx, ey2 = a.y + t1 * dy;
            area2 += sx * ey2 - sy * ex2;
        }
    }
}

// General quads (reflex vertex, self-intersecting "bow-tie": what a head with untrained weights emits, 79 % of the listed
// pairs of the bench pipeline).  polyiou.cpp's intersectArea is sum_ij s_i s_j |T_i n T_j| over the origin fans of the two
// orientation-normalised polygons = the integral of W_A * W_B, W = the polygon's winding-number function -- and W does not
// depend on the fan's apex.  So the same number comes from the fans at vertex 0: two triangles per quad, four
// triangle pairs, each two CONVEX polygons for boundary_part (made counter-clockwise, the sign carried outside).  A fan
// triangle with |cross| < 1e-3 (area < 5e-4 px^2) is dropped: <= 4 such terms, <= 1e-3 px^2 of the integral, < 7e-5 in
// IoU at union >= 16 px^2; with the reference's own sliver bound (above; every cut line >= 1 px: `sane`) the geometric
// IoU is within 2e-4 of polyiou.cpp's, and the +-1e-3 margin decides.
__device__ __forceinline__ void quad_fan(const Quad& q, Poly<3> (&t)[2], double (&sg)[2], bool (&live)[2]) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const P2 a = q.v[0], b = q.v[1 + k], c = q.v[2 + k];
        const double cr = (b.x - a.x) * (c.y - a.y) - (c.x - a.x) * (b.y - a.y);
        live[k] = fabs(cr) >= 1e-3;
        sg[k] = cr < 0.0 ? -1.0 : 1.0;
        t[k].v[0] = a;
        t[k].v[1] = cr < 0.0 ? c : b;
        t[k].v[2] = cr < 0.0 ? b : c;
    }
}

__device__ __forceinline__ int fast_decision(Scratch s, Quad A, Quad B, double thresh) {
    double aa, ab;
    const int ca = quad_fast_class(A, aa), cb = quad_fast_class(B, ab);       // both counter-clockwise afterwards
    if (!(ca & cb & 2)) return 2;
    const P2 o = A.v[0];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        A.v[k].x -= o.x; A.v[k].y -= o.y;
        B.v[k].x -= o.x; B.v[k].y -= o.y;
    }
    bool shaky = false;
    double inter;
    if (ca & cb & 1) {
        double area2 = 0.0;
        boundary_part<4, 4>(A.v, B.v, area2, shaky);
        boundary_part<4, 4>(B.v, A.v, area2, shaky);
        inter = fmax(0.5 * area2, 0.0);
    } else {
        Poly<3> ta[2], tb[2];
        double sa[2], sb[2];
        bool la[2], lb[2];
        quad_fan(A, ta, sa, la);
        quad_fan(B, tb, sb, lb);
        double sum2 = 0.0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (la[i] && lb[j]) {
                    double a2 = 0.0;
                    boundary_part<3, 3>(ta[i].v, tb[j].v, a2, shaky);
                    boundary_part<3, 3>(tb[j].v, ta[i].v, a2, shaky);
                    sum2 += sa[i] * sb[j] * fmax(a2, 0.0);
                }
        inter = 0.5 * sum2;
    }
    if (shaky) return 2;
    const double uni = aa + ab - inter;
    if (!(uni >= 16.0)) return 2;
    const double iou = inter / uni;
    if (iou > thresh + 1e-3) return 1;
    if (iou < thresh - 1e-3) return 0;
    return 2;
}

// ------------------------------------------------------------------- nms_iou
// Persistent waves over the pair list: 16 lanes clip one pair, 4 pairs per wave step.
__global__ void __launch_bounds__(64) nms_iou_kernel(const int* __restrict__ counts, int m_cap, double thresh,
                                                     NmsWs w, long long ntiles) {
    // one wave per block: the per-lane clip scratch (22.5 KB per wave) is what limits
    // residency, so small blocks let ~6 waves share a CU and hide each other's LDS latency
    __shared__ P2 lds_p[1][kCapP * kTile];
    __shared__ P2 lds_pp[1][kCapPP * kTile];
    __shared__ float4 rhull_s[1][kTile];
    __shared__ float rarea_s[1][kTile];
    __shared__ u64 cand_s[1][kTile];
    __shared__ int pre_s[1][kTile];
    const int img = blockIdx.y;
    const int M = img_count(counts, img, m_cap);
    if (M == 0) return;
    const int wv = 0, lane = threadIdx.x & 63;
    const int nb = w.nblk;
    const size_t ibase = (size_t)img * w.Mp;
    Scratch s{lds_p[wv] + lane, lds_pp[wv] + lane};
    const int gw = blockIdx.x, nw = gridDim.x;                   // wave id / waves per image
    const int nbu = (M + kTile - 1) / kTile;
    // Phase A per 64-pair chunk: one lane per pair takes the convex fast path; the undecided pairs are
    // queued and clipped 4 at a time (16 lanes each) in polyiou.cpp's own operation order.
    // The undecided pairs are POOLED across chunks (a stack in LDS): a chunk leaves 1-2 of them on average, and the
    // exact path costs the same for 1 as for 4 pairs, so it only runs on full groups of 4 (and once at the end).
    __shared__ u64 exq[2 * kTile];
    const int chunks = w.pair_cap / kTile;
    auto record = [&](int r, int c) {
        atomicOr(&w.mask[(size_t)img * w.mask_words + tile_id(r >> 6, c >> 6, nb) * kTile + (r & 63)], 1ull << (c & 63));
        atomicOr(&w.rowflag[(size_t)img * nb + (r >> 6)], 1ull << (r & 63));
    };
    int nq = 0;                                                  // wave-uniform stack height
    unsigned n_yes = 0u, n_no = 0u, n_exact = 0u, n_ovf = 0u;    // wave-uniform path counters (w.stats)
    auto exact4 = [&](int n_live) {                              // pops min(4, nq) pairs
        const int k = nq - 1 - (lane >> 4);
        const bool live = (lane >> 4) < n_live;
        const u64 e2 = exq[live ? k : nq - 1];
        const int r = (int)(unsigned)e2, c = (int)(e2 >> 32);
        Quad A = w.dbox ? load_quad_f64(w.dbox + (ibase + r) * 8) : load_quad_f32(w.sbox + (ibase + r) * 8);
        Quad B = w.dbox ? load_quad_f64(w.dbox + (ibase + c) * 8) : load_quad_f32(w.sbox + (ibase + c) * 8);
        const double iou = iou_group16(s, A, B, lane);
        if (live && (lane & 15) == 0 && iou > thresh && (!w.strict || hulls_overlap_strict(A, B))) record(r, c);
        nq -= n_live;
        __builtin_amdgcn_wave_barrier();
    };
    // Items (row block, 64-pair chunk) are dealt round-robin to the waves; most chunk slots of a row block are empty,
    // so a wave first fetches the pair counts of its next 64 items with ONE load (lane l: item base + l*nw) and then
    // walks only the non-empty ones -- instead of one dependent global load per item.
    const long long total = (long long)nbu * chunks;
    for (long long base = gw; base < total; base += 64LL * nw) {
      const long long myit = base + (long long)lane * nw;
      unsigned mycnt = 0u;
      if (myit < total) mycnt = min(w.pair_cnt[(size_t)img * nb + (int)(myit / chunks)], (unsigned)w.pair_cap);
      u64 todo = __ballot(myit < total && (unsigned)(myit % chunks) * (unsigned)kTile < mycnt);
      while (todo) {
        const int il = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long long item = base + (long long)il * nw;
        const int lrb = (int)(item / chunks);
        const unsigned p0 = (unsigned)(item % chunks) * (unsigned)kTile;
        const unsigned n_pairs = (unsigned)__shfl((int)mycnt, il, 64);
        const u64* list = w.pairs + ((size_t)img * nb + lrb) * w.pair_cap;
        int dec = 0;
        u64 en = 0ull;
        if (p0 + (unsigned)lane < n_pairs) {
            en = list[p0 + lane];
            if (en != ~0ull) {                                   // not a skip marker
                dec = 2;
                if (w.fast) {
                    const int r = (int)(unsigned)en, c = (int)(en >> 32);
                    const Quad A = w.dbox ? load_quad_f64(w.dbox + (ibase + r) * 8) : load_quad_f32(w.sbox + (ibase + r) * 8);
                    const Quad B = w.dbox ? load_quad_f64(w.dbox + (ibase + c) * 8) : load_quad_f32(w.sbox + (ibase + c) * 8);
                    dec = fast_decision(s, A, B, thresh);
                    if (dec == 1 && (!w.strict || hulls_overlap_strict(A, B))) record(r, c);
                }
            }
        }
        const u64 need = __ballot(dec == 2);
        n_yes += (unsigned)__popcll(__ballot(dec == 1));
        n_no += (unsigned)__popcll(__ballot(dec == 0 && en != 0ull && en != ~0ull));
        n_exact += (unsigned)__popcll(need);
        if (need == 0ull) continue;
        if (dec == 2) exq[nq + __popcll(need & ((1ull << lane) - 1ull))] = en;
        nq += __popcll(need);
        __builtin_amdgcn_wave_barrier();
        while (nq >= 4) exact4(4);
      }
    }
    auto flush_stats = [&]() {
        if (lane == 0) {
            if (n_yes) atomicAdd(&w.stats[img * 4 + 0], n_yes);
            if (n_no) atomicAdd(&w.stats[img * 4 + 1], n_no);
            if (n_exact) atomicAdd(&w.stats[img * 4 + 2], n_exact);
            if (n_ovf) atomicAdd(&w.stats[img * 4 + 3], n_ovf);
        }
    };
    if (w.meta[img * 4 + 3] != 0u) {
        // overflow phase: tiles whose pairs did not fit the row block's list are clipped in place -- same two paths as
        // above (one lane per pair on the decision fast path, the undecided ones pooled for the reference-order clip).
        // A wave reads the flags of 64 tiles with one load.  (The first version walked the tiles one dependent load at a
        // time and sent EVERY pair of a flagged tile through the 16-lane exact path: one such tile cost its wave ~1 ms,
        // the dense set's nms_iou went from 70 us to 576 us.)
        for (long long tb = (long long)gw * 64; tb < ntiles; tb += (long long)nw * 64) {
            const long long myt = tb + lane;
            u64 todo = __ballot(myt < ntiles && w.tile_flag[(size_t)img * ntiles + myt] != 0);
            while (todo) {
                const int il = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const long long t = tb + il;
                int rb, cb;
                tile_rc(t, nb, rb, cb);
                const u64 mycand = tile_candidates(w, img, M, rb, cb, thresh, rhull_s[wv], rarea_s[wv]);
                const int cnt = __popcll(mycand);
                int incl = cnt;
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += v;
                }
                const int total = __shfl(incl, 63, 64);
                n_ovf += (unsigned)total;
                cand_s[wv][lane] = mycand;
                pre_s[wv][lane] = incl - cnt;
                __builtin_amdgcn_wave_barrier();
                for (int base = 0; base < total; base += 64) {
                    const int k = base + lane;
                    const bool live = k < total;
                    const int kk = live ? k : total - 1;
                    int lo = 0, hi = 63;       // largest row with pre[row] <= kk
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (pre_s[wv][mid] <= kk) lo = mid; else hi = mid - 1;
                    }
                    const int row = lo;
                    const int col = kth_set_bit(cand_s[wv][row], kk - pre_s[wv][row]);
                    const int r = rb * kTile + row, c = cb * kTile + col;
                    int dec = live ? 2 : 0;
                    if (live && w.fast) {
                        const Quad A = w.dbox ? load_quad_f64(w.dbox + (ibase + r) * 8) : load_quad_f32(w.sbox + (ibase + r) * 8);
                        const Quad B = w.dbox ? load_quad_f64(w.dbox + (ibase + c) * 8) : load_quad_f32(w.sbox + (ibase + c) * 8);
                        dec = fast_decision(s, A, B, thresh);
                        if (dec == 1 && (!w.strict || hulls_overlap_strict(A, B))) record(r, c);
                    }
                    const u64 need = __ballot(dec == 2);
                    if (need == 0ull) continue;
                    if (dec == 2) exq[nq + __popcll(need & ((1ull << lane) - 1ull))] = (u64)(unsigned)r | ((u64)(unsigned)c << 32);
                    nq += __popcll(need);
                    __builtin_amdgcn_wave_barrier();
                    while (nq >= 4) exact4(4);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (nq > 0) exact4(nq);
    flush_stats();
}

// ------------------------------------------------- nms_class_reduce / nms_compact
__device__ __forceinline__ u64 readlane64(u64 v, int l) {
    unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

constexpr int kReduceThreads = 1024;
constexpr int kFastBlk = 176;      // tiles of the current row block are parked in LDS, up to 176 of them (89 KiB)
constexpr int kMaxBlk = 1024;      // up to 65536 rows per image
constexpr int kMaxClsWG = 16;      // workgroups per image of the greedy pass (classes c, c + 16, .. each)

// Greedy scan over the suppression matrix, ONE WORKGROUP PER (image, class).
// After the class offsets of nms.py:81-83 the classes are independent greedy problems (class-major images, see
// nms_sort_prep_kernel); an image that is not class-major (plain [M,9] rows, fp64 ResultMerge rows, >= 2 zero-area
// boxes) is one problem handled by workgroup 0.  A problem is a serial chain over its 64-row blocks:
//  * the 64 diagonal words of a block sit one per lane in wave 0 and the in-block scan is wave-uniform scalar code
//    reading them with v_readlane (no LDS round trip per step); the next block's diagonal word is prefetched;
//  * (<= kFastBlk later blocks) the words of tiles (b, b+1..) are fetched one iteration AHEAD into registers,
//    coalesced (a tile is 512 contiguous bytes), then parked in LDS, so the OR phase of block b -- every later block
//    gets a thread -- reads LDS instead of chasing global loads that depend on the scan's result.
// (The first version gave every class one WAVE of a single workgroup per image: the OR phase was that wave's
// dependent global loads, ~3.4-5.9 us per block -- 579 us for the 169-block class of the skewed 27 000-row set.)
// Kept rows go to w.keptw (tile order, atomicOr: a block on a class boundary is written by two workgroups);
// nms_compact_kernel turns them into the keep list.
struct ChainCtx {
    const u64* mask;
    const u64* rowflag;
    u64* keptw;
    u64* remv;        // LDS, indexed b - b0
    u64* kcur;        // LDS
    u64* tbuf;        // LDS [kFastBlk][65]
    int nb, r0, r1, b0, b1;
};

// in-block scan of block b by wave 0 (wave-uniform scalar code over the diagonal words held one per lane)
// (the diagonal word AND the row flags of block b + 1 are fetched while block b is scanned: neither depends on the
// scan, and a load issued inside the chain costs every block a global-memory latency)
__device__ __forceinline__ void chain_scan(const ChainCtx& C, int b, u64& dcur, u64& rfcur) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        u64 dnext = 0ull, rfnext = 0ull;
        if (b < C.b1) {
            dnext = C.mask[tile_id(b + 1, b + 1, C.nb) * kTile + tid];
            rfnext = C.rowflag[b + 1];
        }
        const int lo = max(C.r0, b * kTile) - b * kTile, hi = min(C.r1, (b + 1) * kTile) - b * kTile;
        const u64 rowmask = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
        const u64 dw = ((rowmask >> tid) & 1ull) ? dcur : 0ull;
        // rem only grows and row r's bit can only be set by rows < r (upper-triangular words), so
        // kept = ~rem_final; only rows whose diag word is non-zero can change rem -> walk just those.
        u64 bits = __ballot(dw != 0ull);
        const unsigned dlo = (unsigned)dw, dhi = (unsigned)(dw >> 32);
        u64 rem = C.remv[b - C.b0];                   // same address in every lane: wave-uniform
        rem = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rem >> 32)) << 32) |
              (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rem);
        rem |= ~rowmask;
        bits &= ~rem;                                 // rows already removed never act: skipped without an iteration
        while (bits) {
            const int r = __builtin_amdgcn_readfirstlane(__ffsll((long long)bits) - 1);
            const unsigned l2 = (unsigned)__builtin_amdgcn_readlane((int)dlo, r);
            const unsigned h2 = (unsigned)__builtin_amdgcn_readlane((int)dhi, r);
            rem |= ((u64)h2 << 32) | (u64)l2;         // row r is kept (not in rem): it removes its columns
            bits &= ~rem & ~((2ull << r) - 1ull);      // next: the first later row that is still alive
        }
        if (tid == 0) {
            const u64 K = ~rem;
            if (K) atomicOr(&C.keptw[b], K);
            *C.kcur = K & rfcur;
        }
        dcur = dnext;
        rfcur = rfnext;
    }
}

// Serial chain over the blocks of one class with the tiles of the next D - 1 row blocks in flight (a ring of D register
// sets of RC words per thread; RC * 1024 / 64 = the most later blocks a row block of this class can have): a step parks
// its set in LDS and refills it with row block b + D, so the global-load latency (~2 us with one workgroup's worth of
// requests in flight) is spread over D steps instead of being paid by every one.  D * RC words must stay inside the
// 128 VGPRs a 1024-thread workgroup gets (RC = 12 with D = 4 spilled: 344 -> 651 us on the 169-block class).
template <int RC, int D>
__device__ __forceinline__ void chain_fast(const ChainCtx& C) {
    static_assert(D == 2 || D == 4, "ring depth");
    const int tid = threadIdx.x;
    u64 pre[D][RC];
    auto fetch = [&](int rb, u64 (&dst)[RC]) {        // tiles (rb, rb+1 ..): e = (wd - rb - 1) * 64 + row
#pragma unroll
        for (int k = 0; k < RC; k++) {
            const int e = tid + k * kReduceThreads;
            const int wd = rb + 1 + (e >> 6);
            dst[k] = (rb <= C.b1 && wd <= C.b1) ? C.mask[tile_id(rb, wd, C.nb) * kTile + (e & 63)] : 0ull;
        }
    };
    auto park = [&](const u64 (&src)[RC]) {
#pragma unroll
        for (int k = 0; k < RC; k++) {
            const int e = tid + k * kReduceThreads;
            C.tbuf[(e >> 6) * 65 + (e & 63)] = src[k];
        }
    };
    u64 dcur = 0ull, rfcur = 0ull;                    // wave 0: this lane's diagonal word / the row flags of the current block
    if (tid < 64) {
        dcur = C.mask[tile_id(C.b0, C.b0, C.nb) * kTile + tid];
        rfcur = C.rowflag[C.b0];
    }
#pragma unroll
    for (int i = 0; i < D; i++) fetch(C.b0 + i, pre[i]);
    auto step = [&](int b, u64 (&set)[RC]) {
        park(set);
        fetch(b + D, set);
        chain_scan(C, b, dcur, rfcur);
        __syncthreads();
        const u64 K2 = *C.kcur;
        const int wd = b + 1 + tid;
        if (K2 && wd <= C.b1) {
            u64 acc = 0ull;
            u64 bits = K2;
            while (bits) {
                const int r = __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                acc |= C.tbuf[tid * 65 + r];
            }
            C.remv[wd - C.b0] |= acc;
        }
        __syncthreads();
    };
    for (int b = C.b0; b <= C.b1; b += D) {
#pragma unroll
        for (int i = 0; i < D; i++)
            if (b + i <= C.b1) step(b + i, pre[i]);
    }
}

__global__ void __launch_bounds__(kReduceThreads) nms_class_reduce_kernel(const int* __restrict__ counts, int m_cap, NmsWs w) {
    const int img = blockIdx.y;
    const int M = img_count(counts, img, m_cap);
    if (M == 0) return;
    const int nb = w.nblk;
    const int ncls = w.use_perm ? (int)w.meta[img * 4 + 2] : 0;
    __shared__ u64 remv[kMaxBlk];                     // indexed b - b0
    __shared__ u64 kcur;
    extern __shared__ u64 tbuf[];                     // [kFastBlk][65] tile rows of the current block
    const int tid = threadIdx.x;
    // workgroup c of an image walks classes c, c + gridDim.x, ..: 16 workgroups per image cover DOTA's 15 / 16 classes one
    // each (512 mostly empty 1024-thread workgroups with 89 KB of LDS each queued behind the convolutions of the
    // concurrent streams: 37 us alone became 244 us in the timed layout)
    for (int c = blockIdx.x; c < (ncls > 0 ? ncls : 1); c += gridDim.x) {
        int r0 = 0, r1 = M;
        if (ncls > 0) {
            r0 = w.cbase[(size_t)img * 65 + c];
            r1 = w.cbase[(size_t)img * 65 + c + 1];
            if (r1 <= r0) continue;
        }
        ChainCtx C;
        C.mask = w.mask + (size_t)img * w.mask_words;
        C.rowflag = w.rowflag + (size_t)img * nb;
        C.keptw = w.keptw + (size_t)img * nb;
        C.remv = remv; C.kcur = &kcur; C.tbuf = tbuf;
        C.nb = nb; C.r0 = r0; C.r1 = r1; C.b0 = r0 >> 6; C.b1 = (r1 - 1) >> 6;
        const int later = C.b1 - C.b0;                // later blocks of the first row block
        __syncthreads();                              // the previous class of this workgroup is done with remv / tbuf
        for (int k = tid; k <= later; k += kReduceThreads) remv[k] = 0ull;
        __syncthreads();
        if (later <= 16) { chain_fast<1, 4>(C); continue; }
        if (later <= 64) { chain_fast<4, 4>(C); continue; }
        if (later <= kFastBlk) { chain_fast<kFastBlk * kTile / kReduceThreads, 2>(C); continue; }
        // more than kFastBlk blocks in one chain (plain rows of a TTA-sized set): OR phase straight from global memory
        u64 dcur = 0ull, rfcur = 0ull;
        if (tid < 64) {
            dcur = C.mask[tile_id(C.b0, C.b0, nb) * kTile + tid];
            rfcur = C.rowflag[C.b0];
        }
        for (int b = C.b0; b <= C.b1; b++) {
            chain_scan(C, b, dcur, rfcur);
            __syncthreads();
            const u64 K2 = kcur;
            if (K2) {
                for (int wd = b + 1 + tid; wd <= C.b1; wd += kReduceThreads) {
                    const u64* trow = C.mask + tile_id(b, wd, nb) * kTile;
                    u64 acc = 0ull;
                    u64 kb = K2;
                    while (kb) {                              // four independent loads in flight per step
                        int rr[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            rr[q] = kb ? __ffsll((long long)kb) - 1 : -1;
                            kb &= kb - 1;                     // (0 stays 0)
                        }
                        u64 v[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) v[q] = rr[q] >= 0 ? trow[rr[q]] : 0ull;
                        acc |= (v[0] | v[1]) | (v[2] | v[3]);
                    }
                    remv[wd - C.b0] |= acc;
                }
            }
            __syncthreads();
        }
    }
}

// Kept bits (tile order) -> keep list in GLOBAL score order + the cap of select_over_all_levels
// (dafne_outputs.py:916-923).  One workgroup per image.
__global__ void __launch_bounds__(kReduceThreads) nms_compact_kernel(
    const int* __restrict__ counts, int m_cap, int post_topk, NmsWs w,
    long long* __restrict__ keep, int* __restrict__ num_keep) {
    const int img = blockIdx.x;
    const int M = img_count(counts, img, m_cap);
    const int nb = w.nblk;
    const int nbu = (M + kTile - 1) / kTile;  // blocks in use
    const size_t ibase = (size_t)img * w.Mp;
    long long* kout = keep + (size_t)img * m_cap;
    __shared__ u64 kept[kMaxBlk];
    __shared__ int kpre[kMaxBlk + 1];
    const int tid = threadIdx.x;
    for (int k = tid; k < nb; k += kReduceThreads) kept[k] = k < nbu ? w.keptw[(size_t)img * nb + k] : 0ull;
    __syncthreads();

    if (w.use_perm) {
        // The kept bits live in tile (class-major) order; the keep list is emitted in GLOBAL score order:
        // thread t walks score-order positions [t*cs, (t+1)*cs) through perm.
        const int* perm = w.perm + ibase;
        __shared__ int tcnt[kReduceThreads + 1];
        __shared__ int n_out2;
        const int cs = (M + kReduceThreads - 1) / kReduceThreads;
        const int p_lo = min(M, tid * cs), p_hi = min(M, p_lo + cs);
        int mine = 0;
        for (int p = p_lo; p < p_hi; p++) {
            const int q = perm[p];
            mine += (int)((kept[q >> 6] >> (q & 63)) & 1ull);
        }
        tcnt[tid] = mine;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < kReduceThreads; k++) { const int c = tcnt[k]; tcnt[k] = run; run += c; }
            tcnt[kReduceThreads] = run;
            n_out2 = run;
        }
        __syncthreads();
        const int total = tcnt[kReduceThreads];
        // cap (dafne_outputs.py:916-923): keep every row whose score >= the post_topk-th best kept score;
        // kept rows are walked in descending score order, so that is a prefix (+ the ties behind it)
        if (post_topk > 0 && total > post_topk) {
            if (tcnt[tid] < post_topk && tcnt[tid] + mine >= post_topk) {      // exactly one thread
                int seen = tcnt[tid], p = p_lo;
                for (; p < p_hi; p++) {
                    const int q = perm[p];
                    if ((kept[q >> 6] >> (q & 63)) & 1ull)
                        if (++seen == post_topk) break;
                }
                const float thr = w.sscore[ibase + p];
                int n = post_topk;
                for (p = p + 1; p < M; p++) {
                    const int q = perm[p];
                    if ((kept[q >> 6] >> (q & 63)) & 1ull) {
                        if (w.sscore[ibase + p] >= thr) n++; else break;
                    }
                }
                n_out2 = n;
            }
            __syncthreads();
        }
        const int nout = n_out2;
        int o = tcnt[tid];
        for (int p = p_lo; p < p_hi && o < nout; p++) {
            const int q = perm[p];
            if ((kept[q >> 6] >> (q & 63)) & 1ull) kout[o++] = (long long)w.order[ibase + q];
        }
        if (tid == 0) num_keep[img] = nout;
        return;
    }

    // ordered compaction of the kept sorted positions
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k < nbu; k++) {
            kpre[k] = run;
            run += __popcll(kept[k]);
        }
        kpre[nbu] = run;
    }
    __syncthreads();
    int total = nbu > 0 ? kpre[nbu] : 0;
    // cap (dafne_outputs.py:916-923): keep every row whose score >= the post_topk-th
    // best kept score; kept rows are in descending score order, so that is a prefix.
    __shared__ int n_out;
    if (tid == 0) n_out = total;
    __syncthreads();
    if (post_topk > 0 && total > post_topk) {
        // find the sorted position of the post_topk-th kept row, then extend over ties
        if (tid == 0) {
            int blk = 0;
            while (kpre[blk + 1] < post_topk) blk++;
            int pos = blk * kTile + kth_set_bit(kept[blk], post_topk - 1 - kpre[blk]);
            float thr = w.sscore[ibase + pos];
            int n = post_topk;
            // walk the following kept rows while they tie with thr
            int p = pos + 1;
            while (p < M) {
                if ((kept[p >> 6] >> (p & 63)) & 1ull) {
                    if (w.sscore[ibase + p] >= thr) n++; else break;
                }
                p++;
            }
            n_out = n;
        }
        __syncthreads();
    }
    const int nout = n_out;
    for (int k = tid; k < nbu; k += kReduceThreads) {
        u64 bits = kept[k];
        int o = kpre[k];
        while (bits && o < nout) {
            int r = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            kout[o++] = (long long)w.order[ibase + k * kTile + r];
        }
    }
    if (tid == 0) num_keep[img] = nout;
}

int run_nms(const float* d_dets9, int row_cap, const int* d_counts, int N, int m_cap, double thresh,
            int post_topk, int64_t* d_keep, int32_t* d_num_keep, NmsWs& w, hipStream_t st,
            bool meta_zeroed, const double* d_dets9_f64 = nullptr) {
    if (!meta_zeroed) {
        DAFNE_HIP_TRY(hipMemsetAsync(w.meta, 0, zero_bytes(w, N), st));
    }
    dim3 gp((m_cap + 255) / 256, N);
    if (d_dets9_f64)
        hipLaunchKernelGGL(nms_prep_f64_kernel, gp, dim3(256), 0, st, d_dets9_f64, row_cap, d_counts, m_cap, w);
    else if (m_cap <= kSortMax) {
        DAFNE_MAX_LDS_ONCE(kSortMax * (int)sizeof(u64) + kSortHistBytes, (const void*)nms_sort_prep_kernel);
        const int n = ((m_cap + 1023) >> 10) << 10;                   // padded sequence of the largest image
        w.use_perm = 1;
        hipLaunchKernelGGL(nms_sort_prep_kernel, dim3(N), dim3(1024), (size_t)n * sizeof(u64) + kSortHistBytes, st, d_dets9, row_cap,
                           d_counts, m_cap, w);
        hipLaunchKernelGGL(nms_gather_kernel, gp, dim3(256), 0, st, d_dets9, row_cap, d_counts, m_cap, w);
    } else {
        w.use_perm = 1;
        if (w.cls) hipLaunchKernelGGL(nms_cls_layout_kernel, dim3(N), dim3(1024), 0, st, d_dets9, row_cap, d_counts, m_cap, w);
        if (m_cap < (1 << 24)) {
            DAFNE_MAX_LDS_ONCE(kSortMax * (int)sizeof(u64) + kSortHistBytes, (const void*)nms_chunk_sort_kernel);
            const int K = (m_cap + kSortMax - 1) / kSortMax;
            hipLaunchKernelGGL(nms_chunk_sort_kernel, dim3(K, N, 2), dim3(1024), (size_t)kSortMax * sizeof(u64) + kSortHistBytes, st,
                               d_dets9, row_cap, d_counts, m_cap, w, N);
            hipLaunchKernelGGL(nms_merge_rank_kernel, gp, dim3(256), 0, st, d_counts, m_cap, w, N, 1);
            hipLaunchKernelGGL(nms_merge_rank_kernel, gp, dim3(256), 0, st, d_counts, m_cap, w, N, 0);
            hipLaunchKernelGGL(nms_gather_kernel, gp, dim3(256), 0, st, d_dets9, row_cap, d_counts, m_cap, w);
        } else {
            hipLaunchKernelGGL(nms_prep_kernel, gp, dim3(256), 0, st, d_dets9, row_cap, d_counts, m_cap, w);
        }
    }
    int rc = dafne::check_launch("nms_prep");
    if (rc) return rc;
    long long ntiles = (long long)w.nblk * (w.nblk + 1) / 2;
    if (ntiles > 0x7fffffffLL) return dafne::fail(DAFNE_E_UNSUPPORTED, "too many NMS tiles");
    hipLaunchKernelGGL(nms_scan_kernel, dim3((unsigned)((ntiles + 3) / 4), N), dim3(256), 0, st, d_counts, m_cap,
                       thresh, w, ntiles);
    rc = dafne::check_launch("nms_scan");
    if (rc) return rc;
    const int iou_blocks = 256 * 6 / (N < 6 ? N : 6) + 1;      // ~6 resident waves per CU in total
    hipLaunchKernelGGL(nms_iou_kernel, dim3(iou_blocks, N), dim3(64), 0, st, d_counts, m_cap, thresh, w, ntiles);
    rc = dafne::check_launch("nms_iou");
    if (rc) return rc;
    DAFNE_MAX_LDS_ONCE(kFastBlk * 65 * (int)sizeof(u64), (const void*)nms_class_reduce_kernel);
    // class-major images: one workgroup per class; others: workgroup 0 walks the whole image (the unused ones exit)
    hipLaunchKernelGGL(nms_class_reduce_kernel, dim3(w.cls && w.use_perm ? kMaxClsWG : 1, N), dim3(kReduceThreads),
                       (size_t)kFastBlk * 65 * sizeof(u64), st, d_counts, m_cap, w);
    rc = dafne::check_launch("nms_class_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_compact_kernel, dim3(N), dim3(kReduceThreads), 0, st, d_counts, m_cap, post_topk, w,
                       reinterpret_cast<long long*>(d_keep), d_num_keep);
    return dafne::check_launch("nms_compact");
}

}  // namespace

extern "C" {

int dafne_poly_iou_pairs_hip(const double* d_p, const double* d_q, int64_t n, double* d_out,
                             void* stream) {
    if (n < 0 || (n > 0 && (!d_p || !d_q || !d_out))) return dafne::fail(DAFNE_E_INVALID, "iou_pairs: bad args");
    if (n == 0) return DAFNE_OK;
    long long blocks = (n + 3) / 4;
    hipLaunchKernelGGL(iou_pairs_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, d_p,
                       d_q, (long long)n, d_out);
    return dafne::check_launch("iou_pairs");
}

size_t dafne_poly_nms_stats_offset(int n_images, int m_cap, int f64_rows) {
    if (n_images <= 0 || m_cap < 0) return 0;
    NmsWs w;
    carve(w, nullptr, n_images, m_cap, f64_rows != 0);
    return (size_t)reinterpret_cast<uintptr_t>(w.stats);
}

size_t dafne_poly_nms_workspace_bytes(int n_images, int m_cap) {
    if (n_images <= 0 || m_cap < 0) return 0;
    NmsWs w;
    return carve(w, nullptr, n_images, m_cap);
}

int dafne_poly_nms_batched_hip(const float* d_dets9, const int32_t* d_counts, int n_images,
                               int m_cap, double thresh, int post_topk, int64_t* d_keep,
                               int32_t* d_num_keep, void* d_ws, size_t ws_bytes, int flags, void* stream) {
    if (n_images <= 0 || m_cap < 0 || !d_num_keep) return dafne::fail(DAFNE_E_INVALID, "poly_nms: bad args");
    if (flags & ~DAFNE_NMS_EXACT_ONLY) return dafne::fail(DAFNE_E_INVALID, "poly_nms: unknown flags 0x%x", flags);
    hipStream_t st = (hipStream_t)stream;
    if (m_cap == 0) {
        DAFNE_HIP_TRY(hipMemsetAsync(d_num_keep, 0, sizeof(int32_t) * n_images, st));
        return DAFNE_OK;
    }
    if (!d_dets9 || !d_keep || !d_ws) return dafne::fail(DAFNE_E_INVALID, "poly_nms: null pointer");
    NmsWs w;
    size_t need = carve(w, d_ws, n_images, m_cap, false, flags);
    if (ws_bytes < need) return dafne::fail(DAFNE_E_WORKSPACE, "poly_nms: workspace %zu < %zu", ws_bytes, need);
    if (w.nblk > kMaxBlk) return dafne::fail(DAFNE_E_UNSUPPORTED, "poly_nms: m_cap %d > %d", m_cap, kMaxBlk * kTile);
    w.cls = nullptr;           // plain [M,9] rows carry no class: score order is the tile order
    return run_nms(d_dets9, m_cap, d_counts, n_images, m_cap, thresh, post_topk, d_keep, d_num_keep, w, st, false);
}

size_t dafne_poly_nms_f64_workspace_bytes(int n_images, int m_cap) {
    if (n_images <= 0 || m_cap < 0) return 0;
    NmsWs w;
    return carve(w, nullptr, n_images, m_cap, true);
}

int dafne_poly_nms_f64_batched_hip(const double* d_dets9, const int32_t* d_counts, int n_images,
                                   int m_cap, double thresh, int strict_hbb, int64_t* d_keep,
                                   int32_t* d_num_keep, void* d_ws, size_t ws_bytes, int flags, void* stream) {
    if (n_images <= 0 || m_cap < 0 || !d_num_keep) return dafne::fail(DAFNE_E_INVALID, "poly_nms_f64: bad args");
    if (flags & ~DAFNE_NMS_EXACT_ONLY) return dafne::fail(DAFNE_E_INVALID, "poly_nms_f64: unknown flags 0x%x", flags);
    hipStream_t st = (hipStream_t)stream;
    if (m_cap == 0) {
        DAFNE_HIP_TRY(hipMemsetAsync(d_num_keep, 0, sizeof(int32_t) * n_images, st));
        return DAFNE_OK;
    }
    if (!d_dets9 || !d_keep || !d_ws) return dafne::fail(DAFNE_E_INVALID, "poly_nms_f64: null pointer");
    NmsWs w;
    size_t need = carve(w, d_ws, n_images, m_cap, true, flags);
    if (ws_bytes < need) return dafne::fail(DAFNE_E_WORKSPACE, "poly_nms_f64: workspace %zu < %zu", ws_bytes, need);
    if (w.nblk > kMaxBlk) return dafne::fail(DAFNE_E_UNSUPPORTED, "poly_nms_f64: m_cap %d > %d", m_cap, kMaxBlk * kTile);
    w.strict = strict_hbb ? 1 : 0;
    return run_nms(nullptr, m_cap, d_counts, n_images, m_cap, thresh, 0, d_keep, d_num_keep, w, st, false, d_dets9);
}

int dafne_poly_nms_hip(const float* d_dets9, int M, double thresh, int64_t* d_keep,
                       int32_t* d_num_keep, void* d_ws, size_t ws_bytes, int flags, void* stream) {
    return dafne_poly_nms_batched_hip(d_dets9, nullptr, 1, M, thresh, 0, d_keep, d_num_keep, d_ws,
                                      ws_bytes, flags, stream);
}

int dafne_select_over_all_levels_hip(const float* d_boxes8, const float* d_scores,
                                     const int32_t* d_classes, const int32_t* d_counts,
                                     int n_images, int m_cap, double nms_thresh, int post_topk,
                                     int64_t* d_keep, int32_t* d_num_keep, void* d_ws,
                                     size_t ws_bytes, int flags, void* stream) {
    if (n_images <= 0 || m_cap < 0 || !d_num_keep) return dafne::fail(DAFNE_E_INVALID, "select: bad args");
    if (flags & ~DAFNE_NMS_EXACT_ONLY) return dafne::fail(DAFNE_E_INVALID, "select: unknown flags 0x%x", flags);
    hipStream_t st = (hipStream_t)stream;
    if (m_cap == 0) {
        DAFNE_HIP_TRY(hipMemsetAsync(d_num_keep, 0, sizeof(int32_t) * n_images, st));
        return DAFNE_OK;
    }
    if (!d_boxes8 || !d_scores || !d_classes || !d_keep || !d_ws)
        return dafne::fail(DAFNE_E_INVALID, "select: null pointer");
    if ((uintptr_t)d_boxes8 & 15) return dafne::fail(DAFNE_E_INVALID, "select: d_boxes8 must be 16-byte aligned");
    if (!(nms_thresh > 0)) return dafne::fail(DAFNE_E_UNSUPPORTED, "select: nms_thresh <= 0 bypasses NMS in the reference; handle on the caller side");
    NmsWs w;
    size_t need = carve(w, d_ws, n_images, m_cap, false, flags);
    if (ws_bytes < need) return dafne::fail(DAFNE_E_WORKSPACE, "select: workspace %zu < %zu", ws_bytes, need);
    if (w.nblk > kMaxBlk) return dafne::fail(DAFNE_E_UNSUPPORTED, "select: m_cap %d > %d", m_cap, kMaxBlk * kTile);
    DAFNE_HIP_TRY(hipMemsetAsync(w.meta, 0, zero_bytes(w, n_images), st));
    hipLaunchKernelGGL(nms_minmax_kernel, dim3(n_images), dim3(1024), 0, st, d_boxes8, d_counts, m_cap, w.meta);
    int rc = dafne::check_launch("nms_minmax");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_offset_kernel, dim3((m_cap + 255) / 256, n_images), dim3(256), 0, st, d_boxes8,
                       d_scores, d_classes, d_counts, m_cap, w.Mp, w.meta, w.dets9, w.cls, w.nzero);
    rc = dafne::check_launch("nms_offset");
    if (rc) return rc;
    return run_nms(w.dets9, w.Mp, d_counts, n_images, m_cap, nms_thresh, post_topk, d_keep, d_num_keep, w, st, true);
}

}  // extern "C"
