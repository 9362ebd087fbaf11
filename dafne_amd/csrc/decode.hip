// Decode / threshold / per-level top-k / canonical corner order / gather for gfx950.
//
// Replaces the chain of small torch ops in DAFNeOutputs.forward_for_single_feature_map
// (dafne/modeling/dafne/dafne_outputs.py:792-905, stride multiply :771-772), the
// centre-to-corner epilogue of DAFNeHead.forward (dafne/modeling/dafne/dafne.py:405-411),
// compute_locations (dafne.py:37-44), sort_quadrilateral (dafne/utils/sort_corners.py:26-92)
// and detector_postprocess + OneStageDetector._postprocess
// (dafne/modeling/one_stage_detector.py:79-98).
//
// HBM-bound integer/selection work, no host round trips (the reference syncs with
// .item() per image and level, :851):
//   decode_hist      every (image, level, location, class): score, threshold test,
//                    11-bit histogram of the candidates' score keys (LDS-private)
//   decode_pick      per (image, level): histogram bin that holds the k-th best score
//   decode_collect   candidates above that bin -> selected list S, inside it -> E
//                    (wave-aggregated appends)
//   decode_finalize  per (image, level): exact radix select inside E (score, then
//                    lowest flat index among equal scores), bitonic sort of S by flat
//                    index (= the reference's nonzero() order), box decode, corner sort
//   gather_kernel    kept rows -> final detection rows (+ rescale / clip / drop empty)
//
// fp32 arithmetic is written in the reference's operation order and this file is
// compiled with -ffp-contract=off.
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int kMaxLevels = 8;
constexpr int kBins = 2048;
constexpr int kChunk = 4096;     // elements per decode_hist / decode_collect block
constexpr int kMaxTopk = 4096;

struct LevelDev {
    const float* logits;
    const float* delta;
    const float* center;
    const float* ctrness;
    int logits_ps, delta_ps, center_ps, ctrness_ps;
    int H, W, stride;
    float scale;
    int chunk0;      // first chunk id of this level
    int n_elem;      // H*W*C
    size_t e_off;    // offset of this level's E list (entries) inside one image's E area
};

struct DecodeDev {
    LevelDev lv[kMaxLevels];
    int n_images, n_levels, C, topk, twc, sortc, m_cap;
    float thresh;
    unsigned key_lo;   // score bits are keyed as (bits - key_lo)
    int shift;         // histogram bin = key >> shift
    int n_chunks;      // per image
    int kp;            // topk rounded up to a power of two (sort width)
    size_t e_per_image;
    // workspace
    unsigned* hist;    // [N][L][kBins]
    int* meta;         // [N][L][4]: b1, k1, ncand, nsel
    unsigned* s_cnt;   // [N][L]
    unsigned* e_cnt;   // [N][L]
    u64* S;            // [N][L][kp]
    u64* E;            // [N][e_per_image]
};

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// score + candidate test for element (loc, c) of one level (dafne_outputs.py:804-829)
__device__ __forceinline__ bool score_of(const DecodeDev& P, const LevelDev& L, int img, int loc, int c,
                                         float& score, float& ctr_out) {
    const size_t px = (size_t)img * L.H * L.W + loc;
    float cls = sigmoidf_ref(L.logits[px * L.logits_ps + c]);
    float ctr = sigmoidf_ref(L.ctrness[px * L.ctrness_ps]);
    ctr_out = ctr;
    bool cand;
    if (P.twc) {
        score = sqrtf(cls * ctr);
        cand = score > P.thresh;
    } else {
        cand = cls > P.thresh;
        score = sqrtf(cls * ctr);
    }
    return cand;
}

__device__ __forceinline__ unsigned key_of(const DecodeDev& P, float score) {
    unsigned b = __float_as_uint(score);
    return b > P.key_lo ? b - P.key_lo : 0u;   // scores are positive; clamp is for safety
}

__device__ __forceinline__ int find_level(const DecodeDev& P, int chunk) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < kMaxLevels; k++)
        if (k < P.n_levels && chunk >= P.lv[k].chunk0) l = k;
    return l;
}

// ----------------------------------------------------------------- decode_hist
__global__ void __launch_bounds__(256) decode_hist_kernel(DecodeDev P) {
    __shared__ unsigned h[kBins];
    const int img = blockIdx.y;
    const int lvl = find_level(P, blockIdx.x);
    const LevelDev& L = P.lv[lvl];
    for (int k = threadIdx.x; k < kBins; k += 256) h[k] = 0;
    __syncthreads();
    const int e0 = (blockIdx.x - L.chunk0) * kChunk;
    for (int e = e0 + threadIdx.x; e < min(e0 + kChunk, L.n_elem); e += 256) {
        int loc = e / P.C, c = e - loc * P.C;
        float s, ctr;
        if (score_of(P, L, img, loc, c, s, ctr)) {
            unsigned bin = min(key_of(P, s) >> P.shift, (unsigned)(kBins - 1));
            atomicAdd(&h[bin], 1u);
        }
    }
    __syncthreads();
    unsigned* g = P.hist + ((size_t)img * P.n_levels + lvl) * kBins;
    for (int k = threadIdx.x; k < kBins; k += 256)
        if (h[k]) atomicAdd(&g[k], h[k]);
}

// ----------------------------------------------------------------- decode_pick
__global__ void __launch_bounds__(64) decode_pick_kernel(DecodeDev P) {
    const int img = blockIdx.y, lvl = blockIdx.x, lane = threadIdx.x;
    const unsigned* g = P.hist + ((size_t)img * P.n_levels + lvl) * kBins;
    // lane owns bins [32*lane, 32*lane+32); suffix sums from the top
    unsigned mine = 0;
    for (int k = 0; k < 32; k++) mine += g[lane * 32 + k];
    unsigned suf = mine;   // inclusive suffix: bins >= 32*lane
    for (int o = 1; o < 64; o <<= 1) {
        unsigned v = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += v;
    }
    const unsigned total = __shfl(suf, 0, 64);
    int* meta = P.meta + ((size_t)img * P.n_levels + lvl) * 4;
    if (total <= (unsigned)P.topk) {
        if (lane == 0) {
            meta[0] = -1;
            meta[1] = 0;
            meta[2] = (int)total;
            meta[3] = (int)total;
        }
        return;
    }
    // the crossing lane: above = suf - mine < k <= suf
    const unsigned above = suf - mine;
    if (above < (unsigned)P.topk && suf >= (unsigned)P.topk) {
        unsigned acc = above;
        int b1 = lane * 32;
        for (int k = 31; k >= 0; k--) {
            unsigned hk = g[lane * 32 + k];
            if (acc + hk >= (unsigned)P.topk) {
                b1 = lane * 32 + k;
                break;
            }
            acc += hk;
        }
        meta[0] = b1;
        meta[1] = P.topk - (int)acc;   // how many to take from bin b1
        meta[2] = (int)total;
        meta[3] = P.topk;
    }
}

// -------------------------------------------------------------- decode_collect
__device__ __forceinline__ void wave_append(bool pred, u64 entry, u64* list, unsigned* counter,
                                            unsigned cap) {
    u64 b = __ballot(pred);
    if (!b) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    const int leader = __ffsll((long long)b) - 1;
    if (lane == leader) base = atomicAdd(counter, (unsigned)__popcll(b));
    base = __shfl(base, leader, 64);
    if (pred) {
        unsigned slot = base + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
        if (slot < cap) list[slot] = entry;
    }
}

__global__ void __launch_bounds__(256) decode_collect_kernel(DecodeDev P) {
    const int img = blockIdx.y;
    const int lvl = find_level(P, blockIdx.x);
    const LevelDev& L = P.lv[lvl];
    const size_t il = (size_t)img * P.n_levels + lvl;
    const int b1 = P.meta[il * 4 + 0];
    u64* S = P.S + il * P.kp;
    u64* E = P.E + (size_t)img * P.e_per_image + L.e_off;
    const int e0 = (blockIdx.x - L.chunk0) * kChunk;
    const int e1 = min(e0 + kChunk, L.n_elem);
    for (int eb = e0; eb < e1; eb += 256) {   // uniform trip count: ballots need whole waves
        int e = eb + threadIdx.x;
        bool live = e < e1, cand = false;
        float s = 0.f, ctr;
        if (live) {
            int loc = e / P.C, c = e - loc * P.C;
            cand = score_of(P, L, img, loc, c, s, ctr);
        }
        int bin = cand ? (int)min(key_of(P, s) >> P.shift, (unsigned)(kBins - 1)) : -2;
        u64 entry = ((u64)__float_as_uint(s) << 32) | (unsigned)e;
        wave_append(cand && bin > b1, entry, S, P.s_cnt + il, (unsigned)P.kp);
        wave_append(cand && bin == b1, entry, E, P.e_cnt + il, (unsigned)L.n_elem);
    }
}

// ------------------------------------------------------- sort_quadrilateral
__device__ __forceinline__ float pick4(float a, float b, float c, float d, int i) {
    float r = a;
    if (i == 1) r = b;
    if (i == 2) r = c;
    if (i == 3) r = d;
    return r;
}

__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) {
    return ax * by - ay * bx;   // sort_corners.py:5-7 (two products, one subtraction)
}

// sort_corners.py:26-92, one box per call; q = x0,y0,..,x3,y3 in place.
__device__ void sort_quad(float* q) {
    const float x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
    int k1 = 0;   // first vertex of minimal x (:46)
    float mx = x0;
    if (x1 < mx) { mx = x1; k1 = 1; }
    if (x2 < mx) { mx = x2; k1 = 2; }
    if (x3 < mx) { mx = x3; k1 = 3; }
    const float p1x = pick4(x0, x1, x2, x3, k1), p1y = pick4(y0, y1, y2, y3, k1);
    float rx[3], ry[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        int src = j < k1 ? j : j + 1;
        rx[j] = pick4(x0, x1, x2, x3, src);
        ry[j] = pick4(y0, y1, y2, y3, src);
    }
    float p3x = 0.f, p3y = 0.f, ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {   // :57-73
        const int i2 = i == 0 ? 1 : 0, i3 = i == 2 ? 1 : 2;
        const float dx = rx[i] - p1x, dy = ry[i] - p1y;
        const float l = cross2(dx, dy, rx[i2] - p1x, ry[i2] - p1y);
        const float r = cross2(dx, dy, rx[i3] - p1x, ry[i3] - p1y);
        const bool cond = (l * r < 0.0f) && !done;
        if (cond) {
            p3x = rx[i]; p3y = ry[i];
            ax = rx[i2]; ay = ry[i2];
            bx = rx[i3]; by = ry[i3];
        }
        done = done || cond;
    }
    // :77-90: iteration 0 tests A, iteration 1 tests B unless A already matched
    const float ex = p3x - p1x, ey = p3y - p1y;
    const bool c0 = cross2(ex, ey, ax - p1x, ay - p1y) > 0.0f;
    const bool c1 = cross2(ex, ey, bx - p1x, by - p1y) > 0.0f;
    const bool swap = !c0 && c1;
    q[0] = p1x; q[1] = p1y;
    q[2] = swap ? bx : ax; q[3] = swap ? by : ay;
    q[4] = p3x; q[5] = p3y;
    q[6] = swap ? ax : bx; q[7] = swap ? ay : by;
}

__global__ void __launch_bounds__(256) sort_quad_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = in[i * 8 + k];
    sort_quad(q);
#pragma unroll
    for (int k = 0; k < 8; k++) out[i * 8 + k] = q[k];
}

// ------------------------------------------------------------- decode_finalize
constexpr int kFinThreads = 1024;

// one radix step inside the workgroup: histogram `nbits` bits of f(entry) at
// `shift` over entries that pass `match`, then walk the bins (from the top when
// `descending`) until `need` is covered.  Returns the chosen bin and updates need.
template <typename KeyFn, typename MatchFn>
__device__ int radix_step(const u64* E, int n, int nbits, unsigned* h, KeyFn key, MatchFn match,
                          bool descending, int& need) {
    const int nb = 1 << nbits;
    for (int k = threadIdx.x; k < nb; k += kFinThreads) h[k] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kFinThreads) {
        u64 en = E[i];
        if (match(en)) atomicAdd(&h[key(en) & (nb - 1)], 1u);
    }
    __syncthreads();
    __shared__ int s_bin, s_need;
    // the bin in which the running count (from the top when `descending`) reaches `need`: wave 0, lane l owns `per`
    // consecutive bins in walk order; a wave scan finds the lane that holds the crossing, which then walks its own bins.
    // (Round 3 walked all 2048 bins on ONE thread -- dependent LDS reads, ~60 us per step, four steps per (image, level):
    // the whole 245 us of this kernel.)
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int per = nb >= 64 ? nb / 64 : 1;
        const int t0 = lane * per;                               // first walk position of this lane
        int mine = 0;
        if (t0 < nb)
            for (int t = 0; t < per; t++) mine += (int)h[descending ? nb - 1 - (t0 + t) : t0 + t];
        int incl = mine;                                         // inclusive prefix over lanes in walk order
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        const int before = incl - mine;
        const int total = __shfl(incl, 63, 64);
        if (need <= 0) {                                         // (not reached by the callers) the serial walk stops at once
            if (lane == 0) { s_bin = descending ? nb - 1 : 0; s_need = need; }
        } else if (total < need) {                               // never reached: the walk's defaults
            if (lane == 0) { s_bin = descending ? nb - 1 : 0; s_need = need - total; }
        } else if (before < need && incl >= need) {              // exactly one lane (need >= 1)
            int acc = before, bin = descending ? nb - 1 - t0 : t0;
            for (int t = 0; t < per; t++) {
                const int b = descending ? nb - 1 - (t0 + t) : t0 + t;
                const int hv = (int)h[b];
                if (acc + hv >= need) { bin = b; break; }
                acc += hv;
            }
            s_bin = bin;
            s_need = need - acc;
        }
    }
    __syncthreads();
    need = s_need;
    int r = s_bin;
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(kFinThreads) decode_finalize_kernel(
    DecodeDev P, float* __restrict__ o_corners, float* __restrict__ o_scores, float* __restrict__ o_ctr,
    int* __restrict__ o_classes, float* __restrict__ o_locs, int* __restrict__ o_levels,
    float* __restrict__ o_hbox, int* __restrict__ o_counts) {
    const int img = blockIdx.y, lvl = blockIdx.x, tid = threadIdx.x;
    const LevelDev& L = P.lv[lvl];
    const size_t il = (size_t)img * P.n_levels + lvl;
    const int* meta = P.meta + il * 4;
    const int b1 = meta[0];
    const int nsel = meta[3];
    u64* S = P.S + il * P.kp;
    const u64* E = P.E + (size_t)img * P.e_per_image + L.e_off;

    __shared__ unsigned h[kBins];
    __shared__ u64 srt[kMaxTopk];
    __shared__ unsigned s_app;

    if (b1 >= 0) {
        // exact selection of k1 entries out of bin b1: largest scores first, equal
        // scores by smallest flat index (oracle/postprocess.py decode_level)
        const int n = (int)min(P.e_cnt[il], (unsigned)L.n_elem);
        int need = meta[1];
        unsigned prefix = (unsigned)b1;      // key >> bits_left so far
        int bits_left = P.shift;
        while (bits_left > 0) {
            const int nbits = min(11, bits_left);
            const int sh = bits_left - nbits;
            const unsigned pf = prefix;
            const int bl = bits_left;
            const unsigned lo = P.key_lo;
            auto keyf = [=](u64 en) { unsigned k = (unsigned)(en >> 32); k = k > lo ? k - lo : 0u; return k >> sh; };
            auto matchf = [=](u64 en) { unsigned k = (unsigned)(en >> 32); k = k > lo ? k - lo : 0u; return (k >> bl) == pf; };
            int bin = radix_step(E, n, nbits, h, keyf, matchf, true, need);
            prefix = (prefix << nbits) | (unsigned)bin;
            bits_left = sh;
        }
        const unsigned tkey = prefix;        // exact key of the k-th best score
        // ties at tkey: take the `need` smallest flat indices
        int ibits = 1;
        while ((1 << ibits) < L.n_elem) ibits++;
        unsigned iprefix = 0;
        int ileft = ibits;
        const unsigned lo = P.key_lo;
        while (ileft > 0) {
            const int nbits = min(11, ileft);
            const int sh = ileft - nbits;
            const unsigned pf = iprefix;
            const int bl = ileft;
            auto keyf = [=](u64 en) { return (unsigned)en >> sh; };
            auto matchf = [=](u64 en) {
                unsigned k = (unsigned)(en >> 32); k = k > lo ? k - lo : 0u;
                return k == tkey && (((unsigned)en) >> bl) == pf;
            };
            int bin = radix_step(E, n, nbits, h, keyf, matchf, false, need);
            iprefix = (iprefix << nbits) | (unsigned)bin;
            ileft = sh;
        }
        const unsigned icut = iprefix;       // ties with flat index <= icut are taken
        if (tid == 0) s_app = P.s_cnt[il];
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += kFinThreads) {
            int i = i0 + tid;
            bool take = false;
            u64 en = 0;
            if (i < n) {
                en = E[i];
                unsigned k = (unsigned)(en >> 32); k = k > lo ? k - lo : 0u;
                take = k > tkey || (k == tkey && (unsigned)en <= icut);
            }
            wave_append(take, en, S, &s_app, (unsigned)P.kp);
        }
        __syncthreads();
    }

    // sort the selected entries by flat index (location major, class minor)
    for (int i = tid; i < P.kp; i += kFinThreads) {
        u64 en = i < nsel ? S[i] : ~0ull;
        srt[i] = i < nsel ? ((en << 32) | (en >> 32)) : ~0ull;   // (flat, score bits)
    }
    __syncthreads();
    for (int k = 2; k <= P.kp; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P.kp; i += kFinThreads) {
                int ixj = i ^ j;
                if (ixj > i) {
                    u64 a = srt[i], b = srt[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { srt[i] = b; srt[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }

    int off = 0;
    for (int l2 = 0; l2 < lvl; l2++) off += P.meta[((size_t)img * P.n_levels + l2) * 4 + 3];
    if (lvl == 0 && tid == 0) {
        int tot = 0;
        for (int l2 = 0; l2 < P.n_levels; l2++) tot += P.meta[((size_t)img * P.n_levels + l2) * 4 + 3];
        o_counts[img] = tot;
    }
    const float fs = (float)L.stride;
    const float half = (float)(L.stride / 2);
    for (int r = tid; r < nsel; r += kFinThreads) {
        const u64 en = srt[r];
        const int flat = (int)(en >> 32);
        const float score = __uint_as_float((unsigned)en);
        const int loc = flat / P.C, c = flat - loc * P.C;
        const int y = loc / L.W, x = loc - y * L.W;
        const float lx = (float)(x * L.stride) + half;   // dafne.py:37-44
        const float ly = (float)(y * L.stride) + half;
        const size_t px = (size_t)img * L.H * L.W + loc;
        const float* dl = L.delta + px * L.delta_ps;
        const float* ce = L.center + px * L.center_ps;
        const float cx = ce[0], cy = ce[1];
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float reg = ((j & 1) ? cy : cx) + dl[j];     // dafne.py:403  center.repeat + delta
            reg = reg * L.scale;                         // dafne.py:406  Scale
            reg = reg * fs;                              // dafne_outputs.py:771-772
            q[j] = ((j & 1) ? ly : lx) + reg;            // dafne_outputs.py:861-873
        }
        if (P.sortc) sort_quad(q);
        const size_t row = (size_t)img * P.m_cap + off + r;
        if (off + r < P.m_cap) {
#pragma unroll
            for (int j = 0; j < 8; j++) o_corners[row * 8 + j] = q[j];
            o_scores[row] = score;
            o_ctr[row] = sigmoidf_ref(L.ctrness[px * L.ctrness_ps]);
            o_classes[row] = c;
            o_locs[row * 2 + 0] = lx;
            o_locs[row * 2 + 1] = ly;
            o_levels[row] = lvl;
            o_hbox[row * 4 + 0] = fminf(fminf(q[0], q[2]), fminf(q[4], q[6]));
            o_hbox[row * 4 + 1] = fminf(fminf(q[1], q[3]), fminf(q[5], q[7]));
            o_hbox[row * 4 + 2] = fmaxf(fmaxf(q[0], q[2]), fmaxf(q[4], q[6]));
            o_hbox[row * 4 + 3] = fmaxf(fmaxf(q[1], q[3]), fmaxf(q[5], q[7]));
        }
    }
}

// -------------------------------------------------------------------- gather
__global__ void __launch_bounds__(1024) gather_kernel(
    const float* __restrict__ corners, const float* __restrict__ scores, const float* __restrict__ ctr,
    const int* __restrict__ classes, const float* __restrict__ locs, const int* __restrict__ levels,
    const float* __restrict__ hbox, const long long* __restrict__ keep, const int* __restrict__ num_keep,
    const float* __restrict__ sizes, int do_post, int m_cap, int k_cap, float* __restrict__ out,
    int* __restrict__ out_counts) {
    const int img = blockIdx.x, tid = threadIdx.x;
    const int nk = min(num_keep[img], m_cap);
    float sx = 1.f, sy = 1.f, cxs = 1.f, cys = 1.f, oh = 0.f, ow = 0.f;
    if (do_post) {
        const float* sz = sizes + img * 6;   // net_h, net_w, out_h, out_w, orig_h, orig_w
        oh = sz[2]; ow = sz[3];
        sx = (float)((double)sz[3] / (double)sz[1]);   // python float division, then fp32 multiply
        sy = (float)((double)sz[2] / (double)sz[0]);
        if (do_post >= 2) {   // OneStageDetector._postprocess (skipped by the TTA path)
            cxs = (float)((double)sz[3] / (double)sz[5]);
            cys = (float)((double)sz[2] / (double)sz[4]);
        }
    }
    __shared__ int wsum[16];
    __shared__ int s_base;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < nk; r0 += 1024) {
        const int r = r0 + tid;
        bool ok = r < nk;
        size_t src = 0;
        float hb[4] = {0, 0, 0, 0};
        if (ok) {
            src = (size_t)img * m_cap + (size_t)keep[(size_t)img * m_cap + r];
#pragma unroll
            for (int j = 0; j < 4; j++) hb[j] = hbox[src * 4 + j];
            if (do_post) {
                hb[0] *= sx; hb[2] *= sx; hb[1] *= sy; hb[3] *= sy;          // Boxes.scale
                hb[0] = fminf(fmaxf(hb[0], 0.f), ow); hb[2] = fminf(fmaxf(hb[2], 0.f), ow);   // Boxes.clip
                hb[1] = fminf(fmaxf(hb[1], 0.f), oh); hb[3] = fminf(fmaxf(hb[3], 0.f), oh);
                ok = (hb[2] - hb[0]) > 0.f && (hb[3] - hb[1]) > 0.f;         // Boxes.nonempty
            }
        }
        // ordered compaction: wave ballot + per-wave offsets
        u64 b = __ballot(ok);
        const int lane = tid & 63, wv = tid >> 6;
        if (lane == 0) wsum[wv] = __popcll(b);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < 16; k++) {
            if (k < wv) woff += wsum[k];
            tot += wsum[k];
        }
        const int base = s_base;
        if (ok) {
            const int o = base + woff + __popcll(b & ((1ull << lane) - 1ull));
            if (o < k_cap) {
                float* d = out + ((size_t)img * k_cap + o) * DAFNE_DET_ROW;
#pragma unroll
                for (int j = 0; j < 8; j++) d[j] = corners[src * 8 + j] * ((j & 1) ? cys : cxs);
                d[8] = scores[src];
                d[9] = ctr[src];
                d[10] = (float)classes[src];
                d[11] = (float)levels[src];
                d[12] = hb[0]; d[13] = hb[1]; d[14] = hb[2]; d[15] = hb[3];
                d[16] = locs[src * 2 + 0] * cxs;
                d[17] = locs[src * 2 + 1] * cys;
            }
        }
        __syncthreads();
        if (tid == 0) s_base = base + tot;
        __syncthreads();
    }
    if (tid == 0) out_counts[img] = s_base;
}

int setup(DecodeDev& D, const dafne_decode_params* prm, const dafne_level_desc* levels, void* ws,
          size_t* need_out) {
    if (!prm || !levels) return dafne::fail(DAFNE_E_INVALID, "decode: null params");
    if (prm->n_levels < 1 || prm->n_levels > kMaxLevels) return dafne::fail(DAFNE_E_UNSUPPORTED, "decode: n_levels %d", prm->n_levels);
    if (prm->n_images < 1 || prm->n_classes < 1) return dafne::fail(DAFNE_E_INVALID, "decode: bad sizes");
    if (prm->pre_nms_topk < 1 || prm->pre_nms_topk > kMaxTopk) return dafne::fail(DAFNE_E_UNSUPPORTED, "decode: pre_nms_topk %d not in [1,%d]", prm->pre_nms_topk, kMaxTopk);
    if (prm->m_cap < prm->n_levels * prm->pre_nms_topk) return dafne::fail(DAFNE_E_INVALID, "decode: m_cap too small");
    D.n_images = prm->n_images; D.n_levels = prm->n_levels; D.C = prm->n_classes;
    D.topk = prm->pre_nms_topk; D.twc = prm->thresh_with_ctr; D.sortc = prm->sort_corners;
    D.m_cap = prm->m_cap; D.thresh = prm->pre_nms_thresh;
    // key = bits - key_lo, top bin index < 2048.  THRESH_WITH_CTR: the ranked score IS the thresholded one, so
    // scores lie in (max(thresh,0), 1] and the key range starts at the threshold.  Otherwise the candidate test is
    // cls > thresh while the ranked score is sqrt(cls * ctr), anywhere in (0, 1] (dafne_outputs.py:812-829): the
    // key range must start at 0, or every score below the threshold would collapse into one key and top-k would
    // pick among them by flat index instead of by score.
    union { float f; unsigned u; } t, one;
    t.f = (prm->thresh_with_ctr && prm->pre_nms_thresh > 0.f) ? prm->pre_nms_thresh : 0.f;
    one.f = 1.0f;
    D.key_lo = t.u < one.u ? t.u : 0u;
    unsigned range = one.u - D.key_lo;
    D.shift = 0;
    while ((range >> D.shift) >= (unsigned)kBins) D.shift++;
    D.kp = 2;
    while (D.kp < D.topk) D.kp <<= 1;
    int chunk = 0;
    size_t eoff = 0;
    for (int l = 0; l < D.n_levels; l++) {
        const dafne_level_desc& s = levels[l];
        LevelDev& L = D.lv[l];
        L.logits = s.d_logits; L.delta = s.d_delta; L.center = s.d_center; L.ctrness = s.d_ctrness;
        L.logits_ps = s.logits_ps; L.delta_ps = s.delta_ps; L.center_ps = s.center_ps; L.ctrness_ps = s.ctrness_ps;
        L.H = s.H; L.W = s.W; L.stride = s.stride; L.scale = s.scale;
        if (s.H < 1 || s.W < 1 || (long long)s.H * s.W * D.C > 0x7fffffffLL) return dafne::fail(DAFNE_E_INVALID, "decode: level %d size", l);
        L.n_elem = s.H * s.W * D.C;
        L.chunk0 = chunk;
        chunk += (L.n_elem + kChunk - 1) / kChunk;
        L.e_off = eoff;
        eoff += (size_t)L.n_elem;
    }
    D.n_chunks = chunk;
    D.e_per_image = eoff;
    dafne::WsCarver c(ws);
    const size_t NL = (size_t)D.n_images * D.n_levels;
    D.hist = c.take<unsigned>(NL * kBins);
    D.s_cnt = c.take<unsigned>(NL);
    D.e_cnt = c.take<unsigned>(NL);      // hist, s_cnt, e_cnt are zeroed per call (contiguous)
    D.meta = c.take<int>(NL * 4);
    D.S = c.take<u64>(NL * D.kp);
    D.E = c.take<u64>((size_t)D.n_images * D.e_per_image);
    *need_out = dafne::align_up(c.off, 256);
    return DAFNE_OK;
}

}  // namespace

extern "C" {

size_t dafne_decode_workspace_bytes(const dafne_decode_params* prm, const dafne_level_desc* levels) {
    DecodeDev D;
    size_t need = 0;
    if (setup(D, prm, levels, nullptr, &need)) return 0;
    return need;
}

int dafne_decode_levels_hip(const dafne_decode_params* prm, const dafne_level_desc* levels,
                            float* d_corners, float* d_scores, float* d_ctr, int32_t* d_classes,
                            float* d_locs, int32_t* d_levels, float* d_hbox, int32_t* d_counts,
                            void* d_ws, size_t ws_bytes, void* stream) {
    DecodeDev D;
    size_t need = 0;
    int rc = setup(D, prm, levels, d_ws, &need);
    if (rc) return rc;
    if (!d_ws || ws_bytes < need) return dafne::fail(DAFNE_E_WORKSPACE, "decode: workspace %zu < %zu", ws_bytes, need);
    if (!d_corners || !d_scores || !d_ctr || !d_classes || !d_locs || !d_levels || !d_hbox || !d_counts)
        return dafne::fail(DAFNE_E_INVALID, "decode: null output");
    for (int l = 0; l < D.n_levels; l++)
        if (!D.lv[l].logits || !D.lv[l].delta || !D.lv[l].center || !D.lv[l].ctrness)
            return dafne::fail(DAFNE_E_INVALID, "decode: null input at level %d", l);
    hipStream_t st = (hipStream_t)stream;
    size_t zbytes = (size_t)((char*)(D.e_cnt + (size_t)D.n_images * D.n_levels) - (char*)D.hist);
    DAFNE_HIP_TRY(hipMemsetAsync(D.hist, 0, zbytes, st));
    hipLaunchKernelGGL(decode_hist_kernel, dim3(D.n_chunks, D.n_images), dim3(256), 0, st, D);
    if ((rc = dafne::check_launch("decode_hist"))) return rc;
    hipLaunchKernelGGL(decode_pick_kernel, dim3(D.n_levels, D.n_images), dim3(64), 0, st, D);
    if ((rc = dafne::check_launch("decode_pick"))) return rc;
    hipLaunchKernelGGL(decode_collect_kernel, dim3(D.n_chunks, D.n_images), dim3(256), 0, st, D);
    if ((rc = dafne::check_launch("decode_collect"))) return rc;
    hipLaunchKernelGGL(decode_finalize_kernel, dim3(D.n_levels, D.n_images), dim3(kFinThreads), 0, st, D,
                       d_corners, d_scores, d_ctr, d_classes, d_locs, d_levels, d_hbox, d_counts);
    return dafne::check_launch("decode_finalize");
}

int dafne_sort_quadrilateral_hip(const float* d_in, float* d_out, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!d_in || !d_out))) return dafne::fail(DAFNE_E_INVALID, "sort_quadrilateral: bad args");
    if (n == 0) return DAFNE_OK;
    hipLaunchKernelGGL(sort_quad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_in, d_out, (long long)n);
    return dafne::check_launch("sort_quadrilateral");
}

int dafne_gather_detections_hip(const float* d_corners, const float* d_scores, const float* d_ctr,
                                const int32_t* d_classes, const float* d_locs,
                                const int32_t* d_levels, const float* d_hbox,
                                const int64_t* d_keep, const int32_t* d_num_keep,
                                const float* d_sizes, int do_postprocess, int n_images, int m_cap,
                                int k_cap, float* d_out, int32_t* d_out_counts, void* stream) {
    if (n_images < 1 || m_cap < 0 || k_cap < 1 || !d_corners || !d_scores || !d_ctr || !d_classes || !d_locs ||
        !d_levels || !d_hbox || !d_keep || !d_num_keep || !d_out || !d_out_counts ||
        (do_postprocess && !d_sizes))
        return dafne::fail(DAFNE_E_INVALID, "gather: bad args");
    hipLaunchKernelGGL(gather_kernel, dim3(n_images), dim3(1024), 0, (hipStream_t)stream, d_corners, d_scores,
                       d_ctr, d_classes, d_locs, d_levels, d_hbox, reinterpret_cast<const long long*>(d_keep),
                       d_num_keep, d_sizes, do_postprocess, m_cap, k_cap, d_out, d_out_counts);
    return dafne::check_launch("gather");
}

}  // extern "C"
