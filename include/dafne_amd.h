/*
 * dafne_amd.h -- C ABI of libdafne_amd.so, the MI355X (gfx950) replacement for the
 * native pieces of braun-steven/DAFNe's inference path.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer named d_* is a DEVICE pointer owned by
 *     the caller; `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - the library never allocates device memory and never synchronises the
 *     device: work is enqueued on `stream`; scratch comes from the caller through
 *     (d_ws, ws_bytes), sized by the matching *_workspace_bytes() query
 *   - return value: 0 = OK, non-zero = error (DAFNE_E_*); the message is
 *     available from dafne_last_error() (thread-local); nothing throws across
 *     the boundary
 *   - re-entrant: no mutable global state; one call <-> one stream
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   poly_nms.poly_gpu_nms(dets[M,9] f32, thresh, device_id)   dafne/modeling/nms/nms.py:6,91
 *       (external CUDA ext, DOTA_devkit poly_nms_gpu)         -> dafne_poly_nms_hip / _batched
 *   batched_nms_poly's offset arithmetic                      dafne/modeling/nms/nms.py:74-90
 *                                                             -> dafne_select_over_all_levels_hip
 *   select_over_all_levels' kthvalue cap                      dafne/modeling/dafne/dafne_outputs.py:907-925
 *                                                             -> dafne_select_over_all_levels_hip
 *   polyiou.iou_poly (SWIG C++)                               tools/prepare_dota/polyiou.cpp:112
 *                                                             -> dafne_poly_iou_pairs_hip
 *   forward_for_single_feature_map (+ sort_quadrilateral)     dafne/modeling/dafne/dafne_outputs.py:792-905,
 *                                                             dafne/utils/sort_corners.py:26-92
 *                                                             -> dafne_decode_levels_hip
 *   torch conv2d / GroupNorm / max_pool2d / interpolate       dafne/modeling/dafne/dafne.py:209-229,318-344,
 *   (cuDNN through torch) in backbone and head                dafne/modeling/backbone/fpn.py:26-37
 *                                                             -> dafne_conv2d_nhwc_bf16_hip and friends
 */
#ifndef DAFNE_AMD_H
#define DAFNE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAFNE_OK 0
#define DAFNE_E_INVALID 1   /* bad argument (null pointer, negative size, ...) */
#define DAFNE_E_WORKSPACE 2 /* workspace too small */
#define DAFNE_E_HIP 3       /* a HIP runtime call / kernel launch failed */
#define DAFNE_E_UNSUPPORTED 4

/* Library / ABI version: major*10000 + minor*100 + patch. */
int dafne_abi_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* dafne_last_error(void);

/* ---------------------------------------------------------------- polygon IoU */
/*
 * fp64 IoU of n quadrilateral pairs, bit-identical to the reference's
 * polyiou.cpp arithmetic.  d_p, d_q: [n,8] float64 (x0,y0,..,x3,y3); d_out: [n].
 */
int dafne_poly_iou_pairs_hip(const double* d_p, const double* d_q, int64_t n,
                             double* d_out, void* stream);

/* --------------------------------------------------------------- rotated NMS */
/*
 * Tile ResultMerge NMS (dafne/utils/ResultMerge_multi_process.py:61-121 py_cpu_nms_poly_fast, called from
 * nmsbynamedict :160-176 for every original image of a Task1_<class>.txt file; tools/prepare_dota has the
 * same file).  Rows are FLOAT64 [x1..y4, score] exactly as mergesingle builds them ((tile poly + offset) /
 * rate, :217-228); order = argsort(score, stable)[::-1]; row j is dropped iff a kept earlier row i has
 * iou_poly(i, j) > thresh (polyiou.cpp, fp64) AND, when strict_hbb != 0, their axis-aligned hulls pass the
 * reference's `hbb_ovr > 0` test (:80-98).  strict_hbb == 0 gives py_cpu_nms_poly (:24-58).
 * d_dets9 [n_images][m_cap][9] doubles, d_counts [n_images] or NULL (= m_cap rows each); d_keep
 * [n_images][m_cap] original row indices in descending-score order, d_num_keep [n_images].
 */
size_t dafne_poly_nms_f64_workspace_bytes(int n_images, int m_cap);
int dafne_poly_nms_f64_batched_hip(const double* d_dets9, const int32_t* d_counts, int n_images, int m_cap,
                                   double thresh, int strict_hbb, int64_t* d_keep, int32_t* d_num_keep,
                                   void* d_ws, size_t ws_bytes, int flags, void* stream);

/*
 * Greedy polygon NMS, the replacement for poly_nms.poly_gpu_nms (nms.py:91).
 *   d_dets9   [M,9] float32, C-contiguous: 8 corner coordinates + score
 *   thresh    a box is suppressed iff IoU(kept, box) > thresh (fp64 compare)
 *   d_keep    [M] int64 out: indices into d_dets9 of the kept rows, in
 *             descending-score order (equal scores: larger index first, i.e.
 *             np.argsort(kind="stable")[::-1])
 *   d_num_keep  [1] int32 out
 * IoU: fp64 triangle-fan clip of polyiou.cpp on the float32 inputs.
 */
size_t dafne_poly_nms_workspace_bytes(int n_images, int m_cap);
/*
 * `flags` (every NMS entry point takes it, per call; the library holds no mutable global state, so concurrent callers
 * on other threads / streams are unaffected): 0 = default.  DAFNE_NMS_EXACT_ONLY is the parity switch.  The in-model
 * NMS decides `iou_poly > thresh` (polyiou.cpp:112-133 in fp64) and takes three analytic shortcuts on the way, each with
 * a proven margin: a guarded hull-separation pre-filter, an IoU upper bound for convex pairs, and
 * a one-lane geometric clip for pairs far from the threshold.  With DAFNE_NMS_EXACT_ONLY all three are off for THIS
 * call (every pair of every live tile runs the reference-order clip; same results, slower): what tests/test_gpu_nms.py
 * uses to show the shortcuts change nothing.  Unknown bits -> DAFNE_E_INVALID.
 *
 * dafne_poly_nms_stats_offset: byte offset inside the NMS workspace of uint32 stats[n_images][4], valid after a
 * call has completed: pairs decided by (0) the fast path as "suppress", (1) the fast path as "keep", (2) the
 * reference-order path, (3) the reference-order path of tiles that overflowed the pair lists.  f64_rows: 1 for the
 * dafne_poly_nms_f64_* workspace layout.
 */
#define DAFNE_NMS_EXACT_ONLY 1
size_t dafne_poly_nms_stats_offset(int n_images, int m_cap, int f64_rows);
int dafne_poly_nms_hip(const float* d_dets9, int M, double thresh, int64_t* d_keep,
                       int32_t* d_num_keep, void* d_ws, size_t ws_bytes, int flags, void* stream);
/*
 * Batched form: n_images independent problems in one set of launches.
 *   d_dets9   [n_images, m_cap, 9]; d_counts [n_images] int32 DEVICE (rows used
 *             per image, <= m_cap; read on the device, no host sync)
 *   post_topk > 0: after NMS keep only rows whose score >= the post_topk-th
 *             best kept score (dafne_outputs.py:916-923; ties may exceed it)
 *   d_keep    [n_images, m_cap] int64; d_num_keep [n_images] int32
 */
int dafne_poly_nms_batched_hip(const float* d_dets9, const int32_t* d_counts, int n_images,
                               int m_cap, double thresh, int post_topk, int64_t* d_keep,
                               int32_t* d_num_keep, void* d_ws, size_t ws_bytes, int flags, void* stream);
/*
 * ml_nms + the post-NMS cap for a batch (nms.py:10-92, dafne_outputs.py:907-925):
 * class 5 -> 4, offset = float(class) * (max(boxes) - min(boxes) + 1) in fp32 per
 * image, NMS at nms_thresh, then the post_topk cap.
 *   d_boxes8 [n_images, m_cap, 8] f32 (16-byte aligned), d_scores [n_images, m_cap] f32,
 *   d_classes [n_images, m_cap] int32, d_counts [n_images] int32 (device).
 */
int dafne_select_over_all_levels_hip(const float* d_boxes8, const float* d_scores,
                                     const int32_t* d_classes, const int32_t* d_counts,
                                     int n_images, int m_cap, double nms_thresh, int post_topk,
                                     int64_t* d_keep, int32_t* d_num_keep, void* d_ws,
                                     size_t ws_bytes, int flags, void* stream);

/* ------------------------------------------------------ decode / top-k / sort */
typedef struct dafne_level_desc {
    const float* d_logits;  /* [N, H, W, C]  (NHWC == the reference's permuted view) */
    const float* d_delta;   /* [N, H, W, 8]  corners_pred output                      */
    const float* d_center;  /* [N, H, W, 2]  center_pred output                       */
    const float* d_ctrness; /* [N, H, W]     ctrness logits                           */
    /* pixel strides in floats (>= C, 8, 2, 1): lets several predictions share one
     * NHWC buffer, e.g. corners_pred+ctrness written by one fused conv          */
    int32_t logits_ps, delta_ps, center_ps, ctrness_ps;
    int32_t H, W, stride;   /* feature size and FPN stride                            */
    float scale;            /* the level's learnable Scale (dafne.py:405-411)         */
} dafne_level_desc;

typedef struct dafne_decode_params {
    int32_t n_images, n_levels, n_classes;
    int32_t pre_nms_topk;     /* PRE_NMS_TOPK_TEST per level            */
    float pre_nms_thresh;     /* INFERENCE_TH_TEST, strict >            */
    int32_t thresh_with_ctr;  /* THRESH_WITH_CTR                        */
    int32_t sort_corners;     /* SORT_CORNERS                           */
    int32_t m_cap;            /* rows per image in the outputs (>= n_levels*pre_nms_topk) */
} dafne_decode_params;

/*
 * forward_for_single_feature_map for every level and image
 * (dafne_outputs.py:771-772,792-905): sigmoid, sqrt(cls*ctr), threshold, per
 * level top-k, corner decode ((center.repeat+delta)*scale*stride + location),
 * optional canonical corner order, hull box.  Outputs are per image, levels
 * concatenated in order, candidates within a level in (location, class) order:
 *   d_corners [N, m_cap, 8], d_scores [N, m_cap], d_ctr [N, m_cap],
 *   d_classes [N, m_cap] int32, d_locs [N, m_cap, 2], d_levels [N, m_cap] int32,
 *   d_hbox [N, m_cap, 4], d_counts [N] int32.
 */
size_t dafne_decode_workspace_bytes(const dafne_decode_params* prm, const dafne_level_desc* levels);
int dafne_decode_levels_hip(const dafne_decode_params* prm, const dafne_level_desc* levels,
                            float* d_corners, float* d_scores, float* d_ctr, int32_t* d_classes,
                            float* d_locs, int32_t* d_levels, float* d_hbox, int32_t* d_counts,
                            void* d_ws, size_t ws_bytes, void* stream);
/* sort_quadrilateral on [n,8] float32 (sort_corners.py:26-92). */
int dafne_sort_quadrilateral_hip(const float* d_in, float* d_out, int64_t n, void* stream);
/*
 * Gather the kept rows and apply detector_postprocess + OneStageDetector._postprocess
 * (one_stage_detector.py:79-98): hull boxes scaled by out/net-input size, clipped,
 * empty ones dropped; corners and locations scaled by out/orig.  d_sizes:
 * [N,6] float32 = (net_h, net_w, out_h, out_w, orig_h, orig_w) per image.
 * do_postprocess: 0 = plain gather; 1 = d2 detector_postprocess only (hull boxes;
 * what forward(do_postprocess=False) still does); 2 = also rescale corners/locations.
 * Output rows [N, k_cap, DAFNE_DET_ROW] float32 = corners8, score, centerness,
 * class, level, hbox4, loc2 (class/level stored as float); d_out_counts [N].
 * Rows beyond k_cap are dropped and still counted (caller checks count <= k_cap).
 */
#define DAFNE_DET_ROW 18
int dafne_gather_detections_hip(const float* d_corners, const float* d_scores, const float* d_ctr,
                                const int32_t* d_classes, const float* d_locs,
                                const int32_t* d_levels, const float* d_hbox,
                                const int64_t* d_keep, const int32_t* d_num_keep,
                                const float* d_sizes, int do_postprocess, int n_images, int m_cap,
                                int k_cap, float* d_out, int32_t* d_out_counts, void* stream);

/* ------------------------------------------------------------- dense engine */
/*
 * Activations: NHWC bf16 with a zero halo of 1 pixel: [N, H+2, W+2, C]; producers
 * write the interior only, so the halo stays zero and a 3x3 tap never needs a
 * bounds check.  Weights: bf16 [Cout_pad][Cin/64][KH][KW][64] (k = slab, kh, kw, cin%64), rows
 * beyond Cout zero; Cout_pad = dafne_conv2d_cout_pad(Cout).  Bias: fp32 [Cout_pad]
 * (FrozenBN folded into weight scale + bias).
 */
#define DAFNE_CONV_RELU 1u         /* ReLU in the epilogue                              */
#define DAFNE_CONV_RESIDUAL 2u     /* += d_res (same shape as the output) before ReLU   */
#define DAFNE_CONV_UPSAMPLE_ADD 4u /* += nearest-2x upsample of d_res [N,H/2+2,W/2+2,C] */
#define DAFNE_CONV_OUT_F32 8u      /* fp32 un-haloed NHWC output [N,Hout,Wout,Cout]     */
#define DAFNE_CONV_GN_STATS 16u    /* emit per-(M tile, group of 8 ch) sum and sum-sq   */
#define DAFNE_CONV_GN_INPUT 32u    /* GroupNorm + ReLU of the INPUT applied on load (3x3 patch / slab kernels) */
#define DAFNE_CONV_GN_FINALIZE 64u /* with GN_STATS, 3x3 patch-kernel layers with Cout == 256 (kernel id 6): the last tile
                                    * of every image finalises mean / rstd into d_gn_stats_out -- same values, bit for bit, as
                                    * dafne_groupnorm_finalize_hip on d_gn_partial, without the extra launch */
#define DAFNE_CONV_EXCLUSIVE 128u  /* scheduling hint, results unchanged: nothing else runs on the GPU next to this launch (one
                                    * stream, whole batch).  Small launches (<= one 128 x 128 tile per CU) then take the whole
                                    * LDS of their CU for a 4-stage operand ring (K loop without a drain per step); launches of
                                    * plans that share the GPU with other streams keep the small footprint */

#define DAFNE_CONV_FRAG16 256u     /* dafne_conv3x3_c256_hip / _pair_hip only: d_wfrag is in the 16x16x32 fragment order (below) and the
                                    * launch runs on v_mfma_f32_16x16x32_bf16 (cheaper per flop on this package); fp32 sums in another
                                    * order than the 32x32x16 form: <= 1 bf16 ulp apart on an output, run-to-run identical */

typedef struct dafne_conv_seg {
    const void* d_in;   /* bf16 [N, Hin+2, Win+2, Cin]; stem: [N, Hin, Win, 4] pre-padded */
    void* d_out;        /* bf16 [N, Hout+2, Wout+2, Cout] or fp32 [N,Hout,Wout,Cout]      */
    const void* d_res;  /* residual / coarser map for UPSAMPLE_ADD, or NULL               */
    int32_t Hin, Win, Hout, Wout;
} dafne_conv_seg;

typedef struct dafne_conv_params {
    int32_t n_images, n_segs;   /* segments share weights (the FPN levels of the head) */
    int32_t Cin, Cout, KH, KW, stride, pad;
    uint32_t flags;
    const void* d_weight;
    const float* d_bias;        /* or NULL */
    float* d_gn_partial;        /* GN_STATS: [num_tiles][Cout/8][2] fp32 */
    /* GN_INPUT: the input maps are the RAW output of the previous tower convolution; GroupNorm(Cin/8
     * groups) + ReLU is applied on load from these statistics (dafne_groupnorm_finalize_hip) */
    const float* d_in_gn_stats; /* [n_segs][n_images][Cin/8][2] mean, rstd */
    const float* d_in_gn_gamma; /* [Cin] */
    const float* d_in_gn_beta;  /* [Cin] */
    /* GN_FINALIZE: */
    float* d_gn_stats_out;      /* [n_segs][n_images][Cout/8][2] mean, rstd of the OUTPUT maps */
    int32_t* d_gn_counters;     /* [n_segs][n_images] int32, zero before the first launch (the kernel resets them) */
    float gn_eps;
} dafne_conv_params;

/*
 * 1x1 (pad 0) and 3x3 (pad 1) convolutions, stride 1 or 2, Cin % 64 == 0; plus the
 * ResNet stem as (Cin=4, 7x7, stride 2) on the image layout that
 * dafne_preprocess_image_hip writes (K = 7 rows x 8 cols x 4 ch, zero weights in
 * the padding taps; weight rows are 256 bf16).
 */
int dafne_conv2d_nhwc_bf16_hip(const dafne_conv_params* prm, const dafne_conv_seg* segs, void* stream);
/*
 * 3x3 / stride 1 / pad 1 layers with 256 input channels (the DAFNe head towers, dafne.py:318-348, and the FPN output
 * convolutions [d2 FPN recalled]) on the resident-patch kernel (conv3x3_rp_kernel): 4 x 32 pixel tiles x 256 output
 * channels, the whole (4+2) x (32+2) x 256-channel input patch resident in LDS, weights streamed L2 -> registers.
 * Same parameter block, flags (RELU, GN_STATS, GN_INPUT, GN_FINALIZE), statistics layout and result definition as
 * dafne_conv2d_nhwc_bf16_hip (same K order, same epilogue expressions: outputs are bit-identical to that call; the
 * GroupNorm partial sums are grouped by THIS kernel's tiles, so d_gn_partial needs dafne_conv3x3_c256_num_tiles rows and
 * a finalised mean / rstd may differ from the other kernel's in the last bit).  prm->d_weight is ignored; d_wfrag: bf16
 * [Cout/256][8 waves][144 k16 steps][64 lanes][8] = rows nt*256 + wave*32 + (lane & 31), K columns 16*step +
 * 8*(lane >> 5) .. +8 of the packed weight [Cout][Cin/64][KH][KW][64]  (engine.pack_conv3x3_frag).
 * With DAFNE_CONV_FRAG16 in prm->flags (round 6): d_wfrag: bf16 [Cout/256][8 waves][144 fragments][64 lanes][8], fragment 2m + cb =
 * rows nt*256 + wave*32 + 16*cb + (lane & 15), K columns 32*m + 8*(lane >> 4) .. +8  (engine.pack_conv3x3_frag16); the launch runs the
 * 16x16x32 form of the kernel: the same tiles, statistics layout and definition, fp32 sums of an output in another order (<= 1 bf16
 * ulp from the 32x32x16 form; not bit-identical to dafne_conv2d_nhwc_bf16_hip).
 * The kernel is persistent (one workgroup per CU walks the tiles) and never predicates a store: rows of out-of-image
 * tile pixels go to d_scratch (>= dafne_conv3x3_c256_scratch_bytes(); holds nothing afterwards; may be shared by calls).
 * Shapes: Cin == 256, Cout % 256 == 0, Cout <= 1024, bias, bf16 output, no residual / top-down add; else
 * DAFNE_E_UNSUPPORTED (dafne_conv3x3_c256_ok: 1 / 0).
 */
int dafne_conv3x3_c256_ok(const dafne_conv_params* prm, const dafne_conv_seg* segs);
int dafne_conv3x3_c256_num_tiles(const dafne_conv_params* prm, const dafne_conv_seg* segs);
int dafne_conv3x3_c256_tiles_per_image(const dafne_conv_params* prm, const dafne_conv_seg* segs, int32_t* out);
/* Two layers of identical shape and flags (cls_tower.i and center_tower.i of the DAFNe head, dafne.py:318-348: independent
 * chains) in ONE launch of the persistent kernel: twice the tiles per launch (2784 instead of 1392 at batch 8: 10.9 rounds on
 * 256 CUs instead of 5.4 -- the last-round loss halves) and half the launch boundaries.  Each layer keeps its own tensors,
 * weights, bias and GroupNorm state; results are those of two dafne_conv3x3_c256_hip calls. */
int dafne_conv3x3_c256_pair_hip(const dafne_conv_params* prm_a, const dafne_conv_seg* segs_a, const void* d_wfrag_a,
                                const dafne_conv_params* prm_b, const dafne_conv_seg* segs_b, const void* d_wfrag_b,
                                void* d_scratch, size_t scratch_bytes, void* stream);
size_t dafne_conv3x3_c256_scratch_bytes(void);
int dafne_conv3x3_c256_hip(const dafne_conv_params* prm, const dafne_conv_seg* segs, const void* d_wfrag, void* d_scratch,
                           size_t scratch_bytes, void* stream);
/*
 * Small-M layers -- res5 (1x1 / 3x3 / stride-2 projections on 32 x 32 maps), the FPN laterals, the P4 / P5 outputs and
 * LastLevelP6P7's stride-2 convolutions (backbone/fpn.py:16-37,58-91; d2 ResNet / FPN [recalled]) -- on conv_wr_kernel
 * (conv_wr.hip): 128-pixel x 256-channel tiles over the flat (image, row, column) pixel order, weights streamed L2 ->
 * registers, pixel operand staged by DMA, and a DETERMINISTIC split-K: dafne_conv2d_wr_splits() workgroups share a tile,
 * each writes its fp32 partial (128 KB) to d_workspace, the last arriver sums them in slice order and runs the epilogue.
 * Same parameter block and result definition as dafne_conv2d_nhwc_bf16_hip for one segment with flags RELU / RESIDUAL /
 * UPSAMPLE_ADD (EXCLUSIVE is accepted and ignored: the slice count depends on the shape and the CU count only); 1x1 pad 0 or 3x3 pad 1, stride 1 / 2,
 * Cin % 64 == 0, Cout % 256 == 0, bias required; else DAFNE_E_UNSUPPORTED (dafne_conv2d_wr_ok: 1 / 0).  One slice: K order
 * and epilogue expressions are dafne_conv2d_nhwc_bf16_hip's -> bit-identical output.  Several slices: the fp32 sum is
 * grouped by slices (fp32 rounding apart from the unsplit sum; run-to-run identical).  prm->d_weight is ignored;
 * d_wfrag: bf16 [Cout/256][8 waves][K/16 steps][64 lanes][8] = rows nt*256 + wave*32 + (lane & 31), K columns 16*step +
 * 8*(lane >> 5) .. +8 of the packed weight [Cout][Cin/64][KH][KW][64]  (engine.pack_conv_frag).
 * d_workspace: >= dafne_conv2d_wr_workspace_bytes() (0 for one slice: may be NULL); its first 64 KB are arrival tickets
 * and MUST BE ZERO before the first call (the kernel leaves them zero); calls on ONE stream may share it.
 */
int dafne_conv2d_wr_ok(const dafne_conv_params* prm, const dafne_conv_seg* segs);
int dafne_conv2d_wr_splits(const dafne_conv_params* prm, const dafne_conv_seg* segs);
size_t dafne_conv2d_wr_workspace_bytes(const dafne_conv_params* prm, const dafne_conv_seg* segs);
int dafne_conv2d_wr_hip(const dafne_conv_params* prm, const dafne_conv_seg* segs, const void* d_wfrag, void* d_workspace,
                        size_t workspace_bytes, void* stream);
/*
 * fp8-weight twin (BASELINE config 5: "fp8 weights, CDNA4 fp8 MFMA conv path"; SURVEY 8(b) item 5
 * dafne_conv2d_nhwc_{bf16,fp8w}_hip).  The reference has no fp8 path: this entry DEFINES it.
 *   d_weight   OCP e4m3 bytes [Cout][Cin/64][KH][KW][64] (same K order as the bf16 layout, 1 byte per element)
 *   d_oscale   fp32 [Cout]: weight dequantisation scale of the output channel divided by in_qscale
 *   in_qscale  > 0: activations (bf16 in HBM, after the optional GN_INPUT GroupNorm + ReLU, in fp32) are multiplied
 *              by it, clamped to +-448 and rounded to e4m3 (round to nearest even) while they are loaded
 * out = epilogue(oscale[c] * sum_k e4m3(w)[c][k] * e4m3(x)[k] + bias[c]) with fp32 accumulation on
 * v_mfma_f32_32x32x64_f8f6f4; flags / GroupNorm statistics / output layout as dafne_conv2d_nhwc_bf16_hip.
 * Shapes: 3x3, stride 1, pad 1, Cin % 64 == 0, Cout % 256 == 0, bias, bf16 output, no residual / top-down add
 * (the head towers and FPN output convolutions: the layers the bf16 patch kernel takes); anything else returns
 * DAFNE_E_UNSUPPORTED -- the other layers of a config-5 model run dafne_conv2d_nhwc_bf16_hip on weights that the
 * host dequantised exactly (power-of-two scales).  Tile geometry (d_gn_partial rows) = the bf16 call's when that
 * call's kernel id is 6.
 */
int dafne_conv2d_nhwc_fp8w_hip(const dafne_conv_params* prm, const dafne_conv_seg* segs, const float* d_oscale,
                               float in_qscale, void* stream);
/* M tiles of the fp8w call (rows of d_gn_partial) and per image of every segment: the fp8 kernel always uses the 3x3 patch
 * kernel's 8 x 32 pixel tiles, whatever kernel the bf16 call of the same layer would pick; -1 / error code when the layer is
 * not one the fp8 kernel takes */
int dafne_conv2d_fp8w_num_tiles(const dafne_conv_params* prm, const dafne_conv_seg* segs);
int dafne_conv2d_fp8w_tiles_per_image(const dafne_conv_params* prm, const dafne_conv_seg* segs, int32_t* out);
int dafne_conv2d_cout_pad(int Cout);
/* output pixels per M tile the call would use (geometry of d_gn_partial rows), -1 on error */
int dafne_conv2d_tile_pixels(const dafne_conv_params* prm, const dafne_conv_seg* segs);
/* number of M tiles the call above launches (= rows of d_gn_partial), -1 on error */
int dafne_conv2d_num_tiles(const dafne_conv_params* prm, const dafne_conv_seg* segs);
/* which kernel the call dispatches to (profiling / bench attribution), -1 on error:
 * 0 conv_igemm_kernel<1,4,1,2> (32 cout x 256 px)   1 conv_igemm_kernel<1,4,2,2> (64 x 256)
 * 2 conv_igemm_kernel<2,2,2,2> (128 x 128)           3 conv_igemm_kernel<4,2,2,4> (256 x 256, 8 waves)
 * 4 conv_stream_kernel (persistent, 1x1, Cin 512)    5 conv_ws_kernel (persistent, weights in registers, 1x1, Cin <= 256)
 * 6 conv3x3_patch_kernel (3x3 s1, 256 cout x 8x32 px tiles, input patch staged once per 64-channel slab)
 * 7 conv3x3_slab_kernel (3x3 s1, Cout <= 32, fp32 output: whole 64-channel slabs of both operands in LDS)
 * 8 conv3x3_pred16_kernel (the same layers with Cin = 256 and Cout <= 16: persistent, all weights resident in LDS) */
int dafne_conv2d_kernel_id(const dafne_conv_params* prm, const dafne_conv_seg* segs);
/* M tiles per image of every segment (out[n_segs]): where a segment's rows sit in d_gn_partial */
int dafne_conv2d_tiles_per_image(const dafne_conv_params* prm, const dafne_conv_seg* segs, int32_t* out);

/*
 * OneStageDetector.preprocess_image (one_stage_detector.py:100-107) + ImageList
 * padding: (x - mean)/std on uint8 BGR images [N,3,H,W] (layout_hwc=0) or [N,H,W,3]
 * (=1), zero-padded to Hn x Wn (multiples of 32), written as bf16
 * [N, Hn+6, Wn+6, 4] (3-pixel zero border for the 7x7 stem, 4th channel 0).
 * d_valid_hw: optional int32 [N,2] true sizes (pixels beyond are padding).
 * mean3/std3 are HOST pointers (3 floats each).
 */
int dafne_preprocess_image_hip(const uint8_t* d_img, int layout_hwc, int n_images, int H, int W,
                               const int32_t* d_valid_hw, const float* mean3, const float* std3,
                               int Hn, int Wn, void* d_out, void* stream);
/*
 * detectron2 ResizeShortestEdge / ResizeTransform.apply_image on uint8 images = PIL.Image.resize(BILINEAR)
 * [recalled; dafne/modeling/tta.py:71-99 builds its views with it], bit-exact to Pillow's 8-bit resampler
 * (two separable passes, 22-bit fixed-point coefficients, uint8 intermediate), plus the horizontal / vertical
 * flip of a TTA view folded into the store.  d_in: [C,H,W] (layout_hwc = 0) or [H,W,C] (= 1) uint8;
 * d_out: [C,new_h,new_w] uint8; d_ws: dafne_resize_workspace_bytes(C, H, new_w) bytes.
 */
size_t dafne_resize_workspace_bytes(int C, int H, int new_w);
int dafne_resize_bilinear_u8_hip(const uint8_t* d_in, int layout_hwc, int C, int H, int W, int new_h, int new_w,
                                 int hflip, int vflip, uint8_t* d_out, void* d_ws, size_t ws_bytes, void* stream);
/*
 * conv2 of the res2 bottlenecks [detectron2 BottleneckBlock, recalled; the ResNet of backbone/fpn.py:58-91]: 3x3 / stride 1 /
 * pad 1, 64 -> 64 channels, + bias (FrozenBN folded), optional ReLU.  d_in / d_out: bf16 NHWC [N,H+2,W+2,64] with a zero
 * 1-pixel halo (interior of d_out written); d_weight: bf16 [64, 576], k = (kh, kw, channel) -- the layout
 * dafne_conv2d_nhwc_bf16_hip takes for this layer (engine.pack_conv).  Persistent workgroups with all weights in registers
 * and the input patch of an 8 x 32 output tile staged once in LDS.  Bit-identical to dafne_conv2d_nhwc_bf16_hip.
 */
int dafne_conv3x3_c64_hip(const void* d_in, const void* d_weight, const float* d_bias, int n_images, int H, int W, int relu,
                          void* d_out, void* stream);
/*
 * Tail of one res4 bottleneck + head of the next in one kernel [detectron2 BottleneckBlock, recalled; the ResNet of
 * backbone/fpn.py:58-91]:  d_out = relu(conv3(d_in) + bias3 + d_res)  (1x1, 256 -> 1024, identity shortcut) and
 * d_next = relu(conv1'(d_out) + bias1)  (1x1, 1024 -> 256, the NEXT block's first convolution, stride 1).
 * All tensors bf16 NHWC with a 1-pixel halo: d_in / d_next [N,H+2,W+2,256], d_res / d_out [N,H+2,W+2,1024] (interior
 * written).  d_wfrag: both weight matrices fragment-major, bf16 [8][8][16][64][8] = [phase][wave][k16 step]
 * [lane][8]: phase 2c = conv3 rows c*256 + wave*32 + (lane & 31), K columns 16*step + 8*(lane >> 5) .. +8;
 * phase 2c+1 = conv1' rows wave*32 + (lane & 31), K columns c*256 + 16*step + 8*(lane >> 5) .. +8  (engine.pack_b2b).
 * Bit-identical to dafne_conv2d_nhwc_bf16_hip(conv3, RELU|RESIDUAL) followed by (conv1', RELU); d_out is written
 * once and not read back.
 */
int dafne_bottleneck_tail_head_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias3,
                                   const float* d_bias1, int n_images, int H, int W, void* d_out, void* d_next,
                                   void* stream);
/*
 * The whole body of a res4 bottleneck + the head of the next in one kernel (same reference block; conv_bneck.hip):
 *   T = relu(conv2(d_in) + bias2)  (3x3, 256 -> 256, pad 1; d_in = the block's conv1 output),
 *   d_out = relu(conv3(T) + bias3 + d_res)  (1x1, 256 -> 1024),  d_next = relu(conv1'(d_out) + bias1)  (1x1, 1024 -> 256).
 * T never reaches HBM.  Tensors as for dafne_bottleneck_tail_head_hip (bf16 NHWC, 1-pixel halo, interior written); any
 * H, W (4 x 32 pixel tiles -- 2 x 32 where a launch would leave seven eighths of the CUs without a tile; stores of out-of-image
 * tile pixels are masked; d_scratch needs dafne_bottleneck_body_scratch_bytes() bytes and holds nothing afterwards).
 * d_wfrag (ABI 140): conv2 fragment-major, bf16 [8 waves][144 k16 steps][64 lanes][8] (row wave*32 + perm[lane & 31]; K columns
 * 16*step + 8*(lane >> 5) .. +8 of dafne_conv2d_nhwc_bf16_hip's packed weight: 64-channel slab, kh, kw, channel), followed by
 * [8 GEMMs][8 waves][16 k16 steps][64 lanes][8]: GEMM 2c = conv3 rows c*256 + wave*32 + perm[lane & 31] over K = 256, GEMM 2c+1 =
 * conv1' rows wave*32 + perm[lane & 31] over K-chunk c.  perm[8g + 4h + i] = 16 (g >> 1) + 8h + 4 (g & 1) + i  (g = 0..3, h = 0..1,
 * i = 0..3): the rows of every 32-block in the order that makes a lane's 16 accumulator registers two runs of 8 consecutive
 * channels (engine.pack_bneck; NOT dafne_bottleneck_tail_head_hip's layout any more).  Bit-identical to
 * dafne_conv2d_nhwc_bf16_hip(conv2, RELU) followed by dafne_bottleneck_tail_head_hip.  d_next == NULL (the stage's last block: no
 * next conv1): only d_out is produced (d_bias1 may be NULL; the conv1' section of d_wfrag is still read: pack zeros),
 * bit-identical to dafne_conv2d_nhwc_bf16_hip(conv2, RELU) followed by (conv3, RELU|RESIDUAL).
 */
size_t dafne_bottleneck_body_scratch_bytes(void);
int dafne_bottleneck_body_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias2,
                              const float* d_bias3, const float* d_bias1, int n_images, int H, int W, void* d_out,
                              void* d_next, void* d_scratch, size_t scratch_bytes, void* stream);
/*
 * The same pair for the narrow stage res2 (same reference block):  d_out = relu(conv3(d_in) + bias3 + d_res)  (1x1,
 * 64 -> 256; d_res = the block's shortcut: the identity input or the projection's output) and
 * d_next = relu(conv1'(d_out) + bias1)  (1x1, 256 -> 64).  d_in / d_next [N,H+2,W+2,64], d_res / d_out [N,H+2,W+2,256],
 * bf16 NHWC with a 1-pixel halo (interior written).  d_wfrag: bf16, conv3 fragment-major [8 waves][4 k16 steps][64 lanes][8]
 * (rows wave*32 + (lane & 31), K columns 16*step + 8*(lane >> 5) .. +8) followed by conv1' [2 halves][16 steps][64 lanes][8]
 * (rows half*32 + (lane & 31), same K columns)  (engine.pack_b2b_narrow).  A streaming kernel: persistent workgroups,
 * both weight matrices in registers, 160 KB of HBM traffic per 128 pixels instead of 224 KB.  Bit-identical to
 * dafne_conv2d_nhwc_bf16_hip(conv3, RELU|RESIDUAL) followed by (conv1', RELU).
 */
int dafne_bottleneck_tail_head_narrow_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias3,
                                          const float* d_bias1, int n_images, int H, int W, void* d_out, void* d_next,
                                          void* stream);
/*
 * The tail + head pair for res3:  d_out = relu(conv3(d_in) + bias3 + d_res)  (1x1, 128 -> 512) and
 * d_next = relu(conv1'(d_out) + bias1)  (1x1, 512 -> 128).  d_in / d_next [N,H+2,W+2,128], d_res / d_out [N,H+2,W+2,512].
 * d_wfrag: bf16 [16 quarter blocks][4 channel quarters][4 k16 steps][64 lanes][8] in the order the kernel consumes them:
 * for each 256-channel chunk c of d_out: conv3 rows c*256 + rp*128 + quarter*32 + (lane & 31) over K columns
 * 16*(4*sh + step) + 8*(lane >> 5) .. +8 for (rp, sh) = (0,0) (0,1) (1,0) (1,1), then conv1' rows quarter*32 + (lane & 31)
 * over K columns c*256 + rp*128 + 16*(4*sh + step) + 8*(lane >> 5) .. +8 in the same (rp, sh) order  (engine.pack_b2b_mid).
 * Persistent workgroups; waves 0-3 issue every HBM access, waves 4-7 the weight stream (separate vmcnt queues).
 * Bit-identical to dafne_conv2d_nhwc_bf16_hip(conv3, RELU|RESIDUAL) followed by (conv1', RELU).
 */
int dafne_bottleneck_tail_head_mid_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias3,
                                       const float* d_bias1, int n_images, int H, int W, void* d_out, void* d_next,
                                       void* stream);
/*
 * Block 0 of res2, whose shortcut is a projection (1x1, 64 -> 256, no ReLU) of the block's input d_x0 [N,H+2,W+2,64]:
 *   shortcut = bf16(proj(d_x0) + bias_sc);  d_out = relu(conv3(d_in) + bias3 + shortcut);  d_next = relu(conv1'(d_out) + bias1).
 * The 256-channel shortcut map is never written or read (112 KB of HBM traffic per 128 pixels instead of 304 KB for the
 * three launches).  d_wfrag: the layout above followed by the projection's weights [8 waves][4 steps][64 lanes][8]
 * (engine.pack_b2b_narrow(w3, w1, wsc)).  Bit-identical to dafne_conv2d_nhwc_bf16_hip(proj), (conv3, RELU|RESIDUAL), (conv1', RELU).
 */
int dafne_bottleneck_proj_tail_head_narrow_hip(const void* d_in, const void* d_x0, const void* d_wfrag, const float* d_bias3,
                                               const float* d_bias_sc, const float* d_bias1, int n_images, int H, int W,
                                               void* d_out, void* d_next, void* stream);
/*
 * A WHOLE res2 bottleneck body, optionally with the head of the next block, in one kernel (same reference block;
 * conv_blk_narrow.hip):  T = relu(conv2(d_in) + bias2)  (3x3, 64 -> 64, pad 1; d_in = the block's conv1 output),
 * d_out = relu(conv3(T) + bias3 + X)  (1x1, 64 -> 256) and, when d_next is given, d_next = relu(conv1'(d_out) + bias1)
 * (1x1, 256 -> 64).  X = d_res [N,H+2,W+2,256] (identity shortcut) or, when d_bias_sc is given (block 0), the projection
 * shortcut conv_sc(d_res) + bias_sc of the block's 64-channel input d_res [N,H+2,W+2,64], rounded to bf16 as the separate
 * launch stores it.  T never reaches HBM (per block at batch 8: 67 MB written + read).  d_wfrag (engine.pack_blk_narrow):
 * conv2 fragment-major [2 channel halves][36 k16 steps][64 lanes][8] (rows half*32 + (lane & 31), K columns 16*step +
 * 8*(lane >> 5) .. +8 of dafne_conv2d_nhwc_bf16_hip's packed weight: kh, kw, channel), conv3 [2 halves of 128][4 quarters]
 * [4 steps][64][8], conv1' [2 halves][16 steps][64][8] (zeros without d_next), the projection in conv3's layout (zeros
 * without d_bias_sc).  4 x 32 pixel tiles, any H, W: rows of out-of-image tile pixels are written to d_scratch
 * (>= dafne_bottleneck_block_narrow_scratch_bytes(); holds nothing afterwards).  Bit-identical to
 * dafne_conv2d_nhwc_bf16_hip(conv2, RELU) followed by dafne_bottleneck_[proj_]tail_head_narrow_hip (or by
 * dafne_conv2d_nhwc_bf16_hip(conv3, RELU|RESIDUAL) without d_next).
 */
size_t dafne_bottleneck_block_narrow_scratch_bytes(void);
int dafne_bottleneck_block_narrow_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias2,
                                      const float* d_bias3, const float* d_bias_sc, const float* d_bias1, int n_images, int H, int W,
                                      void* d_out, void* d_next, void* d_scratch, size_t scratch_bytes, void* stream);
/*
 * A WHOLE res3 bottleneck body, optionally with the head of the next block, in one kernel (same reference block;
 * conv_blk_mid.hip):  T = relu(conv2(d_in) + bias2)  (3x3, 128 -> 128, pad 1; d_in = the block's conv1 output),
 * d_out = relu(conv3(T) + bias3 + d_res)  (1x1, 128 -> 512; d_res = the block's shortcut: identity input or projection
 * output) and, when d_next is given, d_next = relu(conv1'(d_out) + bias1)  (1x1, 512 -> 128).  d_in / d_next
 * [N,H+2,W+2,128], d_res / d_out [N,H+2,W+2,512].  T never reaches HBM and every weight byte is fetched once per 128 pixels
 * (dafne_bottleneck_tail_head_mid_hip: once per 64).  d_wfrag (engine.pack_blk_mid): conv2 fragment-major [4 channel
 * groups][72 k16 steps][64 lanes][8] (rows group*32 + (lane & 31), K columns 16*step + 8*(lane >> 5) .. +8 of
 * dafne_conv2d_nhwc_bf16_hip's packed weight: 64-channel slab, kh, kw, channel), conv3 [2 halves of 256][8 groups][8 steps]
 * [64][8], conv1' [4 groups][32 steps][64][8] (zeros without d_next: the section is still read).  4 x 32 pixel tiles, any
 * H, W: rows of out-of-image tile pixels are written to d_scratch (>= dafne_bottleneck_block_mid_scratch_bytes(); holds
 * nothing afterwards).  Bit-identical to dafne_conv2d_nhwc_bf16_hip(conv2, RELU) followed by
 * dafne_bottleneck_tail_head_mid_hip (or by dafne_conv2d_nhwc_bf16_hip(conv3, RELU|RESIDUAL) without d_next).
 */
size_t dafne_bottleneck_block_mid_scratch_bytes(void);
int dafne_bottleneck_block_mid_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias2,
                                   const float* d_bias3, const float* d_bias1, int n_images, int H, int W, void* d_out,
                                   void* d_next, void* d_scratch, size_t scratch_bytes, void* stream);
/*
 * detectron2 BasicStem in one kernel [recalled; the backbone of backbone/fpn.py:58-91]: conv 7x7 / s2 / p3
 * (FrozenBN folded into d_weight / d_bias) + ReLU + max-pool 3x3 / s2 / p1.  d_in: the layout
 * dafne_preprocess_image_hip writes, bf16 [N, H+6, W+6, 4]; d_weight: bf16 [64, 256] with k = (kh 0..7, kw 0..7,
 * c 0..3), zero where kh = 7, kw = 7 or c = 3 (the stem weight of dafne_conv2d_nhwc_bf16_hip); d_out: bf16
 * [N, H/4+2, W/4+2, 64] (interior written, halo untouched).  H, W multiples of 4.  Bit-identical to the stem
 * through dafne_conv2d_nhwc_bf16_hip followed by dafne_maxpool3x3s2_nhwc_bf16_hip; the half-resolution map
 * never reaches HBM.
 */
int dafne_stem_pool_hip(const void* d_in, const void* d_weight, const float* d_bias, int n_images, int H, int W,
                        void* d_out, void* stream);
/*
 * The same, plus the first convolution of res2.0 (detectron2 BottleneckBlock.conv1 [recalled]: 1x1, 64 -> 64, FrozenBN folded,
 * + ReLU) computed on the pooled tile while it is in LDS: d_w1 bf16 [64, 64] (cout, cin: the weight of
 * dafne_conv2d_nhwc_bf16_hip), d_b1 fp32 [64], d_out1 bf16 [N, H/4+2, W/4+2, 64] (interior written).  d_out as above.
 * Bit-identical to dafne_stem_pool_hip followed by dafne_conv2d_nhwc_bf16_hip(1x1, RELU) on d_out.
 */
int dafne_stem_pool_conv1_hip(const void* d_in, const void* d_weight, const float* d_bias, const void* d_w1, const float* d_b1,
                              int n_images, int H, int W, void* d_out, void* d_out1, void* stream);
/* 3x3 stride-2 pad-1 max pool of a post-ReLU map: [N,Hin+2,Win+2,C] -> [N,Hin/2+2,Win/2+2,C] */
int dafne_maxpool3x3s2_nhwc_bf16_hip(const void* d_in, void* d_out, int n_images, int Hin, int Win,
                                     int C, void* stream);

typedef struct dafne_gn_seg {
    void* d_x;                      /* bf16 [N, H+2, W+2, C], normalised in place */
    int32_t H, W;
    int32_t tile0, tiles_per_img;   /* where this segment's tiles sit in d_partial */
} dafne_gn_seg;
/*
 * Statistics half of the call below: tile partials -> mean / rstd in d_stats, no normalisation pass.
 * For layers whose consumer applies GroupNorm + ReLU on load (DAFNE_CONV_GN_INPUT).  d_x of the segments
 * is not touched.
 */
int dafne_groupnorm_finalize_hip(const dafne_gn_seg* segs, int n_segs, int n_images, int C,
                                 const float* d_partial, float* d_stats, float eps, void* stream);
/*
 * GroupNorm(C/8 groups, eps) + ReLU (dafne.py:330-344) from the conv's tile
 * partials: fixed-order reduction -> mean/rstd per (segment, image, group) in
 * d_stats [n_segs][N][C/8][2], then y = relu((x-mean)*rstd*gamma+beta) in place.
 */
int dafne_groupnorm_relu_nhwc_bf16_hip(const dafne_gn_seg* segs, int n_segs, int n_images, int C,
                                       const float* d_partial, float* d_stats, const float* d_gamma,
                                       const float* d_beta, float eps, void* stream);
/* out = relu(in) on n_elems bf16 values (n_elems % 8 == 0); halo zeros stay zero. */
int dafne_relu_copy_bf16_hip(const void* d_in, void* d_out, int64_t n_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DAFNE_AMD_H */
