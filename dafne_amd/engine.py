"""Dense engine: weight packing, activation buffers and the launch plan that
drives libdafne_amd.so's conv / pool / GroupNorm kernels.

PyTorch is only the allocator and stream owner here: every arithmetic step is a
HIP kernel behind include/dafne_amd.h.  Activations are NHWC bf16 with a 1-pixel
zero halo ([N, H+2, W+2, C]); a plan (list of pre-built ctypes calls) is created
once per input shape and replayed, optionally from a HIP graph.

Graph semantics follow the reference: detectron2 ResNet/FPN,
dafne/modeling/backbone/fpn.py:16-37,58-91 and dafne/modeling/dafne/dafne.py:
350-494 (center-to-corner branch).
"""
import ctypes
import os

import torch

from . import _lib

BF16 = torch.bfloat16
STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

F_RELU, F_RES, F_UP, F_F32, F_GN, F_GNIN, F_GNFIN, F_EXCL = 1, 2, 4, 8, 16, 32, 64, 128
F_FRAG16 = 256      # dafne_conv3x3_c256_hip only: the weights are pack_conv3x3_frag16's, the launch runs the 16x16x32 form


# ------------------------------------------------------------------ activations
class Act:
    """Haloed NHWC bf16 activation [N, H+2, W+2, C]; the halo is zero forever
    because kernels only ever write the interior."""

    __slots__ = ("t", "n", "h", "w", "c")

    def __init__(self, n, h, w, c, device):
        self.t = torch.zeros(n, h + 2, w + 2, c, dtype=BF16, device=device)
        self.n, self.h, self.w, self.c = n, h, w, c

    def nchw_float(self):
        return self.t[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).float().contiguous()

    @staticmethod
    def from_nchw(x):
        n, c, h, w = x.shape
        a = Act(n, h, w, c, x.device)
        a.t[:, 1:-1, 1:-1, :] = x.permute(0, 2, 3, 1).to(BF16)
        return a


class Pool:
    """Shape-keyed buffer pool: same shape => same halo positions => reusable."""

    def __init__(self, device):
        self.device = device
        self.free = {}
        self.bytes = 0

    def get(self, n, h, w, c):
        lst = self.free.setdefault((n, h, w, c), [])
        if lst:
            return lst.pop()
        self.bytes += n * (h + 2) * (w + 2) * c * 2
        return Act(n, h, w, c, self.device)

    def put(self, a):
        self.free.setdefault((a.n, a.h, a.w, a.c), []).append(a)


# --------------------------------------------------------------- weight packing
def pack_conv(weight, bias, device):
    """[Cout,Cin,KH,KW] fp32 (+bias) -> bf16 [Cout_pad, Cin/64, KH, KW, 64] flattened
    (k = 64-channel slab, kh, kw, channel in slab: the kernel walks the taps of one
    slab back to back for L2 locality), fp32 bias [Cout_pad]."""
    L = _lib.load()
    cout, cin, kh, kw = weight.shape
    cpad = L.dafne_conv2d_cout_pad(cout)
    w = torch.zeros(cpad, kh * kw * cin, dtype=torch.float32)
    w[:cout] = weight.detach().float().cpu().reshape(cout, cin // 64, 64, kh, kw).permute(0, 1, 3, 4, 2).reshape(cout, -1)
    b = torch.zeros(cpad, dtype=torch.float32)
    if bias is not None:
        b[:cout] = bias.detach().float().cpu()
    return w.to(BF16).to(device).contiguous(), b.to(device).contiguous()


E4M3_MAX = 448.0


def quantize_weight_e4m3(weight):
    """[Cout, ...] fp32 -> (q, scale, deq): OCP e4m3 values (torch.float8_e4m3fn, round to nearest even) of
    weight / scale with ONE POWER-OF-TWO scale per output channel (the smallest 2^k with amax / 2^k <= 448), and the
    dequantised weight q * scale.  Power-of-two scales make q * scale exact in fp32 AND in bf16 (an e4m3 value has
    <= 4 significant bits), so the layers that stay on the bf16 kernels compute with exactly the fp8 model's weights.
    This is the definition of BASELINE config 5's "fp8 weights" (the reference has no fp8 path)."""
    w = weight.detach().float().cpu()
    amax = w.reshape(w.shape[0], -1).abs().amax(dim=1)
    k = torch.ceil(torch.log2(torch.clamp(amax, min=1e-30) / E4M3_MAX))
    scale = torch.where(amax > 0, torch.exp2(k), torch.ones_like(amax))
    scale = torch.where(amax / scale > E4M3_MAX, scale * 2, scale)          # log2 rounding guard
    q = (w / scale.reshape(-1, *([1] * (w.dim() - 1)))).to(torch.float8_e4m3fn)
    deq = q.float() * scale.reshape(-1, *([1] * (w.dim() - 1)))
    return q, scale, deq


def pack_conv_fp8(weight, device):
    """[Cout,Cin,3,3] fp32 -> (e4m3 bytes [Cout, (Cin/64)*9*64] in the kernel's K order (slab, kh, kw, channel),
    fp32 scale [Cout]) for dafne_conv2d_nhwc_fp8w_hip."""
    cout, cin, kh, kw = weight.shape
    q, scale, _ = quantize_weight_e4m3(weight)
    qb = q.view(torch.uint8).reshape(cout, cin // 64, 64, kh, kw).permute(0, 1, 3, 4, 2).reshape(cout, -1)
    return qb.contiguous().to(device), scale.to(device).contiguous()


def act_qscale_from_amax(amax, margin=2.0):
    """fp8 (config 5) activation scale of a layer whose input is NOT a GroupNorm output: the largest power of two q with
    margin * amax * q <= 448 (e4m3's largest finite value; margin 2: inputs up to twice the calibration batch's amax
    still do not clip).  Powers of two keep x * q exact, so the only rounding is the e4m3 one.  amax 0 -> 1."""
    import math
    if not (amax > 0.0) or not math.isfinite(amax):
        return 1.0
    return float(2.0 ** math.floor(math.log2(E4M3_MAX / (margin * amax))))


def check_act_qscales(scales):
    """An activation scale is a positive finite power of two (act_qscale_from_amax); raises ValueError otherwise.  Pure
    host check: checkpoint.load_weights runs it BEFORE it touches the model."""
    import math
    for k, v in dict(scales).items():
        v = float(v)
        if not (math.isfinite(v) and v > 0 and math.log2(v) == round(math.log2(v))):
            raise ValueError("fp8 activation scale %r of %s is not a positive power of two" % (v, k))


def reduce_amax_over_ranks(calib, group=None, device=None):
    """fp8 calibration in a multi-process job: every rank must end up with the SAME activation scales, whatever images its
    shard holds -- element-wise MAX of the per-layer amax values over the ranks (one all-reduce of a small float64 vector
    in sorted-key order; RCCL on the GPU, gloo in the CPU tests).  Returns the reduced dict; a no-op without a process
    group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return dict(calib)
    keys = sorted(calib)
    # every rank calibrates the same architecture: same keys.  A mismatch would silently pair different layers.
    sig = [len(keys), sum(hash_str(k) for k in keys) % (1 << 52)]
    dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if dist.get_backend(group) == "nccl" \
        else torch.device("cpu")
    t = torch.tensor([float(calib[k]) for k in keys] + [float(s) for s in sig] + [-float(s) for s in sig],
                     dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    v = t.cpu().tolist()
    n = len(keys)
    if v[n:n + 2] != [float(s) for s in sig] or v[n + 2:n + 4] != [-float(s) for s in sig]:
        raise RuntimeError("fp8 calibration: the ranks calibrated different layer sets")
    return {k: v[i] for i, k in enumerate(keys)}


def hash_str(s):
    """Process-independent string hash (python's hash() is salted per process)."""
    h = 1469598103934665603
    for ch in s.encode():
        h = ((h ^ ch) * 1099511628211) & ((1 << 64) - 1)
    return h % (1 << 40)


class AmaxProbe:
    """Calibration-plan entry: max |x| of the listed activations (interior only; the halo is zero) into calib[key]."""

    flops = 0
    bytes = 0

    def __init__(self, calib, key, acts):
        self.calib, self.key, self.acts = calib, key, list(acts)

    def kernel_name(self):
        return "amax_probe"

    def __call__(self, stream):
        m = max(float(a.t.float().abs().max().item()) for a in self.acts)
        self.calib[self.key] = max(self.calib.get(self.key, 0.0), m)


def pack_stem(weight, bias, device):
    """[64,3,7,7] -> bf16 [64, 256]: k = (kh 0..7, kw 0..7, c 0..3), zero where kh=7, kw=7 or c=3."""
    cout = weight.shape[0]
    w = torch.zeros(cout, 8, 8, 4, dtype=torch.float32)
    w[:, :7, :7, :3] = weight.detach().float().cpu().permute(0, 2, 3, 1)
    b = torch.zeros(cout, dtype=torch.float32)
    if bias is not None:
        b[:] = bias.detach().float().cpu()
    return w.reshape(cout, 256).to(BF16).to(device).contiguous(), b.to(device).contiguous()


def pack_b2b(w3, w1):
    """Fragment-major weights of dafne_bottleneck_tail_head_hip from the packed 1x1 weights of conv3 ([1024, 256] bf16)
    and the next block's conv1 ([256, 1024] bf16): bf16 [8 phases][8 waves][16 k16-steps][64 lanes][8];
    phase 2c = conv3 rows c*256 + wave*32 + (lane & 31) over K = 256, phase 2c+1 = conv1 rows wave*32 + (lane & 31)
    over K-chunk c; K columns 16*step + 8*(lane >> 5) .. +8."""
    assert tuple(w3.shape) == (1024, 256) and tuple(w1.shape) == (256, 1024) and w3.dtype == BF16 and w1.dtype == BF16
    a1 = w3.reshape(4, 8, 32, 16, 2, 8).permute(0, 1, 3, 4, 2, 5)          # c, w, t, h, r, e
    a2 = w1.reshape(8, 32, 4, 16, 2, 8).permute(2, 0, 3, 4, 1, 5)          # c, w, t, h, r, e
    return torch.stack([a1, a2], dim=1).contiguous().reshape(8, 8, 16, 64, 8)


def _bneck_row_perm(device):
    """conv_bneck.hip's row order inside a wave's 32 output channels: MFMA row 8g + 4h + i (g = 0..3, h = 0..1, i = 0..3) holds
    channel 16 (g >> 1) + 8h + 4 (g & 1) + i, so that the 16 accumulator registers of a lane (fixed h) are two runs of 8
    consecutive channels -- two 16-byte pieces of a pixel's row, loaded / stored straight from registers."""
    r = torch.arange(32, device=device)
    g, h, i = r >> 3, (r >> 2) & 1, r & 3
    return 16 * (g >> 1) + 8 * h + 4 * (g & 1) + i


def pack_bneck(w2, w3, w1):
    """Fragment-major weights of dafne_bottleneck_body_hip: the 3x3 conv2 ([256, 2304] bf16 in pack_conv's K order = 64-channel
    slab, kh, kw, channel) as [8 waves][144 k16 steps][64 lanes][8] (rows wave*32 + perm[lane & 31], K columns 16*step +
    8*(lane >> 5) .. +8), followed by conv3 ([1024, 256]) and the next block's conv1 ([256, 1024]) as [8 GEMMs][8 waves][16 k16
    steps][64 lanes][8]: GEMM 2c = conv3 rows c*256 + wave*32 + perm[lane & 31] over K = 256, GEMM 2c+1 = conv1 rows wave*32 +
    perm[lane & 31] over K-chunk c (pack_b2b's layout with the rows of every 32-block in _bneck_row_perm's order)."""
    assert tuple(w2.shape) == (256, 2304) and w2.dtype == BF16
    assert tuple(w3.shape) == (1024, 256) and tuple(w1.shape) == (256, 1024) and w3.dtype == BF16 and w1.dtype == BF16
    perm = _bneck_row_perm(w2.device)
    a0 = w2.reshape(8, 32, 144, 2, 8)[:, perm].permute(0, 2, 3, 1, 4)                       # w, j, h, r, e
    a1 = w3.reshape(4, 8, 32, 16, 2, 8)[:, :, perm].permute(0, 1, 3, 4, 2, 5)               # c, w, t, h, r, e
    a2 = w1.reshape(8, 32, 4, 16, 2, 8)[:, perm].permute(2, 0, 3, 4, 1, 5)                  # c, w, t, h, r, e
    pb = torch.stack([a1, a2], dim=1).contiguous().reshape(-1)
    return torch.cat([a0.contiguous().reshape(-1), pb]).contiguous()


def pack_conv3x3_frag(w):
    """Fragment-major weights of dafne_conv3x3_c256_hip from a packed 3x3 weight ([Cout, 2304] bf16, pack_conv's K order):
    bf16 [Cout/256][8 waves][144 k16 steps][64 lanes][8]: rows nt*256 + wave*32 + (lane & 31), K columns 16*step +
    8*(lane >> 5) .. +8."""
    cout = w.shape[0]
    assert w.dim() == 2 and w.shape[1] == 2304 and cout % 256 == 0 and w.dtype == BF16
    return w.reshape(cout // 256, 8, 32, 144, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)   # nt, w, j, h, r, e


def pack_conv3x3_frag16(w):
    """The same weights in the fragment order of the kernel's 16x16x32 form (flag F_FRAG16, include/dafne_amd.h): bf16
    [Cout/256][8 waves][144 fragments][64 lanes][8], fragment 2m + cb = rows nt*256 + wave*32 + 16*cb + (lane & 15), K columns
    32*m + 8*(lane >> 4) .. +8."""
    cout = w.shape[0]
    assert w.dim() == 2 and w.shape[1] == 2304 and cout % 256 == 0 and w.dtype == BF16
    return w.reshape(cout // 256, 8, 2, 16, 72, 4, 8).permute(0, 1, 4, 2, 5, 3, 6).contiguous().reshape(-1)   # nt, w, m, cb, q, r, e


def rp_frag16():
    """The resident-patch kernel's matrix instruction: v_mfma_f32_16x16x32_bf16 (default; 6-8 % fewer joules per flop on this package)
    or, DAFNE_RP_MFMA16=0, v_mfma_f32_32x32x16_bf16 (bit-identical to the generic kernels; A/B runs and the parity tests of that form)."""
    return os.environ.get("DAFNE_RP_MFMA16", "1") != "0"


def pack_rp(w):
    """(fragment-major weights, flag bits) of a resident-patch call in the form rp_frag16() selects."""
    return (pack_conv3x3_frag16(w), F_FRAG16) if rp_frag16() else (pack_conv3x3_frag(w), 0)


def pack_conv_frag(w):
    """Fragment-major weights of dafne_conv2d_wr_hip from a packed weight ([Cout, K] bf16 in pack_conv's K order, Cout % 256 ==
    0, K % 64 == 0): bf16 [Cout/256][8 waves][K/16 steps][64 lanes][8]: rows nt*256 + wave*32 + (lane & 31), K columns
    16*step + 8*(lane >> 5) .. +8  (pack_conv3x3_frag for any K)."""
    cout, k = w.shape
    assert w.dim() == 2 and cout % 256 == 0 and k % 64 == 0 and w.dtype == BF16
    return w.reshape(cout // 256, 8, 32, k // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)   # nt, w, j, h, r, e


FP8_KERNELS = ("patch",)


_RP_SCRATCH = {}


def rp_scratch(device):
    """Dump area of dafne_conv3x3_c256_hip (rows of out-of-image tile pixels; contents are never read): one per device."""
    key = str(device)
    if key not in _RP_SCRATCH:
        _RP_SCRATCH[key] = torch.empty(_lib.load().dafne_conv3x3_c256_scratch_bytes(), dtype=torch.uint8, device=device)
    return _RP_SCRATCH[key]


def use_rp_kernel():
    """conv3x3_rp_kernel (resident patch, weights streamed to registers) for the 256-channel 3x3 layers; DAFNE_CONV_RP=0
    keeps them on conv3x3_patch_kernel (A/B runs)."""
    return os.environ.get("DAFNE_CONV_RP", "1") != "0"


def pack_b2b_narrow(w3, w1, wsc=None):
    """Fragment-major weights of dafne_bottleneck_tail_head_narrow_hip from the packed 1x1 weights of conv3 ([256, 64] bf16)
    and the next block's conv1 ([64, 256] bf16): bf16 [8 waves][4 steps][64 lanes][8] then [2 halves][16 steps][64 lanes][8];
    row = wave (half) * 32 + (lane & 31), K columns 16*step + 8*(lane >> 5) .. +8.  wsc: the projection shortcut's
    weights ([256, 64], block 0), appended in conv3's layout for dafne_bottleneck_proj_tail_head_narrow_hip."""
    assert tuple(w3.shape) == (256, 64) and tuple(w1.shape) == (64, 256) and w3.dtype == BF16 and w1.dtype == BF16
    parts = [w3.reshape(8, 32, 4, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1),               # w, t, h, r, e
             w1.reshape(2, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1)]              # c, t, h, r, e
    if wsc is not None:
        assert tuple(wsc.shape) == (256, 64) and wsc.dtype == BF16
        parts.append(wsc.reshape(8, 32, 4, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1))
    return torch.cat(parts).contiguous()


def pack_blk_narrow(w2, w3, w1=None, wsc=None):
    """Weights of dafne_bottleneck_block_narrow_hip (a whole res2 block) from the packed weights of conv2 ([64, 576] bf16, K
    order kh, kw, channel), conv3 ([256, 64]), optionally the next block's conv1 ([64, 256]) and the projection shortcut
    ([256, 64]): conv2 [2 channel halves][36 k16 steps][64 lanes][8] | conv3 [2 halves of 128][4 quarters][4 steps][64][8] |
    conv1' [2 halves][16 steps][64][8] | projection in conv3's layout; absent matrices are zeros (fixed section offsets)."""
    assert tuple(w2.shape) == (64, 576) and tuple(w3.shape) == (256, 64) and w2.dtype == BF16 and w3.dtype == BF16
    a2 = w2.reshape(2, 32, 36, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1)                    # ct, j, hl, r, e

    def quarters(w):                                                                       # [256, 64] -> h, cq, s, hl, r, e
        return w.reshape(2, 4, 32, 4, 2, 8).permute(0, 1, 3, 4, 2, 5).reshape(-1)
    if w1 is not None:
        assert tuple(w1.shape) == (64, 256) and w1.dtype == BF16
        a1 = w1.reshape(2, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1)                # ct, step, hl, r, e
    else:
        a1 = torch.zeros(2 * 16 * 64 * 8, dtype=BF16, device=w2.device)
    if wsc is not None:
        assert tuple(wsc.shape) == (256, 64) and wsc.dtype == BF16
        asc = quarters(wsc)
    else:
        asc = torch.zeros(256 * 64, dtype=BF16, device=w2.device)
    return torch.cat([a2, quarters(w3), a1, asc]).contiguous()


def pack_blk_mid(w2, w3, w1=None):
    """Weights of dafne_bottleneck_block_mid_hip (a whole res3 block) from the packed weights of conv2 ([128, 1152] bf16, K
    order 64-channel slab, kh, kw, channel), conv3 ([512, 128]) and optionally the next block's conv1 ([128, 512]): conv2 [4
    channel groups][72 k16 steps][64 lanes][8] | conv3 [2 halves of 256][8 groups][8 steps][64][8] | conv1' [4 groups][32
    steps][64][8] (zeros when absent: the kernel still walks the section)."""
    assert tuple(w2.shape) == (128, 1152) and tuple(w3.shape) == (512, 128) and w2.dtype == BF16 and w3.dtype == BF16
    a2 = w2.reshape(4, 32, 72, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1)                    # cg, j, hl, r, e
    a3 = w3.reshape(2, 8, 32, 8, 2, 8).permute(0, 1, 3, 4, 2, 5).reshape(-1)               # h, w, s, hl, r, e
    if w1 is not None:
        assert tuple(w1.shape) == (128, 512) and w1.dtype == BF16
        a1 = w1.reshape(4, 32, 32, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1)                # cg, step, hl, r, e
    else:
        a1 = torch.zeros(128 * 512, dtype=BF16, device=w2.device)
    return torch.cat([a2, a3, a1]).contiguous()


def pack_b2b_mid(w3, w1):
    """Weights of dafne_bottleneck_tail_head_mid_hip (res3) from the packed 1x1 weights of conv3 ([512, 128] bf16) and the next
    block's conv1 ([128, 512] bf16): bf16 [16 quarter blocks][4 channel quarters][4 k16 steps][64 lanes][8], in the order the
    kernel consumes them (per 256-channel chunk c: four conv3 blocks (rp, sh), then four conv1 blocks (rp, sh))."""
    assert tuple(w3.shape) == (512, 128) and tuple(w1.shape) == (128, 512) and w3.dtype == BF16 and w1.dtype == BF16
    # conv3: rows (c, rp, ct, r), cols (sh, s, hl, e)  ->  (c, rp, sh, ct, s, hl, r, e)
    a1 = w3.reshape(2, 2, 4, 32, 2, 4, 2, 8).permute(0, 1, 4, 2, 5, 6, 3, 7).reshape(2, 4, -1)
    # conv1: rows (ct, r), cols (c, rp, sh, s, hl, e)  ->  (c, rp, sh, ct, s, hl, r, e)
    a2 = w1.reshape(4, 32, 2, 2, 2, 4, 2, 8).permute(2, 3, 4, 0, 5, 6, 1, 7).reshape(2, 4, -1)
    return torch.cat([a1, a2], dim=1).contiguous().reshape(-1)


def fold_frozen_bn(weight, bn_w, bn_b, bn_mean, bn_var, eps=1e-5):
    s = bn_w * torch.rsqrt(bn_var + eps)
    return weight * s[:, None, None, None], bn_b - bn_mean * s


# --------------------------------------------------------------------- launches
class ConvCall:
    """One dafne_conv2d_nhwc_bf16_hip launch with its argument structs kept alive."""

    def __init__(self, w, b, cin, cout, k, stride, pad, flags, segs, n_images, gn_partial=None, gn_in=None, fp8=None,
                 gn_fin=None, wfrag=None, shared_gpu=False, frag16=False):
        """gn_in: (stats [n_segs,N,Cin/8,2], gamma [Cin], beta [Cin]) of the INPUT maps when they hold the raw
        output of the previous tower convolution (flag F_GNIN: GroupNorm + ReLU applied on load).
        fp8: (oscale fp32 [Cout], in_qscale) -> `w` holds e4m3 bytes (pack_conv_fp8) and the call goes to
        dafne_conv2d_nhwc_fp8w_hip (fp8 MFMA, activations quantised on load).
        gn_fin: (stats out [n_segs,N,Cout/8,2] fp32, counters [n_segs,N] int32 zeros, eps) with flag F_GNFIN: the last
        tile of every image finalises the GroupNorm statistics of the OUTPUT (no dafne_groupnorm_finalize_hip launch).
        wfrag: fragment-major bf16 weights (pack_conv3x3_frag; frag16: pack_conv3x3_frag16, flag F_FRAG16) -> the call goes to dafne_conv3x3_c256_hip (resident-patch
        kernel: 3x3 s1 p1, Cin 256, Cout % 256 == 0; its own tile geometry).
        shared_gpu: the plan this call belongs to runs next to other plans on concurrent streams (no F_EXCL hint)."""
        L = _lib.load()
        self.fp8 = fp8
        self.wfrag = wfrag
        assert fp8 is None or wfrag is None
        assert not frag16 or wfrag is not None
        if frag16:
            flags |= F_FRAG16
        self.keep = (w, b, gn_partial, [s for s in segs], gn_in, fp8, gn_fin, wfrag)
        gi = [t.data_ptr() for t in gn_in] if gn_in is not None else [None, None, None]
        gf = (gn_fin[0].data_ptr(), gn_fin[1].data_ptr(), float(gn_fin[2])) if gn_fin is not None else (None, None, 0.0)
        if not shared_gpu or os.environ.get("DAFNE_SHARED_EXCL", "0") == "1":
            flags |= F_EXCL             # hint (results unchanged): the plan this call belongs to has the GPU to itself
        self.prm = _lib.ConvParams(n_images, len(segs), cin, cout, k, k, stride, pad, flags,
                                   w.data_ptr(), b.data_ptr() if b is not None else None,
                                   gn_partial.data_ptr() if gn_partial is not None else None, gi[0], gi[1], gi[2],
                                   gf[0], gf[1], gf[2])
        arr = (_lib.ConvSeg * len(segs))()
        for i, (tin, tout, tres, hin, win, hout, wout) in enumerate(segs):
            arr[i] = _lib.ConvSeg(tin.data_ptr(), tout.data_ptr(), tres.data_ptr() if tres is not None else None,
                                  hin, win, hout, wout)
        self.segs = arr
        self.fn = L.dafne_conv2d_nhwc_bf16_hip
        self.flops = 0
        self.bytes = w.numel() * w.element_size()       # algorithmic HBM bytes: every operand once
        for (_, tout, tres, hin, win, hout, wout) in segs:
            kk = 49 * 3 if (cin == 4 and k == 7) else k * k * cin
            self.flops += 2 * n_images * hout * wout * cout * kk
            self.bytes += n_images * (hin * win * cin * 2 + hout * wout * cout * (4 if flags & F_F32 else 2)
                                      + (hout * wout * cout * 2 if (flags & F_RES) else 0)
                                      + (hout * wout * cout // 2 if (flags & F_UP) else 0))

    def rp_ok(self):
        """Can dafne_conv3x3_c256_hip take this call (shape / flags)?"""
        return self.fp8 is None and bool(_lib.load().dafne_conv3x3_c256_ok(ctypes.byref(self.prm), self.segs))

    def num_tiles(self):
        if self.wfrag is not None:
            return _lib.load().dafne_conv3x3_c256_num_tiles(ctypes.byref(self.prm), self.segs)
        if self.fp8 is not None:
            return _lib.load().dafne_conv2d_fp8w_num_tiles(ctypes.byref(self.prm), self.segs)
        return _lib.load().dafne_conv2d_num_tiles(ctypes.byref(self.prm), self.segs)

    def tile_pixels(self):
        return _lib.load().dafne_conv2d_tile_pixels(ctypes.byref(self.prm), self.segs)

    KERNEL_NAMES = ("conv_igemm<1,4,1,2>", "conv_igemm<1,4,2,2>", "conv_igemm<2,2,2,2>", "conv_igemm<4,2,2,4>",
                    "conv_stream", "conv_ws", "conv3x3_patch", "conv3x3_slab", "conv3x3_pred16")

    def kernel_id(self):
        """-1 when the library cannot run this call (e.g. F_GNIN on a layer the patch kernel does not take)."""
        return _lib.load().dafne_conv2d_kernel_id(ctypes.byref(self.prm), self.segs)

    def kernel_name(self):
        """The HIP kernel this call dispatches to (conv.hip), for per-kernel attribution in bench.py."""
        if self.fp8 is not None:
            return "conv3x3_patch_fp8"
        if self.wfrag is not None:
            return "conv3x3_rp"
        return self.KERNEL_NAMES[self.kernel_id()]

    def tiles_per_image(self):
        out = (ctypes.c_int32 * self.prm.n_segs)()
        fn = _lib.load().dafne_conv2d_fp8w_tiles_per_image if self.fp8 is not None else _lib.load().dafne_conv2d_tiles_per_image
        if self.wfrag is not None:
            fn = _lib.load().dafne_conv3x3_c256_tiles_per_image
        _lib.check(fn(ctypes.byref(self.prm), self.segs, out), "dafne_conv2d_tiles_per_image")
        return list(out)

    def __call__(self, stream):
        if self.fp8 is not None:
            rc = _lib.load().dafne_conv2d_nhwc_fp8w_hip(ctypes.byref(self.prm), self.segs, _lib.ptr(self.fp8[0]),
                                                        ctypes.c_float(self.fp8[1]), stream)
            if rc:
                _lib.check(rc, "dafne_conv2d_nhwc_fp8w_hip")
            return
        if self.wfrag is not None:
            scr = rp_scratch(self.wfrag.device)
            rc = _lib.load().dafne_conv3x3_c256_hip(ctypes.byref(self.prm), self.segs, _lib.ptr(self.wfrag), _lib.ptr(scr), scr.numel(), stream)
            if rc:
                _lib.check(rc, "dafne_conv3x3_c256_hip")
            return
        rc = self.fn(ctypes.byref(self.prm), self.segs, stream)
        if rc:
            _lib.check(rc, "dafne_conv2d_nhwc_bf16_hip")


class ConvPairCall:
    """Two resident-patch ConvCalls of identical shape and flags (cls_tower.i, center_tower.i) as ONE launch of the persistent
    kernel (dafne_conv3x3_c256_pair_hip): twice the tiles per launch, half the launch boundaries."""

    def __init__(self, a, b):
        assert a.wfrag is not None and b.wfrag is not None and a.prm.flags == b.prm.flags and a.prm.Cout == b.prm.Cout
        self.a, self.b = a, b
        self.flops, self.bytes = a.flops + b.flops, a.bytes + b.bytes

    def kernel_name(self):
        return "conv3x3_rp"

    def __call__(self, stream):
        a, b = self.a, self.b
        scr = rp_scratch(a.wfrag.device)
        rc = _lib.load().dafne_conv3x3_c256_pair_hip(ctypes.byref(a.prm), a.segs, _lib.ptr(a.wfrag), ctypes.byref(b.prm), b.segs,
                                                     _lib.ptr(b.wfrag), _lib.ptr(scr), scr.numel(), stream)
        if rc:
            _lib.check(rc, "dafne_conv3x3_c256_pair_hip")


def use_wr_kernel():
    """conv_wr_kernel (128 px x 256 ch tiles, weights -> registers, deterministic split-K) for the small-M layers it takes
    (res5, FPN laterals / P4-P5 outputs / P6 / P7, the stride-2 projections); DAFNE_CONV_WR=0 keeps them on conv_igemm /
    conv_stream (A/B runs)."""
    return os.environ.get("DAFNE_CONV_WR", "1") != "0"


WR_NOMINAL_BATCH = 8      # conv_wr.hip kNominalBatch: the choice below and the kernel's slice count look at 8 images' worth of pixels,


def wr_takes(k, stride, cin, cout, npx):
    """npx: WR_NOMINAL_BATCH x the output pixels of ONE image -- not the batch's: which kernel runs a layer (and how conv_wr groups
    its fp32 sum) must not depend on the batch an image sits in, or its detections would (tta.py's grouped views).
    Which small-M layers go to conv_wr: the shapes where it measured faster than the kernel dafne_conv2d_nhwc_bf16_hip picks,
    both as a whole batch of 8 alone on the GPU and as a 3-image sub-batch beside two others (scratch/wr_micro.py,
    profiles/NOTES_r04.md): res5 conv2 (3x3, 512 -> 512: x1.09 / x1.23), res5 conv3 (512 -> 2048 + residual: x1.02 / x1.11),
    FPN lateral 5 / 4 (x1.05-1.17), FPN output 5, P6, P7 (3x3 on <= 8192 pixels: x1.05-1.14), res4.0 conv1 (512 -> 256, stride
    2: x1.28 / x1.09).  Not: res5.0 conv1 / shortcut, res5 conv1 (2048 -> 512), the 64 x 64 3x3 layers (x0.75-1.0)."""
    if k == 3:
        return stride in (1, 2) and npx <= 10000 and cin in (256, 512) and cout == cin
    if stride == 1:
        return (cin == 512 and cout == 2048 and npx <= 10000) or (cout == 256 and cin >= 1024 and npx <= 40000)
    return cin == 512 and cout == 256 and npx <= 40000


class WrWorkspace:
    """Split-K workspace of a plan's dafne_conv2d_wr_hip calls (arrival tickets + fp32 partial slabs).  One per plan: its calls
    run on ONE stream, back to back, and each leaves the tickets zero.  Grown while the plan is built, allocated (zeroed) when
    the plan is complete (DensePlan.__init__) -- never at a launch: the fill would run on torch's current stream, not the
    launch's."""

    def __init__(self, device):
        self.device, self.bytes, self.t = device, 0, None

    def need(self, nbytes):
        assert self.t is None
        self.bytes = max(self.bytes, int(nbytes))

    def tensor(self):
        if self.t is None:
            self.t = torch.zeros(max(self.bytes, 1 << 16), dtype=torch.uint8, device=self.device)
        return self.t


class WrCall:
    """One dafne_conv2d_wr_hip launch built from a ConvCall's parameter block."""

    def __init__(self, conv, wfrag, ws):
        self.conv, self.wfrag, self.ws = conv, wfrag, ws
        self.prm, self.segs = conv.prm, conv.segs
        self.flops, self.bytes = conv.flops, conv.bytes
        L = _lib.load()
        self.splits = L.dafne_conv2d_wr_splits(ctypes.byref(self.prm), self.segs)
        assert self.splits >= 1
        ws.need(L.dafne_conv2d_wr_workspace_bytes(ctypes.byref(self.prm), self.segs))

    def kernel_name(self):
        return "conv_wr"

    def __call__(self, stream):
        w = self.ws.tensor()
        rc = _lib.load().dafne_conv2d_wr_hip(ctypes.byref(self.prm), self.segs, _lib.ptr(self.wfrag), _lib.ptr(w), w.numel(), stream)
        if rc:
            _lib.check(rc, "dafne_conv2d_wr_hip")


class FnCall:
    def __init__(self, fn, args, keep, name, flops=0, nbytes=0):
        self.fn, self.args, self.keep, self.name = fn, args, keep, name
        self.flops, self.bytes = flops, nbytes      # > 0: a matrix kernel that bench.py attributes like a ConvCall

    def kernel_name(self):
        return self.name

    def __call__(self, stream):
        rc = self.fn(*self.args, stream)
        if rc:
            _lib.check(rc, self.name)


def conv_out_hw(h, w, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


# ------------------------------------------------------------------------- plan
class DensePlan:
    """Backbone + FPN + head for one (N, H, W): buffers + ordered launches."""

    def __init__(self, weights, n, h, w, depth, num_classes, device, with_head=True, head_outputs=None, calib=None,
                 shared_gpu=False):
        """calib (dict): build the CALIBRATION plan of an fp8 model -- every layer that could take its plain (not
        GroupNorm-fed) input in e4m3 runs on the bf16 kernel and records max |input| under its weight key; the scales derived
        from it (act_qscale_from_amax) go into weights["act_q8"], and plans built afterwards route those layers to the fp8
        MFMA kernel (dafne_conv2d_nhwc_fp8w_hip with in_qscale = the layer's scale)."""
        assert h % 32 == 0 and w % 32 == 0
        # shared_gpu: the plan runs next to other plans on concurrent streams (the sub-batches of the pipelined step): launches
        # carry no F_EXCL hint.  (Until round 5 the tower layers were not paired there either: -1..-4 % in the in-phase layouts;
        # with unequal sub-batches the pairs gain +0.7 %, HeadPlan)
        self.shared_gpu = bool(shared_gpu)
        self._build(weights, n, h, w, depth, num_classes, device, with_head, head_outputs, calib)

    def _build(self, weights, n, h, w, depth, num_classes, device, with_head, head_outputs, calib):
        self.calib = calib
        self.n, self.h, self.w = n, h, w
        self.device = device
        self.calls = []
        self.flops = 0
        self.graph = None
        L = _lib.load()
        P = weights
        pool = Pool(device)
        self.pool = pool

        act_q8 = P.get("act_q8") or {}
        wr_on = use_wr_kernel()
        self.wr_ws = WrWorkspace(device)
        # conv3x3_c64.hip: 65 -> 49 us per launch when it has the GPU to itself, but nothing in the timed 3-stream layout (its
        # persistent workgroups hold every CU's LDS, so the other sub-batches' kernels cannot share the chip with it): opt-in
        use_c64 = os.environ.get("DAFNE_CONV_C64", "0") == "1"
        # whole-block kernels write Y = relu(conv3(T) + X) OVER X when X is dead after the block (an identity block's input, or
        # block 0's projection output): a tile reads its own X rows and nobody else's (the halo is on the 3x3's input only), so
        # the stores go to DRAM pages the same workgroup has just read (open rows, lines still in L2)
        inplace_res = os.environ.get("DAFNE_INPLACE_RES", "1") != "0"

        def res_dead(sc_, x_, b_):
            # the shortcut operand is not needed after the block: block 0's projection output always; an identity block's input
            # unless it is a retained feature
            if not inplace_res or sc_ is None:
                return False
            if b_ == 0:
                return sc_ is not x_
            return not any(sc_ is f for f in feats.values())

        def conv(key, tin, k, stride, pad, flags, res=None, out=None, cout=None):
            wgt, bias = P[key]
            cin = tin.c
            cout = cout or wgt.shape[0]
            ho, wo = conv_out_hw(tin.h, tin.w, k, stride, pad)
            o = out or pool.get(n, ho, wo, cout)
            if (use_c64 and k == 3 and stride == 1 and pad == 1 and cin == 64 and cout == 64 and res is None
                    and (flags & ~F_RELU) == 0 and tuple(wgt.shape) == (64, 576) and bias is not None):
                # res2 conv2: persistent kernel with the weights in registers (conv3x3_c64.hip)
                fl = 2 * n * ho * wo * 64 * 576
                self.calls.append(FnCall(L.dafne_conv3x3_c64_hip,
                                         (_lib.ptr(tin.t), _lib.ptr(wgt), _lib.ptr(bias), n, ho, wo, 1 if flags & F_RELU else 0, _lib.ptr(o.t)),
                                         (tin, wgt, bias, o), "conv3x3_c64", flops=fl,
                                         nbytes=2 * n * ho * wo * 64 * 2 + wgt.numel() * 2))
                self.flops += fl
                return o
            q8 = P.get(key + ".fp8")
            fp8 = None
            if q8 is not None and k == 3 and stride == 1 and res is None and not (flags & (F_F32 | F_UP | F_RES)):
                if calib is not None:
                    self.calls.append(AmaxProbe(calib, key, [tin]))
                elif key in act_q8:
                    fp8 = (q8[1] / act_q8[key], act_q8[key])       # oscale = weight scale / in_qscale
            c = ConvCall(q8[0] if fp8 else wgt, bias, cin, cout, k, stride, pad, flags,
                         [(tin.t, o.t, res.t if res is not None else None, tin.h, tin.w, ho, wo)], n, fp8=fp8,
                         shared_gpu=self.shared_gpu)
            if (fp8 is None and wr_on and wr_takes(k, stride, cin, cout, WR_NOMINAL_BATCH * ho * wo) and bias is not None
                    and L.dafne_conv2d_wr_ok(ctypes.byref(c.prm), c.segs)):
                # small-M layer (res5, FPN top): 128 px x 256 ch tiles, weights -> registers, split-K
                if key + ".wr" not in P:
                    P[key + ".wr"] = pack_conv_frag(wgt)
                w = WrCall(c, P[key + ".wr"], self.wr_ws)
                self.calls.append(w)
                self.flops += w.flops
                return o
            if fp8 is None and k == 3 and cin == 256 and use_rp_kernel() and c.kernel_id() == 6 and c.rp_ok():
                # 256-channel 3x3 layers the patch kernel would take (FPN outputs): resident-patch kernel
                f16 = rp_frag16()
                fkey = key + (".frag16" if f16 else ".frag")
                if fkey not in P:
                    P[fkey] = pack_rp(wgt)[0]
                c = ConvCall(wgt, bias, cin, cout, k, stride, pad, flags,
                             [(tin.t, o.t, None, tin.h, tin.w, ho, wo)], n, wfrag=P[fkey], shared_gpu=self.shared_gpu, frag16=f16)
            self.calls.append(c)
            self.flops += c.flops
            return o

        # stem: preprocess output [N, H+6, W+6, 4] -> conv7x7/s2 -> maxpool
        self.stem_in = torch.zeros(n, h + 6, w + 6, 4, dtype=BF16, device=device)
        wgt, bias = P["stem"]
        x = pool.get(n, h // 4, w // 4, 64)
        stem_flops = 2 * n * (h // 2) * (w // 2) * 64 * 7 * 7 * 3
        stem_y1 = None            # res2.0's conv1 output when the stem kernel computed it
        if os.environ.get("DAFNE_FUSE_STEM", "1") != "0":
            # conv7x7/s2 + ReLU + max-pool in one kernel: the half-resolution map never reaches HBM
            c1 = P.get("res2.0.conv1")
            if (os.environ.get("DAFNE_FUSE_STEM_CONV1", "1") != "0" and c1 is not None and tuple(c1[0].shape) == (64, 64)
                    and c1[1] is not None):
                # ... and res2.0's first convolution (1x1, 64 -> 64, ReLU) on the pooled tile while it is in LDS
                # (stem_pool.hip CONV1): the pooled map is not read back by a launch of its own (67 MB at batch 8)
                stem_y1 = pool.get(n, h // 4, w // 4, 64)
                c1_flops = 2 * n * (h // 4) * (w // 4) * 64 * 64
                fc = FnCall(L.dafne_stem_pool_conv1_hip, (_lib.ptr(self.stem_in), _lib.ptr(wgt), _lib.ptr(bias), _lib.ptr(c1[0]),
                                                          _lib.ptr(c1[1]), n, h, w, _lib.ptr(x.t), _lib.ptr(stem_y1.t)),
                            (self.stem_in, wgt, bias, c1[0], c1[1], x, stem_y1), "stem_pool_conv1", flops=stem_flops + c1_flops,
                            nbytes=n * ((h + 6) * (w + 6) * 8 + (h // 4) * (w // 4) * 256))
                self.calls.append(fc)
                self.flops += stem_flops + c1_flops
            else:
                fc = FnCall(L.dafne_stem_pool_hip, (_lib.ptr(self.stem_in), _lib.ptr(wgt), _lib.ptr(bias), n, h, w, _lib.ptr(x.t)),
                            (self.stem_in, wgt, bias, x), "stem_pool", flops=stem_flops,
                            nbytes=n * ((h + 6) * (w + 6) * 8 + (h // 4) * (w // 4) * 128))
                self.calls.append(fc)
                self.flops += stem_flops
        else:
            stem_out = pool.get(n, h // 2, w // 2, 64)
            c = ConvCall(wgt, bias, 4, 64, 7, 2, 3, F_RELU,
                         [(self.stem_in, stem_out.t, None, h + 6, w + 6, h // 2, w // 2)], n, shared_gpu=self.shared_gpu)
            self.calls.append(c)
            self.flops += c.flops
            self.calls.append(FnCall(L.dafne_maxpool3x3s2_nhwc_bf16_hip,
                                     (_lib.ptr(stem_out.t), _lib.ptr(x.t), n, h // 2, w // 2, 64),
                                     (stem_out, x), "maxpool"))
            pool.put(stem_out)

        feats = {}
        fuse_b2b = os.environ.get("DAFNE_FUSE_B2B", "1") != "0"
        fuse_narrow = fuse_b2b and os.environ.get("DAFNE_FUSE_B2B_NARROW", "1") != "0"
        fuse_mid = fuse_b2b and os.environ.get("DAFNE_FUSE_B2B_MID", "1") != "0"
        fuse_bneck = fuse_b2b and os.environ.get("DAFNE_FUSE_BNECK", "1") != "0"
        bneck_scratch = None
        blk_scratch = None
        blk_mid_scratch = None
        fuse_blk_mid = fuse_mid and os.environ.get("DAFNE_FUSE_BLK_MID", "1") != "0"
        fuse_blk_narrow = fuse_narrow and os.environ.get("DAFNE_FUSE_BLK_NARROW", "1") != "0"
        for si, nb in enumerate(STAGE_BLOCKS[depth]):
            y1_next = stem_y1 if si == 0 else None
            for b in range(nb):
                p = "res%d.%d." % (si + 2, b)
                stride = 2 if (b == 0 and si > 0) else 1
                w3, b3 = P[p + "conv3"]
                nxt = "res%d.%d.conv1" % (si + 2, b + 1)
                # res2.0: the projection shortcut is computed inside the fused tail (conv_b2b_narrow.hip, PROJ)
                proj_fused = (fuse_narrow and b == 0 and stride == 1 and nb > 1 and tuple(w3.shape) == (256, 64)
                              and tuple(P[p + "shortcut"][0].shape) == (256, 64) and tuple(P[nxt][0].shape) == (64, 256))
                # (the whole-block kernel below takes block 0 under the same conditions: its projection is fused as well)
                if b == 0:
                    sc = None if proj_fused else conv(p + "shortcut", x, 1, stride, 0, 0)
                else:
                    sc = x
                if y1_next is not None:
                    y1, y1_next = y1_next, None                       # computed by the previous block's fused tail
                else:
                    y1 = conv(p + "conv1", x, 1, stride, 0, F_RELU)       # STRIDE_IN_1X1
                w2, b2 = P[p + "conv2"]
                q8_2 = P.get(p + "conv2.fp8")
                bn_head = b + 1 < nb                  # the stage's last block has no next conv1: the kernel's no-head form
                body_fused = (fuse_bneck and tuple(w3.shape) == (1024, 256) and (not bn_head or tuple(P[nxt][0].shape) == (256, 1024))
                              and (bn_head or os.environ.get("DAFNE_FUSE_BNECK_LAST", "1") != "0")
                              and tuple(w2.shape) == (256, 2304) and y1.c == 256
                              # an fp8 model's conv2 takes e4m3 activations on the fp8 MFMA kernel (its definition): not fused
                              and not (q8_2 is not None and (calib is not None or (p + "conv2") in act_q8)))
                if body_fused:
                    # res4: conv2 (3x3) + conv3 + residual + ReLU + the next block's conv1 + ReLU in ONE kernel
                    # (conv_bneck.hip): neither the 3x3's output nor conv3's is read back from HBM
                    w1, b1 = P[nxt] if bn_head else (torch.zeros(256, 1024, dtype=BF16, device=w2.device), None)
                    key = p + "bneck"
                    if key not in P:
                        P[key] = pack_bneck(w2, w3, w1)
                    if bneck_scratch is None:
                        bneck_scratch = torch.empty(L.dafne_bottleneck_body_scratch_bytes(), dtype=torch.uint8, device=device)
                    inpl = res_dead(sc, x, b)
                    y3 = sc if inpl else pool.get(n, y1.h, y1.w, 1024)
                    y1_next = pool.get(n, y1.h, y1.w, 256) if bn_head else None
                    fl = 2 * n * y1.h * y1.w * (256 * 2304 + 256 * 1024 + (1024 * 256 if bn_head else 0))
                    nb_ = n * y1.h * y1.w * (256 + 1024 + 1024 + (256 if bn_head else 0)) * 2 + (256 * 2304 + 2 * 1024 * 256) * 2
                    self.calls.append(FnCall(L.dafne_bottleneck_body_hip,
                                             (_lib.ptr(y1.t), _lib.ptr(sc.t), _lib.ptr(P[key]), _lib.ptr(b2), _lib.ptr(b3), _lib.ptr(b1),
                                              n, y1.h, y1.w, _lib.ptr(y3.t), _lib.ptr(y1_next.t) if bn_head else None, _lib.ptr(bneck_scratch),
                                              bneck_scratch.numel()),
                                             (y1, sc, P[key], b2, b3, b1, y3, y1_next, bneck_scratch),
                                             "conv_bneck" if bn_head else "conv_bneck_last", flops=fl, nbytes=nb_))
                    self.flops += fl
                    pool.put(y1)
                    if b == 0 and sc is not None and sc is not y3:
                        pool.put(sc)
                    if x is not y3 and not any(x is f for k, f in feats.items() if k != "res2"):
                        pool.put(x)
                    x = y3
                    continue
                if (fuse_blk_narrow and tuple(w2.shape) == (64, 576) and tuple(w3.shape) == (256, 64) and y1.c == 64 and stride == 1
                        and (b > 0 or proj_fused) and (b + 1 >= nb or tuple(P[nxt][0].shape) == (64, 256))):
                    # res2: the whole block -- conv2 (3x3) + conv3 + shortcut (identity / projection) + ReLU, and the next
                    # block's conv1 + ReLU when there is one -- in ONE kernel (conv_blk_narrow.hip): the 64-channel map between
                    # the 3x3 and the tail is never written or read
                    head = b + 1 < nb
                    proj = b == 0
                    w1, b1 = P[nxt] if head else (None, None)
                    wsc, bsc = P[p + "shortcut"] if proj else (None, None)
                    key = p + "blk"
                    if key not in P:
                        P[key] = pack_blk_narrow(w2, w3, w1, wsc)
                    if blk_scratch is None:
                        blk_scratch = torch.empty(L.dafne_bottleneck_block_narrow_scratch_bytes(), dtype=torch.uint8, device=device)
                    src = x if proj else sc               # projection: the block input; identity: the previous block's output
                    inpl = (not proj) and res_dead(sc, x, b)
                    y3 = sc if inpl else pool.get(n, y1.h, y1.w, 256)
                    y1_next = pool.get(n, y1.h, y1.w, 64) if head else None
                    px = n * y1.h * y1.w
                    fl = 2 * px * (64 * 576 + 64 * 256 + (256 * 64 if head else 0) + (64 * 256 if proj else 0))
                    nb_ = px * (64 + (64 if proj else 256) + 256 + (64 if head else 0)) * 2 + P[key].numel() * 2
                    self.calls.append(FnCall(L.dafne_bottleneck_block_narrow_hip,
                                             (_lib.ptr(y1.t), _lib.ptr(src.t), _lib.ptr(P[key]), _lib.ptr(b2), _lib.ptr(b3), _lib.ptr(bsc),
                                              _lib.ptr(b1), n, y1.h, y1.w, _lib.ptr(y3.t), _lib.ptr(y1_next.t) if head else None,
                                              _lib.ptr(blk_scratch), blk_scratch.numel()),
                                             (y1, src, P[key], b2, b3, bsc, b1, y3, y1_next, blk_scratch),
                                             "conv_blk_narrow" + ("_proj" if proj else "") + ("" if head else "_last"), flops=fl, nbytes=nb_))
                    self.flops += fl
                    pool.put(y1)
                    if x is not y3 and not any(x is f for k, f in feats.items() if k != "res2"):
                        pool.put(x)
                    x = y3
                    continue
                if (fuse_blk_mid and tuple(w2.shape) == (128, 1152) and tuple(w3.shape) == (512, 128) and y1.c == 128
                        and (b + 1 >= nb or tuple(P[nxt][0].shape) == (128, 512))):
                    # res3: the whole block -- conv2 (3x3) + conv3 + shortcut + ReLU, and the next block's conv1 + ReLU when
                    # there is one -- in ONE kernel (conv_blk_mid.hip)
                    head = b + 1 < nb
                    w1, b1 = P[nxt] if head else (None, None)
                    key = p + "blk"
                    if key not in P:
                        P[key] = pack_blk_mid(w2, w3, w1)
                    if blk_mid_scratch is None:
                        blk_mid_scratch = torch.empty(L.dafne_bottleneck_block_mid_scratch_bytes(), dtype=torch.uint8, device=device)
                    inpl = res_dead(sc, x, b)
                    y3 = sc if inpl else pool.get(n, y1.h, y1.w, 512)
                    y1_next = pool.get(n, y1.h, y1.w, 128) if head else None
                    px = n * y1.h * y1.w
                    fl = 2 * px * (128 * 1152 + 128 * 512 + (512 * 128 if head else 0))
                    nb_ = px * (128 + 512 + 512 + (128 if head else 0)) * 2 + P[key].numel() * 2
                    self.calls.append(FnCall(L.dafne_bottleneck_block_mid_hip,
                                             (_lib.ptr(y1.t), _lib.ptr(sc.t), _lib.ptr(P[key]), _lib.ptr(b2), _lib.ptr(b3), _lib.ptr(b1),
                                              n, y1.h, y1.w, _lib.ptr(y3.t), _lib.ptr(y1_next.t) if head else None,
                                              _lib.ptr(blk_mid_scratch), blk_mid_scratch.numel()),
                                             (y1, sc, P[key], b2, b3, b1, y3, y1_next, blk_mid_scratch),
                                             "conv_blk_mid" + ("" if head else "_last"), flops=fl, nbytes=nb_))
                    self.flops += fl
                    pool.put(y1)
                    if b == 0 and sc is not None and sc is not y3:
                        pool.put(sc)
                    if x is not y3 and not any(x is f for k, f in feats.items() if k != "res2"):
                        pool.put(x)
                    x = y3
                    continue
                y2 = conv(p + "conv2", y1, 3, 1, 1, F_RELU)
                pool.put(y1)
                if proj_fused:
                    w1, b1 = P[nxt]
                    wsc, bsc = P[p + "shortcut"]
                    key = p + "b2b"
                    if key not in P:
                        P[key] = pack_b2b_narrow(w3, w1, wsc)
                    y3 = pool.get(n, y2.h, y2.w, 256)
                    y1_next = pool.get(n, y2.h, y2.w, 64)
                    fl = 2 * n * y2.h * y2.w * (64 * 256 * 2 + 256 * 64)
                    nb_ = n * y2.h * y2.w * (64 + 64 + 256 + 64) * 2 + 3 * 256 * 64 * 2
                    self.calls.append(FnCall(L.dafne_bottleneck_proj_tail_head_narrow_hip,
                                             (_lib.ptr(y2.t), _lib.ptr(x.t), _lib.ptr(P[key]), _lib.ptr(b3), _lib.ptr(bsc), _lib.ptr(b1),
                                              n, y2.h, y2.w, _lib.ptr(y3.t), _lib.ptr(y1_next.t)),
                                             (y2, x, P[key], b3, bsc, b1, y3, y1_next), "conv_b2b_narrow_proj", flops=fl, nbytes=nb_))
                    self.flops += fl
                elif fuse_b2b and b + 1 < nb and tuple(w3.shape) == (1024, 256) and tuple(P[nxt][0].shape) == (256, 1024):
                    # conv3 + residual + ReLU and the next block's conv1 + ReLU in one kernel (conv_b2b.hip)
                    w1, b1 = P[nxt]
                    key = p + "b2b"
                    if key not in P:
                        P[key] = pack_b2b(w3, w1)
                    y3 = pool.get(n, y2.h, y2.w, 1024)
                    y1_next = pool.get(n, y2.h, y2.w, 256)
                    fl = 2 * n * y2.h * y2.w * (256 * 1024 + 1024 * 256)
                    nb_ = n * y2.h * y2.w * (256 + 1024 + 1024 + 256) * 2 + 2 * 1024 * 256 * 2
                    self.calls.append(FnCall(L.dafne_bottleneck_tail_head_hip,
                                             (_lib.ptr(y2.t), _lib.ptr(sc.t), _lib.ptr(P[key]), _lib.ptr(b3), _lib.ptr(b1),
                                              n, y2.h, y2.w, _lib.ptr(y3.t), _lib.ptr(y1_next.t)),
                                             (y2, sc, P[key], b3, b1, y3, y1_next), "conv_b2b", flops=fl, nbytes=nb_))
                    self.flops += fl
                elif fuse_mid and b + 1 < nb and tuple(w3.shape) == (512, 128) and tuple(P[nxt][0].shape) == (128, 512):
                    # res3: the pair with the weights streamed through LDS (conv_b2b_mid.hip)
                    w1, b1 = P[nxt]
                    key = p + "b2b"
                    if key not in P:
                        P[key] = pack_b2b_mid(w3, w1)
                    y3 = pool.get(n, y2.h, y2.w, 512)
                    y1_next = pool.get(n, y2.h, y2.w, 128)
                    fl = 2 * n * y2.h * y2.w * (128 * 512 + 512 * 128)
                    nb_ = n * y2.h * y2.w * (128 + 512 + 512 + 128) * 2 + 2 * 512 * 128 * 2
                    self.calls.append(FnCall(L.dafne_bottleneck_tail_head_mid_hip,
                                             (_lib.ptr(y2.t), _lib.ptr(sc.t), _lib.ptr(P[key]), _lib.ptr(b3), _lib.ptr(b1),
                                              n, y2.h, y2.w, _lib.ptr(y3.t), _lib.ptr(y1_next.t)),
                                             (y2, sc, P[key], b3, b1, y3, y1_next), "conv_b2b_mid", flops=fl, nbytes=nb_))
                    self.flops += fl
                elif fuse_narrow and b + 1 < nb and tuple(w3.shape) == (256, 64) and tuple(P[nxt][0].shape) == (64, 256):
                    # res2: the same pair as a streaming kernel (conv_b2b_narrow.hip)
                    w1, b1 = P[nxt]
                    key = p + "b2b"
                    if key not in P:
                        P[key] = pack_b2b_narrow(w3, w1)
                    y3 = pool.get(n, y2.h, y2.w, 256)
                    y1_next = pool.get(n, y2.h, y2.w, 64)
                    fl = 2 * n * y2.h * y2.w * (64 * 256 + 256 * 64)
                    nb_ = n * y2.h * y2.w * (64 + 256 + 256 + 64) * 2 + 2 * 256 * 64 * 2
                    self.calls.append(FnCall(L.dafne_bottleneck_tail_head_narrow_hip,
                                             (_lib.ptr(y2.t), _lib.ptr(sc.t), _lib.ptr(P[key]), _lib.ptr(b3), _lib.ptr(b1),
                                              n, y2.h, y2.w, _lib.ptr(y3.t), _lib.ptr(y1_next.t)),
                                             (y2, sc, P[key], b3, b1, y3, y1_next), "conv_b2b_narrow", flops=fl, nbytes=nb_))
                    self.flops += fl
                else:
                    y3 = conv(p + "conv3", y2, 1, 1, 0, F_RELU | F_RES, res=sc)
                pool.put(y2)
                if b == 0 and sc is not None:
                    pool.put(sc)
                if not any(x is f for k, f in feats.items() if k != "res2"):
                    pool.put(x)          # res3/res4 outputs stay alive for the FPN laterals
                x = y3
            feats["res%d" % (si + 2)] = x
        self.stage_feats = feats
        # FPN: laterals (+ top-down add fused) and 3x3 outputs
        prev = None
        outs = {}
        for lvl in (5, 4, 3):
            f = feats["res%d" % lvl]
            lat = conv("fpn_lateral%d" % lvl, f, 1, 1, 0, F_UP if prev is not None else 0, res=prev)
            outs["p%d" % lvl] = conv("fpn_output%d" % lvl, lat, 3, 1, 1, 0)
            prev = lat
        p6 = conv("p6", outs["p5"], 3, 2, 1, 0)
        p6r = pool.get(n, p6.h, p6.w, p6.c)
        self.calls.append(FnCall(L.dafne_relu_copy_bf16_hip, (_lib.ptr(p6.t), _lib.ptr(p6r.t), p6.t.numel()),
                                 (p6, p6r), "relu_copy"))
        p7 = conv("p7", p6r, 3, 2, 1, 0)
        outs["p6"], outs["p7"] = p6, p7
        self.features = [outs[k] for k in ("p3", "p4", "p5", "p6", "p7")]
        self.head_start = len(self.calls)          # launches [0, head_start) are backbone + FPN, the rest the head
        self.head = HeadPlan(weights, self.features, num_classes, device, pool, self, head_outputs) if with_head else None
        # the split-K workspace is allocated and ZEROED here, on the stream the plan is built on, like every other buffer of the
        # plan.  (Round 4 zeroed it at the first launch: torch.zeros then ran on torch's current stream while the launch went
        # to the sub-batch's own stream -- with small images, where the GPU keeps up with the eager enqueue, the fill landed
        # on the arrival tickets / partial slabs of a running conv_wr: one call of garbage features in ~1 of 8 processes,
        # found in round 5 as test_batch_invariance_and_determinism / test_tta_packed_chunks... failing now and then.)
        if any(isinstance(c, WrCall) for c in self.calls):
            self.wr_ws.tensor()

    def run(self, stream=None):
        if self.graph is not None and stream is None:
            self.graph.replay()
            return
        stream = stream if stream is not None else _lib.current_stream()
        for c in self.calls:
            c(stream)

    def capture(self):
        """Capture the launch list into a HIP graph (one host call per replay instead of
        ~150): the kernels, their arguments and buffers are static for a given plan."""
        if self.graph is not None:
            return
        self.run()                       # warm-up outside capture (function attributes, lazy init)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
            for c in self.calls:
                c(_lib.current_stream())
        self.graph = g

    def capture_parts(self):
        """The launch list as TWO HIP graphs -- backbone + FPN, then the head -- so that an event can be recorded on the stream
        where the head towers begin (OneStageDetector.detect_packed(defer=True) starts the previous step's post-process there)."""
        if getattr(self, "graph_parts", None) is not None:
            return
        self.run(_lib.current_stream())  # warm-up outside capture
        torch.cuda.synchronize()
        parts = []
        for lo, hi in ((0, self.head_start), (self.head_start, len(self.calls))):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                for c in self.calls[lo:hi]:
                    c(_lib.current_stream())
            parts.append(g)
        self.graph_parts = tuple(parts)


# Stream-capture mode of the launch-plan graphs.  The default ("global") makes a capture fail when ANY thread of the process
# calls a capture-unsafe runtime function meanwhile; a multi-process job has such threads by construction (RCCL's watchdog
# polling the detection gather's events, the test loader's staging of the next batch).  Only this thread enqueues into the
# captured stream, so thread-local checking loses nothing.
_CAPTURE_MODE = "thread_local"


class HeadOutputs:
    """Whole-batch head outputs shared by the sub-batch plans of a split batch; quacks like
    a HeadPlan for dafne.head_levels()."""

    def __init__(self, n, h, w, num_classes, scales, device):
        p5 = (h // 32, w // 32)
        p6 = ((p5[0] + 1) // 2, (p5[1] + 1) // 2)             # 3x3 stride-2 pad-1 convs
        p7 = ((p6[0] + 1) // 2, (p6[1] + 1) // 2)
        sizes = [(h // 8, w // 8), (h // 16, w // 16), p5, p6, p7]
        f32 = torch.float32
        self.logits = [torch.empty(n, a, b, num_classes, dtype=f32, device=device) for a, b in sizes]
        self.center = [torch.empty(n, a, b, 2, dtype=f32, device=device) for a, b in sizes]
        self.delta_ctr = [torch.empty(n, a, b, 9, dtype=f32, device=device) for a, b in sizes]
        self.scales = scales

    def views(self, lo, hi):
        return {"logits": [t[lo:hi] for t in self.logits], "center": [t[lo:hi] for t in self.center],
                "delta_ctr": [t[lo:hi] for t in self.delta_ctr]}


class CallList:
    """Bare launch list (what HeadPlan needs from its owner)."""

    def __init__(self):
        self.calls = []
        self.flops = 0

    def run(self, stream=None):
        stream = stream if stream is not None else _lib.current_stream()
        for c in self.calls:
            c(stream)


class HeadPlan:
    """DAFNeHead over the 5 levels in single launches (weights are shared across
    levels: dafne.py:350-494): 12 tower convs (+GroupNorm+ReLU) and 3 prediction
    convs; outputs fp32 NHWC logits / [delta8|ctrness] / center."""

    def __init__(self, P, feats, num_classes, device, pool, plan, outputs=None):
        """outputs: optional {"logits"|"center"|"delta_ctr": [5 fp32 NHWC tensors]} to write
        into (views of a larger batch's buffers when the batch is split over streams)."""
        L = _lib.load()
        n = feats[0].n
        self.levels = feats
        C = feats[0].c
        self.num_classes = num_classes
        calls = plan.calls
        sg = bool(getattr(plan, "shared_gpu", False))
        fuse_gn = os.environ.get("DAFNE_FUSE_GN", "1") != "0"
        fuse_gnfin = os.environ.get("DAFNE_FUSE_GNFIN", "1") != "0"

        def seg_list(ins, outs, f32=False):
            return [(i.t, (o if f32 else o.t), None, i.h, i.w, i.h, i.w) for i, o in zip(ins, outs)]

        # (round 5: pairs in the sub-batch plans of the pipelined step as well -- with three sub-batches of unequal size 1371-1373
        # against 1356-1365 img/s, two alternating runs; in round 3's in-phase layout the pairs cost 1-4 % there)
        pair_towers = os.environ.get("DAFNE_RP_PAIR", "1") != "0" and (not getattr(plan, "shared_gpu", False)
                                                                       or os.environ.get("DAFNE_RP_PAIR_SHARED", "1") == "1")
        deferred = []              # intermediate maps of cls_tower / center_tower: released when BOTH towers are built (see below)

        def tower(name, ins, in_gn, consumers):
            """4 x [conv3x3 -> GroupNorm(32) -> ReLU].  When the library's 3x3 patch kernel takes the layer
            (kernel id 6 with F_GNIN), the GroupNorm + ReLU of layer i is applied by layer i+1 while it loads
            its input patch: only the statistics are finalised between the two convolutions, and the separate
            normalisation pass (a read + write of all five levels) disappears.  The same holds for the
            layers behind the tower (`consumers`: (weight key, Cout, flags) of the prediction convolution --
            slab kernel, id 7 -- and, for the center tower, of the corners tower's first layer): when every one
            of them takes F_GNIN the tower returns its last layer RAW together with (stats, gamma, beta)."""
            cur, cur_gn = ins, in_gn
            for i in range(4):
                wgt, bias = P["%s.%d" % (name, 3 * i)]
                gamma, beta = P["%s.%d.gn" % (name, 3 * i + 1)]
                outs = [pool.get(n, f.h, f.w, C) for f in cur]
                flags = F_GN | (F_GNIN if cur_gn is not None else 0)
                lkey = "%s.%d" % (name, 3 * i)
                q8 = P.get(lkey + ".fp8")
                aq = 1.0 if cur_gn is not None else (P.get("act_q8") or {}).get(lkey)
                calib = getattr(plan, "calib", None)
                if q8 is not None and cur_gn is None and calib is not None:
                    calls.append(AmaxProbe(calib, lkey, cur))            # FPN-fed layer: one scale for the five levels
                    aq = None                                            # ... and bf16 while calibrating
                # M-tile geometry comes from the library (the kernel choice fixes the tile shape; the fp8 kernel always
                # uses the patch kernel's tiles, the bf16 one only when the launch has enough of them)
                probe = ConvCall(wgt, bias, C, C, 3, 1, 1, flags & ~F_GN, seg_list(cur, outs), n, gn_in=cur_gn, shared_gpu=sg)
                rp_on = use_rp_kernel() and C == 256
                use_fp8 = q8 is not None and aq is not None and (cur_gn is None or probe.kernel_id() == 6 or (rp_on and probe.rp_ok()))
                # (the FPN-fed layer 0 of a SMALL plan -- the 2-image sub-batch of the timed layout -- takes the generic tile, kernel
                # id != 6, and is then followed by ONE dafne_groupnorm_finalize_hip launch per tower: the two gn_finalize launches
                # per step in the pipelined rocprof stats.  DAFNE_RP_LAYER0=1 puts it on the persistent kernel with fused finalize:
                # measured -0.3 .. 0 % in the timed layout (round 4), so it stays opt-in)
                rp_small = os.environ.get("DAFNE_RP_LAYER0", "0") == "1"
                use_rp = rp_on and not use_fp8 and probe.rp_ok() and (cur_gn is not None or probe.kernel_id() == 6 or rp_small)
                wfrag = None
                f16 = use_rp and rp_frag16()
                if use_rp:
                    fkey = lkey + (".frag16" if f16 else ".frag")
                    if fkey not in P:
                        P[fkey] = pack_rp(wgt)[0]
                    wfrag = P[fkey]
                    probe = ConvCall(wgt, bias, C, C, 3, 1, 1, flags & ~F_GN, seg_list(cur, outs), n, gn_in=cur_gn, wfrag=wfrag, shared_gpu=sg,
                                     frag16=f16)
                if use_fp8:
                    probe = ConvCall(q8[0], bias, C, C, 3, 1, 1, flags & ~F_GN, seg_list(cur, outs), n, gn_in=cur_gn,
                                     fp8=(q8[1] / aq, aq), shared_gpu=sg)
                is_patch = use_fp8 or use_rp or probe.kernel_id() == 6
                nt = probe.num_tiles()
                partial = torch.empty(nt, C // 8, 2, dtype=torch.float32, device=device)
                stats = torch.empty(len(outs), n, C // 8, 2, dtype=torch.float32, device=device)
                # does every consumer of this layer normalise on load?  (decided before the producer is built: the
                # producer then finalises the statistics itself, flag F_GNFIN, and no launch sits between the two convs)
                nxt_specs = [("%s.%d" % (name, 3 * (i + 1)), C, 0)] if i < 3 else consumers
                fuse_next = fuse_gn and len(nxt_specs) > 0
                for key, cout, fl in (nxt_specs if fuse_next else ()):
                    wn, bn_ = P[key]
                    f32 = bool(fl & F_F32)
                    dst = [torch.empty(1, dtype=torch.float32, device=device)] * len(outs) if f32 else outs
                    nxt = ConvCall(wn, bn_, C, cout, 3, 1, 1, fl | F_GNIN, seg_list(outs, dst, f32=f32), n,
                                   gn_in=(stats, gamma, beta), shared_gpu=sg)
                    fuse_next = fuse_next and (nxt.kernel_id() in ((7, 8) if f32 else (6,)) or (not f32 and rp_on and nxt.rp_ok()))
                fuse_fin = fuse_next and fuse_gnfin and is_patch and C == 256
                fin = (stats, torch.zeros(len(outs), n, dtype=torch.int32, device=device), 1e-5) if fuse_fin else None
                if fuse_fin:
                    flags |= F_GNFIN
                if use_fp8:
                    # fp8 model: e4m3 weights on the fp8 MFMA kernel.  GroupNorm-fed layers: the GroupNorm + ReLU output is
                    # quantised on load (in_qscale 1: a normalised, rectified map sits well inside e4m3's range); the two
                    # layers that read FPN features use the calibrated scale of their input
                    c = ConvCall(q8[0], bias, C, C, 3, 1, 1, flags, seg_list(cur, outs), n, gn_partial=partial,
                                 gn_in=cur_gn, fp8=(q8[1] / aq, aq), gn_fin=fin, shared_gpu=sg)
                else:
                    c = ConvCall(wgt, bias, C, C, 3, 1, 1, flags, seg_list(cur, outs), n, gn_partial=partial, gn_in=cur_gn,
                                 gn_fin=fin, wfrag=wfrag, shared_gpu=sg, frag16=f16)
                c.tower_tag = (name, i)
                calls.append(c)
                plan.flops += c.flops
                gsegs = (_lib.GnSeg * len(outs))()
                t0 = 0
                for k, (o, tpi) in enumerate(zip(outs, c.tiles_per_image())):
                    gsegs[k] = _lib.GnSeg(o.t.data_ptr(), o.h, o.w, t0, tpi)
                    t0 += tpi * n
                assert t0 == nt == c.num_tiles()
                if fuse_fin:
                    nxt_gn = (stats, gamma, beta)
                elif fuse_next:
                    calls.append(FnCall(L.dafne_groupnorm_finalize_hip,
                                        (gsegs, len(outs), n, C, _lib.ptr(partial), _lib.ptr(stats), ctypes.c_float(1e-5)),
                                        (outs, partial, stats), "groupnorm_finalize"))
                    nxt_gn = (stats, gamma, beta)
                else:
                    calls.append(FnCall(L.dafne_groupnorm_relu_nhwc_bf16_hip,
                                        (gsegs, len(outs), n, C, _lib.ptr(partial), _lib.ptr(stats), _lib.ptr(gamma),
                                         _lib.ptr(beta), ctypes.c_float(1e-5)), (outs, partial, stats, gamma, beta),
                                        "groupnorm"))
                    nxt_gn = None
                if i > 0:
                    for a in cur:
                        # cls_tower.i and center_tower.i may run in ONE launch: a map the cls tower has released must not be
                        # handed to the center tower while the pair that still reads it is in flight
                        (deferred.append(a) if pair_towers and name != "corners_tower" else pool.put(a))
                cur, cur_gn = outs, nxt_gn
            return cur, cur_gn

        fuse_pred = os.environ.get("DAFNE_FUSE_GN_PRED", "1") != "0"
        cls_t, cls_gn = tower("cls_tower", feats, None, [("cls_logits", num_classes, F_F32)] if fuse_pred else [])
        ctr_t, ctr_gn = tower("center_tower", feats, None,
                              [("center_pred", 2, F_F32), ("corners_tower.0", C, 0)] if fuse_pred else [])
        for a in deferred:
            pool.put(a)
        cor_t, cor_gn = tower("corners_tower", ctr_t, ctr_gn, [("corners_ctrness", 9, F_F32)] if fuse_pred else [])
        # cls_tower.i and center_tower.i are independent chains of identical shape: one launch of the persistent kernel for
        # both (the pair takes cls_tower.i's place in the launch list: center_tower.i only moves EARLIER, behind the pair that
        # holds center_tower.i-1 -- legal when that layer's statistics are finalised inside its kernel, F_GNFIN)
        if pair_towers:
            tagged = {c.tower_tag: c for c in calls if isinstance(c, ConvCall) and hasattr(c, "tower_tag")}
            for i in range(4):
                a, b = tagged.get(("cls_tower", i)), tagged.get(("center_tower", i))
                if a is None or b is None or a.wfrag is None or b.wfrag is None or a.prm.flags != b.prm.flags:
                    break
                if i > 0 and not (tagged[("cls_tower", i - 1)].prm.flags & tagged[("center_tower", i - 1)].prm.flags & F_GNFIN):
                    break
                calls[calls.index(a)] = ConvPairCall(a, b)
                calls.remove(b)

        def pred(key, ins, cout, name, gn_in):
            wgt, bias = P[key]
            if outputs is not None:
                outs = outputs[name]
                assert all(o.is_contiguous() and tuple(o.shape) == (n, f.h, f.w, cout) for o, f in zip(outs, ins))
            else:
                outs = [torch.empty(n, f.h, f.w, cout, dtype=torch.float32, device=device) for f in ins]
            c = ConvCall(wgt, bias, C, cout, 3, 1, 1, F_F32 | (F_GNIN if gn_in is not None else 0),
                         seg_list(ins, outs, f32=True), n, gn_in=gn_in, shared_gpu=sg)
            calls.append(c)
            plan.flops += c.flops
            return outs

        self.logits = pred("cls_logits", cls_t, num_classes, "logits", cls_gn)
        self.center = pred("center_pred", ctr_t, 2, "center", ctr_gn)
        self.delta_ctr = pred("corners_ctrness", cor_t, 9, "delta_ctr", cor_gn)     # corners_pred (8) + ctrness (1) fused
        self.scales = P["scales"]


# ---------------------------------------------------------- weights from a state dict
def _wq(w, fp8):
    """fp8 model (ENGINE.WEIGHT_DTYPE fp8_e4m3): the weight every kernel sees is the dequantised e4m3 weight."""
    return quantize_weight_e4m3(w)[2] if fp8 else w


def pack_backbone_weights(sd, depth, device, prefix="backbone.", fp8=False):
    """state dict with the reference checkpoint's names -> packed
    backbone weights; FrozenBN folded into weight scale + bias.  fp8: the folded weights are quantised to e4m3 with
    power-of-two per-channel scales (quantize_weight_e4m3) and run, exactly dequantised, on the bf16 kernels."""
    P = {}
    bu = prefix + "bottom_up."

    def cb(k):
        w, b = fold_frozen_bn(sd[k + ".weight"].float(), sd[k + ".norm.weight"].float(),
                              sd[k + ".norm.bias"].float(), sd[k + ".norm.running_mean"].float(),
                              sd[k + ".norm.running_var"].float())
        return _wq(w, fp8), b

    w, b = cb(bu + "stem.conv1")
    P["stem"] = pack_stem(w, b, device)
    for si, nb in enumerate(STAGE_BLOCKS[depth]):
        for blk in range(nb):
            for cname in (("shortcut",) if blk == 0 else ()) + ("conv1", "conv2", "conv3"):
                w, b = cb("%sres%d.%d.%s" % (bu, si + 2, blk, cname))
                P["res%d.%d.%s" % (si + 2, blk, cname)] = pack_conv(w, b, device)
                if fp8 and cname == "conv2" and w.shape[0] % 256 == 0:
                    # res4 / res5 3x3 layers: e4m3 bytes of the (BN-folded) weight for the fp8 MFMA kernel; used once
                    # the layer's activation scale has been calibrated (DensePlan, act_q8)
                    P["res%d.%d.%s.fp8" % (si + 2, blk, cname)] = pack_conv_fp8(w, device)
    for lvl in (3, 4, 5):
        for kind in ("lateral", "output"):
            k = "%sfpn_%s%d" % (prefix, kind, lvl)
            P["fpn_%s%d" % (kind, lvl)] = pack_conv(_wq(sd[k + ".weight"], fp8), sd[k + ".bias"], device)
            if fp8 and kind == "output":
                P["fpn_output%d.fp8" % lvl] = pack_conv_fp8(sd[k + ".weight"], device)
    for nme in ("p6", "p7"):
        k = prefix + "top_block." + nme
        P[nme] = pack_conv(_wq(sd[k + ".weight"], fp8), sd[k + ".bias"], device)
    return P


def pack_head_weights(sd, device, prefix="proposal_generator.dafne_head.", fp8=False):
    """DAFNeHead parameters -> packed weights.  corners_pred and ctrness both read
    the corners tower (dafne.py:403,467-468) and are fused into one 9-channel conv.
    fp8: every weight is the dequantised e4m3 weight; the tower layers additionally get key + ".fp8" =
    (e4m3 bytes, scale) for the fp8 MFMA kernel (HeadPlan uses it where the input is normalised on load)."""
    P = {}
    hp = prefix
    for tower in ("cls_tower", "center_tower", "corners_tower"):
        for i in range(4):
            k = "%s%s.%d" % (hp, tower, 3 * i)
            P["%s.%d" % (tower, 3 * i)] = pack_conv(_wq(sd[k + ".weight"], fp8), sd[k + ".bias"], device)
            if fp8:
                P["%s.%d.fp8" % (tower, 3 * i)] = pack_conv_fp8(sd[k + ".weight"], device)
            g = "%s%s.%d" % (hp, tower, 3 * i + 1)
            P["%s.%d.gn" % (tower, 3 * i + 1)] = (sd[g + ".weight"].float().to(device).contiguous(),
                                                  sd[g + ".bias"].float().to(device).contiguous())
    P["cls_logits"] = pack_conv(_wq(sd[hp + "cls_logits.weight"], fp8), sd[hp + "cls_logits.bias"], device)
    P["center_pred"] = pack_conv(_wq(sd[hp + "center_pred.weight"], fp8), sd[hp + "center_pred.bias"], device)
    wcc = torch.cat([sd[hp + "corners_pred.weight"].float(), sd[hp + "ctrness.weight"].float()], 0)
    bcc = torch.cat([sd[hp + "corners_pred.bias"].float(), sd[hp + "ctrness.bias"].float()], 0)
    P["corners_ctrness"] = pack_conv(_wq(wcc, fp8), bcc, device)
    P["scales"] = [float(sd["%sscales.%d.scale" % (hp, l)].reshape(-1)[0]) for l in range(5)]
    return P


def pack_model_weights(sd, depth, device, weight_dtype="bf16", fp8_kernel="patch"):
    """weight_dtype: "bf16" or "fp8_e4m3" (cfg.ENGINE.WEIGHT_DTYPE; BASELINE config 5); fp8_kernel: cfg.ENGINE.FP8_CONV3X3_KERNEL
    (FP8_KERNELS: "patch" only since round 6; kept in the packed weights because a kernel's rounding is part of the model)."""
    if weight_dtype not in ("bf16", "fp8_e4m3"):
        raise NotImplementedError("ENGINE.WEIGHT_DTYPE %r (bf16 or fp8_e4m3)" % (weight_dtype,))
    if fp8_kernel not in FP8_KERNELS:
        raise NotImplementedError("ENGINE.FP8_CONV3X3_KERNEL %r (%s)" % (fp8_kernel, " or ".join(FP8_KERNELS)))
    fp8 = weight_dtype == "fp8_e4m3"
    P = pack_backbone_weights(sd, depth, device, fp8=fp8)
    P.update(pack_head_weights(sd, device, fp8=fp8))
    P["fp8_kernel"] = fp8_kernel
    return P
