#!/usr/bin/env python3
"""bench.py -- images/sec of the DAFNe inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the whole path over one batch of synthetic 1024x1024 BGR
uint8 tiles that already sit in HBM: normalise/pad -> ResNet-FPN -> DAFNe head ->
decode/top-k -> rotated NMS -> rescale/gather, and for N>1 the final detection
gather to rank 0 over RCCL.  Weak scaling: every rank runs the same per-GPU batch
(images are independent; weights replicated); value = images of all ranks / the
slowest rank's time.
The timed layout is a loop's: step i enqueues its convolutions (sub-batches on concurrent streams) and then the decode / NMS /
gather of step i - 1, which start where step i reaches its head towers; the last step's are completed by a flush INSIDE the
timed region, the warm-up's before it -- K steps = K whole passes (--no-defer: every step's post-process right behind its own
convolutions, round 3's form).

Workload (config.workload): BASELINE.json's metric names R101-FPN on 1024x1024
DOTA tiles at 1/2/4/8 GPUs, i.e. the per-GPU shard of configs[2] (batch 8 per GPU,
DOTA-1.0 head: 15 classes, THRESH_WITH_CTR, SORT_CORNERS).  configs[1] (R50-FPN,
batch 8, one GPU) is measured in the same run at N=1 and reported under
"configs1_r50_b8".  Random-init weights (seeded), synthetic images.

Extra objects on the JSON line:
  roofline      dominant kernel = the kernel with the most GPU time per step (whole batch on one stream: every launch
                has the GPU to itself, the layout `--mode serial` runs and profiles/r04b_kernel_stats_isolated.txt
                profiles with rocprofv3): algorithmic FLOPs of its launches / their HIP-event time vs the dense bf16
                MFMA peak (2.5 PFLOP/s); roofline.library_gemm = what torch.matmul (hipBLASLt) sustains in the same run on
                that kernel's GEMM shape and on an 8192^3 bf16 GEMM (the practical ceiling: the matrix pipe's clock follows
                its power draw).  Its sibling "roofline_timed" repeats the bookkeeping for the timed region's
                layout (2 sub-batches on concurrent streams, where a launch's duration includes sharing the GPU);
                every instantiation is listed under "kernels" (timed layout) and "kernels_isolated".  Rounds 1's
                "roofline" was the timed layout, round 2 nested it as roofline.shared_stream: compare like with like
  roofline_nms  the rotated NMS on the SURVEY 8(d) candidate sets: class-filtered pairs per second and algorithmic bytes
                per second against HBM
  cpu_baseline  the torch fp32 oracle (oracle/model.py + oracle/postprocess.py, a
                port) timed on the host cores on a bounded sample, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0        # HBM3E spec (6.29 TB/s measured float4 copy, same guide)
GFLOP_PER_IMG = {50: 505.97, 101: 661.13}   # SURVEY 8(d), 1024^2, C=15


def seeded_state_dict(model, seed, tower_std=None):
    """Random-init weights of the architecture: He-normal convs, FrozenBN scale
    U(0.5,1.5) / shift N(0,0.1), head towers N(0, 0.03), class prior -4.595
    (dafne.py:269-285).  He init (instead of the reference's N(0,0.01) towers /
    pretrained trunk) keeps activations O(1) through the trunk so the bench does
    not time an all-zero network (zeros clock higher: guide rule 25)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.state_dict().items():
        leaf = k.split(".")[-1]
        if v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            std = (2.0 / fan_in) ** 0.5
            if "_tower" in k and tower_std is not None:
                std = tower_std               # dafne.py:269-285: the reference initialises the head convolutions N(0, 0.01)
            if any(t in k for t in ("cls_logits", "ctrness", "corners_pred", "center_pred")):
                std = 0.01
            sd[k] = torch.randn(v.shape, generator=g) * std
        elif leaf == "running_var":
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif leaf == "running_mean":
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        elif leaf == "scale":
            sd[k] = torch.ones(v.shape)
        elif leaf == "weight":
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif leaf == "bias":
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
            if "cls_logits" in k:
                sd[k] = torch.full(v.shape, -4.59512)
        else:
            sd[k] = v.clone()
    return sd


def build_model(depth, device, seed=0, cfgname=None, cls_prior=None, tower_std=None):
    """cls_prior: class-logit bias instead of the reference's -4.595.  The configs that threshold the RAW class score
    (THRESH_WITH_CTR false: DOTA-1.5, UCAS-AOD, HRSC) yield no candidates at -4.595 with random weights; the side
    metrics on those configs raise it so that decode / NMS see full candidate sets."""
    import dafne_amd.modeling  # noqa: F401
    from dafne_amd.config import load_cfg
    from dafne_amd.registry import build_model as bm
    cfg = load_cfg(os.path.join(ROOT, "configs", cfgname or "dota-1.0_r%d.yaml" % depth))
    m = bm(cfg)
    sd = seeded_state_dict(m, seed, tower_std=tower_std)
    if cls_prior is not None:
        kb = "proposal_generator.dafne_head.cls_logits.bias"
        sd[kb] = torch.full_like(sd[kb], float(cls_prior))
    m.load_state_dict(sd)
    m.to(device)
    m.invalidate()
    return cfg, m, sd


PRIME = 4       # steps that build everything built on first use: both plan sets run once eagerly, then their graphs are captured


def time_regions(step_fn, steps, warmup, distributed, regions, device="cuda", per_rank=None, flush_fn=None):
    """`regions` consecutive timed regions of `steps` steps each (every one bracketed as time_steps brackets its own; the
    warm-up runs once, before the first) -> (seconds of the MEDIAN region, [seconds of every region]).  The line's `value` /
    `ms_per_step` are the median region's, so `ms_per_step x steps` is still ONE region of exactly K steps; min / max go on
    the line beside it (one 0.12-s region on boxes that differ by +-4 % carries no spread).  per_rank: the
    ranks' own times of the median region."""
    times, ranks = [], []
    for r in range(max(1, int(regions))):
        pr = []
        times.append(time_steps(step_fn, steps, warmup if r == 0 else 0, distributed, device, per_rank=pr, flush_fn=flush_fn))
        ranks.append(pr)
    order = sorted(range(len(times)), key=lambda i: times[i])
    med = order[(len(order) - 1) // 2]           # lower median: an actually measured region
    if per_rank is not None:
        per_rank[:] = ranks[med]
    return times[med], times


def time_steps(step_fn, steps, warmup, distributed, device="cuda", per_rank=None, flush_fn=None):
    """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + device synchronize on both sides; the
    result is the MAX over ranks (the slowest rank's time).  device "cpu" is the gloo test's layout (no GPU).
    flush_fn: a step function that leaves part of its step to the NEXT call (the deferred post-process of the timed layout)
    completes it here -- after the warm-up (nothing of an untimed step leaks into the timed region) and after the last timed
    step, INSIDE the timed region (all of the K steps' work is timed)."""
    sync = torch.cuda.synchronize if str(device).startswith("cuda") else (lambda: None)
    for _ in range(warmup):
        step_fn()
    if flush_fn is not None:
        flush_fn()
    if distributed:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    if flush_fn is not None:
        flush_fn()
    if distributed:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if distributed:
        mine = torch.tensor([dt], dtype=torch.float64, device=device)
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if per_rank is not None:                 # every rank's own time, for the line's self-check (min / max over ranks)
            every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(every, mine)
            per_rank[:] = [float(e.item()) for e in every]
        dt = float(t.item())
    elif per_rank is not None:
        per_rank[:] = [dt]
    return dt


def conv_kernel_profile(model, batch, splits, reps=3):
    """Per-launch HIP-event timing of every conv launch of one step, on the SAME plans,
    streams and enqueue order as the timed region (sub-batch plans on concurrent streams;
    each event pair is recorded on the stream its kernel is launched on, so a launch's
    duration includes the sharing of the GPU with the other sub-batch, as rocprofv3 sees it)."""
    from dafne_amd import engine, _lib
    n, _, h, w = batch.shape
    import ctypes
    st = model._pipe[(n, h, w, max(1, min(splits, n)))]
    plans = st["plans"][0]
    cs = st["cs"]
    sp = [ctypes.c_void_p(s.cuda_stream) for s in cs]
    torch.cuda.synchronize()
    stats = {}
    for _ in range(reps):
        evs = []
        ref = torch.cuda.Event(enable_timing=True)
        ref.record(cs[0])
        # same enqueue order as the timed region: launch j of every sub-batch, each on its stream
        for j in range(max(len(p.calls) for p in plans)):
            for k, plan in enumerate(plans):
                if j >= len(plan.calls):
                    continue
                c = plan.calls[j]
                if getattr(c, "flops", 0) > 0:        # ConvCall or a matrix FnCall (stem_pool)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(cs[k])
                    c(sp[k])
                    b.record(cs[k])
                    evs.append((c, a, b))
                else:
                    c(sp[k])
        torch.cuda.synchronize()
        spans = {}
        for c, a, b in evs:
            cfgname = c.kernel_name()
            s = stats.setdefault(cfgname, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "union_ms": 0.0})
            s["ms"] += a.elapsed_time(b)
            s["flops"] += c.flops
            s["bytes"] += c.bytes
            s["launches"] += 1
            spans.setdefault(cfgname, []).append((ref.elapsed_time(a), ref.elapsed_time(b)))
        # wall time during which at least one launch of the kernel was running (its launches on the three streams
        # overlap each other): the kernel family's throughput while it is on the GPU
        for name, iv in spans.items():
            iv.sort()
            tot, (lo, hi) = 0.0, iv[0]
            for a0, b0 in iv[1:]:
                if a0 > hi:
                    tot += hi - lo
                    lo, hi = a0, b0
                else:
                    hi = max(hi, b0)
            stats[name]["union_ms"] += tot + (hi - lo)
    for s in stats.values():
        s["ms"] /= reps
        s["union_ms"] /= reps
        s["flops"] /= reps
        s["bytes"] /= reps
        s["launches"] //= reps
        s["tflops"] = s["flops"] / (s["ms"] * 1e-3) / 1e12 if s["ms"] > 0 else 0.0
        s["avg_launch_us"] = 1e3 * s["ms"] / max(s["launches"], 1)
    return stats


def conv_kernel_profile_isolated(model, batch, reps=5):
    """Same per-launch event timing, but the whole batch as ONE plan on one stream: every
    launch has the GPU to itself (kernel quality without stream sharing).  Per launch the MEDIAN over `reps` passes: one
    disturbed pass (a clock dip: seen once, conv_bneck 125 instead of 77 us in every launch of a run) does not end up in the
    roofline record."""
    from dafne_amd import engine, _lib
    n, _, h, w = batch.shape
    plan = model.plan(n, h, w)
    model.detect_packed(batch)
    torch.cuda.synchronize()
    stream = _lib.current_stream()
    timed = [c for c in plan.calls if getattr(c, "flops", 0) > 0]        # ConvCall or a matrix FnCall (stem_pool)
    per_call = [[] for _ in timed]
    for _ in range(reps):
        evs = []
        for c in plan.calls:
            if getattr(c, "flops", 0) > 0:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                c(stream)
                b.record()
                evs.append((a, b))
            else:
                c(stream)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            per_call[i].append(a.elapsed_time(b))
    stats = {}
    for c, ts in zip(timed, per_call):
        s = stats.setdefault(c.kernel_name(), {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        s["ms"] += sorted(ts)[len(ts) // 2]
        s["flops"] += c.flops
        s["bytes"] += c.bytes
        s["launches"] += 1
    return {k: {"tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                "hbm_gbps_algorithmic": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                "hbm_frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                "ms_per_step": v["ms"], "launches": v["launches"], "flops": v["flops"], "bytes": v["bytes"]}
            for k, v in stats.items()}


def energy_ledger(model, batch, device, seconds=2.0):
    """Joules, not only cycles: the package runs at its power cap in the timed loop, so a kernel's TIME there
    is roughly its ENERGY / cap.  Every instantiation of `kernels_isolated` is looped alone (all its launches of the serial plan,
    in plan order, back to back) for >= `seconds` while amdgpu's hwmon power / clock are sampled; the same for the practical
    ceilings: the 8192^3 library GEMM and a plain copy.  Per entry: socket_w, sclk_mhz, ms_per_step (looped: the chip is warm and
    at the clock its power allows -- slower than the single-shot `kernels_isolated` figure), joules_per_step = socket_w x
    ms_per_step, pj_per_flop, and the same minus the idle draw (`*_above_idle`).  The hwmon figure is a ~1-s average: the first
    second of every loop is dropped.  Inputs of an entry are whatever the serial plan left in its buffers (random-weight
    activations); in-place launches (res4's Y over X) feed on their own output inside a loop -- values grow, bit patterns stay
    dense."""
    import collections
    from dafne_amd import _lib
    idx = device.index if getattr(device, "index", None) is not None else 0
    n, _, h, w = batch.shape
    plan = model.plan(n, h, w)
    model.detect_packed(batch)
    torch.cuda.synchronize()
    stream = _lib.current_stream()
    groups = collections.OrderedDict()
    for c in plan.calls:
        if getattr(c, "flops", 0) > 0:
            groups.setdefault(c.kernel_name(), []).append(c)

    def looped(fn, seconds):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record()
        torch.cuda.synchronize()
        per = max(a.elapsed_time(b) / 5, 1e-3)                       # ms per pass
        chunk = max(1, int(200.0 / per))                             # host syncs every ~0.2 s: a bounded launch queue
        npass = 0
        with TelemetrySampler(idx) as ts:
            t0 = time.perf_counter()
            a.record()
            while time.perf_counter() - t0 < seconds + 1.0:
                for _ in range(chunk):
                    fn()
                npass += chunk
                torch.cuda.synchronize()
            b.record()
            torch.cuda.synchronize()
        tel = ts.summary(skip_s=1.0)
        return a.elapsed_time(b) / npass, tel

    torch.cuda.synchronize()
    time.sleep(1.5)
    with TelemetrySampler(idx) as ts:
        time.sleep(1.2)
    idle = ts.summary(skip_s=0.0)
    idle_w = idle.get("socket_power_w", {}).get("mean")
    out = {"idle": idle, "seconds_per_entry": seconds, "kernels": {}, "unit": "joules per step of %d images" % n,
           "method": "each entry looped alone for >= %.1f s, hwmon socket power (first second dropped) x looped time" % seconds}

    def entry(ms, tel, flops, nbytes, launches=None):
        wv = tel.get("socket_power_w", {}).get("mean")
        e = {"ms_per_step_looped": ms, "socket_w": wv, "sclk_mhz": tel.get("sclk_mhz", {}).get("mean"), "samples": tel.get("samples")}
        if launches is not None:
            e["launches"] = launches
        if wv is not None:
            e["joules_per_step"] = wv * ms * 1e-3
            if flops:
                e["pj_per_flop"] = wv * ms * 1e-3 / flops * 1e12
            if nbytes:
                e["pj_per_algorithmic_byte"] = wv * ms * 1e-3 / nbytes * 1e12
            if idle_w is not None:
                e["joules_per_step_above_idle"] = (wv - idle_w) * ms * 1e-3
                if flops:
                    e["pj_per_flop_above_idle"] = (wv - idle_w) * ms * 1e-3 / flops * 1e12
        return e

    for name, calls in groups.items():
        def fn(calls=calls):
            for c in calls:
                c(stream)
        ms, tel = looped(fn, seconds)
        out["kernels"][name] = entry(ms, tel, sum(c.flops for c in calls), sum(c.bytes for c in calls), len(calls))
    model.detect_packed(batch)                                       # the plan's buffers back to a forward pass's values
    torch.cuda.synchronize()
    a = torch.randn(8192, 8192, device=device).to(torch.bfloat16)
    b = torch.randn(8192, 8192, device=device).to(torch.bfloat16)
    ms, tel = looped(lambda: a @ b, seconds)
    out["library_gemm_8192"] = entry(ms, tel, 2.0 * 8192 ** 3, 0)
    out["library_gemm_8192"]["tflops_looped"] = 2.0 * 8192 ** 3 / (ms * 1e-3) / 1e12
    del a, b
    nb = 1 << 30
    x = torch.empty(nb // 2, dtype=torch.bfloat16, device=device).normal_()
    y = torch.empty_like(x)
    ms, tel = looped(lambda: y.copy_(x), seconds)
    out["library_copy_1gib"] = entry(ms, tel, 0, 2.0 * nb)
    out["library_copy_1gib"]["gbps_looped"] = 2.0 * nb / (ms * 1e-3) / 1e9
    del x, y
    tot = sum(v.get("joules_per_step", 0.0) for v in out["kernels"].values())
    out["joules_per_step_sum_of_kernels"] = tot
    for v in out["kernels"].values():
        if tot and "joules_per_step" in v:
            v["share_of_joules"] = v["joules_per_step"] / tot
    return out


def nms_ms_per_image(device, m=10000, n_images=8, reps=5, kind="uniform", stats=None):
    """Rotated-NMS ms/img (A10+A11 only) on the synthetic candidate sets of SURVEY
    8(d) (tests/conftest.py nms_candidate_set: uniform / dense / skewed), seed 1234.  stats (dict): filled with the
    algorithmic work of the set and the path counters of the last call."""
    from dafne_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import nms_candidate_set
    L = _lib.load()
    rng = np.random.default_rng(1234)
    sets = [nms_candidate_set(kind, m, rng) for _ in range(n_images)]
    b = np.stack([x[0] for x in sets])
    s = np.stack([x[1] for x in sets])
    c = np.stack([x[2] for x in sets]).astype(np.int32)
    tb, ts, tc = (torch.from_numpy(a).to(device) for a in (b, s, c))
    tn = torch.full((n_images,), m, dtype=torch.int32, device=device)
    keep = torch.empty((n_images, m), dtype=torch.int64, device=device)
    nk = torch.zeros(n_images, dtype=torch.int32, device=device)
    nbytes = L.dafne_poly_nms_workspace_bytes(n_images, m)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def run():
        _lib.check(L.dafne_select_over_all_levels_hip(_lib.ptr(tb), _lib.ptr(ts), _lib.ptr(tc), _lib.ptr(tn), n_images,
                                                      m, 0.1, 1000, _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(ws), nbytes, 0,
                                                      _lib.current_stream()))
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps / n_images
    if stats is not None:
        cm = np.where(c == 5, 4, c)                                 # nms.py:77-79
        pairs_cls = sum(int(n) * (int(n) - 1) // 2 for img in cm for n in np.bincount(img))
        off = L.dafne_poly_nms_stats_offset(n_images, m, 0)
        st = ws[off:off + 16 * n_images].view(torch.int32).reshape(n_images, 4).sum(0).cpu().tolist()
        nblk = (m + 63) // 64
        stats.update({"pairs_all": n_images * m * (m - 1) // 2, "pairs_same_class": pairs_cls,
                      "algorithmic_bytes": n_images * (36 * m + 2 * 8 * m * nblk),        # rows in, mask write + read
                      "pairs_listed_after_hull_and_bound": int(sum(st)), "pairs_decided_fast_path": int(st[0] + st[1]),
                      "pairs_decided_reference_order_path": int(st[2] + st[3]),
                      "kept_mean": float(nk.float().mean().item()), "ms_per_call": ms * n_images})
    return ms


def _nms_one(args):
    from oracle import postprocess as opp
    b, s_, c = args
    return len(opp.batched_nms_poly(b, s_, c, 0.1, fast=True))


def cpu_baseline(cfg, sd, depth, budget_s=4.0, keep=None):
    """Oracle (port) on the host cores, bounded (~30 s of CPU work in total): the full path for single 1024^2 images
    (batch 1, ~budget_s) and for ONE batch of 8 ("batch 1 and batch 8"); the rotated NMS alone on the
    M = 10 000 set, one thread and all cores (8 images over a process pool: the C oracle is single-threaded)."""
    from oracle import model as om
    from oracle import postprocess as opp
    d = cfg.MODEL.DAFNE
    P = {k: v.float() for k, v in sd.items()}
    g = torch.Generator().manual_seed(0)
    # threads the host can really run: the scheduler affinity capped by the cgroup CPU quota (256 visible CPUs under a quota of
    # 16 on this project's MI355X boxes; a 128-thread pool there is throttled as a group and measures the throttle)
    from dafne_amd.utils.host import usable_cpus
    threads_before = torch.get_num_threads()
    cores = min(threads_before, usable_cpus())
    torch.set_num_threads(cores)

    def full_path(imgs, emulate_bf16=False, sink=None):
        with torch.no_grad():
            x, sizes = om.preprocess(imgs, cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
            f = om.backbone_forward(P, x, depth, emulate_bf16=emulate_bf16)
            lg, rg, ce, ct = om.head_forward(P, [f[k] for k in ("p3", "p4", "p5", "p6", "p7")], emulate_bf16=emulate_bf16)
        for i in range(len(imgs)):
            levels = [(lg[l][i].numpy(), rg[l][i].numpy(), ct[l][i].numpy()) for l in range(5)]
            det = opp.predict_proposals(levels, d.FPN_STRIDES, thresh=d.INFERENCE_TH_TEST, topk=d.PRE_NMS_TOPK_TEST,
                                        nms_thresh=d.NMS_TH, post_topk=d.POST_NMS_TOPK_TEST,
                                        thresh_with_ctr=d.THRESH_WITH_CTR, sort_corners=d.SORT_CORNERS, fast=True)
            det = opp.detector_postprocess(det, (1024, 1024), (1024, 1024), (1024, 1024))
            if sink is not None:
                sink.append({k: np.asarray(det[k]) for k in ("pred_corners", "scores", "pred_classes")})

    n_done, t_total = 0, 0.0
    while n_done < 1 or (t_total < budget_s and n_done < 8):
        img = torch.randint(0, 256, (3, 1024, 1024), generator=g, dtype=torch.uint8)
        t0 = time.perf_counter()
        full_path([img])
        t_total += time.perf_counter() - t0
        n_done += 1
    imgs8 = [torch.randint(0, 256, (3, 1024, 1024), generator=g, dtype=torch.uint8) for _ in range(8)]
    t0 = time.perf_counter()
    full_path(imgs8, sink=keep["fp32"] if keep is not None else None)
    t_b8 = time.perf_counter() - t0
    if keep is not None:         # the checker side of `equivalence_ap` (not timed): the same 8 images through the oracle's bf16 emulation
        keep["images"] = imgs8
        for k in range(0, 8, 2):
            full_path(imgs8[k:k + 2], emulate_bf16=True, sink=keep["bf16_emulation"])
    # rotated NMS alone on the CPU (SURVEY 8(d) "CPU baseline beside it"): the C oracle (hull pre-filter on) on the
    # M = 10 000 uniform candidate set of nms_ms_per_image -- one thread, then all cores
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import nms_candidate_set
    rng = np.random.default_rng(1234)
    sets = [nms_candidate_set("uniform", 10000, rng) for _ in range(8)]
    t0 = time.perf_counter()
    kq = opp.batched_nms_poly(*sets[0], 0.1, fast=True)
    nms_cpu_ms = 1e3 * (time.perf_counter() - t0)
    nms_all = {}
    try:
        import multiprocessing as mp
        nproc = max(1, min(8, usable_cpus()))
        with mp.get_context("spawn").Pool(nproc) as pool:      # spawn: never fork a process that holds a HIP context
            pool.map(_nms_one, sets[:nproc])                    # warm (imports, page-in)
            t0 = time.perf_counter()
            pool.map(_nms_one, sets)
            nms_all = {"rotated_nms_ms_per_img_m10000_all_cores": 1e3 * (time.perf_counter() - t0) / len(sets),
                       "rotated_nms_all_cores_processes": nproc}
    except Exception as e:      # noqa: BLE001
        nms_all = {"rotated_nms_all_cores_error": "%s: %s" % (type(e).__name__, e)}
    torch.set_num_threads(threads_before)
    out = {"rotated_nms_ms_per_img_m10000": nms_cpu_ms, "rotated_nms_cores": 1, "rotated_nms_kept": int(len(kq)),
           "value": n_done / t_total, "unit": "images/sec", "cores": cores, "cpus_visible": os.cpu_count(), "kind": "port",
           "batch8_images_per_sec": 8 / t_b8,
           "sample": "%d x 1024x1024 image(s) at batch 1 (%.1f s) + one batch of 8 (%.1f s), R%d-FPN fp32 torch-CPU + "
                     "C/numpy post-process" % (n_done, t_total, t_b8, depth)}
    out.update(nms_all)
    return out


def equivalence_side_metric(model, keep, device):
    """"Detections equivalent to the reference" as a number (BASELINE.json north_star): VOC07 AP (the evaluator's own scoring,
    dafne/evaluation/voc_eval.py:41-224, polygon IoU on the device) of the ENGINE's detections for the cpu_baseline leg's batch of
    8 images against the FP32 ORACLE's detections of the same images as ground truth, at IoU 0.5 and 0.75, per class and mean --
    next to the same AP for the oracle's own bf16 emulation (what bf16 arithmetic costs whoever implements it).  The oracle is the
    checker here, outside every timed region."""
    from dafne_amd.evaluation.equivalence import equivalence_ap
    imgs = keep["images"]
    outs = model([{"image": im.to(device), "height": 1024, "width": 1024} for im in imgs])
    eng = []
    for o in outs:
        inst = o["instances"]
        eng.append({"pred_corners": inst.pred_corners.float().cpu().numpy(), "scores": inst.scores.float().cpu().numpy(),
                    "pred_classes": inst.pred_classes.cpu().numpy()})
    res = {"engine_vs_fp32_oracle": equivalence_ap(eng, keep["fp32"]),
           "bf16_emulation_vs_fp32_oracle": equivalence_ap(keep["bf16_emulation"], keep["fp32"]),
           "ground_truth": "the fp32 oracle's own detections (oracle/model.py + oracle/postprocess.py) of 8 seeded 1024x1024 images, "
                           "random-init weights (bench.build_model)", "entry": "model(batched_inputs)"}
    for thr in ("iou_0.50", "iou_0.75"):
        res["engine_minus_emulation_" + thr] = (res["engine_vs_fp32_oracle"][thr]["weighted_mean"]
                                                - res["bf16_emulation_vs_fp32_oracle"][thr]["weighted_mean"])
    return res


def headline(args, world, dt, det_mean, region_times=None):
    """The contract's JSON line (without the extras): value = images of ALL ranks / the slowest rank's time.  region_times:
    every timed region's seconds (dt is the median one) -> value_min / value_max / regions on the line."""
    out = {
        "metric": "images/sec on 1024x1024 DOTA tiles, R%d-FPN" % args.depth,
        "value": args.batch * world * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "DOTA-1.0 %dx%d R%d-FPN bf16, batch %d per GPU (per-GPU shard of configs[2]; "
                               "the model BASELINE.json's metric names), uint8 tiles resident in HBM -> detections"
                               % (args.size, args.size, args.depth, args.batch),
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "classes": 15,
                   "parallelism": "dp%d (independent images, RCCL gather of detections)" % world,
                   "detections_per_image_mean": det_mean},
    }
    if region_times:
        vals = [args.batch * world * args.steps / t for t in region_times]
        out["value_min"], out["value_max"] = min(vals), max(vals)
        out["regions"] = {"count": len(vals), "steps_each": args.steps, "images_per_sec": vals,
                          "value_is": "the median region (ms_per_step x steps = that one region)"}
    return out


_TELEMETRY_DIR = {}


def _telemetry_dir(index):
    """sysfs directory (/sys/class/drm/cardN/device) of HIP device `index`, matched by PCI address: a box shows every GPU of
    the node in sysfs (cards 0, 8, .., 56) while the process sees one of them as device 0."""
    import glob
    if index not in _TELEMETRY_DIR:
        found = None
        try:
            pr = torch.cuda.get_device_properties(index)
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
                if os.path.basename(os.path.realpath(c)) == addr and glob.glob(os.path.join(c, "hwmon", "hwmon*")):
                    found = c
        except Exception:      # noqa: BLE001
            found = None
        _TELEMETRY_DIR[index] = found
    return _TELEMETRY_DIR[index]


def device_telemetry(index=0):
    """Shader clock (MHz), socket power (W), temperature of HIP device `index` as the driver reports them right now (amdgpu
    hwmon: no subprocess, microseconds per read; the same figures `rocm-smi --showpower --showclocks` prints): sampled WHILE
    the timed loop and the library GEMM run, so that a 0.416 box can be told from a 0.445 box.  Fields that
    cannot be read are absent."""
    import glob
    out = {}
    try:
        base = _telemetry_dir(index)
        if base is None:
            return {"error": "no sysfs card matches the device's PCI address"}
        hw = sorted(glob.glob(os.path.join(base, "hwmon", "hwmon*")))
        for name, key, scale in (("power1_input", "socket_power_w", 1e-6), ("power1_average", "socket_power_w", 1e-6),
                                 ("power1_cap", "power_cap_w", 1e-6), ("freq1_input", "sclk_mhz", 1e-6),
                                 ("temp2_input", "temp_c", 1e-3), ("temp1_input", "temp_c", 1e-3)):
            f = os.path.join(hw[0], name)
            if key not in out and os.path.exists(f):
                try:
                    out[key] = float(open(f).read().strip()) * scale
                except (OSError, ValueError):
                    pass
    except Exception as e:      # noqa: BLE001  (telemetry must never cost the line)
        out["error"] = repr(e)[:120]
    return out


class TelemetrySampler:
    """Background thread: device_telemetry() every `period` seconds while a block runs -> mean / max clock and power."""

    def __init__(self, index=0, period=0.1):
        import threading
        self.index, self.period, self.samples = index, period, []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._loop, daemon=True)

    def _loop(self):
        while not self._stop.is_set():
            self.samples.append(device_telemetry(self.index))
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._th.join(timeout=2)

    def summary(self, skip_s=1.0):
        """The driver's power figure is an average over about a second: the first `skip_s` of samples are dropped when the
        block ran long enough to leave twice as many."""
        k = int(skip_s / self.period)
        smp = self.samples[k:] if len(self.samples) >= 3 * k else self.samples
        out = {"samples": len(smp), "period_s": self.period}
        for key in ("sclk_mhz", "socket_power_w", "temp_c"):
            v = [s[key] for s in smp if key in s]
            if v:
                out[key] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v)}
        return out


def library_gemm_reference(device, batch):
    """What the vendor GEMM (hipBLASLt through torch.matmul) sustains on THIS box, measured live: (a) the dominant kernel's
    own GEMM shape with the im2col already done (M = batch x 21 824 tower pixels of a 1024^2 image, N = 256, K = 2304),
    (b) a large square bf16 GEMM, both with random operands.  The matrix pipe's clock follows its power draw (the same
    8192^3 GEMM runs 30 % faster on all-zero operands), so (b) is the practical ceiling `roofline.frac` is to be read
    against; `peak` stays the guide's nominal 2.5 PFLOP/s."""
    def run(M, N, K):
        a = torch.randn(M, K, device=device).to(torch.bfloat16)
        b = torch.randn(K, N, device=device).to(torch.bfloat16)
        for _ in range(3):
            a @ b
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            a @ b
        e.record()
        torch.cuda.synchronize()
        return 2.0 * M * N * K * 10 / (s.elapsed_time(e) * 1e-3) / 1e12
    M = batch * 21824
    out = {"same_gemm_shape_tflops": run(M, 256, 2304), "same_gemm_shape": [M, 256, 2304]}
    # clock / power WHILE the square GEMM runs: a ~3-s burst of it (the driver's power reading is a ~1-s average)
    a = torch.randn(8192, 8192, device=device).to(torch.bfloat16)
    b = torch.randn(8192, 8192, device=device).to(torch.bfloat16)
    idx = device.index if getattr(device, "index", None) is not None else 0
    with TelemetrySampler(idx) as ts:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            for _ in range(50):
                a @ b
            torch.cuda.synchronize()
    del a, b
    out.update({"square_8192_tflops": run(8192, 8192, 8192), "operands": "randn bf16", "telemetry_during_square_gemm": ts.summary(),
                "note": "torch.matmul (hipBLASLt), im2col not included; measured in this run"})
    return out


def library_hbm_reference(device, mb=1024):
    """What plain streaming kernels (torch's elementwise launches) sustain on THIS box over a buffer far beyond L2 + MALL, measured
    live: a pure write (fill), a copy (read one buffer, write another) and an in-place update (read and write the same lines).
    The convolution blocks' traffic is a read + write mix, so `copy_total` / `inplace_total` -- not the 8 TB/s nominal peak -- is the
    practical ceiling `roofline_hbm.frac` and the res2 / res3 rows of `kernels_isolated` are to be read against."""
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=device).normal_()
    y = torch.empty_like(x)
    def rate(fn, nbytes, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return nbytes * reps / (s.elapsed_time(e) * 1e-3) / 1e9
    nb = 2 * n
    out = {"buffer_mb": mb, "unit": "GB/s", "fill_write_only": rate(lambda: y.fill_(1.5), nb), "copy_total": rate(lambda: y.copy_(x), 2 * nb),
           "inplace_total": rate(lambda: torch.relu_(y), 2 * nb),
           "note": "torch elementwise kernels, read + write bytes counted; measured in this run"}
    del x, y
    return out


def distributed_record(args, world, distributed, per_rank_s, gathered):
    """What the collective DELIVERED, so that an N > 1 line proves itself: `n_gpus` in the headline is the launcher's
    WORLD_SIZE; this records the process group's own size and backend, the number of images the last step's detection
    gather landed on rank 0 (must be per_gpu_batch x world) with their detection count, and every rank's own timed-region
    time (value is computed from the MAX).  gathered: (rows_all, counts_all) on rank 0, None elsewhere / at N = 1."""
    rec = {"world_size_from_env": world, "process_group_size": dist.get_world_size() if distributed else 1,
           "backend": dist.get_backend() if distributed else None,
           "per_rank_ms_per_step": {"min": 1e3 * min(per_rank_s) / args.steps, "max": 1e3 * max(per_rank_s) / args.steps,
                                    "ranks": len(per_rank_s)} if per_rank_s else None}
    if distributed and gathered is not None:
        rows_all, counts_all = gathered
        rec["gathered_images"] = int(rows_all.shape[0])
        rec["gathered_detections"] = int(counts_all.sum().item())
        rec["expected_images"] = args.batch * world
    return rec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--depth", type=int, default=101, choices=[50, 101])
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--splits", type=int, default=3,
                    help="dense part runs as this many sub-batches on concurrent HIP streams")
    ap.add_argument("--mode", choices=["pipelined", "serial"], default="pipelined",
                    help="pipelined (default, the timed configuration): sub-batches on concurrent streams, post-process of step i "
                         "under the convolutions of step i+1.  serial: the whole batch, every kernel alone on ONE stream -- the "
                         "layout the isolated roofline is quoted on (profiles/r03_kernel_stats_isolated.txt)")
    ap.add_argument("--no-defer", action="store_true",
                    help="pipelined mode: every step's decode + NMS enqueued right behind its own convolutions (round 3's form) instead "
                         "of at the next step's head towers (detect_packed(defer=True)); for A/B runs")
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of --steps steps each, back to back; value = the median region, value_min / value_max beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline profile pass, R50 and NMS side metrics")
    ap.add_argument("--no-energy", action="store_true", help="skip the per-kernel energy ledger (about 45 s of looped kernels)")
    ap.add_argument("--energy-seconds", type=float, default=2.0, help="telemetry window per ledger entry (after a 1-s lead-in)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawned(local_rank, args, port, target):
    """One worker of a self-launched N-GPU run: the environment torch.distributed.run would have set."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(local_rank), LOCAL_RANK=str(local_rank),
                      WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    target(args)


def launch(args, target=None):
    """`python bench.py --gpus N` starts itself: with no WORLD_SIZE in the environment and N > 1 it spawns one worker
    per GPU (rank = local rank = GPU index, 127.0.0.1 rendezvous on a free port), the counterpart of the reference's
    `launch(main, num_gpus, ...)` (tools/plain_train_net.py:660-671).  Under torch.distributed.run (WORLD_SIZE set) the
    process IS one worker and runs `target` directly.  target: picklable callable(args); default `run`."""
    target = target or run
    if "WORLD_SIZE" in os.environ or args.gpus <= 1:
        return target(args)
    import torch.multiprocessing as mp
    mp.spawn(_spawned, args=(args, _free_port(), target), nprocs=args.gpus, join=True)
    return None


def run(args, make_step=None, backend="nccl", device_kind="cuda"):
    """One worker.  make_step(args, rank, world, device) -> (step_fn, finish_fn) replaces the detector (tests: a stub on
    CPU over gloo); finish_fn(out) may add keys to the JSON line on rank 0.  Returns the JSON dict on rank 0."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # host threads: never more than this worker's share of the CPUs the process group may use (affinity, cgroup quota): a
    # pool sized by the visible CPU count is throttled as a group and stalls the HIP runtime's threads with it
    from dafne_amd.utils.host import usable_cpus
    share = max(1, usable_cpus() // max(world, 1))
    if torch.get_num_threads() > share:
        torch.set_num_threads(share)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launched world size is %d (torch.distributed.run --nproc-per-node "
                         "must equal --gpus)" % (args.gpus, world))
    if device_kind == "cuda":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    if distributed:
        if device_kind == "cuda":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        out = _run_worker(args, make_step, rank, world, distributed, device)
    except BaseException:
        raise                                   # no barrier on the error path: the launcher must see the failure, not a hang
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return out if rank == 0 else None


def _run_worker(args, make_step, rank, world, distributed, device):
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if make_step is not None:
        step, finish = make_step(args, rank, world, device)
        per_rank = []
        dt, region_times = time_regions(step, args.steps, args.warmup, distributed, getattr(args, "regions", 1), device, per_rank=per_rank)
        out = headline(args, world, dt, None, region_times)
        last = step()
        out["distributed"] = distributed_record(args, world, distributed, per_rank, last)
        if rank == 0 and finish is not None:
            finish(out)
        return out

    from dafne_amd.evaluation.gather import gather_detections
    cfg, model, sd = build_model(args.depth, device, seed=0)
    g = torch.Generator().manual_seed(rank)
    batch = torch.randint(0, 256, (args.batch, 3, args.size, args.size), generator=g, dtype=torch.uint8).to(device)

    gathered = {"out": None}

    def step():
        # decode + NMS + gather (+ the RCCL detection gather) ride a side stream and overlap the
        # next step's convolutions; the dense part runs as --splits sub-batches on concurrent
        # streams; every step still runs the whole path on its own batch.
        if args.mode == "serial":
            rows, counts = model.detect_packed(batch)
            if distributed:
                gathered["out"] = gather_detections(rows, counts, dst=0)
            return rows, counts
        # the loop's form of the pipelined step: this call's convolutions, then the PREVIOUS call's decode + NMS, which start on
        # the side stream when this call's sub-batches reach their head towers (detect_packed(defer=True)); flush() completes the
        # last step.  Every step's whole path runs once; time_steps flushes before and INSIDE the end of the timed region.
        if args.no_defer:
            return after(model.detect_packed(batch, pipelined=True, splits=args.splits))
        return after(model.detect_packed(batch, pipelined=True, splits=args.splits, defer=True))

    def after(res):
        if res is not None and distributed:
            with torch.cuda.stream(model.side_stream):
                gathered["out"] = gather_detections(res[0], res[1], dst=0)
        return res

    def flush():
        return after(model.flush_deferred()) if (args.mode != "serial" and not args.no_defer) else None

    for _ in range(PRIME):                 # untimed, before the W warm-up steps: launch plans, packed weights, HIP graphs
        step()
    per_rank = []
    tel0 = device_telemetry(local_rank)
    dt, region_times = time_regions(step, args.steps, args.warmup, distributed, args.regions, device, per_rank=per_rank, flush_fn=flush)
    tel1 = device_telemetry(local_rank)
    last = step()
    rows, counts = flush() if (args.mode != "serial" and not args.no_defer) else (last if args.mode != "serial" else step())
    torch.cuda.synchronize()
    out = headline(args, world, dt, float(counts.float().mean().item()), region_times)
    out["config"]["mode"] = args.mode
    # the path that was just timed against the immediate form of the same step, OUTSIDE the timed region (the parity tests pin
    # both to the oracle at this size: tests/test_gpu_headline.py::test_headline_timed_layout_vs_oracle)
    if args.mode != "serial":
        ri, ci = model.detect_packed(batch, pipelined=True, splits=args.splits)
        torch.cuda.synchronize()
        out["timed_path_equals_immediate"] = bool(torch.equal(ci, counts) and all(
            torch.equal(rows[i, :int(ci[i])], ri[i, :int(ci[i])]) for i in range(args.batch)))
    out["device_telemetry"] = {"before_timed_regions": tel0, "after_timed_regions": tel1}
    out["distributed"] = distributed_record(args, world, distributed, per_rank, gathered["out"])
    # extras only at N=1: at N>1 the other ranks would sit in the final barrier while rank 0 measures side metrics
    if rank == 0 and world == 1 and not args.no_extras:
      try:
        if args.mode == "serial":
            model.detect_packed(batch, pipelined=True, splits=args.splits)     # builds the timed layout's plans for the profile
            torch.cuda.synchronize()
        # clock / socket power WHILE the timed step runs: ~3 s of the same loop (the regions above are 0.1-0.3 s each, shorter than
        # the driver's power-averaging window), so that a fast box can be told from a kernel change
        if args.mode != "serial":
            with TelemetrySampler(local_rank) as ts_loop:
                t0 = time.perf_counter()
                nloop = 0
                while time.perf_counter() - t0 < 3.0:
                    for _ in range(25):
                        step()
                    nloop += 25
                    torch.cuda.synchronize()
                flush()
                torch.cuda.synchronize()
                dt_loop = time.perf_counter() - t0
            out["device_telemetry"]["during_timed_loop"] = dict(ts_loop.summary(), images_per_sec=args.batch * nloop / dt_loop, steps=nloop)
        prof = conv_kernel_profile(model, batch, args.splits)
        iso = conv_kernel_profile_isolated(model, batch)
        # dominant kernel = the one with the most GPU time per step when every launch has the GPU to itself (whole batch,
        # one stream): the layout rocprofv3 profiles in profiles/r02_kernel_stats_isolated.txt (`--mode serial`)
        domname = max(iso, key=lambda k: iso[k]["ms_per_step"])
        di, ds = iso[domname], prof.get(domname)
        kname = domname.replace("conv_igemm", "conv_igemm_kernel") if "igemm" in domname else domname + "_kernel"
        out["roofline"] = {"bound": "mfma", "achieved": di["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": di["frac"], "traffic": None, "kernel": kname,
                           "layout": "isolated: whole batch of %d on one stream, launches do not overlap (bench.py --mode serial)" % args.batch,
                           "launches_per_step": di["launches"], "avg_launch_us": 1e3 * di["ms_per_step"] / max(di["launches"], 1),
                           "algorithmic_gflop_per_step": di["flops"] / 1e9, "algorithmic_bytes_per_step": di["bytes"],
                           "share_of_gpu_time_per_step": di["ms_per_step"] / sum(v["ms_per_step"] for v in iso.values())}
        # HBM bytes per launch of that kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs of `bench.py --mode serial`; counters cannot be read from inside the process).
        # null when no PMC summary is present for the kernel.
        def pmc_traffic(fname, kernel=None):
            kernel = kernel or kname
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", fname)))
                cands = [v for k, v in pmc["kernels"].items() if k == kernel or k.startswith(kernel + "<")]
                if cands:       # the kernel's template instantiations, weighted by their launches
                    nl = sum(c.get("launches", 1) for c in cands)
                    return 1e6 * sum(c["hbm_mb_per_launch"] * c.get("launches", 1) for c in cands) / nl, pmc["source"]
            except (OSError, KeyError, ValueError):
                pass
            return None, None
        try:
            out["roofline"]["library_gemm"] = library_gemm_reference(device, args.batch)
            out["roofline"]["frac_of_library_square_gemm"] = di["tflops"] / out["roofline"]["library_gemm"]["square_8192_tflops"]
        except Exception as ex:                      # a reference figure must never cost the line
            out["roofline"]["library_gemm"] = {"error": repr(ex)[:200]}
        tr, src = pmc_traffic("pmc_traffic_isolated.json")
        if tr is not None:
            out["roofline"]["traffic"] = tr
            out["roofline"]["traffic_unit"] = "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), " + src
        if ds:
            out["roofline_timed"] = {
                "bound": "mfma", "kernel": kname, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                "layout": "timed region: %d sub-batches on concurrent streams (a launch's duration includes sharing the GPU)" % args.splits,
                "achieved": ds["tflops"], "frac": ds["tflops"] / PEAK_BF16_TFLOPS, "launches_per_step": ds["launches"],
                "avg_launch_us": ds["avg_launch_us"], "concurrent_streams": args.splits,
                "achieved_over_union_of_launches": ds["flops"] / (ds["union_ms"] * 1e-3) / 1e12 if ds.get("union_ms", 0) > 0 else None,
                "note": "the timed region's layout: launches of %d sub-batches share the GPU on concurrent streams, so a "
                        "launch's duration includes that sharing (as rocprofv3 reports it for the default command)" % args.splits}
            tr2, src2 = pmc_traffic("pmc_traffic.json")
            if tr2 is not None:
                out["roofline_timed"]["traffic"] = tr2
                out["roofline_timed"]["traffic_unit"] = "bytes per launch at sub-batch size, " + src2
        out["kernels"] = {k: {"tflops": v["tflops"], "ms_per_step": v["ms"], "launches": v["launches"],
                              "hbm_gbps_algorithmic": v["bytes"] / (v["ms"] * 1e-3) / 1e9} for k, v in prof.items()}
        # the HBM-bound kernel family next to the MFMA-bound dominant one: persistent weight-stationary 1x1 layers
        if "conv_ws" in iso:
            wi, wsk = iso["conv_ws"], prof.get("conv_ws")
            # `achieved` = bytes the HBM counters saw per launch (profiles/pmc_traffic_isolated.json: 2 x FETCH_SIZE + WRITE_SIZE)
            # / the launch time measured here.  The layer's input was written by the launch before it and is largely L2 / MALL
            # resident, so the ALGORITHMIC bytes / time (kept as a second field) can exceed the HBM peak -- round 4 printed
            # that as frac 1.07 ("weak 8"); it is a cache-inclusive rate, not HBM traffic.
            avg_us = 1e3 * wi["ms_per_step"] / max(wi["launches"], 1)
            out["roofline_hbm"] = {"bound": "hbm", "kernel": "conv_ws_kernel", "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                   "layout": "isolated (as roofline)", "launches_per_step": wi["launches"], "avg_launch_us": avg_us,
                                   "algorithmic_bytes_per_step": wi["bytes"],
                                   "algorithmic_gbps_cache_inclusive": wi["hbm_gbps_algorithmic"]}
            trw, srcw = pmc_traffic("pmc_traffic_isolated.json", "conv_ws_kernel")
            if trw is not None:
                out["roofline_hbm"].update({"achieved": trw / (avg_us * 1e-6) / 1e9, "frac": trw / (avg_us * 1e-6) / 1e9 / PEAK_HBM_GBPS,
                                            "traffic": trw, "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), " + srcw})
            else:
                out["roofline_hbm"].update({"achieved": None, "frac": None, "traffic": None})
            try:
                out["roofline_hbm"]["library_rates"] = library_hbm_reference(device)
                if out["roofline_hbm"].get("achieved"):
                    out["roofline_hbm"]["frac_of_library_copy"] = out["roofline_hbm"]["achieved"] / out["roofline_hbm"]["library_rates"]["copy_total"]
            except Exception as e:       # a side measurement must never cost the headline line
                out["roofline_hbm"]["library_rates"] = {"error": repr(e)}
            if wsk:
                out["roofline_hbm"]["timed_layout"] = {"algorithmic_gbps_cache_inclusive": wsk["bytes"] / (wsk["ms"] * 1e-3) / 1e9,
                                                        "launches_per_step": wsk["launches"], "avg_launch_us": wsk["avg_launch_us"]}
        out["kernels_isolated"] = {k: {kk: vv for kk, vv in v.items() if kk not in ("flops", "bytes")} for k, v in iso.items()}
        if not args.no_energy:
            try:
                led = energy_ledger(model, batch, device, seconds=args.energy_seconds)
                for k, v in led["kernels"].items():
                    if k in out["kernels_isolated"]:
                        out["kernels_isolated"][k].update({kk: v.get(kk) for kk in ("socket_w", "sclk_mhz", "joules_per_step", "pj_per_flop",
                                                                                     "ms_per_step_looped", "share_of_joules")})
                out["energy_ledger"] = led
                wl = out["device_telemetry"].get("during_timed_loop", {}).get("socket_power_w", {}).get("mean")
                if wl:
                    out["energy_ledger"]["timed_loop"] = {"socket_w": wl, "joules_per_step": wl * dt / args.steps,
                                                           "pj_per_flop": wl * dt / args.steps / sum(v["flops"] for v in prof.values()) * 1e12}
            except Exception as ex:                  # a side measurement must never cost the headline line
                out["energy_ledger"] = {"error": repr(ex)[:300]}
        tot_flops = sum(v["flops"] for v in prof.values())
        out["model_tflops_end_to_end"] = tot_flops / (dt / args.steps) / 1e12
        out["mfma_frac_end_to_end"] = out["model_tflops_end_to_end"] / PEAK_BF16_TFLOPS
        # rotated NMS (A10 + A11), SURVEY 8(d) candidate sets: sizes 500 / 2000 / 10 000 per image (batch of 8) and the
        # 27 000-row TTA merge (one image); uniform, DENSE (70 % of the boxes in one 256^2 window) and the DOTA-1.5
        # SKEWED class histogram (60 % in {4,5,6})
        nms_stats = {}
        by_m = {}
        for kind in ("uniform", "dense", "skewed"):
            for m, n_img in ((500, 8), (2000, 8), (10000, 8), (27000, 1)):
                if kind != "uniform" and m < 10000:
                    continue
                st = {}
                key = "%dx%d" % (m, n_img) + ("" if kind == "uniform" else "_" + kind)
                by_m[key] = nms_ms_per_image(device, m=m, n_images=n_img, kind=kind, stats=st)
                nms_stats[key] = st
        out["rotated_nms_ms_per_img"] = by_m["10000x8"]
        out["rotated_nms_ms_per_img_by_m"] = by_m
        # NMS roofline on the hardest full-size set: the unit of work is a pair of same-class boxes (the class offsets of
        # nms.py:81-83 make every other pair a certain "no"); bytes = rows in + suppression mask written and read
        hard = max((k for k in by_m if k.startswith("10000x8")), key=lambda k: by_m[k])
        hs = nms_stats[hard]
        sec = hs["ms_per_call"] * 1e-3
        out["roofline_nms"] = {"set": hard, "bound": "latency / fp64 vector ALU (not MFMA, not HBM)", "ms_per_img": by_m[hard],
                               "same_class_pairs_per_sec": hs["pairs_same_class"] / sec, "all_pairs_per_sec": hs["pairs_all"] / sec,
                               "algorithmic_bytes_per_call": hs["algorithmic_bytes"],
                               "achieved": hs["algorithmic_bytes"] / sec / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                               "frac": hs["algorithmic_bytes"] / sec / 1e9 / PEAK_HBM_GBPS, "per_set": nms_stats}
        if world == 1 and args.depth == 101:
            # side metrics on the other BASELINE configs: each one is guarded -- a failure there must never cost the
            # headline line
            def side_r50():
                cfg50, m50, _ = build_model(50, device, seed=0)
                f50 = lambda: m50.detect_packed(batch, pipelined=True, splits=args.splits, defer=True)     # the timed layout's loop form
                dt50 = min(time_steps(f50, max(args.steps // 2, 3), PRIME, False, flush_fn=m50.flush_deferred),
                           time_steps(f50, max(args.steps // 2, 3), 1, False, flush_fn=m50.flush_deferred))     # side metric: best of two
                out["configs1_r50_b8"] = {"images_per_sec": args.batch * max(args.steps // 2, 3) / dt50,
                                          "workload": "DOTA-1.0 1024x1024 R50-FPN bf16, batch 8, 1 GPU"}
                del m50

            def side_fp8():
                # configs[4]: R101-FPN, 2 classes, fp8 (e4m3) weights, 16 images per GPU -- reported beside the bf16 metric,
                # never as `value` (reduced precision); 41 of the 3x3 layers run the fp8 MFMA kernel (calibrate_fp8 on the batch, before the timed
                # region, pins the activation scales)
                cfg8, m8, _ = build_model(101, device, seed=0, cfgname="ucas_aod_r101_fp8.yaml", cls_prior=-1.5)
                b16 = torch.cat([batch, batch.flip(0)])[:16]
                n8 = max(args.steps // 4, 3)
                m8.calibrate_fp8(b16)                  # explicit: the activation scales are part of the model
                dt8 = time_steps(lambda: m8.detect_packed(b16, pipelined=True, splits=args.splits, defer=True), n8, PRIME, False,
                                 flush_fn=m8.flush_deferred)
                r8, c8 = m8.detect_packed(b16, pipelined=True, splits=args.splits)
                torch.cuda.synchronize()
                out["configs4_fp8w_r101_b16"] = {"images_per_sec": b16.shape[0] * n8 / dt8, "detections_per_image_mean": float(c8.float().mean().item()), "dtype": "fp8 e4m3 weights; 41 3x3 layers (res4/res5, FPN outputs, head towers) on fp8 MFMA with calibrated e4m3 activations, the rest bf16",
                                                 "workload": "UCAS-AOD head (2 classes) 1024x1024 R101-FPN, batch 16, 1 GPU"}
                m8b = build_model(101, device, seed=0, cfgname="ucas_aod_r101.yaml", cls_prior=-1.5)[1]
                dt8b = time_steps(lambda: m8b.detect_packed(b16, pipelined=True, splits=args.splits, defer=True), n8, PRIME, False,
                                  flush_fn=m8b.flush_deferred)
                out["configs4_fp8w_r101_b16"]["bf16_same_workload_images_per_sec"] = b16.shape[0] * n8 / dt8b
                del m8, m8b

            def side_tta():
                # configs[3]: DOTA-1.5 R101-FPN with multi-scale + flip TTA (9 sizes x 3 views = 27 forward passes per
                # image, one merged rotated NMS over <= 27 000 boxes).  The class prior is raised so that every view fills
                # its 1000 post-NMS slots (this config thresholds the raw class score; the bench weights keep the
                # reference's -4.6 prior and would yield no candidates), i.e. the merge sees its worst case.
                from dafne_amd.modeling.tta import OneStageRCNNWithTTA
                cfg15, m15, sd15 = build_model(101, device, seed=0, cfgname="dota-1.5_r101.yaml", cls_prior=-1.5)
                tta = OneStageRCNNWithTTA(cfg15, m15)
                # the wrapper's own call shape: tta(batched_inputs).  Six images per call: the same-size views of up to three
                # images share a detector call (chunks of 9 views), and a group's convolutions run under the previous group's
                # read-back + inverse transforms + merged NMS; per image the result equals the one-image call's
                # (tests/test_gpu_model.py::test_tta_groups_of_images_match_the_per_image_calls)
                n_img = min(6, args.batch)
                ins = [{"image": batch[k], "height": args.size, "width": args.size} for k in range(n_img)]
                for _ in range(PRIME):       # every view shape's two plan sets run once eagerly, then their HIP graphs are captured
                    tta(ins[:3])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nd = [len(o["instances"]) for o in tta(ins)]
                torch.cuda.synchronize()
                dt_group = (time.perf_counter() - t0) / n_img
                for _ in range(PRIME):       # the one-image call's chunks of 3 views are plans of their own
                    tta([ins[0]])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(min(3, n_img)):
                    tta([ins[k]])
                torch.cuda.synchronize()
                dt_one = (time.perf_counter() - t0) / min(3, n_img)
                out["configs3_tta_r101"] = {"ms_per_image": 1e3 * dt_group, "ms_per_image_one_image_per_call": 1e3 * dt_one,
                                            "images_per_call": n_img, "views_per_image": 27, "detections_per_image": nd,
                                            "workload": "DOTA-1.5 1024x1024 R101-FPN bf16, TTA sizes %s x {none, hflip, vflip}, merged NMS"
                                                        % (list(cfg15.TEST.AUG.MIN_SIZES),)}
                del tta, m15

            def side_forward():
                nthreads = torch.get_num_threads()      # forward_streamed caps torch's intra-op pool: restore it for the CPU baseline
                # the SAME model through the reference-shaped entry point: inference_on_dataset(model, loader, evaluator)
                # (tools/plain_train_net.py:316-336) -> OneStageDetector.forward_streamed, list[dict] in, list[{"instances"}] out,
                # Instances built on the host for every image.  Device-resident tiles (as the headline) and host tiles (pageable
                # CPU tensors as a data loader yields them: pinned staging + upload under the previous batch).
                from dafne_amd.evaluation.inference import DafneEvaluator, inference_on_dataset
                nb = max(args.steps // 2, 24)         # long enough that the loop's fill / drain (one step each) is < 5 %
                res = {}
                for where in ("device", "host"):
                    imgs = batch if where == "device" else batch.cpu()
                    loader = [[{"image": imgs[k], "height": args.size, "width": args.size, "image_id": j * args.batch + k}
                               for k in range(args.batch)] for j in range(nb)]
                    ev = DafneEvaluator("synthetic", cfg, distributed=False)
                    inference_on_dataset(model, loader[:PRIME], ev)                 # warm-up: plans, graphs, pinned buffers
                    st = {}
                    r = inference_on_dataset(model, loader, ev, st)
                    assert r["num_images"] == nb * args.batch
                    res[where] = st["images_per_sec"]
                torch.set_num_threads(nthreads)
                # the SYNCHRONOUS call detectron2's own loop makes -- outputs = model(batched_inputs), Instances on the host side
                # of every call (tools/plain_train_net.py:331) -- next to the immediate (not deferred) form of the timed step,
                # which is what that call runs underneath since round 5
                loader = [[{"image": batch[k], "height": args.size, "width": args.size} for k in range(args.batch)] for j in range(nb)]
                for b in loader[:PRIME]:
                    model(b)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for b in loader:
                    model(b)
                torch.cuda.synchronize()
                sync_ips = nb * args.batch / (time.perf_counter() - t0)
                f_imm = lambda: model.detect_packed(batch, pipelined=True, splits=args.splits)
                imm_ips = args.batch * nb / time_steps(f_imm, nb, PRIME, False)
                out["through_forward_sync"] = {"images_per_sec": sync_ips, "no_defer_images_per_sec": imm_ips,
                                               "fraction_of_no_defer": sync_ips / imm_ips, "fraction_of_value": sync_ips / out["value"],
                                               "batches": nb, "entry": "model(batched_inputs): OneStageDetector.forward, one host sync per call"}
                out["through_forward_images_per_sec"] = res["device"]
                out["through_forward"] = {"images_per_sec_device_tiles": res["device"], "images_per_sec_host_tiles": res["host"],
                                          "fraction_of_value": res["device"] / out["value"], "batches": nb,
                                          "entry": "evaluation.inference.inference_on_dataset -> OneStageDetector.forward_streamed "
                                                   "(ENGINE.PIPELINE_SPLITS %d), DafneEvaluator.process per batch" % cfg.ENGINE.PIPELINE_SPLITS}

            def side_files():
                # the same model fed from IMAGE FILES through the reference-shaped loader: build_test_loader(cfg, dir)
                # (tools/plain_train_net.py:280-313) -> inference_on_dataset.  PNG tiles written to a temporary directory first
                # (smooth content + noise: 2 MB each); PIL decode on host threads is the bound here, not the GPU -- reported with
                # the CPU count it ran on, never as `value`.
                import shutil
                import tempfile
                from dafne_amd.data import build_test_loader
                from dafne_amd.evaluation.inference import DafneEvaluator, inference_on_dataset
                from dafne_amd.utils.host import usable_cpus
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import eval_net
                nthreads = torch.get_num_threads()
                root = tempfile.mkdtemp(prefix="dafne_tiles_")
                try:
                    n_files = 12 * args.batch
                    eval_net.write_synthetic_tiles(root, n_files, args.size, args.size, 0)
                    mk = lambda: build_test_loader(cfg, root, batch_size=args.batch, device=device, num_workers=64, prefetch_batches=4)
                    ev = DafneEvaluator("tiles", cfg, distributed=False)
                    inference_on_dataset(model, list(mk())[:2], ev)               # warm-up (plans exist already; pinned staging)
                    st = {}
                    ld = mk()
                    r = inference_on_dataset(model, ld, ev, st)
                    assert r["num_images"] == n_files
                    out["through_files"] = {"images_per_sec": st["images_per_sec"], "files": n_files, "decode_threads": ld.num_workers,
                                            "cpus_usable": usable_cpus(), "cpus_visible": os.cpu_count(),
                                            "bound": "host PNG decode (PIL), not the GPU",
                                            "entry": "data.build_test_loader -> evaluation.inference.inference_on_dataset"}
                finally:
                    shutil.rmtree(root, ignore_errors=True)
                    torch.set_num_threads(nthreads)

            def side_latency_b1():
                # batch 1 in the reference's OWN loop shape (tools/benchmark.py:117-145 `benchmark_eval`: 5 warm-up calls, then
                # 400 x model(d) with d = one image from the test loader; plain_train_net.py:316-336 evaluates at batch 1 per GPU
                # too): ms per image of the synchronous call, and of the streamed loop over single-image batches
                # (inference_on_dataset -> forward_streamed).  The headline model at 1024^2 and BASELINE configs[0]'s workload
                # (HRSC2016 R50-FPN, one 800 x 1216 image, 1 class).  Device-resident images, as everywhere in this file.
                from dafne_amd.evaluation.inference import inference_on_dataset
                nthreads = torch.get_num_threads()
                res = {}
                cases = (("r101_1024x1024", lambda: model, (args.size, args.size)),
                         ("configs0_hrsc_r50_800x1216", lambda: build_model(50, device, seed=0, cfgname="hrsc_r50.yaml", cls_prior=-1.5)[1], (800, 1216)))
                for name, mk, (h, w) in cases:
                    m1 = mk()
                    g1 = torch.Generator().manual_seed(7)
                    img = torch.randint(0, 256, (3, h, w), generator=g1, dtype=torch.uint8).to(device)
                    d = [{"image": img, "height": h, "width": w, "image_id": 0}]
                    for _ in range(5):
                        o = m1(d)
                    torch.cuda.synchronize()
                    iters = 400
                    t0 = time.perf_counter()
                    for _ in range(iters):
                        o = m1(d)
                    torch.cuda.synchronize()
                    ms_call = 1e3 * (time.perf_counter() - t0) / iters
                    loader = [d] * iters
                    inference_on_dataset(m1, loader[:8])
                    st = {}
                    inference_on_dataset(m1, loader, None, st)
                    res[name] = {"ms_per_image_model_call": ms_call, "ms_per_image_streamed_loop": 1e3 * st["seconds"] / iters,
                                 "images_per_sec_model_call": 1e3 / ms_call, "images_per_sec_streamed_loop": st["images_per_sec"],
                                 "iters": iters, "warmup": 5, "detections": len(o[0]["instances"])}
                    if m1 is not model:
                        del m1
                torch.set_num_threads(nthreads)
                res["loop"] = "tools/benchmark.py:117-145 (5 warm-up + 400 x model([one image])); streamed: inference_on_dataset over 400 single-image batches"
                out["latency_b1"] = res

            for fn in (side_forward, side_latency_b1, side_r50, side_fp8, side_tta, side_files):
                try:
                    fn()
                except Exception as e:      # noqa: BLE001
                    out.setdefault("side_metric_errors", {})[fn.__name__] = "%s: %s" % (type(e).__name__, e)
                torch.cuda.synchronize()
      except Exception as e:      # noqa: BLE001  (the headline line is printed regardless)
        out.setdefault("side_metric_errors", {})["extras"] = "%s: %s" % (type(e).__name__, e)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            keep = {"fp32": [], "bf16_emulation": []} if not args.no_extras else None
            out["cpu_baseline"] = cpu_baseline(cfg, sd, args.depth, keep=keep)
        except Exception as e:      # noqa: BLE001
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            keep = None
        if keep is not None and len(keep["fp32"]) == 8:
            try:
                out["equivalence_ap"] = equivalence_side_metric(model, keep, device)
            except Exception as e:      # noqa: BLE001
                out["equivalence_ap"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def main(argv=None):
    return launch(parse_args(argv))


if __name__ == "__main__":
    main()
