"""fwd_sync_probe.py preceded by what bench.py's side_forward does first (the streamed loop over device and host tiles)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.evaluation.inference import DafneEvaluator, inference_on_dataset
d = torch.device("cuda", 0)
if os.environ.get("NT"): torch.set_num_threads(int(os.environ["NT"]))
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
def sync_loop(tag):
    loader = [[{"image": batch[k], "height": 1024, "width": 1024} for k in range(8)] for j in range(24)]
    for b in loader[:4]: m(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in loader: m(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %.1f img/s (%.2f ms per call), torch threads %d" % (tag, 192 / dt, 1e3 * dt / 24, torch.get_num_threads()), flush=True)
sync_loop("fresh")
for where in sys.argv[1:] or ["device", "host"]:
    imgs = batch if where == "device" else batch.cpu()
    loader = [[{"image": imgs[k], "height": 1024, "width": 1024, "image_id": j * 8 + k} for k in range(8)] for j in range(24)]
    ev = DafneEvaluator("synthetic", cfg, distributed=False)
    inference_on_dataset(m, loader[:4], ev)
    st = {}
    inference_on_dataset(m, loader, ev, st)
    print("streamed %s tiles: %.1f img/s" % (where, st["images_per_sec"]), flush=True)
    sync_loop("after the streamed loop over %s tiles" % where)
if os.environ.get("DEFER_FIRST"):
    for _ in range(40): m.detect_packed(batch, pipelined=True, splits=3, defer=True)
    m.flush_deferred(); torch.cuda.synchronize()
    sync_loop("after 40 deferred steps")
    iso = bench.conv_kernel_profile_isolated(m, batch)
    sync_loop("after the isolated kernel profile")
    prof = bench.conv_kernel_profile(m, batch, 3)
    sync_loop("after the timed-layout kernel profile")
    bench.library_hbm_reference(d)
    sync_loop("after the HBM library probe")
    bench.library_gemm_reference(d, 8)
    sync_loop("after the GEMM library probe")
if os.environ.get("NMS_FIRST"):
    for kind in ("uniform", "dense", "skewed"):
        for mm, n_img in ((500, 8), (2000, 8), (10000, 8), (27000, 1)):
            if kind != "uniform" and mm < 10000: continue
            bench.nms_ms_per_image(d, m=mm, n_images=n_img, kind=kind, stats={})
    sync_loop("after the NMS side benchmarks")
    f_imm = lambda: m.detect_packed(batch, pipelined=True, splits=3)
    print("immediate form, no host wait: %.1f img/s" % (8 * 24 / bench.time_steps(f_imm, 24, 4, False)))
    sync_loop("after the immediate loop")
