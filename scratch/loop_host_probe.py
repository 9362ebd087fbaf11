"""Host time of the streamed evaluation loop per batch of 8 device tiles: forward_streamed() and evaluator.process() separately,
against the loop's wall clock per batch (GPU-bound when the host parts sum to less)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.evaluation.inference import DafneEvaluator, inference_on_dataset
d = torch.device("cuda", 0)
from dafne_amd.utils.host import usable_cpus
torch.set_num_threads(min(16, usable_cpus()))
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
nb = 40
loader = [[{"image": batch[k], "height": 1024, "width": 1024, "image_id": j * 8 + k} for k in range(8)] for j in range(nb)]
ev = DafneEvaluator("synthetic", cfg, distributed=False)
inference_on_dataset(m, loader[:6], ev)
tf, tp = [], []
of, op = m.forward_streamed, ev.process
def wf(x):
    t = time.perf_counter(); r = of(x); tf.append(time.perf_counter() - t); return r
def wp(i, o):
    t = time.perf_counter(); r = op(i, o); tp.append(time.perf_counter() - t); return r
m.forward_streamed = wf; ev.process = wp
st = {}
inference_on_dataset(m, loader, ev, st)
import statistics as S
print("loop: %.1f img/s = %.2f ms per batch; forward_streamed host %.2f ms mean (median %.2f, max %.2f); evaluator.process %.2f ms mean (max %.2f)"
      % (st["images_per_sec"], 1e3 * st["seconds"] / nb, 1e3 * S.mean(tf), 1e3 * S.median(tf), 1e3 * max(tf), 1e3 * S.mean(tp), 1e3 * max(tp)))
m.forward_streamed = of
f = lambda: m.detect_packed(batch, pipelined=True, splits=3, defer=True)
dt = bench.time_steps(f, 40, 5, False, flush_fn=m.flush_deferred)
print("bare deferred loop: %.1f img/s = %.2f ms per batch" % (320 / dt, 1e3 * dt / 40))
