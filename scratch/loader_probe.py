"""Where the file-fed loop's time goes: decoded-batch wait / finish (stage + upload + resize) / forward_streamed, per batch.
usage: loader_probe.py DIR [workers] [backend]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.data.loader import InferenceLoader, list_image_records
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
recs = list_image_records(sys.argv[1])
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
backend = sys.argv[3] if len(sys.argv) > 3 else "process"
ld = InferenceLoader(cfg, recs, batch_size=8, device=d, num_workers=W, prefetch_batches=4, backend=backend)
# warm the model
g = torch.Generator().manual_seed(0)
b = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
for _ in range(5):
    m.forward_streamed([{"image": b[i], "height": 1024, "width": 1024} for i in range(8)])
m.flush(); torch.cuda.synchronize()
tw = tf = tm = 0.0
t_start = time.perf_counter()
it = ld._decoded_batches()
done = 0
nb = 0
first = None
while True:
    t0 = time.perf_counter()
    try:
        imgs = next(it)
    except StopIteration:
        break
    t1 = time.perf_counter()
    if first is None:
        first = t1 - t_start
    batch = [ld.mapper.finish(ld.records[done + k], im, ld._staging) for k, im in enumerate(imgs)]
    done += len(imgs)
    t2 = time.perf_counter()
    m.forward_streamed(batch)
    t3 = time.perf_counter()
    if nb > 0:
        tw += t1 - t0; tf += t2 - t1; tm += t3 - t2
    nb += 1
m.flush(); torch.cuda.synchronize()
tot = time.perf_counter() - t_start
print("%s x%d: %d batches, first batch after %.2f s; per batch: wait %.2f ms, finish %.2f ms, forward_streamed %.2f ms; total %.2f s = %.1f img/s (%.1f after the first batch)"
      % (backend, W, nb, first, 1e3 * tw / (nb - 1), 1e3 * tf / (nb - 1), 1e3 * tm / (nb - 1), tot, done / tot, (done - 8) / (tot - first)))
