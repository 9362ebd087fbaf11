"""Is forward_streamed slow because decode workers are alive, or because of what the loader hands it?
usage: loader_probe2.py DIR [workers]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from dafne_amd.data.loader import InferenceLoader, list_image_records
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
recs = list_image_records(sys.argv[1])[:256]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator().manual_seed(0)
b = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
for _ in range(6):
    m.forward_streamed([{"image": b[i], "height": 1024, "width": 1024} for i in range(8)])
m.flush(); torch.cuda.synchronize()
ld = InferenceLoader(cfg, recs, batch_size=8, device=d, num_workers=W, prefetch_batches=4)
decoded = list(ld._decoded_batches())          # workers have exited when this returns
def loop(tag, batches_fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tf = tm = 0.0; n = 0
    for imgs in batches_fn():
        t1 = time.perf_counter()
        batch = [ld.mapper.finish(ld.records[n + k], im, ld._staging) for k, im in enumerate(imgs)]
        n += len(imgs)
        t2 = time.perf_counter()
        m.forward_streamed(batch)
        tm += time.perf_counter() - t2; tf += t2 - t1
    m.flush(); torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("%-42s %.1f img/s; per batch: finish %.2f ms, forward_streamed %.2f ms" % (tag, n / tot, 1e3 * tf / (n / 8), 1e3 * tm / (n / 8)), flush=True)
loop("pre-decoded, no workers alive", lambda: iter(decoded))
loop("pre-decoded again", lambda: iter(decoded))
loop("decoding live, %d workers" % W, ld._decoded_batches)
# persistent device images, workers alive (a second loader decoding in the background, its output dropped)
import threading
def drain():
    for _ in InferenceLoader(cfg, recs, batch_size=8, device=None, num_workers=W, prefetch_batches=4)._decoded_batches():
        pass
th = threading.Thread(target=drain); th.start()
time.sleep(1.0)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
while th.is_alive() and n < 400:
    m.forward_streamed([{"image": b[i], "height": 1024, "width": 1024} for i in range(8)]); n += 8
m.flush(); torch.cuda.synchronize()
print("resident tiles while %d workers decode: %.1f img/s" % (W, n / (time.perf_counter() - t0)))
th.join()
