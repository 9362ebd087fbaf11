import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev)
imgs = [torch.randint(0, 256, (3, 1024, 1024), dtype=torch.uint8) for _ in range(8)]
inputs = [{"image": im, "height": 1024, "width": 1024, "image_id": i} for i, im in enumerate(imgs)]
for _ in range(3):
    model.forward_streamed(inputs)
model.flush(); torch.cuda.synchronize()
# idle GPU
for rep in range(3):
    t0 = time.perf_counter(); b, v, o = model._pack_inputs(inputs, staged=True); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("idle: pack %.2f ms, sync %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
# in the loop
ts = []
for rep in range(12):
    t0 = time.perf_counter()
    b, v, o = model._pack_inputs(inputs, staged=True)
    t1 = time.perf_counter()
    rows, counts = model.detect_packed(b, valid_hw=v, out_hw=o, pipelined=True, splits=3)
    t2 = time.perf_counter()
    ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
torch.cuda.synchronize()
print("loop (pack ms, enqueue ms):", " ".join("%.1f/%.1f" % t for t in ts))
t0 = time.perf_counter()
for rep in range(12):
    model.forward_streamed(inputs)
model.flush(); torch.cuda.synchronize()
print("forward_streamed host: %.1f ms per batch" % (1e3 * (time.perf_counter() - t0) / 12))
