import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
splits = int(sys.argv[1]); ndummy = int(sys.argv[2])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dummies = [torch.cuda.Stream() for _ in range(ndummy)]
for d in dummies:
    with torch.cuda.stream(d): torch.zeros(1, device=dev)
cfg, model, sd = bench.build_model(101, dev)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8).to(dev)
def run(splits, steps=20):
    for _ in range(3): model.detect_packed(batch, pipelined=True, splits=splits)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): model.detect_packed(batch, pipelined=True, splits=splits)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print("cs=%s side=%s hwq=%s dummy=%d splits %d: %.3f ms/step -> %.1f img/s" % (os.environ.get("DAFNE_CS_PRIO"), os.environ.get("DAFNE_SIDE_PRIO"), os.environ.get("GPU_MAX_HW_QUEUES"), ndummy, splits, dt * 1e3, 8 / dt))
run(splits)
