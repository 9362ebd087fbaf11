"""Upper bound of what fusing groupnorm_finalize into its consumer could buy: time the pipelined step with the
finalize launches removed from the plans (results are then WRONG -- stale statistics -- only the timing is meaningful)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), dtype=torch.uint8).to(dev)
step = lambda: model.detect_packed(batch, pipelined=True, splits=3)
def run(tag):
    dt = bench.time_steps(step, 30, 5, False)
    print(tag, round(8 * 30 / dt, 1), "img/s")
run("with finalize   ")
run("with finalize   ")
for st in model._pipe.values():
    for plans in st["plans"]:
        for p in plans:
            p.calls[:] = [c for c in p.calls if getattr(c, "name", "") != "groupnorm_finalize"]
run("without finalize")
run("without finalize")
