import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
order = sys.argv[1:] or ["graphs", "eager", "graphs"]
cfg, m, _ = bench.build_model(101, d, seed=0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=torch.Generator().manual_seed(0), dtype=torch.uint8).to(d)
for mode in order:
    m.use_graphs = mode == "graphs"
    f = lambda: m.detect_packed(batch, pipelined=True, splits=3)
    dt = bench.time_steps(f, 30, 5, False)
    print(mode, "%.1f img/s" % (8 * 30 / dt), flush=True)
