"""The synchronous call detectron2's loop makes -- outputs = model(batched_inputs), tools/plain_train_net.py:316-336 -- at batch 8,
R101, 1024^2: ms per call for the layouts of forward() (DAFNE_FWD_PPS / DAFNE_FWD_SPLITS / DAFNE_SPLIT_SIZES are read per call) and
the equality of their results.  usage: fwd_sync_probe.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
cfg, m, _ = bench.build_model(101, d, seed=0)
g = torch.Generator().manual_seed(0)
batch = torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(d)
inputs = [{"image": batch[k], "height": 1024, "width": 1024} for k in range(8)]
def run(env, nb=40):
    for k in ("DAFNE_FWD_PPS", "DAFNE_FWD_SPLITS", "DAFNE_SPLIT_SIZES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(6): out = m(inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nb): out = m(inputs)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / nb, out
ref = None
for name, env in (("whole-batch post-process, 4+4", {"DAFNE_FWD_PPS": "0"}),
                  ("per sub-batch, 5+3", {"DAFNE_FWD_PPS": "1"}),
                  ("per sub-batch, 6+2", {"DAFNE_FWD_PPS": "1", "DAFNE_SPLIT_SIZES": "6,2"}),
                  ("per sub-batch, 4+4", {"DAFNE_FWD_PPS": "1", "DAFNE_SPLIT_SIZES": "4,4"}),
                  ("per sub-batch, 3+2+3", {"DAFNE_FWD_PPS": "1", "DAFNE_FWD_SPLITS": "3"}),
                  ("per sub-batch, 4+3+1", {"DAFNE_FWD_PPS": "1", "DAFNE_FWD_SPLITS": "3", "DAFNE_SPLIT_SIZES": "4,3,1"}),
                  ("whole-batch post-process, 4+4 (again)", {"DAFNE_FWD_PPS": "0"}),
                  ("per sub-batch, 5+3 (again)", {"DAFNE_FWD_PPS": "1"})):
    ms, out = run(env)
    same = ""
    if ref is None:
        ref = out
    else:
        same = "  same as first: %s" % all(torch.equal(a["instances"].pred_corners, b["instances"].pred_corners) and
                                           torch.equal(a["instances"].scores, b["instances"].scores) for a, b in zip(out, ref))
    print("%-42s %.3f ms per call of 8 (%.0f img/s)%s" % (name, ms, 8e3 / ms, same))
