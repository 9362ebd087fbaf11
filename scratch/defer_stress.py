"""Run-to-run / layout-to-layout equality at the headline size: the immediate pipelined step against the deferred two-part-graph
loop (and against itself), many repetitions.  usage: defer_stress.py [reps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
cfg, model, sd = bench.build_model(101, dev, seed=0)
nb = 6
batches = []
for j in range(nb):
    g = torch.Generator().manual_seed(100 + j if j else 0)
    batches.append(torch.randint(0, 256, (8, 3, 1024, 1024), generator=g, dtype=torch.uint8).to(dev))


def same(a, b):
    (r, c), (rw, cw) = a, b
    if not torch.equal(c, cw):
        return False
    return all(torch.equal(r[i, :int(cw[i])], rw[i, :int(cw[i])]) for i in range(8))


want = []
for b in batches:
    r, c = model.detect_packed(b, pipelined=True, splits=2)
    torch.cuda.synchronize()
    want.append((r.clone(), c.clone()))
bad = {"immediate": 0, "deferred": 0, "serial_vs_first": 0}
ser = []
for b in batches[:2]:
    r, c = model.detect_packed(b)
    torch.cuda.synchronize()
    ser.append((r.clone(), c.clone()))
for rep in range(reps):
    for j, b in enumerate(batches):
        r, c = model.detect_packed(b, pipelined=True, splits=2)
        torch.cuda.synchronize()
        if not same((r, c), want[j]):
            bad["immediate"] += 1
            print("immediate mismatch rep", rep, "batch", j)
    got = []
    for b in batches:
        res = model.detect_packed(b, pipelined=True, splits=2, defer=True)
        if res is not None:
            got.append(res)
    got.append(model.flush_deferred())
    torch.cuda.synchronize()
    for j in range(nb):
        if not same(got[j], want[j]):
            bad["deferred"] += 1
            print("deferred mismatch rep", rep, "batch", j)
    for j, b in enumerate(batches[:2]):
        r, c = model.detect_packed(b)
        torch.cuda.synchronize()
        if not same((r, c), ser[j]):
            bad["serial_vs_first"] += 1
            print("serial mismatch rep", rep, "batch", j)
print("STRESS", os.environ.get("DAFNE_AMD_LIB", "tree"), bad)
