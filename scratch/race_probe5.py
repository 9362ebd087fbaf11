"""Concurrent single-image calls on rotating streams, the SAME image every time: after every burst of 4 calls (one per plan set) the
four sets must hold identical buffers.  Compares FPN features and head outputs across the sets; reports which differ."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_inference_loop as T
bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 150
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
cfg, m = T._gpu_model()
g = torch.Generator().manual_seed(21)
img = torch.randint(0, 256, (1, 3, H, W), generator=g, dtype=torch.uint8).cuda()
names = None
bad = {}
rot = 0
graphs = os.environ.get("GRAPHS", "1") == "1"
for b in range(bursts):
    for c in range(4):
        m.detect_packed(img, pipelined=True, splits=1, defer=True, stream_offset=rot, graphs=graphs)
        rot = (rot + 1) % 3
    m.flush_deferred()
    torch.cuda.synchronize()
    st = m._pipe[(1, H, W, 1)]
    sets = st["plans"]
    ref = sets[0][0]
    for s in range(1, len(sets)):
        p = sets[s][0]
        for lv, (f0, f1) in enumerate(zip(ref.features, p.features)):
            if not torch.equal(f0.t, f1.t):
                bad.setdefault("feature p%d" % (lv + 3), []).append((b, s, int((f0.t != f1.t).sum()), float((f0.t.float() - f1.t.float()).abs().max())))
        h0, h1 = st["ho"][0], st["ho"][s]
        for nm in ("logits", "center", "delta_ctr"):
            for lv, (t0, t1) in enumerate(zip(getattr(h0, nm), getattr(h1, nm))):
                if not torch.equal(t0, t1):
                    bad.setdefault("head %s[%d]" % (nm, lv), []).append((b, s, int((t0 != t1).sum()), float((t0 - t1).abs().max())))
print("%d bursts of 4 calls at %dx%d, graphs %s: %s" % (bursts, H, W, graphs, "all four plan sets identical every time" if not bad else ""))
for k, v in bad.items():
    print("  %s: %d mismatches; first (burst, set, elements, max |d|): %s" % (k, len(v), v[:3]))
