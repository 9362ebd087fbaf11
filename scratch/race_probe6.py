"""Same image, concurrent single-image calls on rotating streams (deferred post-process): every call's DETECTIONS against the first
call's, and against the synchronous model([image]).  Conv buffers were shown identical (race_probe5): this isolates decode + NMS + hand-out."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_inference_loop as T
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
cfg, m = T._gpu_model()
g = torch.Generator().manual_seed(21)
img = torch.randint(0, 256, (1, 3, H, W), generator=g, dtype=torch.uint8).cuda()
r0, c0 = m.detect_packed(img); torch.cuda.synchronize()
k0 = int(c0[0]); ref = r0[0, :k0].clone()
rot = 0; bad = []; pend = []
for i in range(calls):
    res = m.detect_packed(img, pipelined=True, splits=1, defer=os.environ.get("DEFER", "1") == "1", stream_offset=rot)
    rot = (rot + 1) % int(os.environ.get("NSTREAMS", "3"))
    if res is not None: pend.append((i - 1, res))
    if os.environ.get("DEFER", "1") != "1" and i == calls - 1: pend.append((i, res)); pend.pop()
    if len(pend) >= 8 or i == calls - 1:
        if i == calls - 1 and os.environ.get("DEFER", "1") == "1":
            pend.append((i, m.flush_deferred()))
        torch.cuda.synchronize()
        for j, (rows, counts) in pend:
            k = int(counts[0])
            if k != k0 or not torch.equal(rows[0, :k], ref):
                d = "count %d vs %d" % (k, k0) if k != k0 else "rows differ in %d elements, max |d| %.4g" % (int((rows[0, :k] != ref).sum()), float((rows[0, :k] - ref).abs().max()))
                bad.append((j, d))
        pend = []
print("%d calls at %dx%d: %d differ from model([image])" % (calls, H, W, len(bad)), bad[:6])
