import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
dev = torch.device("cuda", 0)
for m, n in [(500, 8), (2000, 8), (10000, 8), (27000, 1), (27000, 4)]:
    print("M=%5d x %d images: %.3f ms/img" % (m, n, bench.nms_ms_per_image(dev, m=m, n_images=n, reps=5)))
