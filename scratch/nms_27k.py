import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
print(bench.nms_ms_per_image(torch.device("cuda", 0), m=27000, n_images=1, reps=10))
