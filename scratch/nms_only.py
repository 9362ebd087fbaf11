import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
print("nms ms/img", bench.nms_ms_per_image(torch.device("cuda", 0), reps=20))
