"""Where does the host-tile path of forward_streamed spend its time?  (pinned staging copy, H2D, evaluator D2H)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
dev = torch.device("cuda", 0)
imgs = [torch.randint(0, 256, (3, 1024, 1024), dtype=torch.uint8) for _ in range(8)]
pinned = torch.zeros(8, 3, 1024, 1024, dtype=torch.uint8).pin_memory()
print("pinned?", pinned.is_pinned(), "threads", torch.get_num_threads())
for rep in range(3):
    t0 = time.perf_counter()
    for k, im in enumerate(imgs):
        pinned[k, :, :1024, :1024].copy_(im)
    t1 = time.perf_counter()
    for k, im in enumerate(imgs):
        pinned[k].copy_(im)
    t2 = time.perf_counter()
    b = torch.empty(8, 3, 1024, 1024, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    b.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    c = torch.stack(imgs)
    t5 = time.perf_counter()
    d = c.to(dev)
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    print("sliced copy %.2f ms | plain copy %.2f ms | H2D pinned %.2f ms | stack %.2f ms | H2D pageable %.2f ms" %
          (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t6 - t5)))
