"""What the library GEMM (hipBLASLt through torch.matmul) sustains on this box for the tower layer's GEMM shape and for a large
square bf16 GEMM, with random and with all-zero operands (the matrix pipe's clock follows its power draw): the practical
ceiling conv3x3_rp's roofline fraction should be read against."""
import torch, time
d = torch.device("cuda", 0)
def run(M, N, K, kind, reps=30):
    if kind == "randn":
        a = torch.randn(M, K, device=d).to(torch.bfloat16); b = torch.randn(K, N, device=d).to(torch.bfloat16)
    elif kind == "relu":
        a = torch.randn(M, K, device=d).clamp_(min=0).to(torch.bfloat16); b = (torch.randn(K, N, device=d) * 0.02).to(torch.bfloat16)
    else:
        a = torch.zeros(M, K, device=d, dtype=torch.bfloat16); b = torch.zeros(K, N, device=d, dtype=torch.bfloat16)
    for _ in range(5): c = a @ b
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): c = a @ b
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print("M %7d N %5d K %5d %-6s %8.1f us  %7.0f TFLOP/s" % (M, N, K, kind, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
for kind in ("randn", "relu", "zeros"):
    run(8 * 21824, 256, 2304, kind)      # one tower layer, batch 8, five levels (im2col'd)
    run(8192, 8192, 8192, kind)
    run(16384, 4096, 4096, kind)
# fp8 (OCP e4m3) library GEMM, if this torch build has it for gfx950
try:
    f8 = torch.float8_e4m3fn
    for (M, N, K) in ((8 * 21824, 256, 2304), (8192, 8192, 8192)):
        a = torch.randn(M, K, device=d).to(f8); b = torch.randn(N, K, device=d).to(f8).t()
        one = torch.ones((), device=d)
        for _ in range(3): c = torch._scaled_mm(a, b, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): c = torch._scaled_mm(a, b, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print("fp8 e4m3 M %7d N %5d K %5d randn %8.1f us  %7.0f TFLOP/s" % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
except Exception as ex:
    print("fp8 library GEMM unavailable:", repr(ex)[:300])
