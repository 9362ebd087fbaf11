"""scratch/ub/pk_probe.hip beside torch.matmul on three streams: launches of each instruction form that differ from the idle result."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
L = ctypes.CDLL(os.path.join(R, "scratch", "ub", "libpk.so"))
L.pk_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
d = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
n = 1 << 15
x = (torch.rand(n, 4, generator=g) * 2 - 1).to(d)
names = ["scalar v_mul_f32 x2", "v_pk_mul_f32", "v_pk_mul_f32 op_sel swizzle", "v_pk_add_f32 neg", "v_pk_fma_f32 op_sel_hi + neg", "v_pk_mov_b32 op_sel", "v_pk_fma_f32"]
side = torch.cuda.Stream()
cs = [torch.cuda.Stream(priority=-1) for _ in range(3)]
A = [torch.randn(2048, 2048, device=d).bfloat16() for _ in range(3)]
B = [torch.randn(2048, 2048, device=d).bfloat16() for _ in range(3)]
def run(kind):
    o = torch.empty_like(x)
    rc = L.pk_run(kind, x.data_ptr(), o.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return o
refs = [run(k).clone() for k in range(7)]; torch.cuda.synchronize()
bad = [0] * 7; cnt = 0
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 600
load = os.environ.get("LOAD", "1") == "1"
pend = []
for it in range(rounds):
    if load:
        for k in range(3):
            with torch.cuda.stream(cs[k]):
                for _ in range(6): A[k] @ B[k]
    with torch.cuda.stream(side):
        pend.append([run(k) for k in range(7)])
    if len(pend) == 16 or it == rounds - 1:
        torch.cuda.synchronize()
        for outs in pend:
            cnt += 1
            for k in range(7):
                if not torch.equal(outs[k], refs[k]): bad[k] += 1
        pend = []
print("%d rounds, matmul load %s; launches differing from the idle result:" % (cnt, load))
for k in range(7): print("   %-32s %d" % (names[k], bad[k]))
