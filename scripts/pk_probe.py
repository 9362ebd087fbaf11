"""Which packed fp32 instruction form returns wrong lanes beside matrix kernels on gfx950?  (round 5 finding)

scripts/pk_probe.hip holds one kernel per instruction form (every thread runs a dependent chain of 256 instructions on its own four
inputs: a pure function of the input).  `run_probe` launches them on a side stream while torch.matmul (hipBLASLt) keeps the matrix
pipes busy on three other streams and counts the launches whose output differs from the idle-GPU output.  Round 5, MI355X, ROCm 7.2:
1194 of 1500 launches wrong for `v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]` (the low result lane reads the HIGH half of a source
pair), 0 for every other form, 0 on an idle GPU.  tests/test_gpu_pk_canary.py runs this on every GPU test pass and writes the rates
to gpurun_out/pk_canary.json, so that a runtime / firmware / compiler change is visible.
usage: pk_probe.py [rounds]      (LOAD=0: no matrix load)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["scalar v_mul_f32 x2", "v_pk_mul_f32", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 neg_lo neg_hi",
         "v_pk_fma_f32 op_sel_hi:[0,1,1] neg", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_fma_f32"]
HAZARD = 2                      # index of the form that is wrong beside matrix kernels (never emitted by this library: check_packed_fp32.py)


def build(out_dir):
    """hipcc scripts/pk_probe.hip -> <out_dir>/libpk_probe.so (gfx950); returns the path."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    so = os.path.join(out_dir, "libpk_probe.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "pk_probe.hip")], check=True,
                   capture_output=True, timeout=600)
    return so


def run_probe(so, rounds=300, load=True, device=0):
    import torch
    L = ctypes.CDLL(so)
    L.pk_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    d = torch.device("cuda", device)
    g = torch.Generator().manual_seed(1)
    n = 1 << 15
    x = (torch.rand(n, 4, generator=g) * 2 - 1).to(d)
    side = torch.cuda.Stream(device=d)
    cs = [torch.cuda.Stream(device=d, priority=-1) for _ in range(3)]
    A = [torch.randn(2048, 2048, device=d).bfloat16() for _ in range(3)]
    B = [torch.randn(2048, 2048, device=d).bfloat16() for _ in range(3)]

    def run(kind):
        o = torch.empty_like(x)
        rc = L.pk_run(kind, x.data_ptr(), o.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return o
    refs = [run(k).clone() for k in range(7)]
    torch.cuda.synchronize()
    bad, cnt, pend = [0] * 7, 0, []
    for it in range(rounds):
        if load:
            for k in range(3):
                with torch.cuda.stream(cs[k]):
                    for _ in range(6):
                        A[k] @ B[k]
        with torch.cuda.stream(side):
            pend.append([run(k) for k in range(7)])
        if len(pend) == 16 or it == rounds - 1:
            torch.cuda.synchronize()
            for outs in pend:
                cnt += 1
                for k in range(7):
                    if not torch.equal(outs[k], refs[k]):
                        bad[k] += 1
            pend = []
    return {"rounds": cnt, "matrix_load": bool(load), "wrong_launches": {NAMES[k]: bad[k] for k in range(7)}}


if __name__ == "__main__":
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        r = run_probe(build(td), int(sys.argv[1]) if len(sys.argv) > 1 else 600, os.environ.get("LOAD", "1") == "1")
    print("%d rounds, matmul load %s; launches differing from the idle result:" % (r["rounds"], r["matrix_load"]))
    for k, v in r["wrong_launches"].items():
        print("   %-44s %d" % (k, v))
