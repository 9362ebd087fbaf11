"""Build libdafne_amd.so (gfx950 only) in-tree with hipcc.

    python -m dafne_amd.build [--force]

One object per csrc/*.hip, linked into dafne_amd/libdafne_amd.so.  The
post-process kernels are compiled with -ffp-contract=off: their fp64/fp32
arithmetic must round exactly like the CPU reference (no FMA contraction).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdafne_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# -fno-slp-vectorize on every translation unit WITHOUT matrix instructions: the SLP vectoriser turns scalar fp32 code into packed
# v_pk_{add,mul,fma}_f32 / v_pk_mov_b32 with op_sel / neg modifiers, and on gfx950 such a kernel returned wrong results when
# matrix (MFMA) kernels of OTHER streams ran on the same CUs -- sort_quad_kernel, a pure elementwise function of its input: 7783 of
# 30000 launches wrong beside plain torch.matmul on three streams, 0 of 30000 without the packed instructions, 0 idle (round 5:
# scratch/race_probe11.py, profiles/NOTES_r05.md; tests/test_gpu_reproducible.py).  The convolution units keep their explicit
# two-wide arithmetic: their outputs are compared bit for bit across thousands of concurrent runs (same test).
PER_FILE = {
    "poly_nms.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "decode.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "resize.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "dense_ops.hip": ["-fno-slp-vectorize"],
}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "dafne_amd.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + COMMON + PER_FILE.get(s, []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode()))
        if verbose and out:
            print(out.decode())
    if force or procs or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode())
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
