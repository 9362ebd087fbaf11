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


# The static ISA guards (scripts/check_async_loads.py, scripts/check_packed_fp32.py) as a BUILD step: `python -m dafne_amd.build --check`
# compiles every unit once more to a listing (hipcc -S, same flags) and fails on a flagged instruction.  ASYNC_CHECKED: the units whose
# hand-scheduled kernels are fully unrolled per tile, which is what the checker's in-order queue walks (program-text order); the
# persistent kernels with RUNTIME loops (conv_b2b_mid / _narrow, conv_blk_mid / _narrow, conv_wr) re-use ring registers across a back
# edge the model does not follow -- their waits are covered by the bit-identity tests under load (tests/test_gpu_reproducible.py).
ASYNC_CHECKED = {"conv.hip": ["conv3x3_rp_kernel"], "conv_bneck.hip": ["conv_bneck_kernel"], "conv_b2b.hip": ["conv_b2b_kernel"],
                 "conv3x3_c64.hip": ["conv3x3_c64_kernel"]}
NO_MATRIX_UNITS = ("decode.hip", "poly_nms.hip", "resize.hip", "dense_ops.hip")


def check(verbose=True):
    """-> list of (unit, kernel or None, text) violations of the two static rules; empty when the build is clean."""
    import tempfile
    sys.path.insert(0, os.path.join(HERE, "..", "scripts"))
    import check_async_loads as cal
    import check_packed_fp32 as cpf
    bad = []
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for s in srcs:
            out = os.path.join(td, s + ".s")
            flags = [f for f in COMMON if f != "-fPIC"] + PER_FILE.get(s, [])
            procs.append((s, out, subprocess.Popen([HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, s)],
                                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        for s, out, p in procs:
            log, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("hipcc -S failed on %s:\n%s" % (s, log.decode()))
            lines = open(out).read().split("\n")
            for n, k, t in (cpf.check_none(lines) if s in NO_MATRIX_UNITS else cpf.check(lines)):
                bad.append((s, k, "packed fp32: " + t))
            for name, body in cal.kernels(lines):
                if any(k in name for k in ASYNC_CHECKED.get(s, [])):
                    for i, l, why in cal.check(body):
                        bad.append((s, name, "async load: %s (%s)" % (l, why)))
            if verbose:
                print("checked %-22s %s" % (s, "async + packed" if s in ASYNC_CHECKED else "packed"))
    return bad


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--check" in sys.argv:
        v = check()
        for s, k, t in v[:40]:
            print("VIOLATION %s [%s] %s" % (s, k, t))
        print("%d violations" % len(v))
        sys.exit(1 if v else 0)
