"""Static guard (CPU, needs hipcc): no translation unit of the library, compiled with ITS build flags, contains a packed fp32
instruction whose low result lane reads the high half of a source pair (`v_pk_{mul,add,fma}_f32 ... op_sel:[..1..]`): on the MI355X
that operand form returns wrong lanes beside matrix kernels of other waves (scripts/check_packed_fp32.py; found in round 5 as mis-ordered
quads out of sort_quadrilateral -- dafne/utils/sort_corners.py:26-92 -- and present in conv3x3_pred16's GroupNorm on load,
dafne/modeling/dafne/dafne.py:330-344).  tests/test_gpu_reproducible.py is the dynamic counterpart."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_no_packed_fp32_instruction_reads_a_high_half_into_its_low_lane(tmp_path):
    from dafne_amd import build as B
    if not os.path.exists(B.HIPCC):
        pytest.skip("hipcc not installed")
    import check_packed_fp32 as chk
    srcs = sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))

    def listing(s):
        out = str(tmp_path / (s + ".s"))
        flags = [f for f in B.COMMON if f not in ("-fPIC",)] + B.PER_FILE.get(s, [])
        r = subprocess.run([B.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(B.CSRC, s)],
                           capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (s, r.stderr[-1500:])
        return s, chk.check(open(out).read().split("\n"))
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(listing, srcs))
    bad = {s: v[:4] for s, v in res if v}
    assert not bad, bad
    # the units without matrix instructions are built without the SLP vectoriser, which is what makes the form out of scalar code
    for s in ("decode.hip", "poly_nms.hip", "resize.hip", "dense_ops.hip"):
        assert "-fno-slp-vectorize" in B.PER_FILE[s]


def test_the_checker_flags_the_form():
    import check_packed_fp32 as chk
    lines = ["_Zk:", "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]", "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1]",
             "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,1,0]", "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] neg_lo:[0,1] neg_hi:[0,1]",
             "\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]"]
    assert [n for n, _, _ in chk.check(lines)] == [2, 4]
