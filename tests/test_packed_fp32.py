"""Static guard (CPU, needs hipcc) for packed fp32 arithmetic, per translation unit with ITS build flags: none at all in the units
without matrix instructions; only the operand forms measured stable in the matrix units -- never a `v_pk_{mul,add,fma}_f32` whose low
result lane reads the HIGH half of a source pair, which on the MI355X returns wrong lanes beside matrix kernels of other waves
(scripts/check_packed_fp32.py; found in round 5 as mis-ordered quads out of sort_quadrilateral -- dafne/utils/sort_corners.py:26-92
-- and present in conv3x3_pred16's GroupNorm on load, dafne/modeling/dafne/dafne.py:330-344).  tests/test_gpu_reproducible.py and
tests/test_gpu_pk_canary.py are the dynamic counterparts."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


NO_MATRIX_UNITS = ("decode.hip", "poly_nms.hip", "resize.hip", "dense_ops.hip")


def test_packed_fp32_rule_per_translation_unit(tmp_path):
    """Every unit with ITS build flags: the units without matrix instructions (-fno-slp-vectorize) contain no packed fp32 arithmetic
    at all; the matrix units only the operand forms measured stable beside matrix kernels (scripts/check_packed_fp32.py
    ALLOWED_FORMS).  The forms each unit contains are written to gpurun_out/packed_fp32_forms.json."""
    import json
    from dafne_amd import build as B
    if not os.path.exists(B.HIPCC):
        pytest.skip("hipcc not installed")
    import check_packed_fp32 as chk
    srcs = sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))

    import _listings

    def listing(s):
        lines = _listings.listing(s)
        has_mfma = any("v_mfma_" in l for l in lines)
        return s, has_mfma, chk.check(lines), chk.check_none(lines), chk.forms(lines)
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(listing, srcs))
    bad = {s: v[:4] for s, _, v, _, _ in res if v}
    assert not bad, bad
    for s, has_mfma, _, anyp, fm in res:
        if s in NO_MATRIX_UNITS:
            # built without the SLP vectoriser, which is what makes packed forms out of scalar code; and nobody wrote one by hand
            assert "-fno-slp-vectorize" in B.PER_FILE[s] and not has_mfma, s
            assert not anyp, (s, anyp[:4])
        else:
            assert set(fm) <= chk.ALLOWED_FORMS, (s, fm)
    # a unit that is neither listed as matrix-free nor contains a matrix instruction would escape both rules
    for s, has_mfma, _, _, _ in res:
        assert has_mfma or s in NO_MATRIX_UNITS or s == "abi.hip", s
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "packed_fp32_forms.json"), "w") as f:
        json.dump({s: {"matrix_unit": m, "forms": fm} for s, m, _, _, fm in res}, f, indent=1)


def test_the_checker_classifies_the_forms():
    import check_packed_fp32 as chk
    lines = ["_Zk:", "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]", "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1]",
             "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,1,0]", "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] neg_lo:[0,1] neg_hi:[0,1]",
             "\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]", "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]",
             "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,0] op_sel_hi:[0,1]", "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] frob:[1,0]"]
    assert [n for n, _, _ in chk.check(lines)] == [2, 4, 9]
    assert [chk.form_of(l) for l in lines[1:]] == ["op_sel_high", "op_sel_hi", "op_sel_high", "neg", None, "plain", "op_sel_hi",
                                                  "unknown:v[0:1], v[2:3], v[4:5] frob:[1,0]"]
    assert len(chk.check_none(lines)) == 7
    assert chk.forms(lines) == {"op_sel_high": 2, "op_sel_hi": 2, "neg": 1, "plain": 1, "unknown:v[0:1], v[2:3], v[4:5] frob:[1,0]": 1}
