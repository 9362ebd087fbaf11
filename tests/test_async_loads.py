"""Static guard for the hand-scheduled kernels' asynchronous register loads (CPU, needs hipcc): in the gfx950 listing of the
resident-patch tower kernels and the res4 block kernel no instruction may read or overwrite the destination registers of an
inline-asm `global_load_dwordx4` before the counted `s_waitcnt vmcnt(n)` that covers it (scripts/check_async_loads.py).

Why: the compiler does not know those registers are not valid yet.  Round 5 found `if (first) wait(kF) else wait(kN)` on a ring
register compiled to a COPY of the register IN FRONT OF one of the waits -- steps 0..7 of every tile but a workgroup's first could
multiply a weight fragment that had not landed.  It showed as run-to-run different detections once a tile's tail got shorter
(profiles/NOTES_r05.md); single-launch bit-identity tests never saw it.  Reference layers: dafne/modeling/dafne/dafne.py:318-344
(towers), detectron2 BottleneckBlock [recalled] (res4)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.parametrize("src,kernels", [("conv.hip", ["conv3x3_rp_kernel"]), ("conv_bneck.hip", ["conv_bneck_kernel"]),
                                         ("conv_b2b.hip", ["conv_b2b_kernel"]), ("conv3x3_c64.hip", ["conv3x3_c64_kernel"])])
def test_no_instruction_touches_an_in_flight_load_destination(tmp_path, src, kernels):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import check_async_loads as chk
    import _listings
    lines = _listings.listing(src)            # (the build's flags; shared with tests/test_packed_fp32.py)
    seen = 0
    for name, body in chk.kernels(lines):
        if not any(k in name for k in kernels):
            continue
        seen += 1
        viol = chk.check(body)
        assert not viol, (name, viol[:6])
    assert seen >= (2 if src in ("conv.hip", "conv_bneck.hip") else 1), seen           # every instantiation of the listed kernels was found


def test_the_checker_sees_a_copy_in_front_of_the_wait():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_async_loads as chk
    good = ["global_load_dwordx4 v[10:13], v1, s[2:3]", "s_waitcnt vmcnt(0)", "v_mov_b64_e32 v[2:3], v[10:11]"]
    bad = ["global_load_dwordx4 v[10:13], v1, s[2:3]", "v_mov_b64_e32 v[2:3], v[10:11]", "s_waitcnt vmcnt(0)"]
    younger = ["global_load_dwordx4 v[10:13], v1, s[2:3]", "global_load_dwordx4 v[14:17], v1, s[2:3]", "s_waitcnt vmcnt(1)",
               "v_mfma_f32_32x32x16_bf16 v[50:65], v[10:13], v[20:23], v[50:65]", "v_add_u32_e32 v14, 1, v14"]
    assert chk.check(good) == []
    assert len(chk.check(bad)) == 1
    v = chk.check(younger)
    assert len(v) == 1 and "v14" in v[0][1]          # the older load has landed behind vmcnt(1), the younger one has not
