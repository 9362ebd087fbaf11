"""GPU canary for the packed-fp32 hazard (round 5; DESIGN.md section 7, scripts/pk_probe.py): one-instruction kernels beside
torch.matmul on three streams.  REPORTS the hazard rate of the one operand form that is wrong on this box (`v_pk_mul_f32
op_sel:[0,1] op_sel_hi:[1,0]`: the library never emits it, scripts/check_packed_fp32.py) into gpurun_out/pk_canary.json -- a
runtime, firmware or compiler change shows there -- and ASSERTS that the forms the matrix translation units do contain (plain
v_pk_{mul,add,fma}_f32, neg modifiers, the op_sel_hi broadcast, v_pk_mov_b32) still give the idle-GPU bits.  Reference function whose
results the hazard corrupted: dafne/utils/sort_corners.py:26-92."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
pytestmark = pytest.mark.gpu


def test_packed_fp32_forms_beside_matrix_kernels(tmp_path):
    import pk_probe
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not installed")
    so = pk_probe.build(str(tmp_path))
    idle = pk_probe.run_probe(so, rounds=60, load=False)
    busy = pk_probe.run_probe(so, rounds=300, load=True)
    rep = {"idle": idle, "beside_matmul": busy, "hazard_form": pk_probe.NAMES[pk_probe.HAZARD],
           "hazard_rate": busy["wrong_launches"][pk_probe.NAMES[pk_probe.HAZARD]] / max(busy["rounds"], 1)}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "pk_canary.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print("PK_CANARY " + json.dumps(rep))
    assert all(v == 0 for v in idle["wrong_launches"].values()), idle
    safe = {k: v for k, v in busy["wrong_launches"].items() if k != pk_probe.NAMES[pk_probe.HAZARD]}
    assert all(v == 0 for v in safe.values()), safe
