"""CPU: the C-ABI library loads, exports every symbol include/dafne_amd.h
declares, and the product never reaches into oracle/."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from dafne_amd import build
    return build.build()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dafne_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dafne_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    from dafne_amd import _lib
    declared = _declared_symbols()
    assert declared, "header parse failed"
    out = subprocess.run(["nm", "-D", "--defined-only", built], capture_output=True, text=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    L = _lib.load()          # binds every symbol; no GPU needed
    assert L.dafne_abi_version() >= 100
    assert L.dafne_poly_nms_workspace_bytes(2, 1000) > 0


def test_product_does_not_touch_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dafne_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                t = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", t, flags=re.M) or "liboracle" in t \
                        or "/root/reference" in t:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch):
    from dafne_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdafne_amd.so")
    with pytest.raises(_lib.DafneHipError):
        _lib.load()


def test_cpu_tensor_is_rejected():
    import torch
    from dafne_amd import _lib
    from dafne_amd.modeling.nms import batched_nms_poly
    with pytest.raises(_lib.DafneHipError):
        batched_nms_poly(torch.zeros(3, 8), torch.zeros(3), torch.zeros(3, dtype=torch.int64), 0.1)
