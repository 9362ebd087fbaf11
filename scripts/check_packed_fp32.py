#!/usr/bin/env python3
"""Static check of a hipcc -S listing (gfx950) for packed fp32 instructions.

Why: on the MI355X `v_pk_{mul,add,fma}_f32` with an `op_sel` that puts the HIGH half of a source pair into the LOW result lane
returned wrong lanes while matrix (MFMA) kernels of other waves ran on the same CUs -- measured with one-instruction kernels beside
torch.matmul on three streams (scripts/pk_probe.hip, scripts/pk_probe.py; tests/test_gpu_pk_canary.py re-measures it on every GPU pass):
1194 of 1500 launches wrong for `v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]`, 0 for the plain form, for `neg_lo / neg_hi`, for the
`op_sel_hi` broadcast and for `v_pk_mov_b32`; idle GPU: 0 everywhere.  The SLP vectoriser produces the form from scalar code
(sort_quad: mis-ordered quads in the post-process, profiles/NOTES_r05.md), and so does a multiplication by one half of a register pair
(conv3x3_pred16's GroupNorm on load).

The rule a kernel of this library follows:
  * a translation unit WITHOUT matrix instructions is built with -fno-slp-vectorize and contains NO packed fp32 arithmetic at all
    (`check_none`);
  * a translation unit with matrix instructions may contain packed fp32 arithmetic only in the operand forms measured stable
    (`form_of` in ALLOWED_FORMS: no modifier, `neg_lo` / `neg_hi`, `op_sel_hi` with low-half selection = the broadcast); anything
    else -- an `op_sel` that reads a high half, a modifier this script does not know -- is flagged (`check`).
usage: check_packed_fp32.py file.s [--none]      exit status 1 when something is flagged"""
import re
import sys

INSN = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b(.*)$")
MOD = re.compile(r"\b(op_sel_hi|op_sel|neg_lo|neg_hi|clamp)(?::\[([01,]+)\])?")
ALLOWED_FORMS = {"plain", "neg", "op_sel_hi", "neg+op_sel_hi"}


def form_of(line):
    """None for a line that is not packed fp32 arithmetic, else the operand form: "plain", "neg", "op_sel_hi" (broadcast / default
    high-half routing), "neg+op_sel_hi", "op_sel_high" (a LOW result lane reads a HIGH source half: the hazard) or "unknown:<text>"."""
    m = INSN.match(line)
    if not m:
        return None
    rest = m.group(2).split(";")[0]
    ops = rest
    kinds = set()
    for mm in MOD.finditer(rest):
        name, bits = mm.group(1), mm.group(2) or ""
        ops = ops.replace(mm.group(0), "")
        if name == "op_sel":
            if "1" in bits:
                return "op_sel_high"
        elif name == "op_sel_hi":
            kinds.add("op_sel_hi")
        elif name in ("neg_lo", "neg_hi"):
            kinds.add("neg")
        else:
            return "unknown:" + name
    if re.search(r"[a-z_]+:\[", ops):                 # a modifier this script has no opinion on
        return "unknown:" + ops.strip()
    return "+".join(sorted(kinds)) if kinds else "plain"


def _walk(lines):
    cur = None
    for n, l in enumerate(lines, 1):
        if l.startswith("_Z") and ":" in l:
            cur = l.split(":")[0]
        f = form_of(l)
        if f is not None:
            yield n, cur, l.strip(), f


def check(lines):
    """-> [(line number, kernel, text)] of packed fp32 instructions in a form that is not known to be stable."""
    return [(n, k, t) for n, k, t, f in _walk(lines) if f not in ALLOWED_FORMS]


def check_none(lines):
    """-> [(line number, kernel, text)] of ALL packed fp32 arithmetic (for the units built with -fno-slp-vectorize)."""
    return [(n, k, t) for n, k, t, f in _walk(lines)]


def forms(lines):
    """-> {form: count} over the listing."""
    out = {}
    for _, _, _, f in _walk(lines):
        out[f] = out.get(f, 0) + 1
    return out


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().split("\n")
    bad = check_none(lines) if "--none" in sys.argv[2:] else check(lines)
    for n, k, t in bad[:20]:
        print("%s:%d [%s] %s" % (sys.argv[1], n, k, t))
    print("forms:", forms(lines))
    print("%d flagged" % len(bad))
    sys.exit(1 if bad else 0)
