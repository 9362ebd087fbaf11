#!/usr/bin/env python3
"""Static check of a hipcc -S listing (gfx950): no packed fp32 instruction may select the HIGH half of a source pair for its LOW
result lane (`v_pk_{mul,add,fma}_f32 ... op_sel:[..1..]`).

Why: on the MI355X that operand form returned wrong lanes while matrix (MFMA) kernels of other waves ran on the same CUs -- measured
with one-instruction kernels beside torch.matmul on three streams (scratch/ub/pk_probe.hip, scratch/pk_probe.py): 1194 of 1500 launches
wrong for `v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]`, 0 for the plain form, for `neg_lo / neg_hi`, for the `op_sel_hi` broadcast and for
`v_pk_mov_b32`; idle GPU: 0 everywhere.  The SLP vectoriser produces the form from scalar code (sort_quad: mis-ordered quads in the
post-process, profiles/NOTES_r05.md), and so does a multiplication by one half of a register pair (conv3x3_pred16's GroupNorm on load).
usage: check_packed_fp32.py file.s      exit status 1 when something is flagged"""
import re
import sys

PAT = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b.*\bop_sel:\[([01,]+)\]")


def check(lines):
    """-> [(line number, kernel, text)] of flagged instructions."""
    out, cur = [], None
    for n, l in enumerate(lines, 1):
        if l.startswith("_Z") and ":" in l:
            cur = l.split(":")[0]
        m = PAT.match(l)
        if m and "1" in m.group(2):
            out.append((n, cur, l.strip()))
    return out


if __name__ == "__main__":
    bad = check(open(sys.argv[1]).read().split("\n"))
    for n, k, t in bad[:20]:
        print("%s:%d [%s] %s" % (sys.argv[1], n, k, t))
    print("%d flagged" % len(bad))
    sys.exit(1 if bad else 0)
