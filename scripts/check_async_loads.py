#!/usr/bin/env python3
"""Static check of a hipcc -S listing: does any instruction other than the matching s_waitcnt read or overwrite the destination
registers of a vector-memory LOAD that may still be in flight?

The hand-scheduled kernels issue `global_load_dwordx4` from inline asm and track readiness themselves (counted `s_waitcnt vmcnt(n)`,
in-order retirement); the compiler does not know that those registers are not valid yet and is free to COPY them (live-range
splitting at a loop's back edge, around a high-pressure region) -- the copy then takes the old bits.  Round 5 found such a copy as
run-to-run different detections (profiles/NOTES_r05.md).  The model: an in-order queue of vector-memory operations per kernel,
popped by every `s_waitcnt vmcnt(n)` down to n entries, walked in program-text order (both sides of a branch are walked: that only
pops more).  usage: check_async_loads.py file.s [kernel-name-substring ...]      exit status 1 when something is flagged"""
import re
import sys


def regset(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def kernels(lines):
    cur, body = None, []
    for l in lines:
        if l.startswith("_Z") and l.rstrip().endswith(":") or (l.startswith("_Z") and ": " in l and "@" in l):
            cur, body = l.split(":")[0], []
        elif cur is not None:
            t = l.strip()
            if t.startswith("s_endpgm"):
                yield cur, body
                cur = None
            elif t and not t.startswith((";", ".")):
                body.append(t)


def check(body):
    queue, viol = [], []
    for i, l in enumerate(body):
        toks = l.replace(",", " ").split()
        op = toks[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", l)
            if m:
                del queue[:max(0, len(queue) - int(m.group(1)))]
            continue
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            dest = regset(toks[1]) if "_load_" in op and "lds" not in op else set()
            if dest and queue:
                infl = set().union(*[q[0] for q in queue])
                if dest & infl:
                    viol.append((i, l, "overwrites an in-flight destination"))
            queue.append((dest, i))
            continue
        if op.startswith("s_") or not queue:
            continue
        infl = set().union(*[q[0] for q in queue])
        used = set()
        for t in toks[1:]:
            used |= regset(t)
        if used & infl:
            viol.append((i, l, "touches the destination of the load at #%d" % [q for q in queue if q[0] & used][0][1]))
    return viol


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2:]
    bad = 0
    for name, body in kernels(lines):
        if want and not any(w in name for w in want):
            continue
        v = check(body)
        print("%-90s %5d instructions, %d flagged" % (name[:90], len(body), len(v)))
        for i, l, why in v[:8]:
            print("      #%d  %s    <- %s" % (i, l, why))
        bad += len(v)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
