"""gfx950 listings (hipcc -S, the build's own flags per unit) of dafne_amd/csrc/*.hip, compiled once per test session: the static ISA
guards (tests/test_async_loads.py, tests/test_packed_fp32.py) read the same files -- conv.hip alone takes three minutes."""
import os
import subprocess
import sys
import tempfile
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_DIR = None
_DONE = {}
_LOCK = threading.Lock()


_PREFETCHED = False


def listing(src, _prefetch=True):
    """-> the lines of `src`'s device listing (thread-safe: one compile per unit).  The first call starts the compiles of ALL units in
    the background (6 at a time), so a session's cost is that of the slowest unit whichever test asks first."""
    global _DIR, _PREFETCHED
    from dafne_amd import build as B
    if _prefetch and not _PREFETCHED:
        _PREFETCHED = True
        from concurrent.futures import ThreadPoolExecutor
        srcs = sorted((f for f in os.listdir(B.CSRC) if f.endswith(".hip")), key=lambda f: -os.path.getsize(os.path.join(B.CSRC, f)))
        ex = ThreadPoolExecutor(max_workers=6)
        for f in srcs:
            ex.submit(listing, f, False)
        ex.shutdown(wait=False)
    with _LOCK:
        if _DIR is None:
            _DIR = tempfile.mkdtemp(prefix="dafne_listings_")
        ev = _DONE.get(src)
        mine = ev is None
        if mine:
            ev = _DONE[src] = {"event": threading.Event(), "lines": None, "err": None}
    if mine:
        out = os.path.join(_DIR, src + ".s")
        flags = [f for f in B.COMMON if f != "-fPIC"] + B.PER_FILE.get(src, [])
        try:
            r = subprocess.run([B.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(B.CSRC, src)],
                               capture_output=True, text=True, timeout=1500)
            if r.returncode != 0:
                ev["err"] = r.stderr[-2000:]
            else:
                ev["lines"] = open(out).read().split("\n")
        except Exception as e:      # noqa: BLE001
            ev["err"] = repr(e)
        ev["event"].set()
    ev["event"].wait()
    assert ev["err"] is None, (src, ev["err"])
    return ev["lines"]
