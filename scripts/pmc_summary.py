"""Summarise the rocprofv3 --pmc passes written by scripts/pmc_passes.sh.

usage: pmc_summary.py <dir with FETCH_SIZE/ WRITE_SIZE/ SQ_VALU_MFMA_BUSY_CYCLES/ sub-directories> <tag>

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB as reported; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads).  Matrix-pipe utilisation per kernel =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch duration x 2.4 GHz nominal): the counter advances 32 cycles per
v_mfma_f32_32x32x16_bf16 (checked against the algorithmic MFMA count), which is that instruction's issue interval at
peak.  (GRBM_GUI_ACTIVE is collected too, but it includes the profiler's per-dispatch set-up, so a "running clock"
figure derived from it is only printed, not stored.)  Dispatches are serialised by the counter collection, so these are
per-launch figures at the sub-batch sizes of the bench (2-3 images), without the overlap of the concurrent streams.
"""
import collections
import csv
import glob
import json
import os
import sys

N_SIMD = 256 * 4
NOMINAL_GHZ = 2.4


def short(name):
    k = name
    for p in ("void (anonymous namespace)::", "(anonymous namespace)::", "void "):
        k = k.replace(p, "")
    return k.split("(")[0].replace(", ", ",").strip()


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    root, tag = sys.argv[1], sys.argv[2]
    lines = []
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for sub in sorted(os.listdir(root)):
        d = os.path.join(root, sub)
        if not os.path.isdir(d):
            continue
        for r in load(d):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].setdefault(sub, []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    want = [k for k in per if any(s in k for s in ("conv", "gn_", "nms", "decode", "stem", "pool", "preprocess"))]
    traffic = {}
    lines.append("%-40s %7s %9s %9s %9s | %9s %8s %8s" % ("kernel", "launch", "fetchMB", "writeMB", "hbmMB",
                                                          "mfma_busy", "util_clk", "util_nom"))
    def tot(k):
        return sum(per[k].get("FETCH_SIZE", [0]))
    for k in sorted(want, key=lambda k: -tot(k)):
        c = per[k]
        n = max(len(v) for v in c.values())
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) / 1024 if c.get("FETCH_SIZE") else float("nan")
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) / 1024 if c.get("WRITE_SIZE") else float("nan")
        hbm = 2 * f + w
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        gui = c.get("GRBM_GUI_ACTIVE")
        u_clk = u_nom = float("nan")
        mb = 0.0
        if busy and sum(busy) > 0:
            mb = sum(busy) / len(busy)
            if gui:
                u_clk = mb / (N_SIMD * (sum(gui) / len(gui)) / 8.0)
            dd = dur[k].get("SQ_VALU_MFMA_BUSY_CYCLES")
            if dd:
                u_nom = mb / (N_SIMD * (sum(dd) / len(dd)) * NOMINAL_GHZ)
        lines.append("%-40s %7d %9.2f %9.2f %9.2f | %9.3g %8.3f %8.3f" % (k[:40], n, f, w, hbm, mb, u_clk, u_nom))
        if f == f:
            traffic[k] = {"fetch_size_mb_per_launch": round(f, 2), "write_size_mb_per_launch": round(w, 2),
                          "hbm_mb_per_launch": round(hbm, 2), "launches": n}
            if u_clk == u_clk:
                traffic[k]["mfma_util_at_nominal_clock"] = round(u_nom, 4)
    txt = "\n".join(lines) + "\n"
    open(os.path.join(root, "summary.txt"), "w").write(txt)
    src = ("profiles/%s_pmc_%s.txt (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, "
           "separate passes; hbm = 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md)" % (os.environ.get("PMC_ROUND", "r02"), tag))
    json.dump({"source": src, "kernels": traffic}, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
    print(txt)


if __name__ == "__main__":
    main()
