"""Where a joule goes (round 6, profiles/NOTES_r06.md): three one-purpose loops on all CUs -- an L2 -> register
stream in the weight-stream pattern of the resident-patch kernels, LDS reads of a resident tile, bare MFMAs -- each looped for seconds
while amdgpu's hwmon power / clock are sampled (bench.TelemetrySampler): rate, W, MHz and the energy per byte / per flop above idle.
scripts/l2_probe.hip is compiled on first use (hipcc, gfx950) into the system's temporary directory."""
import ctypes, os, subprocess, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
_so = os.path.join(tempfile.gettempdir(), "libdafne_l2probe.so")
if not os.path.exists(_so) or os.path.getmtime(_so) < os.path.getmtime(os.path.join(R, "scripts", "l2_probe.hip")):
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", _so,
                    os.path.join(R, "scripts", "l2_probe.hip")], check=True, timeout=600)
L = ctypes.CDLL(_so)
L.probe_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
d = torch.device("cuda", 0)
w = torch.randint(0, 2**31 - 1, (4 * 1024 * 1024 // 4,), dtype=torch.int32, device=d)
out = torch.zeros(64, dtype=torch.float32, device=d)
seed = torch.randn(1024, device=d)
st = torch.cuda.current_stream().cuda_stream
def loop(kind, region, iters, blocks, secs=3.0):
    fn = lambda: L.probe_run(kind, w.data_ptr(), region, iters, blocks, out.data_ptr(), seed.data_ptr(), st)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    with bench.TelemetrySampler(0) as ts:
        t0 = time.perf_counter(); a.record()
        while time.perf_counter() - t0 < secs:
            for _ in range(20): fn()
            n += 20; torch.cuda.synchronize()
        b.record(); torch.cuda.synchronize()
    tel = ts.summary(skip_s=1.0)
    return a.elapsed_time(b) / n, tel["socket_power_w"]["mean"], tel["sclk_mhz"]["mean"]
time.sleep(1.5)
with bench.TelemetrySampler(0) as ts: time.sleep(1.2)
idle = ts.summary(skip_s=0.0)["socket_power_w"]["mean"]
print("idle %.0f W" % idle)
for blocks in (256, 512):
    for region in (1179648, 2 * 1024 * 1024, 256 * 1024):
        iters = 4096
        ms, wv, mhz = loop(0, region, iters, blocks)
        nbytes = blocks * 8 * iters * 8 * 1024.0
        print("L2 stream  blocks %d region %7d: %.3f ms  %.1f TB/s  %.1f GB/s/CU  %.0f W %4.0f MHz  -> %.1f pJ/B above idle" % (
            blocks, region, ms, nbytes / ms / 1e9, nbytes / ms / 1e6 / min(blocks, 256), wv, mhz, (wv - idle) * ms * 1e-3 / nbytes * 1e12))
ms, wv, mhz = loop(1, 0, 8192, 256)
nbytes = 256 * 512 * 16 * 8 * 8192.0
print("LDS stream: %.3f ms  %.1f TB/s  %.0f W %4.0f MHz -> %.2f pJ/B above idle" % (ms, nbytes / ms / 1e9, wv, mhz, (wv - idle) * ms * 1e-3 / nbytes * 1e12))
for blocks in (256, 512):
    iters = 8192
    ms, wv, mhz = loop(2, 0, iters, blocks)
    fl = blocks * 8 * iters * 4 * 32768.0
    print("MFMA only blocks %d: %.3f ms  %.0f TFLOP/s  %.0f W %4.0f MHz -> %.3f pJ/flop total, %.3f above idle" % (
        blocks, ms, fl / ms / 1e9, wv, mhz, wv * ms * 1e-3 / fl * 1e12, (wv - idle) * ms * 1e-3 / fl * 1e12))
tab = torch.randn(2, 8, 64, 8, device=d)
tab[1] = torch.relu(tab[0])
tab = tab.bfloat16().contiguous()
def loop_tab(kind, iters, blocks, secs=3.0):
    global w
    keep, w = w, tab
    try:
        return loop(kind, 0, iters, blocks, secs)
    finally:
        w = keep
for kind, name in ((7, "32x32x16, one constant pair (asm)"), (3, "32x32x16, 8 rotating operand pairs"), (4, "32x32x16, rotating, B behind a ReLU"),
                   (5, "16x16x32, one constant pair"), (6, "16x16x32, 8 rotating operand pairs")):
    iters = 8192
    ms, wv, mhz = loop_tab(kind, iters, 256)
    fl = 256 * 8 * iters * 4 * 32768.0
    print("MFMA %-36s: %.3f ms  %.0f TFLOP/s  %.0f W %4.0f MHz -> %.3f pJ/flop total, %.3f above idle" % (
        name, ms, fl / ms / 1e9, wv, mhz, wv * ms * 1e-3 / fl * 1e12, (wv - idle) * ms * 1e-3 / fl * 1e12))
