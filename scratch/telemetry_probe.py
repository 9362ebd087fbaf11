"""Which sysfs files carry the LIVE shader clock / socket power of the GPU under load?  (bench.py device_telemetry)"""
import glob, os, subprocess, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
d = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=d).to(torch.bfloat16); b = torch.randn(8192, 8192, device=d).to(torch.bfloat16)
print("cards:", sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk")))
for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
    hw = glob.glob(c + "/hwmon/hwmon*")
    print(c, "hwmon:", hw, [os.path.basename(f) for f in glob.glob(hw[0] + "/*")][:40] if hw else None)
stop = threading.Event()
def load():
    torch.cuda.set_device(0)
    while not stop.is_set():
        for _ in range(50): a @ b
        torch.cuda.synchronize()
th = threading.Thread(target=load); th.start()
for k in range(8):
    time.sleep(0.5)
    print(k, bench.device_telemetry(0))
    if k in (3, 7):
        print(subprocess.run("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk'", shell=True, capture_output=True, text=True).stdout)
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            for f in ("pp_dpm_sclk",):
                try: print(c, [l.strip() for l in open(os.path.join(c, f)) if "*" in l])
                except OSError as e: print(c, e)
            for hw in glob.glob(c + "/hwmon/hwmon*"):
                for f in ("power1_average", "power1_input", "freq1_input"):
                    p = os.path.join(hw, f)
                    if os.path.exists(p):
                        try: print("  ", p, open(p).read().strip())
                        except OSError as e: print("  ", p, e)
stop.set(); th.join()
