"""What the part draws and clocks at while the cfg2 kernel runs back to back: hwmon power / sclk samples (sysfs) beside the launch
time, for random, constant and all-zero input, and for 8 / 12 waves per CU (lab library through SS_LIB_PATH + SS_WAVES)."""
import glob
import os
import sys
import threading
import time

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "mfcc-rust_amd"))
import torch

import speechsauce_amd as ss


def read(path):
    try:
        return int(open(path).read().split()[0])
    except Exception:
        return None


hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
pw = [p for h in hw for p in glob.glob(h + "/power1_average") + glob.glob(h + "/power1_input")]
fq = [p for h in hw for p in glob.glob(h + "/freq1_input")]
cap = [p for h in hw for p in glob.glob(h + "/power1_cap")]
print("hwmon:", hw, "power files", pw, "freq files", fq, "cap", [read(c) for c in cap])


def run(name, x, seconds=2.0):
    samples = []
    stop = False

    def sampler():
        while not stop:
            samples.append((read(pw[0]) if pw else None, read(fq[0]) if fq else None))
            time.sleep(0.02)

    for _ in range(200):
        ss.mfcc_batch(x, 16000)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(500):
            ss.mfcc_batch(x, 16000)
        n += 500
    e1.record()
    torch.cuda.synchronize()
    stop = True
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    p = [s[0] for s in samples[5:] if s[0]]
    f = [s[1] for s in samples[5:] if s[1]]
    print("%-10s %.2f us per launch | power W: mean %s max %s | sclk MHz: mean %s min %s | %d samples" % (
        name, us, round(sum(p) / len(p) / 1e6) if p else None, round(max(p) / 1e6) if p else None,
        round(sum(f) / len(f) / 1e6) if f else None, round(min(f) / 1e6) if f else None, len(samples)))


g = torch.Generator(device="cuda").manual_seed(1)
xs = torch.randn((5 * 1024, 16000), device="cuda", generator=g) * 0.1
run("random", xs[:1024].contiguous())
run("constant", torch.full((1024, 16000), 0.25, device="cuda"))
run("zeros", torch.zeros((1024, 16000), device="cuda"))
run("random", xs[1024:2048].contiguous())
