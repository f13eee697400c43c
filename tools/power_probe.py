"""What the part draws and clocks at while one of bench.py's workloads runs back to back: hwmon socket power and shader clock of
THIS process's GPU (sysfs, found by PCI address) sampled every 20 ms beside the launch time.

    python tools/power_probe.py [--workload cfg2|cfg3|cfg5] [--seconds 2]
    SS_LIB_PATH=.../libspeechsauce_amd_lab.so SS_WAVES=8 python tools/power_probe.py      (lab build: eight waves per CU)

Inputs: `ring` = bench.py's own rotated ring of distinct N(0, 0.1) batches (> 256 MiB: streams from HBM); `one` = one such batch
again and again (stays in the Infinity Cache); `pcm16` = the ring rounded to 16-bit PCM steps (what decoded audio looks like: the
low mantissa bits are zero); `const` / `zeros` = every sample 0.25 / 0.  Energy per launch = mean power x time per launch."""
import argparse
import ctypes as C
import glob
import os
import sys
import threading
import time

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "mfcc-rust_amd"))
sys.path.insert(0, R)
import torch

import bench
from speechsauce_amd import SpeechConfig, _lib, make_params

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg5"])
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--inputs", default="ring,one,pcm16,const,zeros,ring")
args = ap.parse_args()


def read(path):
    try:
        return int(open(path).read().split()[0])
    except Exception:
        return None


# the hwmon directory of this process's GPU (a box has eight; /sys/class/drm order is not the runtime's order): by PCI address
pr = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
hw = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
pw = [p for h in hw for p in glob.glob(h + "/power1_average") + glob.glob(h + "/power1_input")]
fq = [p for h in hw for p in glob.glob(h + "/freq1_input")]
cap = [read(p) for h in hw for p in glob.glob(h + "/power1_cap")]
desc, pkw, n_samples, clips, kind = bench.WORKLOADS[args.workload]
print("# %s | build: %s" % (desc, "lab, " + " ".join(k + "=" + v for k, v in os.environ.items() if k in ("SS_WAVES", "SS_MEL_WAVES")) if "lab" in os.environ.get("SS_LIB_PATH", "") else "product"))
print("# device %s, power cap %s W, idle %s W at %s MHz" % (
    bdf, [round(c / 1e6) for c in cap if c], round(read(pw[0]) / 1e6) if pw and read(pw[0]) else None,
    round(read(fq[0]) / 1e6) if fq and read(fq[0]) else None))

cfg = SpeechConfig(make_params(**pkw))
lib = _lib.lib()
if kind == "mfcc":
    out = torch.empty((clips, cfg.num_frames(n_samples), cfg.params.num_cepstral), device="cuda")
else:
    out = torch.empty((clips, cfg.params.num_filters, cfg.stft_rows(n_samples)[0]), device="cuda")
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def launch(x):
    if kind == "mfcc":
        rc = lib.ss_mfcc_batch_device(cfg.handle, x.data_ptr(), clips, n_samples, n_samples, out.data_ptr(), sp)
    else:
        rc = lib.ss_mel_spectrogram_device(cfg.handle, x.data_ptr(), clips, n_samples, n_samples, out.data_ptr(), sp)
    assert rc == 0, rc


def run(name, xs):
    samples = []
    stop = False

    def sampler():
        while not stop:
            samples.append((read(pw[0]) if pw else None, read(fq[0]) if fq else None))
            time.sleep(0.02)

    for i in range(300):
        launch(xs[i % len(xs)])
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    e0.record()
    while time.time() - t0 < args.seconds:
        for _ in range(500):
            launch(xs[n % len(xs)])
            n += 1
    e1.record()
    torch.cuda.synchronize()
    stop = True
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    p = [s[0] / 1e6 for s in samples[5:] if s[0]]
    f = [s[1] / 1e6 for s in samples[5:] if s[1]]
    mp = sum(p) / len(p) if p else None
    # shader clock by the library's one-wave probe beside the same launches (s_memtime / s_memrealtime): cycles per launch
    clk = bench.probe_clock(torch, lib, lambda i: launch(xs[i % len(xs)]), us * 1e-6, torch.device("cuda", 0), probe_us=4000)
    print("%-6s %6.2f us per launch | power W: mean %s max %s | sclk MHz: mean %s min %s | energy per launch %s mJ | probe clock %s GHz -> %s k cycles per launch | kernel %s" % (
        name, us, round(mp) if p else None, round(max(p)) if p else None, round(sum(f) / len(f)) if f else None,
        round(min(f)) if f else None, round(mp * us * 1e-3, 1) if p else None, round(clk, 3) if clk else None,
        round(us * clk, 1) if clk else None, lib.ss_last_kernel_name().decode()))


n_buf = max(2, -(-300 * 1024 * 1024 // (4 * clips * n_samples)))
ring = [bench.synth_batch(torch, clips, n_samples, 1 + i, "cuda") for i in range(n_buf)]
sets = {
    "ring": lambda: ring,
    "one": lambda: ring[:1],
    "pcm16": lambda: [torch.round(x * 32768.0).clamp_(-32768, 32767).div_(32768.0) for x in ring],
    "const": lambda: [torch.full((clips, n_samples), 0.25, device="cuda")],
    "zeros": lambda: [torch.zeros((clips, n_samples), device="cuda")],
}
for name in args.inputs.split(","):
    run(name, sets[name]())
