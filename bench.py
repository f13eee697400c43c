#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native speechsauce hot path.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|cfg4|cfg5] [--gather-mode root|all|none]

Metric (BASELINE.json): frames/sec (+ real-time factor) for 16 kHz MFCC n_fft=512 at 1/2/4/8 MI355X.
A "step" is one pass of the hot path over one batch of synthetic clips that already sit in HBM:
cfg2 = 1024 x 1 s clips @16 kHz, SpeechConfig defaults (n_fft 512, hop 160, 40 mels, 13 ceps),
one fused kernel launch per step.  To keep the 256 MiB Infinity Cache from serving the input,
steps rotate over enough distinct input batches to exceed it (5 x 65.5 MB for cfg2).

Multi-GPU: one process per GPU (torch.distributed, backend nccl == RCCL).  `python bench.py --gpus N`
starts the N ranks itself (fresh child processes, spawned before this process touches the GPU); under
torch.distributed.run it uses the ranks it is given.  Clips are independent, so every rank processes its
own batch of the same size (weak scaling) with no collective inside the path.  The north-star's "RCCL
gather over xGMI of the final [n_frames x n_mfcc] blocks" follows the path: the blocks of --gather-every
consecutive steps form one bucket that is gathered to rank 0 (--gather-mode root, the default: grouped
ncclSend / ncclRecv, one direct link per peer) or to every rank (all: ncclAllGather) on a side stream while
the next steps' kernels run.  An N > 1 run times TWO regions of K steps back to back, each between
barrier + synchronize pairs: the path alone, then the path with the collective.  `value` is the second
(the whole job the north-star describes), `value_path_only` the first; `gather` carries the collective's
own time and the per-link rate it reached, so the line says how much of the step is interconnect.
cfg4 (the 100 h corpus, 360 000 clips) is the one strong-scaling workload: the corpus is split into
contiguous clip shards (speechsauce_amd.distributed.shard_bounds), one launch per shard per step.

A default N = 1 run (headline workload, no measurement switches) also times, after the headline region and outside it: the headline
workload over 1000 single-stream steps with time AND shader clock taken from those same launches (`secondary.cfg2`: the stable figure
rounds are compared on; its `cycles_per_launch` is the only one whose factors share launches), four independent batches per launch
(`secondary.cfg2_x4`, ss_mfcc_batches_device), the other BASELINE configurations -- `secondary.cfg3`, `secondary.cfg5` (1000 steps
each) and `secondary.cfg4` (the whole 360 000-clip corpus in one launch, 5 steps) -- and the headline workload once more with successive
steps going round four HIP streams (`value_pipelined`, with `pipelined.value_one_stream` = secondary.cfg2's value beside it: one launch's
tail under the next ones' heads; whole-job throughput, not a kernel duration and no part of `roofline`).

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch (4 B per input
sample + 4 B per output element) / average launch duration measured with HIP events
on the launch stream over the timed (path-only when N > 1) region.  `cpu_baseline` = the oracle's
reference-shaped single-thread f32 port (oracle/ss_oracle.c, "port") timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "mfcc-rust_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 64 FLOP/clk/SIMD
# algorithmic flop per output row (FFT 2.5 N log2 N + magnitudes + mel taps + logs + DCT)
ALGO_FLOP_PER_ROW = {"cfg2": 14.2e3, "cfg4": 14.2e3, "cfg3": 63.0e3, "cfg5": 157.0e3}

WORKLOADS = {
    # name: (description, params kwargs, clip samples, clips per GPU, kind)
    "cfg2": ("cfg2: 1024 x 1 s clips @16 kHz, MFCC n_fft=512 hop=160 n_mels=40 n_mfcc=13",
             dict(sample_rate=16000), 16000, 1024, "mfcc"),
    "cfg3": ("cfg3: 1024 x 1 s clips @16 kHz, mel_spectrogram n_fft=2048 hop=512 n_mels=128",
             dict(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128,
                  high_frequency=8000.0), 16000, 1024, "mel"),
    "cfg4": ("cfg4: 360 000 x 1 s clips @16 kHz (100 h corpus, split over the ranks), MFCC n_fft=512 hop=160 n_mels=40 n_mfcc=13",
             dict(sample_rate=16000), 16000, 360000, "mfcc"),
    "cfg5": ("cfg5: 512 x 1 s clips @44.1 kHz, MFCC n_fft=4096 hop=1024 n_mels=256 n_mfcc=40",
             dict(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100,
                  num_cepstral=40, num_filters=256, high_frequency=22050.0), 44100, 512, "mfcc"),
}


# The legs a default N = 1 run appends behind the headline (`secondary.<name>`): name -> (workload, measure_simple arguments).
# `python bench.py --leg <name>` runs one of them alone (what tools/profile.sh wraps in rocprofv3 for the batch-table builds).
#   cfg2: the headline workload and kernel over 1000 single-stream steps, run inside the library -- HIP events around the launches AND
#     the in-kernel stamps of ALL of those same launches (every wave's cycles / lifetime, summed on the device): the one
#     cycles_per_launch whose time and clock share launches -- stamping only the region's last quarter spreads 1.2 % between regions
#     on one box, stamping all of it 0.3 % (profiles/r06/stamp_cost.txt) --, and the figure rounds are compared on (a 20-step headline
#     region is 0.6 ms: the same binary reads 26 - 30 us there, profiles/r05/box_spread.txt).
#   cfg2_x4 / cfg2_x8 / cfg3_x4 / cfg5_x4: four (eight) independent batches of the workload per ss_mfcc_batches_device /
#     ss_mel_spectrogram_batches_device call (ONE persistent launch): what the start-up + tail of a launch cost, recovered without
#     streams; eight batches reach the one-launch corpus rate of cfg4.  Per-batch figures; never `value`.  The input ring grows
#     with the batches in flight (measure_simple), so that the Infinity Cache serves none of it.
#   cfg3 / cfg5: 1000 steps, 50 - 60 ms each (regions of 200 steps read 5 - 10 % slower on the same box, profiles/r05/secondary_probe.txt),
#     run inside the library like cfg2 (their twelve-wave builds stamp every wave's lifetime too: time and clock from the same launches)
#   cfg4: the whole 360 000-clip corpus in one launch, 5 steps
LEGS = {
    "cfg2": ("cfg2", dict(steps=1000, warmup=100, prewarm_ms=300.0, stamped=1000)),
    "cfg2_x4": ("cfg2", dict(steps=1000, warmup=100, prewarm_ms=100.0, group=4, probe_board=False)),
    "cfg2_x8": ("cfg2", dict(steps=1000, warmup=96, prewarm_ms=100.0, group=8, probe_board=False)),
    "cfg3": ("cfg3", dict(steps=1000, warmup=100, prewarm_ms=300.0, stamped=1000)),
    "cfg3_x4": ("cfg3", dict(steps=1000, warmup=100, prewarm_ms=100.0, group=4, probe_board=False)),
    "cfg5": ("cfg5", dict(steps=1000, warmup=100, prewarm_ms=300.0, stamped=1000)),
    "cfg5_x4": ("cfg5", dict(steps=1000, warmup=100, prewarm_ms=100.0, group=4, probe_board=False)),
    "cfg4": ("cfg4", dict(steps=5, warmup=1, prewarm_ms=300.0)),
}


def synth_batch(torch, batch, n, seed, device):
    """N(0, 0.1) clips (the distribution of the reference's own tests, lib.rs:18-22), generated on device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randn((batch, n), generator=g, device=device, dtype=torch.float32).mul_(0.1)


class BoardProbe:
    """Socket power and shader clock of this process's GPU, read from its hwmon directory (found by PCI address: a box has eight)
    every 10 ms on a thread while the untimed pre-roll launches run -- the timed region itself is too short to sample.  Context for
    `roofline.frac`: on random samples the kernels sit at the board's power cap and below its peak clock.  None where
    the files cannot be read."""

    def __init__(self, torch, device):
        import glob
        import threading

        self.samples, self.done, self.thread = [], False, None
        try:
            pr = torch.cuda.get_device_properties(device)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            hw = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            self.pw = [p for h in hw for n in ("power1_average", "power1_input") for p in glob.glob(f"{h}/{n}")]
            self.fq = [p for h in hw for p in glob.glob(f"{h}/freq1_input")]
            self.fq2 = [p for h in hw for p in glob.glob(f"{h}/freq2_input")]  # memory clock, where the driver exposes it
            self.cap = self._read([p for h in hw for p in glob.glob(f"{h}/power1_cap")])
        except Exception:
            self.pw, self.fq, self.fq2, self.cap = [], [], [], None
        if self.pw or self.fq:
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

    @staticmethod
    def _read(paths):
        try:
            return int(open(paths[0]).read().split()[0])
        except Exception:
            return None

    def _run(self):
        while not self.done:
            self.samples.append((self._read(self.pw), self._read(self.fq), self._read(self.fq2) if self.fq2 else None))
            time.sleep(0.01)

    def stop(self):
        self.done = True
        if not self.thread:
            return None
        self.thread.join()
        tail = self.samples[len(self.samples) // 2:]  # the second half: the sensors lag the load by tens of milliseconds
        p = [a / 1e6 for a, _, _ in tail if a]
        f = [b / 1e6 for _, b, _ in tail if b]
        m = [c / 1e6 for _, _, c in tail if c]
        if not p and not f:
            return None
        return {"power_w_mean": round(sum(p) / len(p)) if p else None, "power_w_max": round(max(p)) if p else None,
                "power_cap_w": round(self.cap / 1e6) if self.cap else None,
                "sclk_mhz_mean": round(sum(f) / len(f)) if f else None, "sclk_mhz_min": round(min(f)) if f else None,
                "mclk_mhz_mean": round(sum(m) / len(m)) if m else None, "mclk_mhz_min": round(min(m)) if m else None,
                "samples": len(tail), "when": "second half of the untimed pre-roll of the same launches (hwmon, 10 ms period)"}


def timed_port(kind, p, probe_clip):
    """The port function bench.py times: the build for THIS host (oracle/Makefile `native`: -O3 -march=native, no fast-math, no FMA
    contraction -- compiled here, now, by gcc), checked against the f64 checker on one clip before it is timed; the portable -O2
    build of libss_oracle.so when the native build is not possible.  Returns (callable, flags text)."""
    import functools

    import numpy as np

    import oracle_c

    base = oracle_c.port_mfcc if kind == "mfcc" else oracle_c.port_mel_spectrogram
    nat, flags = oracle_c.native_port()
    if nat is None:
        return base, f"portable build (gcc -O2 -ffp-contract=off; {flags})"
    fn = functools.partial(base, from_lib=nat)
    want = (oracle_c.mfcc if kind == "mfcc" else oracle_c.mel_spectrogram)(p, probe_clip)
    got = fn(p, probe_clip)
    err = float(np.abs(got - want).max() / np.abs(want).max())
    if not err <= 1e-4:
        # never take the measured GPU line down with it: time the portable build instead and say so
        return base, f"portable build (gcc -O2 -ffp-contract=off; the native build [{flags}] disagreed with the f64 checker: rel err {err:.1e})"
    return fn, flags + f"; checked against the f64 oracle on one clip first (rel err {err:.1e})"


def cpu_baseline(kind, pkw, n_samples, budget_s=12.0):
    """Time the oracle's reference-shaped f32 port, single thread, on fresh clips until ~budget_s."""
    import numpy as np

    import oracle_c

    p = oracle_c.make_params(**pkw)
    rng = np.random.default_rng(1234)
    rows_per_clip = oracle_c.num_frames(p, n_samples) if kind == "mfcc" else oracle_c.stft_rows(p, n_samples)[0]
    pool = (rng.standard_normal((64, n_samples)) * 0.1).astype(np.float32)
    fn, flags = timed_port(kind, p, pool[0])
    fn(p, pool[0])  # warm
    clips, t0 = 0, time.perf_counter()
    while True:
        fn(p, pool[clips % 64])
        clips += 1
        el = time.perf_counter() - t0
        if el >= budget_s or clips >= 200000:
            break
    return {
        "value": clips * rows_per_clip / el,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "note": "lower bound on the Rust crate: a textbook radix-2 FFT with a per-call plan where rustfft picks SIMD mixed-radix plans; "
                "build flags: " + flags,
        "sample": f"{clips} x {n_samples}-sample clips ({clips * rows_per_clip} frames) in {el:.1f} s, single thread, "
                  f"oracle/ss_oracle.c port_{'mfcc' if kind == 'mfcc' else 'mel_spectrogram'}_f32 built with [{flags.split(';')[0]}]; "
                  f"host has {os.cpu_count()} logical cores",
    }


def cpu_baseline_all_cores(kind, pkw, n_samples, budget_s=6.0):
    """The same port on every host core at once (ctypes releases the GIL): what a data-parallel CPU run of the reference
    path could reach on this box.  Extra information next to the contract's single-thread `cpu_baseline`."""
    import threading

    import numpy as np

    import oracle_c

    p = oracle_c.make_params(**pkw)
    rows_per_clip = oracle_c.num_frames(p, n_samples) if kind == "mfcc" else oracle_c.stft_rows(p, n_samples)[0]
    cores = os.cpu_count() or 1
    pool = (np.random.default_rng(4321).standard_normal((16, n_samples)) * 0.1).astype(np.float32)
    fn, _ = timed_port(kind, p, pool[0])
    fn(p, pool[0])
    counts = [0] * cores
    stop = time.perf_counter() + budget_s

    def work(i):
        k = 0
        while time.perf_counter() < stop:
            fn(p, pool[(i + k) % 16])
            k += 1
        counts[i] = k

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    el = time.perf_counter() - t0
    clips = sum(counts)
    return {"value": clips * rows_per_clip / el, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{clips} clips in {el:.1f} s on {cores} threads (one per logical core), same port as cpu_baseline"}


def load_profile(kernel_name, workload, headline):
    """Stored PMC figures of this kernel + workload from profiles/pmc_traffic.json (written by tools/make_traffic_json.py
    from a rocprofv3 --pmc run of this same command; NOT measured in this run).  A headline configuration whose kernel is
    not the profiled one fails loudly instead of reporting null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    none = {"traffic": None, "traffic_source": "none (this configuration has no stored PMC profile)"}
    if not headline:
        return none
    with open(path) as f:
        d = json.load(f)
    e = d.get(workload)
    if not e:
        return none
    if kernel_name.split("<")[0] not in e.get("kernel_full", ""):
        raise SystemExit(f"bench.py: profiles/pmc_traffic.json holds {e.get('kernel_full')!r} for {workload}, but the launch ran "
                         f"{kernel_name!r}: re-run tools/profile.sh + tools/make_traffic_json.py")
    out = {"traffic": e.get("hbm_bytes_per_launch"),
           "traffic_source": "profiles/pmc_traffic.json (stored: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                             f"{e.get('profiled', 'round 1')}; 2 x FETCH_SIZE + WRITE_SIZE per launch)"}
    # the instruction count is a property of the kernel binary, not of the profiled run: it is what this run's VALU-floor
    # fraction is computed from (with this run's duration and this run's clock).  The profiled run's own shares (LDS busy,
    # wave-time shares, ...) stay in profiles/: they belong to that run's duration, not to this line.
    if "valu_insts_per_launch" in e:
        out["valu_insts_per_launch"] = e["valu_insts_per_launch"]
    return out


def load_pmc(workload):
    """Stored PMC entry of a workload (profiles/pmc_traffic.json) or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(workload)
    except Exception:
        return None


def probe_clock(torch, lib, step, avg_s, device, probe_us=2000, want=3):
    """Shader clock (GHz) the device holds while `step` launches run: the median of `want` accepted readings of the library's
    one-wave probe (ss_shader_clock_probe_async: s_memtime against the 100 MHz s_memrealtime, asleep in between, a lead-in not
    counted).  Each probe is queued on a fresh side stream FIRST and enough launches to outlast it follow at once on the launch
    stream: the probe wave is resident before they start and reads the clock under their load.  A reading is ACCEPTED only if
    the launches took no longer than they do alone: HIP multiplexes streams onto a few hardware queues, and a side stream that
    shares the launch stream's queue makes the launches wait behind the sleeping probe, which then reads the idle clock (2.4 GHz;
    tools/clock_probe_ab.py, profiles/r05/clock_probe_check.txt).  Outside every timed region.  None if no reading was accepted."""
    n = max(8, int(2.5 * probe_us * 1e-6 / max(avg_s, 1e-6)))
    words = torch.zeros((3 * want, 2), dtype=torch.int64, device=device)
    got = []
    for t in range(3 * want):
        if len(got) >= want:
            break
        side = torch.cuda.Stream(device=device)
        for i in range(min(n, 200)):  # the load is established and the clock has settled before the probe goes out
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.ss_shader_clock_probe_async(C.c_void_p(side.cuda_stream), int(probe_us), C.c_void_p(words[t].data_ptr()))
        for i in range(n):
            step(i)
        torch.cuda.current_stream().synchronize()
        alone = (time.perf_counter() - t0) <= 1.2 * n * avg_s + 100e-6
        torch.cuda.synchronize()
        w = words[t].tolist()
        if rc == 0 and alone and w[1] > 0 and 0.3 < w[0] / (10.0 * w[1]) < 2.6:
            got.append(w[0] / (10.0 * w[1]))
    got.sort()
    return got[len(got) // 2] if got else None


def measure_simple(torch, ss_mod, workload, device, steps, warmup, prewarm_ms=300.0, streams=1, probe_board=True, ring_mib=300,
                   group=1, stamped=0):
    """One N = 1 measurement of a BASELINE configuration outside the headline region: `steps` launches of the hot path over batches
    resident in HBM (rotated over > 256 MiB of distinct inputs, like the headline), HIP events on the launch stream around them,
    wall clock between synchronize pairs.  streams > 1: successive steps alternate over that many streams (independent batches in
    flight), so the per-step figure is wall time, not a kernel duration.
    group > 1: each launch takes `group` independent batches of the workload at once (ss_mfcc_batches_device: one persistent launch
    over all of them); a step is still ONE batch, so `steps` must be a multiple of `group` and the per-step figures are launch / group.
    stamped > 0 (512-point MFCC kernel, one stream): the timed region runs inside the library (ss_mfcc_timed_region: the same
    launches, HIP events on the launch stream around them) and the shader clock comes from the in-kernel stamps of the last
    `stamped` of THOSE launches -- time and clock of `cycles_per_launch` share launches."""
    from speechsauce_amd import SpeechConfig, _lib, make_params

    lib = _lib.lib()
    desc, pkw, n_samples, clips, kind = WORKLOADS[workload]
    cfg = SpeechConfig(make_params(**pkw))
    if kind == "mfcc":
        rows = cfg.num_frames(n_samples)
        out_shape = (clips, rows, cfg.params.num_cepstral)
    else:
        rows, _ = cfg.stft_rows(n_samples)
        out_shape = (clips, cfg.params.num_filters, rows)
    out_elems = out_shape[0] * out_shape[1] * out_shape[2]
    bytes_per_launch = 4 * clips * n_samples + 4 * out_elems
    big = 4 * clips * n_samples > ring_mib * 2**20  # cfg4: one 23 GB batch is its own ring
    # With several batches in flight at once (`group` per launch, or `streams` launches side by side) the ring grows by that factor:
    # a 300 MiB ring is five 65 MB batches, and a launch of four of them re-reads three that its predecessor read ~260 MB ago -- at the
    # edge of the 256 MiB Infinity Cache, which then serves part of the input (cfg2, one box, four batches per launch: 25.2 - 25.8 us
    # per batch on a 300 MiB ring, 26.4 - 26.8 on 1200 or 2400 MiB; four streams 26.5 against 27.6 - 28.6; one batch per launch reads
    # 28.4 - 28.9 on every ring: profiles/r06/ring_check.txt).  Round 5's `value_pipelined` was measured on the small ring.
    in_flight = max(1, group, streams)
    n_buf = 1 if big else max(2, -(-ring_mib * in_flight * 2**20 // (4 * clips * n_samples)))
    xs = [synth_batch(torch, clips, n_samples, 7001 + i, device) for i in range(n_buf)]
    main = torch.cuda.current_stream()
    sts = [main] + [torch.cuda.Stream(device=device) for _ in range(max(0, streams - 1))]
    sptrs = [C.c_void_p(st.cuda_stream) for st in sts]
    outs = [torch.empty(out_shape, dtype=torch.float32, device=device) for _ in range(1 if big else max(2, 2 * streams))]
    fn = lib.ss_mfcc_batch_device if kind == "mfcc" else lib.ss_mel_spectrogram_device
    if group > 1:
        # launch k takes batches k*group .. k*group + group - 1 of the ring (distinct inputs, distinct output blocks)
        assert steps % group == 0 and warmup % group == 0 and streams == 1
        batches_fn = lib.ss_mfcc_batches_device if kind == "mfcc" else lib.ss_mel_spectrogram_batches_device
        outs = [torch.empty(out_shape, dtype=torch.float32, device=device) for _ in range(2 * group)]
        nb = (C.c_size_t * group)(*([clips] * group))
        tabs = []
        for k in range(max(n_buf, 2) * 2):
            px = (C.c_void_p * group)(*[xs[(k * group + g) % n_buf].data_ptr() for g in range(group)])
            po = (C.c_void_p * group)(*[outs[((k % 2) * group + g)].data_ptr() for g in range(group)])
            tabs.append((px, po))

        def step(i):  # one call = `group` steps; called for i = 0, group, 2 group, ...
            if i % group:
                return
            px, po = tabs[(i // group) % len(tabs)]
            rc = batches_fn(cfg.handle, group, px, nb, n_samples, n_samples, po, sptrs[0])
            if rc:
                _lib.check(rc)
    else:
        def step(i):
            rc = fn(cfg.handle, xs[i % n_buf].data_ptr(), clips, n_samples, n_samples, outs[i % len(outs)].data_ptr(), sptrs[i % len(sptrs)])
            if rc:
                _lib.check(rc)

    board = None
    if prewarm_ms > 0:
        probe = BoardProbe(torch, device) if probe_board else None
        t_end = time.perf_counter() + prewarm_ms * 1e-3
        k = 0
        while True:
            for _ in range(1 if big else 50 * group):
                step(k)
                k += 1
            torch.cuda.synchronize()
            if time.perf_counter() >= t_end:
                break
        if probe:
            board = probe.stop()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    clock_ghz = clock_source = None
    if stamped and streams == 1 and group == 1:
        # the library runs the region: the same launches (input ring, output blocks), events on the launch stream, per-wave stamps
        # of the last `stamped` of them
        px = (C.c_void_p * n_buf)(*[x.data_ptr() for x in xs])
        po = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        ms, ghz, wall = C.c_float(0.0), C.c_float(0.0), C.c_float(0.0)
        region = lib.ss_mfcc_timed_region if kind == "mfcc" else lib.ss_mel_spectrogram_timed_region
        rc = region(cfg.handle, px, n_buf, clips, n_samples, n_samples, po, len(outs), sptrs[0], steps, stamped,
                    C.byref(ms), C.byref(ghz), C.byref(wall))
        torch.cuda.synchronize()
        if rc:
            _lib.check(rc)
        elapsed = wall.value * 1e-3  # host clock inside the call: first launch .. last launch done (not the stamps' read-back)
        dev_s = ms.value * 1e-3 * steps
        if ghz.value > 0:
            clock_ghz = float(ghz.value)
            clock_source = (f"in-kernel stamps of {'all' if stamped >= steps else 'the last ' + str(stamped) + ' of'} the {steps} timed launches themselves "
                            f"(ss_{'mfcc' if kind == 'mfcc' else 'mel_spectrogram'}_timed_region)")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for st in sts[1:]:  # the other streams start behind the start event
            st.wait_event(e0)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        for st in sts[1:]:
            ev = torch.cuda.Event()
            ev.record(st)
            main.wait_event(ev)
        e1.record(main)
        while not e1.query():
            pass
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        dev_s = e0.elapsed_time(e1) * 1e-3
    kernel = lib.ss_last_kernel_name().decode()
    avg = dev_s / steps
    if clock_ghz is None and streams == 1:
        # (group > 1: one call per `group` steps)
        clock_ghz = probe_clock(torch, lib, (lambda i: step(i * group)) if group > 1 else step, avg * group, device)
        clock_source = "a one-wave probe beside FURTHER launches of the same step right after the timed ones (ss_shader_clock_probe)"
    del xs, outs
    torch.cuda.empty_cache()
    res = {
        "workload": desc, "kernel": kernel, "steps": steps, "warmup": warmup, "streams": streams,
        "metric": "mfcc_frames_per_sec" if kind == "mfcc" else "mel_rows_per_sec",
        "value": clips * rows * steps / elapsed, "ms_per_step": elapsed * 1e3 / steps,
        "avg_launch_us": avg * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch,
        "achieved": bytes_per_launch / avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_per_launch / avg / 1e9 / HBM_PEAK_GBS,
        "frames_per_launch": clips * rows,
    }
    if group > 1:
        # a launch covers `group` batches: the per-launch figures above are per BATCH (region / steps); the launch itself is group x
        res["batches_per_launch"] = group
        res["launch_us"] = avg * 1e6 * group
        res["note"] = (f"{group} independent batches of the workload per ss_{'mfcc' if kind == 'mfcc' else 'mel_spectrogram'}_batches_device call = one "
                       f"persistent launch; avg_launch_us, bytes and frac are per {clips}-clip batch (launch / group): the start-up and the one-unit "
                       f"tail of a launch are paid once per {group} batches")
    e = load_pmc(workload if group == 1 else f"{workload}_x{group}")
    # (the batch-table builds report themselves as <kernel>m<...>; rocprofv3 lists them as one more instantiation of <kernel><...>)
    kbase = kernel.split("<")[0]
    if group > 1 and kbase.endswith("m"):
        kbase = kbase[:-1]
    if e and kbase in e.get("kernel_full", ""):
        # (a launch of `group` batches: the stored per-launch counters are divided down to one batch, like every figure of this leg)
        res["traffic"] = e.get("hbm_bytes_per_launch") / group if e.get("hbm_bytes_per_launch") else None
        res["traffic_source"] = (f"profiles/pmc_traffic.json (stored, {e.get('profiled', '?')}; 2 x FETCH_SIZE + WRITE_SIZE per launch"
                                 + (f", / {group} batches" if group > 1 else "") + ")")
        if e.get("valu_insts_per_launch"):
            res["valu_insts_per_launch"] = e["valu_insts_per_launch"] / group
    else:
        res["traffic"] = None
    if board:
        res["board"] = board
        if board.get("power_w_mean") and streams == 1:
            # socket energy per launch (mean power of the pre-roll of these same launches x this region's time per launch): at the
            # power cap, time = energy / cap -- the figure a kernel change has to move
            res["energy_mj_per_launch"] = board["power_w_mean"] * avg * 1e3
    if clock_ghz:
        # shader cycles per launch at the clock a probe wave read beside these same launches right after the timed ones (the hwmon
        # figure in `board` reads up to 10 % higher): the figure to compare across boxes that hold different clocks at the power cap
        res["clock_ghz_measured"] = clock_ghz
        res["clock_source"] = clock_source
        res["cycles_per_launch"] = avg * 1e9 * clock_ghz
        if res.get("valu_insts_per_launch"):
            res["valu_floor_frac"] = res["valu_insts_per_launch"] / 1024.0 * 2.14 / res["cycles_per_launch"]
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE set) and
    relay rank 0's JSON line.  Runs before this process imports torch or touches the GPU; nothing is re-exec'd."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SS_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL needs it on this driver)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU (default: the workload's)")
    ap.add_argument("--gather-mode", default="root", choices=["root", "all", "none"],
                    help="N > 1: gather of the output blocks to rank 0 (grouped send / recv), to every rank (all-gather), or no collective")
    ap.add_argument("--gather", action="store_true", help="(kept for old command lines; the gather is on by default for --gpus > 1)")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather-mode none")
    ap.add_argument("--gather-every", type=int, default=8, help="steps per gather bucket (fewer, larger collectives)")
    ap.add_argument("--force-generic", action="store_true", help="run on the generic kernel (ss_debug_force_generic): the 'generic us' columns of DESIGN.md")
    ap.add_argument("--prewarm-ms", type=float, default=500.0,
                    help="untimed launches before the warm-up steps: leaves the idle power state (200 ms do) and lets the board's power sensor settle for roofline.board")
    ap.add_argument("--streams", type=int, default=1, help="issue successive steps round-robin on this many HIP streams "
                    "(independent batches in flight: one launch's tail overlaps the next one's head); the roofline block "
                    "is then per-step wall time, not a kernel duration -- not the headline setting")
    ap.add_argument("--params", default="", help='JSON dict of extra ss_params switches, e.g. \'{"mfcc_window": 1, "preemph_coef": 0.97}\' (not the headline config)')
    ap.add_argument("--kind", default="", choices=["", "mfcc", "mel"], help="run the workload's clips through the other path (mfcc / mel_spectrogram); not the headline config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="single-thread CPU baseline budget (the all-cores run takes half of it)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (cfg2 over 1000 steps, the batch-table legs, cfg3, cfg5, cfg4) and value_pipelined that a default N = 1 run appends")
    ap.add_argument("--corpus-clips", type=int, default=0, help="cfg4: size of the corpus that is split over the ranks (default 360 000); keeps strong scaling, unlike --clips")
    ap.add_argument("--gather-chunks", type=int, default=8, help="cfg4 with a gather: a rank's shard is computed and gathered in this many chunks, "
                    "chunk i's collective under chunk i+1's kernel")
    ap.add_argument("--ring-mib", type=int, default=300, help="size of the rotated input ring (measurement aid: below 256 MiB the Infinity Cache serves the input)")
    ap.add_argument("--one-device", action="store_true", help="test aid: every rank uses device 0 (gloo instead of RCCL; control flow only)")
    ap.add_argument("--link-gbps", type=float, default=64.0, help="one-way xGMI rate per link assumed by scaling_model (unmeasured here)")
    ap.add_argument("--leg", default="", choices=[""] + sorted(LEGS), help="run ONE of the default run's secondary legs alone and print its JSON "
                    "(an explicit --steps / --warmup replaces the leg's own; measurement aid: rocprofv3 passes of the batch-table builds)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    if args.gpus == 1 and not args.no_cpu_baseline:
        # the CPU leg's port is compiled for this host NOW, before anything touches the GPU (a compiler run is a child process;
        # none is started once the device is initialised), not in the middle of the run
        import oracle_c

        oracle_c.native_port()

    import torch
    import torch.distributed as dist

    import speechsauce_amd as ss
    from speechsauce_amd import SpeechConfig, _lib, make_params

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if n_dev == 0:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    shared_device = world > n_dev or args.one_device  # several ranks per GPU: control-flow check only
    local_rank = 0 if args.one_device else local_rank % n_dev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL needs it on this driver)
        if not shared_device:
            dist.init_process_group(backend="nccl", device_id=device)  # RCCL; a failure here is fatal, not papered over
            backend = "nccl"
        else:  # RCCL refuses two ranks on one device: the control flow (sharding, buckets, barriers) runs over gloo
            dist.init_process_group(backend="gloo")
            backend = "gloo"
            print(f"[bench] rank {rank}: {world} ranks share {n_dev} device(s): gloo instead of RCCL, timings are not a scaling result",
                  file=sys.stderr, flush=True)
    gather_mode = "none" if (world == 1 or args.no_gather) else args.gather_mode
    do_gather = gather_mode != "none"
    if do_gather and args.streams > 1:
        # a bucket's hand-off to the side stream is ordered by events on ONE launch stream; with steps spread over several
        # streams the gather could read slots a kernel is still writing
        raise SystemExit("bench.py: --streams > 1 cannot be combined with the gather (use --gather-mode none)")

    desc, pkw, n_samples, clips, kind = WORKLOADS[args.workload]
    if args.params:
        pkw = dict(pkw, **json.loads(args.params))
        desc += " + " + args.params
    if args.kind and args.kind != kind:
        kind = args.kind
        desc += " through the " + ("mel_spectrogram" if kind == "mel" else "mfcc") + " path"
    strong = args.workload == "cfg4" and not args.clips
    corpus = shard_max = None
    if args.clips:
        clips = args.clips
    elif strong:  # fixed corpus: this rank's contiguous shard
        from speechsauce_amd.distributed import shard_bounds, shard_sizes

        corpus = args.corpus_clips or clips
        desc = desc.replace("360 000", f"{corpus:,}".replace(",", " ")) if args.corpus_clips else desc
        lo, hi = shard_bounds(corpus, world, rank)
        clips = hi - lo
        shard_max = max(shard_sizes(corpus, world))
        if clips == 0:
            raise SystemExit("bench.py: --corpus-clips is smaller than the number of ranks")
    if args.leg:
        wl, kw = LEGS[args.leg]
        kw = dict(kw)
        if "--steps" in sys.argv:
            g = kw.get("group", 1)
            kw["steps"] = max(g, args.steps // g * g)
            if kw.get("stamped"):
                kw["stamped"] = kw["steps"]
        if "--warmup" in sys.argv:
            g = kw.get("group", 1)
            kw["warmup"] = args.warmup // g * g
        print(json.dumps(dict(measure_simple(torch, ss, wl, device, **kw), leg=args.leg)), flush=True)
        return
    if args.force_generic:
        # a process-wide kernel-selection override: a test aid of the LAB library (include/speechsauce_amd_debug.h), so this
        # measurement aid runs the whole bench on that build
        _lib._lib = _lib.lab()
        _lib._lib.ss_debug_force_generic(1)
    cfg = SpeechConfig(make_params(**pkw))
    lib = _lib.lib()

    if kind == "mfcc":
        rows = cfg.num_frames(n_samples)
        out_shape = (clips, rows, cfg.params.num_cepstral)
    else:
        rows, _ = cfg.stft_rows(n_samples)
        out_shape = (clips, cfg.params.num_filters, rows)
    out_elems = out_shape[0] * out_shape[1] * out_shape[2]
    bytes_per_launch = 4 * clips * n_samples + 4 * out_elems  # algorithmic: input once + output once
    frames_per_launch = clips * rows

    # distinct input batches totalling > 256 MiB so the Infinity Cache cannot hold the stream
    ring_mib = args.ring_mib * max(1, args.streams)  # (several launches side by side: see measure_simple)
    n_buf = max(1 if strong else 2, -(-ring_mib * 1024 * 1024 // (4 * clips * n_samples)))
    xs = [synth_batch(torch, clips, n_samples, 1 + rank * 100 + i, device) for i in range(n_buf)]
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    extra_streams = [torch.cuda.Stream(device=device) for _ in range(max(0, args.streams - 1))]
    sptrs = [sptr] + [C.c_void_p(st.cuda_stream) for st in extra_streams]
    outs = [torch.empty(out_shape, dtype=torch.float32, device=device) for _ in range(max(2, 2 * args.streams))]

    # The gather (north-star: "RCCL gather over xGMI of the final [n_frames x n_mfcc] blocks"): the outputs of
    # `gather_every` consecutive steps land in one bucket [G, clips, rows, ceps]; a full bucket is gathered into
    # [world, G, ...] (on rank 0, or on every rank) on a side stream while the following steps compute into the other bucket.
    from speechsauce_amd.distributed import all_gather_into, gather_into

    # cfg4 (strong): a rank's shard is computed AND gathered in CH chunks -- chunk c's collective runs on the side stream under chunk
    # c+1's kernel, so only the last chunk's transfer is exposed (one block of 229 MB per rank at N = 8 would take ~3 ms over a link
    # after 1.1 ms of compute).  Shards are padded to the largest one for the collective (360 000 divides evenly by 1, 2, 4, 8).
    G = 1 if strong else max(1, args.gather_every)
    CH = max(1, min(args.gather_chunks, shard_max)) if strong else 1
    buckets = gathered = comm_stream = chunk_bounds = None
    if do_gather:
        if strong:
            from speechsauce_amd.distributed import shard_bounds as _sb

            chunk_bounds = [_sb(shard_max, CH, c) for c in range(CH)]
            buckets = [torch.empty((shard_max,) + out_shape[1:], dtype=torch.float32, device=device) for _ in range(2)]
            if gather_mode == "all" or rank == 0:
                gathered = [torch.empty((world, shard_max) + out_shape[1:], dtype=torch.float32, device=device) for _ in range(2)]
        else:
            buckets = [torch.empty((G,) + out_shape, dtype=torch.float32, device=device) for _ in range(2)]
            if gather_mode == "all" or rank == 0:
                gathered = [torch.empty((world * G,) + out_shape, dtype=torch.float32, device=device) for _ in range(2)]
        comm_stream = torch.cuda.Stream(device=device)
    row_elems = out_shape[1] * out_shape[2]

    class Region:
        """One timed (or warm-up) sequence of steps; with_gather routes the outputs through the buckets and the collective."""

        def __init__(self, with_gather):
            self.g = bool(with_gather and do_gather)
            self.bucket_done = [None, None]
            self.comm_events = []

        def out_for(self, i):
            if not self.g:
                return outs[i % len(outs)]
            b = (i // G) % 2
            if i % G == 0 and self.bucket_done[b] is not None:  # the bucket's previous gather must have read it
                stream.wait_event(self.bucket_done[b])
            return buckets[b][i % G]

        def after_step(self, i, last):
            """Launch the gather of a bucket once its last step has been issued."""
            if not self.g or strong or not ((i + 1) % G == 0 or last):
                return
            b = (i // G) % 2
            n = i % G + 1  # filled slots (a run's last bucket may be partial: only what was computed is gathered)
            ev = torch.cuda.Event()
            ev.record(stream)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev)
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record(comm_stream)
                if gather_mode == "all":
                    all_gather_into(gathered[b][: world * n], buckets[b][:n])
                else:
                    gather_into(gathered[b][: world * n] if rank == 0 else None, buckets[b][:n], dst=0)
                c1.record(comm_stream)
            self.comm_events.append((c0, c1, n))
            self.bucket_done[b] = c1

        def launch(self, x_ptr, n, o_ptr, sp):
            fn = lib.ss_mfcc_batch_device if kind == "mfcc" else lib.ss_mel_spectrogram_device
            rc = fn(cfg.handle, x_ptr, n, n_samples, n_samples, o_ptr, sp)
            if rc:
                _lib.check(rc)

        def step_chunked(self, i):
            """cfg4 with a gather: the shard in CH chunks, each chunk's gather issued right behind its kernel."""
            b = i % 2
            x = xs[i % n_buf]
            if self.bucket_done[b] is not None:  # the bucket's previous gathers must have read it
                stream.wait_event(self.bucket_done[b])
            for c0, c1 in chunk_bounds:
                n = min(c1, clips) - c0  # a shorter shard computes what it has and sends the padded chunk
                if n > 0:
                    self.launch(x.data_ptr() + 4 * c0 * n_samples, n, buckets[b].data_ptr() + 4 * c0 * row_elems, sptr)
                ev = torch.cuda.Event()
                ev.record(stream)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(ev)
                    e_a, e_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e_a.record(comm_stream)
                    parts = [gathered[b][r, c0:c1] for r in range(world)] if gathered is not None else None
                    if gather_mode == "all":
                        all_gather_into(None, buckets[b][c0:c1], parts=parts)
                    else:
                        gather_into(None, buckets[b][c0:c1], dst=0, parts=parts)
                    e_b.record(comm_stream)
                self.comm_events.append((e_a, e_b, 1.0 / CH))
                self.bucket_done[b] = e_b
            return buckets[b]

        def step(self, i):
            if self.g and strong:
                return self.step_chunked(i)
            x, o, sp = xs[i % n_buf], self.out_for(i), sptrs[i % len(sptrs)]
            if kind == "mfcc":
                rc = lib.ss_mfcc_batch_device(cfg.handle, x.data_ptr(), clips, n_samples, n_samples, o.data_ptr(), sp)
            else:
                rc = lib.ss_mel_spectrogram_device(cfg.handle, x.data_ptr(), clips, n_samples, n_samples, o.data_ptr(), sp)
            if rc:
                _lib.check(rc)
            return o

        def drain(self):
            if comm_stream is not None:
                comm_stream.synchronize()
            torch.cuda.synchronize()

    def run_timed(with_gather, steps):
        """barrier + synchronize, EXACTLY `steps` steps, barrier + synchronize; wall time is the MAX over ranks."""
        reg = Region(with_gather)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # ten intermediate events on the launch stream: median / spread of the per-launch time over tenths of the timed region
        seg_every = max(1, steps // 10)
        seg_events = []
        e0.record(stream)
        t0 = time.perf_counter()
        for i in range(steps):
            reg.step(i)
            reg.after_step(i, i + 1 == steps)
            if args.streams == 1 and steps >= 200 and (i + 1) % seg_every == 0 and i + 1 < steps:  # (an event between launches costs ~1 us)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(stream)
                seg_events.append((i + 1, ev))
        for st in extra_streams:  # the end event on the launch stream waits for the other streams' work
            ev = torch.cuda.Event()
            ev.record(st)
            stream.wait_event(ev)
        e1.record(stream)
        while not e1.query():  # poll instead of sleeping in the driver: the blocking wait's wake-up latency is tens of microseconds,
            pass               # which is several steps' worth when only a few steps are timed
        reg.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        marks = [(0, e0)] + seg_events + [(steps, e1)]
        segs = sorted(a_ev.elapsed_time(b_ev) * 1e3 / (b_i - a_i) for (a_i, a_ev), (b_i, b_ev) in zip(marks[:-1], marks[1:]) if b_i > a_i)
        comm_ms = [c0.elapsed_time(c1) for c0, c1, _ in reg.comm_events]
        comm_steps = sum(n for _, _, n in reg.comm_events)
        return {"elapsed": elapsed, "dev_ms": e0.elapsed_time(e1), "segs": segs, "comm_ms": comm_ms, "comm_steps": comm_steps}

    warm = Region(True)
    board = None
    if args.prewarm_ms > 0:  # leave the idle power state before the (possibly few) warm-up steps
        pre = Region(False)
        probe = BoardProbe(torch, device) if rank == 0 else None  # socket power / shader clock while these launches run
        t_end = time.perf_counter() + args.prewarm_ms * 1e-3
        k = 0
        while time.perf_counter() < t_end:
            for _ in range(50):
                pre.step(k)
                k += 1
            torch.cuda.synchronize()
        if probe:
            board = probe.stop()
    for i in range(args.warmup):  # warm-up runs the full step (with the collective when there is one)
        warm.step(i)
        warm.after_step(i, i + 1 == args.warmup)
    warm.drain()

    # N > 1 with a gather: the path alone first, then the path with the collective -- both K steps, both in this run
    t_path = run_timed(False, args.steps)
    t_full = run_timed(True, args.steps) if do_gather else t_path
    elapsed, dev_ms, segs = t_full["elapsed"], t_path["dev_ms"], t_path["segs"]

    # Shader clock during the launches (512-point MFCC kernel only): thirty more launches right behind the timed ones through the
    # library's per-call diagnostic ss_mfcc_shader_clock (per-wave lifetime in shader cycles / on the 100 MHz clock, written into
    # a buffer that call owns)
    clock_ghz = None
    if world == 1 and kind == "mfcc" and lib.ss_last_kernel_name().decode().startswith("ss_mfcc_c256<") and args.streams == 1:
        g = C.c_float(0.0)
        torch.cuda.synchronize()
        rc = lib.ss_mfcc_shader_clock(cfg.handle, xs[0].data_ptr(), clips, n_samples, n_samples, outs[0].data_ptr(), sptr, 30, C.byref(g))
        if rc == 0 and g.value > 0:
            clock_ghz = float(g.value)
    clock_source = "in-kernel stamps of the same launches (ss_mfcc_shader_clock)"
    if clock_ghz is None and world == 1 and args.streams == 1:
        # every other kernel: the library's one-wave probe beside more launches of the same step (ss_shader_clock_probe)
        clock_ghz = probe_clock(torch, lib, lambda i: Region(False).step(i), dev_ms * 1e-3 / args.steps, device)
        clock_source = "one-wave probe beside the same launches (ss_shader_clock_probe)"

    if rank == 0:
        kernel = lib.ss_last_kernel_name().decode()
        total_frames = (corpus * rows if strong else frames_per_launch * world) * args.steps  # strong: the corpus once per step, whatever the shard sizes
        value = total_frames / elapsed
        avg_launch_s = dev_ms * 1e-3 / args.steps
        achieved = bytes_per_launch / avg_launch_s / 1e9
        gather_info = None
        if do_gather:
            cm = sorted(t_full["comm_ms"])
            step_s = t_full["elapsed"] / args.steps
            gather_info = {
                "mode": gather_mode,
                "collective": "ncclAllGather (all_gather_into_tensor)" if gather_mode == "all" else "grouped ncclSend/ncclRecv to rank 0 (torch.distributed.gather)",
                "bucket_steps": G,
                "chunks_per_step": CH,
                "bytes_per_rank_per_step": 4 * out_elems,
                "bytes_into_root_per_step" if gather_mode == "root" else "bytes_received_per_rank_per_step": 4 * out_elems * (world - 1),
                # what one direct link (peer -> root, or peer -> peer in the all-gather) carried per second of the timed region
                "achieved_gbps_per_link": 4 * out_elems / step_s / 1e9,
                # the collective's own duration on the side stream (rank 0's events), per step of output it moved
                "collective_ms_per_step_median": (cm[len(cm) // 2] * len(cm) / max(1, t_full["comm_steps"])) if cm else None,
                "collectives_timed": len(cm),
                "hardware": "unmeasured on a multi-GPU box until a SCALE record exists" if shared_device else "this run",
            }
        res = {
            "metric": "mfcc_frames_per_sec" if kind == "mfcc" else "mel_rows_per_sec",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic N(0,0.1) clips generated on device, resident in HBM; "
                    f"{n_buf} distinct input batches rotated ({n_buf * 4 * clips * n_samples / 2**20:.0f} MiB > Infinity Cache)",
            "config": {
                "workload": desc,
                "clips_per_gpu": clips,
                "samples_per_clip": n_samples,
                "frames_per_clip": rows,
                "parallelism": f"clip-sharded x{world}" + ((f" + gather of the output blocks ({gather_mode}) over {backend} in "
                                                               + (f"{CH} chunks per shard" if strong else f"buckets of {G} steps") + ", overlapped")
                                                              if do_gather else ", no collective")
                               + (f", {args.streams} streams" if args.streams > 1 else ""),
            },
            "backend": backend,
            "rccl_ranks": world if backend == "nccl" else 0,
            "ranks_share_a_device": bool(shared_device) if world > 1 else False,
            # N > 1: the same K steps without the collective, timed in this run just before the region `value` comes from
            "value_path_only": total_frames / t_path["elapsed"],
            "ms_per_step_path_only": t_path["elapsed"] * 1e3 / args.steps,
            "gather": gather_info,
            "prewarm_ms": args.prewarm_ms,
            "real_time_factor": value / rows * (n_samples / pkw["sample_rate"]),
            "roofline": {
                "bound": "hbm",
                "kernel": kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                **load_profile(kernel, args.workload, headline=not (args.params or args.kind or args.clips or args.force_generic)),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_us": avg_launch_s * 1e6,
                "launch_us_median_min_max_over_tenths": [segs[len(segs) // 2], segs[0], segs[-1]],
                "frames_per_sec_kernel_only": frames_per_launch / avg_launch_s,
                # secondary ceiling: FP32 vector peak (MI355X_MICROARCH.md: 157.3 TFLOP/s) against the algorithmic flop count of SURVEY 8d
                "valu": ({"algorithmic_flop_per_launch": ALGO_FLOP_PER_ROW[args.workload] * frames_per_launch,
                          "achieved_tflops": ALGO_FLOP_PER_ROW[args.workload] * frames_per_launch / avg_launch_s / 1e12,
                          "peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
                          "frac": ALGO_FLOP_PER_ROW[args.workload] * frames_per_launch / avg_launch_s / 1e12 / FP32_VECTOR_PEAK_TFLOPS}
                         if not (args.params or args.kind) else None),
            },
        }
        if world > 1:
            # How to read an N > 1 line (UNMEASURED ON HARDWARE until a SCALE record exists): the path shards with no collective, so
            # `value_path_only` is N independent runs (efficiency against N x this rank's kernel-only rate: launch gaps + the barrier's
            # skew); with the gather every rank's features cross ONE direct xGMI link to the root (root mode) or to each peer (all),
            # and a rank makes features faster than a link carries them -- `value` is bounded by N links x link rate / bytes per frame.
            out_bytes_per_frame = 4.0 * out_shape[2] if kind == "mfcc" else 4.0 * out_shape[1]
            kernel_only = frames_per_launch / avg_launch_s
            res["scaling_model"] = {
                "path_only_efficiency": res["value_path_only"] / (world * kernel_only) if not strong else res["value_path_only"] / (corpus * rows / avg_launch_s),
                "path_only_efficiency_against": "N x rank 0's kernel-only frames/s of this run (HIP events)",
                "feature_bytes_per_frame": out_bytes_per_frame,
                "features_gbps_per_rank_at_kernel_speed": kernel_only * out_bytes_per_frame / 1e9,
                "link_gbps_assumed_one_way": args.link_gbps,
                "gather_bound_frames_per_s": world * args.link_gbps * 1e9 / out_bytes_per_frame,
                "gather_bound_efficiency": min(1.0, args.link_gbps * 1e9 / out_bytes_per_frame / kernel_only),
                "value_over_gather_bound": value / (world * args.link_gbps * 1e9 / out_bytes_per_frame),
                # one word for whoever reads the first SCALE record: is `value` held by the links into the root, or by the path?
                # ("interconnect": the gather bound lies below what the ranks' kernels produce, or `value` sits within 15 % of it)
                "bound": (("interconnect" if (world * args.link_gbps * 1e9 / out_bytes_per_frame < (corpus * rows / avg_launch_s if strong else world * kernel_only)
                                              or value >= 0.85 * world * args.link_gbps * 1e9 / out_bytes_per_frame) else "path")
                          if do_gather else "path"),
                "note": "root mode: (N-1) peers each fill one direct link into rank 0; the bound counts N producers at one link's rate each "
                        "(rank 0's own block needs no link, so it is slightly pessimistic); unmeasured on hardware",
            }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(kind, pkw, n_samples, args.cpu_seconds)
            res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(kind, pkw, n_samples, args.cpu_seconds / 2)
        if board:
            res["roofline"]["board"] = board
            if board.get("power_w_mean") and args.streams == 1:
                res["roofline"]["energy_mj_per_launch"] = board["power_w_mean"] * avg_launch_s * 1e3  # (see measure_simple)
        headline = world == 1 and args.workload == "cfg2" and args.streams == 1 and not (
            args.params or args.kind or args.clips or args.force_generic or args.no_secondary or args.ring_mib != 300)
        if clock_ghz is not None:
            rf = res["roofline"]
            rf["clock_ghz_measured"] = clock_ghz
            rf["clock_source"] = clock_source
            if headline:
                # The clock above was read on LATER launches than the ones `avg_launch_us` times (and a 20-step region is 0.6 ms): their
                # product is not a cycle count of anything.  The coherent figure -- time and clock from the same 1000 launches -- is
                # secondary.cfg2.cycles_per_launch below.
                rf["cycles_per_launch"] = None
                rf["cycles_per_launch_see"] = "secondary.cfg2 (time and clock from the same launches)"
            else:
                # shader cycles per launch: the figure to compare between boxes / rounds (boxes hold 1.9-2.25 GHz at the power cap).
                # Long regions only (--steps >= a few hundred): the clock is read on further launches right behind the timed ones
                rf["cycles_per_launch"] = avg_launch_s * 1e9 * clock_ghz
                if rf.get("valu_insts_per_launch"):
                    # ONE VALU-issue-floor fraction: the kernel's instruction count (2.14 cycles per instruction per SIMD, 1024 SIMDs;
                    # tools/ubench/valu_issue.hip) over THIS run's launch duration at the clock THIS device held in THIS run
                    rf["valu_floor_frac"] = rf["valu_insts_per_launch"] / 1024.0 * 2.14 / (avg_launch_s * 1e9 * clock_ghz)
        if headline:
            # the other BASELINE configurations and the pipelined headline, after and outside the headline's timed region
            del xs[:], outs[:]
            torch.cuda.empty_cache()
            # Everything below is extra: none of it may take the measured headline down with it (each leg reports {"error": ...}).
            res["secondary"] = {}
            for name, (wl, kw) in LEGS.items():
                try:
                    res["secondary"][name] = measure_simple(torch, ss, wl, device, **kw)
                except Exception as e:
                    res["secondary"][name] = {"error": repr(e)}
            # value_pipelined: the headline workload once more with successive steps going round four HIP streams, 1000 steps with the
            # settings of secondary.cfg2 -- `value_one_stream` beside it is that leg's value (same steps, same ring, one stream), so the
            # ratio of the two is streams and nothing else (+8 - 11 % on input rings that grow with the batches in flight; round 5's
            # +10 - 16 %, profiles/r05/streams.txt, came from the fixed 300 MiB ring).
            try:
                pl = measure_simple(torch, ss, "cfg2", device, steps=1000, warmup=100, prewarm_ms=100.0, streams=4, probe_board=False)
                res["value_pipelined"] = pl["value"]
                one = res["secondary"]["cfg2"].get("value")
                res["pipelined"] = {"streams": 4, "steps": pl["steps"], "ms_per_step": pl["ms_per_step"], "kernel": pl["kernel"],
                                    "value_one_stream": one, "over_one_stream": (pl["value"] / one) if one else None,
                                    "note": "same workload, successive steps go round four HIP streams (independent batches): a launch's one-unit tail "
                                            "and the next ones' wait for their first samples overlap; wall time per step over 1000 steps, not a kernel "
                                            "duration, not part of `roofline`; value_one_stream = secondary.cfg2 (same step count, one stream; the input "
                                            "ring is 300 MiB per batch in flight in both): compare with THAT, not with the 20-step headline; outputs of this concurrent use are checked bit for "
                                            "bit against serial launches in tests/test_concurrency.py"}
            except Exception as e:
                res["value_pipelined"] = None
                res["pipelined"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
