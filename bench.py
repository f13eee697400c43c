#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native speechsauce hot path.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|cfg4|cfg5] [--gather]

Metric (BASELINE.json): frames/sec (+ real-time factor) for 16 kHz MFCC n_fft=512 at 1/2/4/8 MI355X.
A "step" is one pass of the hot path over one batch of synthetic clips that already sit in HBM:
cfg2 = 1024 x 1 s clips @16 kHz, SpeechConfig defaults (n_fft 512, hop 160, 40 mels, 13 ceps),
one fused kernel launch per step.  To keep the 256 MiB Infinity Cache from serving the input,
steps rotate over enough distinct input batches to exceed it (8 x 65.5 MB for cfg2).

Multi-GPU: one process per GPU (torch.distributed, backend nccl == RCCL).  Clips are independent,
so every rank processes its own batch of the same size (weak scaling) and there is no data-path
collective; --gather adds an RCCL all-gather of the [frames x n_mfcc] blocks, overlapped on a
side stream, for the north-star's "gather over xGMI" variant.  cfg4 (the 100 h corpus, 360 000
clips) is the one strong-scaling workload: the corpus is split into contiguous clip shards
(speechsauce_amd.distributed.shard_bounds), one per rank, one launch per shard per step.

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch (4 B per input
sample + 4 B per output element) / average launch duration measured with HIP events
on the launch stream over the timed region.  `cpu_baseline` = the oracle's reference-shaped
single-thread f32 port (oracle/ss_oracle.c, "port") timed on this host on a bounded sample.
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


def synth_batch(torch, batch, n, seed, device):
    """N(0, 0.1) clips (the distribution of the reference's own tests, lib.rs:18-22), generated on device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randn((batch, n), generator=g, device=device, dtype=torch.float32).mul_(0.1)


def cpu_baseline(kind, pkw, n_samples, budget_s=12.0):
    """Time the oracle's reference-shaped f32 port, single thread, on fresh clips until ~budget_s."""
    import numpy as np

    import oracle_c

    p = oracle_c.make_params(**pkw)
    rng = np.random.default_rng(1234)
    fn = oracle_c.port_mfcc if kind == "mfcc" else oracle_c.port_mel_spectrogram
    rows_per_clip = oracle_c.num_frames(p, n_samples) if kind == "mfcc" else oracle_c.stft_rows(p, n_samples)[0]
    pool = (rng.standard_normal((64, n_samples)) * 0.1).astype(np.float32)
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
        "sample": f"{clips} x {n_samples}-sample clips ({clips * rows_per_clip} frames) in {el:.1f} s, single thread, "
                  f"oracle/ss_oracle.c port_{'mfcc' if kind == 'mfcc' else 'mel_spectrogram'}_f32; host has {os.cpu_count()} logical cores",
    }


def cpu_baseline_all_cores(kind, pkw, n_samples, budget_s=6.0):
    """The same port on every host core at once (ctypes releases the GIL): what a data-parallel CPU run of the reference
    path could reach on this box.  Extra information next to the contract's single-thread `cpu_baseline`."""
    import threading

    import numpy as np

    import oracle_c

    p = oracle_c.make_params(**pkw)
    fn = oracle_c.port_mfcc if kind == "mfcc" else oracle_c.port_mel_spectrogram
    rows_per_clip = oracle_c.num_frames(p, n_samples) if kind == "mfcc" else oracle_c.stft_rows(p, n_samples)[0]
    cores = os.cpu_count() or 1
    pool = (np.random.default_rng(4321).standard_normal((16, n_samples)) * 0.1).astype(np.float32)
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


def load_traffic(kernel_name, workload):
    """HBM bytes per launch from the committed PMC profile of this kernel+workload (or None)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        e = d.get(workload)
        if e and kernel_name.split("<")[0] in e.get("kernel_full", ""):
            return e.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU (default: the workload's)")
    ap.add_argument("--gather", action="store_true", help="add an overlapped RCCL all-gather of the outputs")
    ap.add_argument("--streams", type=int, default=1, help="issue successive steps round-robin on this many HIP streams "
                    "(independent batches in flight: one launch's tail overlaps the next one's head); the roofline block "
                    "is then per-step wall time, not a kernel duration -- not the headline setting")
    ap.add_argument("--params", default="", help='JSON dict of extra ss_params switches, e.g. \'{"mfcc_window": 1, "preemph_coef": 0.97}\' (not the headline config)')
    ap.add_argument("--kind", default="", choices=["", "mfcc", "mel"], help="run the workload's clips through the other path (mfcc / mel_spectrogram); not the headline config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import speechsauce_amd as ss
    from speechsauce_amd import SpeechConfig, _lib, make_params

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    if os.environ.get("SS_BENCH_ONE_DEVICE"):  # test aid: several ranks on one GPU (exercises the multi-rank control flow)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL needs it on this driver)
        try:
            dist.init_process_group(backend="nccl", device_id=device)  # RCCL
        except Exception as e:  # the control plane (barrier, max over ranks) also works over gloo; --gather needs RCCL
            if args.gather:
                raise
            print(f"[bench] rank {rank}: RCCL init failed ({e}); using gloo for the barriers", file=sys.stderr, flush=True)
            dist.init_process_group(backend="gloo")

    desc, pkw, n_samples, clips, kind = WORKLOADS[args.workload]
    if args.params:
        pkw = dict(pkw, **json.loads(args.params))
        desc += " + " + args.params
    if args.kind and args.kind != kind:
        kind = args.kind
        desc += " through the " + ("mel_spectrogram" if kind == "mel" else "mfcc") + " path"
    strong = args.workload == "cfg4"
    if args.clips:
        clips = args.clips
    elif strong:  # fixed corpus: this rank's contiguous shard
        from speechsauce_amd.distributed import shard_bounds

        lo, hi = shard_bounds(clips, world, rank)
        clips = hi - lo
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
    n_buf = max(1 if strong else 2, -(-300 * 1024 * 1024 // (4 * clips * n_samples)))
    xs = [synth_batch(torch, clips, n_samples, 1 + rank * 100 + i, device) for i in range(n_buf)]
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    extra_streams = [torch.cuda.Stream(device=device) for _ in range(max(0, args.streams - 1))]
    sptrs = [sptr] + [C.c_void_p(st.cuda_stream) for st in extra_streams]
    outs = [torch.empty(out_shape, dtype=torch.float32, device=device) for _ in range(max(2, 2 * args.streams))]

    def step(i):
        x, o, sp = xs[i % n_buf], outs[i % len(outs)], sptrs[i % len(sptrs)]
        if kind == "mfcc":
            rc = lib.ss_mfcc_batch_device(cfg.handle, x.data_ptr(), clips, n_samples, n_samples, o.data_ptr(), sp)
        else:
            rc = lib.ss_mel_spectrogram_device(cfg.handle, x.data_ptr(), clips, n_samples, n_samples, o.data_ptr(), sp)
        if rc:
            _lib.check(rc)
        return o

    gather_bufs, comm_stream, pending = None, None, []
    if args.gather and world > 1:
        gather_bufs = [torch.empty((world,) + out_shape, dtype=torch.float32, device=device) for _ in range(2)]
        comm_stream = torch.cuda.Stream(device=device)

    def gather(i, o):
        # all-gather of step i's block on a side stream; overlaps the next step's kernel
        ev = torch.cuda.Event()
        ev.record(stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            dist.all_gather_into_tensor(gather_bufs[i % 2], o)
            done = torch.cuda.Event()
            done.record(comm_stream)
        pending.append(done)
        if len(pending) > 1:  # the buffer pair is reused two steps later
            stream.wait_event(pending.pop(0))

    for i in range(args.warmup):
        o = step(i)
        if gather_bufs is not None:
            gather(i, o)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # ten intermediate events on the launch stream: median / spread of the per-launch time over tenths of the timed region
    seg_every = max(1, args.steps // 10)
    seg_events = []
    t0 = time.perf_counter()
    e0.record(stream)
    for i in range(args.steps):
        o = step(i)
        if gather_bufs is not None:
            gather(i, o)
        if args.streams == 1 and (i + 1) % seg_every == 0 and i + 1 < args.steps:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            seg_events.append((i + 1, ev))
    for st in extra_streams:  # the end event on the launch stream waits for the other streams' work
        ev = torch.cuda.Event()
        ev.record(st)
        stream.wait_event(ev)
    e1.record(stream)
    if comm_stream is not None:
        comm_stream.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)  # HIP events on the launch stream over the timed region

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        kernel = lib.ss_last_kernel_name().decode()
        total_frames = frames_per_launch * args.steps * world  # equal shards (360 000 divides by 1, 2, 4, 8)
        value = total_frames / elapsed
        avg_launch_s = dev_ms * 1e-3 / args.steps
        achieved = bytes_per_launch / avg_launch_s / 1e9
        marks = [(0, e0)] + seg_events + [(args.steps, e1)]
        segs = sorted(a_ev.elapsed_time(b_ev) * 1e3 / (b_i - a_i) for (a_i, a_ev), (b_i, b_ev) in zip(marks[:-1], marks[1:]) if b_i > a_i)
        res = {
            "metric": "mfcc_frames_per_sec" if kind == "mfcc" else "mel_rows_per_sec",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong and not args.clips else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic N(0,0.1) clips generated on device, resident in HBM; "
                    f"{n_buf} distinct input batches rotated ({n_buf * 4 * clips * n_samples / 2**20:.0f} MiB > Infinity Cache)",
            "config": {
                "workload": desc,
                "clips_per_gpu": clips,
                "samples_per_clip": n_samples,
                "frames_per_clip": rows,
                "parallelism": f"clip-sharded x{world}" + (" + RCCL all-gather (overlapped)" if gather_bufs is not None else ", no collective")
                               + (f", {args.streams} streams" if args.streams > 1 else ""),
            },
            "real_time_factor": value / rows * (n_samples / pkw["sample_rate"]),
            "roofline": {
                "bound": "hbm",
                "kernel": kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_traffic(kernel, args.workload),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_us": avg_launch_s * 1e6,
                "launch_us_median_min_max_over_tenths": [segs[len(segs) // 2], segs[0], segs[-1]],
                "frames_per_sec_kernel_only": frames_per_launch / avg_launch_s,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(kind, pkw, n_samples, args.cpu_seconds)
            res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(kind, pkw, n_samples, args.cpu_seconds / 2)
        print(json.dumps(res), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
