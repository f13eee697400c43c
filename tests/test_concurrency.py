"""Concurrent use of ONE config handle: several HIP streams, several host threads.

The reference's SpeechConfig is `!Sync` (RefCell + STFT carry-over state, config.rs:126-130) and its functions run one call at a
time; include/speechsauce_amd.h promises more -- "a config handle is immutable after creation ... and may be used from several
threads / streams concurrently" -- and bench.py's `value_pipelined` (four streams on one handle) times exactly that.  These tests
own the promise: every output of launches that overlap on the device must equal, BIT FOR BIT, the output of the same launch issued
alone on one stream, on the three kernel builds the bench times (asserted by name), and sampled clips must match the oracle
(semantics per clip: feature.rs:99-174).
"""
import ctypes as C
import threading

import numpy as np
import pytest

from common import BENCH_KERNELS

pytestmark = pytest.mark.gpu

RTOL = 1e-4
STEPS = 64     # launches per workload
STREAMS = 4
DISTINCT = 16  # distinct input batches per workload (step i reads batch i % DISTINCT and writes its own output block)

# name: (params, clip samples, clips per launch, path) -- the shapes of bench.py's WORKLOADS
WL = {
    "cfg2": (dict(sample_rate=16000), 16000, 1024, "mfcc"),
    "cfg3": (dict(sample_rate=16000, fft_points=2048, frame_length=0.032, frame_stride=0.032, num_filters=128,
                  high_frequency=8000.0), 16000, 1024, "mel"),
    "cfg5": (dict(sample_rate=44100, fft_points=4096, frame_length=4096 / 44100, frame_stride=1024 / 44100,
                  num_cepstral=40, num_filters=256, high_frequency=22050.0), 44100, 512, "mfcc"),
}


def _rel(got, want):
    return float(np.abs(got.astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


class Work:
    """One workload: ONE config handle, DISTINCT input batches in HBM, one output block per step for each of the two runs."""

    def __init__(self, name, seed):
        import torch
        from speechsauce_amd import SpeechConfig, make_params

        self.name = name
        pkw, self.n, self.clips, self.kind = WL[name]
        self.pkw = pkw
        self.cfg = SpeechConfig(make_params(**pkw))
        if self.kind == "mfcc":
            self.shape = (self.clips, self.cfg.num_frames(self.n), self.cfg.params.num_cepstral)
        else:
            self.shape = (self.clips, self.cfg.params.num_filters, self.cfg.stft_rows(self.n)[0])
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        # amplitudes differ from batch to batch: a launch that read another step's input cannot pass for the right one
        self.xs = [torch.randn((self.clips, self.n), generator=g, device="cuda", dtype=torch.float32).mul_(0.05 + 0.01 * i)
                   for i in range(DISTINCT)]
        self.serial = [torch.full(self.shape, float("nan"), device="cuda") for _ in range(STEPS)]
        self.conc = [torch.full(self.shape, float("nan"), device="cuda") for _ in range(STEPS)]

    def launch(self, lib, i, out, stream_ptr):
        fn = lib.ss_mfcc_batch_device if self.kind == "mfcc" else lib.ss_mel_spectrogram_device
        return fn(self.cfg.handle, self.xs[i % DISTINCT].data_ptr(), self.clips, self.n, self.n, out[i].data_ptr(), stream_ptr)

    def check_against_oracle(self, oracle, outs):
        p = oracle.make_params(**self.pkw)
        for i in (0, 21, 42, 63):
            b = (i * 37) % self.clips
            x = self.xs[i % DISTINCT][b].cpu().numpy()
            want = oracle.mfcc(p, x) if self.kind == "mfcc" else oracle.mel_spectrogram(p, x[None, :])[0]
            assert _rel(outs[i][b].cpu().numpy(), want) <= RTOL, (self.name, i, b)


@pytest.fixture(scope="module")
def works(ss, sslib):
    import torch

    ws = [Work("cfg2", 101), Work("cfg3", 102), Work("cfg5", 103)]
    # the reference run: every step alone on ONE stream, one after the other
    for w in ws:
        for i in range(STEPS):
            assert w.launch(sslib, i, w.serial, None) == 0, sslib.ss_last_error_string()
        assert sslib.ss_last_kernel_name() == BENCH_KERNELS[w.name], sslib.ss_last_kernel_name()
    torch.cuda.synchronize()
    for w in ws:
        w.cfg.device_status()
        assert all(bool(torch.isfinite(o).all()) for o in w.serial)
    yield ws
    del ws
    torch.cuda.empty_cache()


def test_serial_reference_matches_the_oracle(works, oracle):
    for w in works:
        w.check_against_oracle(oracle, w.serial)
        # distinct inputs gave distinct outputs (the comparison below cannot pass on a stale block)
        assert not bool((w.serial[0] == w.serial[1]).all())


def test_four_streams_on_one_handle_equal_the_serial_run_bit_for_bit(works, sslib, oracle):
    """What bench.py's `value_pipelined` does (successive steps go round four streams), on all three bench kernels AT ONCE: step i of
    workload w goes to stream (i + w) % 4 -- each handle is used from all four streams, and launches of different kernels overlap
    on the device as well (3 x 64 launches, nothing synchronised in between)."""
    import torch

    streams = [torch.cuda.Stream() for _ in range(STREAMS)]
    ptrs = [C.c_void_p(s.cuda_stream) for s in streams]
    torch.cuda.synchronize()
    for i in range(STEPS):
        for k, w in enumerate(works):
            assert w.launch(sslib, i, w.conc, ptrs[(i + k) % STREAMS]) == 0, sslib.ss_last_error_string()
            assert sslib.ss_last_kernel_name() == BENCH_KERNELS[w.name]
    for s in streams:
        s.synchronize()
    for w in works:
        w.cfg.device_status()  # no kernel reported a protocol error (the error word the handle shares between its launches)
        for i in range(STEPS):
            assert torch.equal(w.conc[i], w.serial[i]), (w.name, i)
        w.check_against_oracle(oracle, w.conc)


def test_four_streams_same_workload_back_to_back(works, sslib):
    """The pipelined bench loop itself: ONE workload, 64 launches round-robin over four streams, for each bench kernel in turn
    (launches of the same kernel and the same handle overlap: shared tables and the handle's error word)."""
    import torch

    streams = [torch.cuda.Stream() for _ in range(STREAMS)]
    ptrs = [C.c_void_p(s.cuda_stream) for s in streams]
    for w in works:
        for o in w.conc:
            o.fill_(float("nan"))
        torch.cuda.synchronize()
        for i in range(STEPS):
            assert w.launch(sslib, i, w.conc, ptrs[i % STREAMS]) == 0, sslib.ss_last_error_string()
        for s in streams:
            s.synchronize()
        w.cfg.device_status()
        for i in range(STEPS):
            assert torch.equal(w.conc[i], w.serial[i]), (w.name, i)


def test_four_host_threads_share_the_handles(works, sslib, oracle):
    """Four host threads, each with a stream of its own, call ss_mfcc_batch_device / ss_mel_spectrogram_device on the SAME three
    handles (ctypes releases the GIL inside the calls: they run in parallel); a fifth thread keeps provoking SS_ERR_SHORT_SIGNAL on
    one of those handles meanwhile.  Every output equals the serial run bit for bit; the failing thread reads its own error text,
    the working threads never see it (ss_last_error_string and ss_last_kernel_name are thread-local)."""
    import torch

    for w in works:
        for o in w.conc:
            o.fill_(float("nan"))
    torch.cuda.synchronize()
    dev = torch.cuda.current_device()
    streams = [torch.cuda.Stream() for _ in range(STREAMS)]
    start = threading.Barrier(STREAMS + 1)
    stop = threading.Event()
    problems = []

    def worker(t):
        try:
            torch.cuda.set_device(dev)  # the current device is per thread
            sp = C.c_void_p(streams[t].cuda_stream)
            # a failing call first: this thread's error text is now a known one of its own ("null argument")
            assert sslib.ss_num_frames(None, 0, None) == 3
            mine = sslib.ss_last_error_string()
            assert mine == b"null argument", mine
            start.wait()
            for i in range(t, STEPS, STREAMS):
                for w in works:
                    rc = w.launch(sslib, i, w.conc, sp)
                    if rc != 0:
                        problems.append((t, w.name, i, rc, sslib.ss_last_error_string()))
                    if sslib.ss_last_kernel_name() != BENCH_KERNELS[w.name]:
                        problems.append((t, w.name, i, "kernel", sslib.ss_last_kernel_name()))
                    if sslib.ss_last_error_string() != mine:  # another thread's failure must not show up here
                        problems.append((t, w.name, i, "error text changed", sslib.ss_last_error_string()))
            streams[t].synchronize()
        except Exception as e:  # noqa: BLE001
            problems.append((t, "exception", repr(e)))

    failures = [0]

    def saboteur():
        try:
            torch.cuda.set_device(dev)
            w = works[0]
            side = torch.cuda.Stream()
            scratch = torch.empty(w.shape, device="cuda")
            start.wait()
            while not stop.is_set():
                # 100 samples < one 320-sample frame: the reference underflows and panics (processing.rs:101), here SS_ERR_SHORT_SIGNAL
                rc = sslib.ss_mfcc_batch_device(w.cfg.handle, w.xs[0].data_ptr(), 4, 100, 16000, scratch.data_ptr(), C.c_void_p(side.cuda_stream))
                txt = sslib.ss_last_error_string()
                if rc != 1 or not txt:
                    problems.append(("saboteur", rc, txt))
                    break
                failures[0] += 1
        except Exception as e:  # noqa: BLE001
            problems.append(("saboteur", "exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(STREAMS)]
    sab = threading.Thread(target=saboteur)
    for th in threads + [sab]:
        th.start()
    for th in threads:
        th.join()
    stop.set()
    sab.join()
    torch.cuda.synchronize()
    assert not problems, problems[:5]
    assert failures[0] > 0
    for w in works:
        w.cfg.device_status()  # the short-signal calls launched nothing and left no device-side error behind
        for i in range(STEPS):
            assert torch.equal(w.conc[i], w.serial[i]), (w.name, i)
        w.check_against_oracle(oracle, w.conc)


def test_batch_table_launches_from_four_threads(works, sslib):
    """ss_mfcc_batches_device (several batches in one launch of the batch-table build) from four host threads on one handle, each on a
    stream of its own: the kernel-argument table is per launch, so concurrent calls cannot see each other's blocks -- every block
    equals the serial single-batch run bit for bit."""
    import torch

    w = works[0]  # cfg2
    for o in w.conc:
        o.fill_(float("nan"))
    torch.cuda.synchronize()
    dev = torch.cuda.current_device()
    streams = [torch.cuda.Stream() for _ in range(STREAMS)]
    problems = []

    def worker(t):
        try:
            torch.cuda.set_device(dev)
            sp = C.c_void_p(streams[t].cuda_stream)
            for g in range(t, STEPS // 4, STREAMS):  # group g = steps 4g .. 4g + 3
                idx = list(range(4 * g, 4 * g + 4))
                px = (C.c_void_p * 4)(*[w.xs[i % DISTINCT].data_ptr() for i in idx])
                po = (C.c_void_p * 4)(*[w.conc[i].data_ptr() for i in idx])
                nb = (C.c_size_t * 4)(*([w.clips] * 4))
                rc = sslib.ss_mfcc_batches_device(w.cfg.handle, 4, px, nb, w.n, w.n, po, sp)
                if rc != 0 or sslib.ss_last_kernel_name() != b"ss_mfcc_c256m<10,exact,bank421,sym>":
                    problems.append((t, g, rc, sslib.ss_last_error_string(), sslib.ss_last_kernel_name()))
            streams[t].synchronize()
        except Exception as e:  # noqa: BLE001
            problems.append((t, "exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(STREAMS)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not problems, problems[:5]
    w.cfg.device_status()
    for i in range(STEPS):
        assert torch.equal(w.conc[i], w.serial[i]), i


def test_host_pointer_calls_from_threads_on_one_handle(ss, sslib, oracle):
    """The synchronous host-pointer entry points (ss_mfcc: H2D + kernel + D2H on the handle's private streams) from four threads on
    one handle: serialised by the handle's mutex, every result equals the single-threaded one bit for bit."""
    from speechsauce_amd import SpeechConfig, make_params

    cfg = SpeechConfig(make_params(sample_rate=16000))
    rng = np.random.default_rng(77)
    clips = (rng.standard_normal((32, 16000)) * 0.1).astype(np.float32)
    T = cfg.num_frames(16000)
    want = np.empty((32, T, 13), np.float32)
    for b in range(32):
        assert sslib.ss_mfcc(cfg.handle, clips[b].ctypes.data, 16000, want[b].ctypes.data) == 0
    got = np.full((32, T, 13), np.nan, np.float32)
    bad = []

    def worker(t):
        import torch

        torch.cuda.set_device(0)
        for b in range(t, 32, 4):
            rc = sslib.ss_mfcc(cfg.handle, clips[b].ctypes.data, 16000, got[b].ctypes.data)
            if rc:
                bad.append((t, b, rc, sslib.ss_last_error_string()))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not bad, bad
    assert np.array_equal(got, want)
    p = oracle.make_params(sample_rate=16000)
    for b in (0, 13, 31):
        assert _rel(got[b], oracle.mfcc(p, clips[b])) <= RTOL
