#!/usr/bin/env python3
"""Secondary measurements for the other benchmark configurations (the one
JSON line of bench.py stays on the 256-track workload).  Run on the GPU box:

    python tools/bench_configs.py [c2] [c4] [c5]

c2  stereo impulse-reverb-stereo, 4 bars of 96 000 frames @ 48 kHz, 2 s stereo IR: us per bar
    (device-resident wp_reverb_block_dev, and through the host-pointer API)
c4  mix-buffers master reduce: S stereo stems x 60 s @ 48 kHz into one stereo bus (pure HBM-bound)
c5  per-GPU share of the 2048-track render: 256 stereo tracks x 60 s x 4 s IR (P = 47 partitions)
c5ir  the same with an impulse response per channel-job
pcie  the bench workload with the hand-over included: inputs start in pinned host memory, every
      go() result and the bus end in pinned host memory (never the headline figure)
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from dsp_amd import capi, device as dev


def timed(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def c2():
    L = M = 96000
    irs = np.stack([dev.synth_ir(M, 500000), dev.synth_ir(M, 500001)])
    ir = dev.DeviceIR(torch.from_numpy(irs).cuda(), L)
    job_ch = (C.c_int32 * 2)(0, 1)
    rv = capi.check_handle(capi.lib().wp_reverb_create(ir.handle, 2, job_ch, 0), "wp_reverb_create")
    x = torch.empty((2, L), dtype=torch.float32, device="cuda")
    dev.synth_noise(x[0], 1000)
    dev.synth_noise(x[1], 1001)
    curr, stride, v = C.c_void_p(), C.c_int64(), capi.WpReverbView()

    def bar():
        capi.check(capi.lib().wp_reverb_block_dev(rv, C.c_void_p(x.data_ptr()), x.stride(0), L, C.byref(curr), C.byref(stride),
                                                  C.byref(v), None), "wp_reverb_block_dev")
    t_dev = timed(bar, 200, 10)
    capi.lib().wp_reverb_destroy(rv)
    # host-pointer path (what the JavaScript shim calls): includes H2D of 2 x L and D2H of 2 x (L+M) floats
    from dsp_amd import impulse

    class Ctx:  # the surface of src/context.js the wrapper uses
        def __init__(self, buffer):
            self.buffer, self.n, self.store = buffer, 0, {}

        def sample(self, url):
            return [irs[0], irs[1]]

        def get(self, k):
            return self.store.get(k, False)

        def set(self, k, v, ttl=None):
            self.store[k] = v
    ctx = Ctx([x[0].cpu().numpy(), x[1].cpu().numpy()])
    fn = impulse.impulse_reverb_stereo(ctx, "ir")
    t_host = timed(lambda: fn(ctx), 50, 5)
    return {"config": "c2 stereo reverb, L=M=96000", "us_per_bar_device_resident": t_dev * 1e6,
            "us_per_bar_host_api": t_host * 1e6, "realtime_factor_host_api": (L / 48000.0) / t_host}


def c4(S=4096):
    F = 2880000
    free, _ = torch.cuda.mem_get_info()
    need = S * 2 * F * 4
    if need > free * 0.9:
        S = int(free * 0.9 // (2 * F * 4))
    stems = torch.empty((S, 2, F), dtype=torch.float32, device="cuda")
    flat = stems.view(-1)
    # one noise fill per stem channel would be 8192 launches; fill in 64 big chunks instead
    n = flat.numel()
    step = (n + 63) // 64
    for i in range(0, n, step):
        dev.synth_noise(flat[i:i + step], 7, i % (1 << 31))
    bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
    mix = dev.DeviceMix([{"ch": [stems[s, 0], stems[s, 1]], "vol": 1.0 / 4096} for s in range(S)])
    t = timed(lambda: mix.run([bus[0], bus[1]]), 5, 1)
    alg = F * (8 * S + 16)
    return {"config": "c4 master reduce, %d stereo stems x 60 s" % S, "ms": t * 1e3, "GBs": alg / t / 1e9,
            "frac_of_8TBs": alg / t / 8e12, "stem_Msamples_per_s": S * 2 * F / t / 1e6, "bytes": alg}


def c5(tracks=256, per_track_ir=False):
    F, M = 2880000, 192000
    njobs = 2 * tracks
    nir = njobs if per_track_ir else 2  # an impulse response per channel-job, or one shared stereo IR
    irs = np.stack([dev.synth_ir(M, 500000 + c) for c in range(nir)])
    ir = dev.DeviceIR(torch.from_numpy(irs).cuda(), F)
    conv = dev.BatchedConvolver(ir, [j % nir for j in range(njobs)])
    x = torch.empty((njobs, F), dtype=torch.float32, device="cuda")
    for j in range(njobs):
        dev.synth_noise(x[j], 1000 + j)
    bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
    vol = [1.0 / 2048] * tracks
    conv.profile(True)

    def step():
        bus.zero_()
        conv.run_mix(x, [bus[0], bus[1]], vol)
    t = timed(step, 3, 1)
    f, m, i = conv.profile_read()
    return {"config": "c5 per-GPU share: %d stereo tracks x 60 s x 4 s IR (P=%d), %s" % (
                tracks, ir.info(4), "%d per-track IRs (%.0f MB of spectra)" % (nir, ir.info(7) / 1e6) if per_track_ir else "shared stereo IR"),
            "ms_per_step": t * 1e3,
            "Msamples_per_s": njobs * (F + M) / t / 1e6, "kernel_ms": {"fwd": f, "mac(2 groups)": m, "inv_mix": i}}


def pcie(tracks=256):
    F, M = 2880000, 96000
    njobs = 2 * tracks
    irs = np.stack([dev.synth_ir(M, 500000), dev.synth_ir(M, 500001)])
    ir = dev.DeviceIR(torch.from_numpy(irs).cuda(), F)
    conv = dev.BatchedConvolver(ir, [j % 2 for j in range(njobs)])
    x = torch.empty((njobs, F), dtype=torch.float32, device="cuda")
    for j in range(njobs):
        dev.synth_noise(x[j], 1000 + j)
    xh = torch.empty((njobs, F), dtype=torch.float32).pin_memory()
    xh.copy_(x)
    out = torch.empty((njobs, F + M), dtype=torch.float32, device="cuda")
    outh = torch.empty((njobs, F + M), dtype=torch.float32).pin_memory()
    bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
    bush = torch.empty((2, F), dtype=torch.float32).pin_memory()
    vol = [1.0 / 256] * tracks

    def step():
        x.copy_(xh, non_blocking=True)
        bus.zero_()
        conv.run_and_mix(x, out, [bus[0], bus[1]], vol)
        outh.copy_(out, non_blocking=True)
        bush.copy_(bus, non_blocking=True)
    t = timed(step, 3, 1)

    def up():
        x.copy_(xh, non_blocking=True)
    tu = timed(up, 3, 1)
    return {"config": "bench workload with PCIe hand-over (pinned host memory, copies and compute serialised on one stream)",
            "ms_per_step": t * 1e3, "Msamples_per_s": njobs * (F + M) / t / 1e6,
            "h2d_GBs": x.numel() * 4 / tu / 1e9, "bytes_up": x.numel() * 4, "bytes_down": (out.numel() + bus.numel()) * 4}


if __name__ == "__main__":
    torch.cuda.set_device(0)
    which = sys.argv[1:] or ["c2", "c4", "c5"]
    for w in which:
        print(json.dumps({"c2": c2, "c4": c4, "c5": c5, "c5ir": lambda: c5(per_track_ir=True), "pcie": pcie}[w]()), flush=True)
        torch.cuda.empty_cache()
