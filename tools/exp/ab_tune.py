#!/usr/bin/env python3
"""A/B of a wp_tune knob on the bench workload, interleaved rounds in ONE process:
    python tools/exp/ab_tune.py --knob 4 --values 1,2 [--config c3|c5] [--rounds 5]
Prints per value the median/min of the three kernel times (HIP events inside the library) and of the step."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", type=int, required=True)
    ap.add_argument("--values", default="1,2")
    ap.add_argument("--config", default="c3")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--tracks", type=int, default=256)
    ap.add_argument("--recreate", action="store_true", help="the knob only affects handles created afterwards")
    ap.add_argument("--tol", type=float, default=0.0, help="allowed max abs difference between values (0: bit-identical)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from dsp_amd import capi
    if os.environ.get("WPDSP_EXP_LIB"):
        capi.LIB_PATH = os.environ["WPDSP_EXP_LIB"]
    from dsp_amd import device as dev
    F = 2880000
    M = 96000 if a.config == "c3" else 192000
    njobs = 2 * a.tracks
    if a.config == "c3":
        irs = np.stack([dev.synth_ir(M, 500000), dev.synth_ir(M, 500001)])
        job_ch = [j % 2 for j in range(njobs)]
    else:
        irs = np.stack([dev.synth_ir(M, 500000 + j) for j in range(njobs)])
        job_ch = list(range(njobs))
    irs_d = torch.from_numpy(irs).cuda()
    x = torch.empty((njobs, F), dtype=torch.float32, device="cuda")
    for j in range(njobs):
        dev.synth_noise(x[j], 1000 + j)
    out = torch.empty((njobs, (F + M + 3) & ~3), dtype=torch.float32, device="cuda")
    bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
    vol = [1.0 / 256] * a.tracks
    values = [int(v) for v in a.values.split(",")]
    plans = {}

    def plan(v):
        if v not in plans or a.recreate:
            capi.check(capi.lib().wp_tune(a.knob, v))
            ir = dev.DeviceIR(irs_d, F)
            conv = dev.BatchedConvolver(ir, job_ch)
            conv.profile(True)
            if a.recreate:
                return ir, conv
            plans[v] = (ir, conv)
        return plans[v]

    res = {v: [] for v in values}
    ref = None
    worst = 0.0
    for r in range(a.rounds + 1):
        for v in values:
            ir, conv = plan(v)
            capi.check(capi.lib().wp_tune(a.knob, v))
            bus.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            conv.run_and_mix(x, out, [bus[0], bus[1]], vol)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if r > 0:
                res[v].append(conv.profile_read() + (dt,))
            if ref is None:
                ref = (bus.clone(), out[0].clone())
            elif a.tol == 0.0:
                assert torch.equal(ref[0].view(torch.int32), bus.view(torch.int32)) and torch.equal(ref[1].view(torch.int32), out[0].view(torch.int32)), "results differ between knob values"
            else:
                worst = max(worst, float((ref[0] - bus).abs().max()), float((ref[1] - out[0]).abs().max()))
                assert worst <= a.tol, "results differ by %.3e between knob values" % worst
            if a.recreate:
                conv.close()
                ir.close()
    capi.lib().wp_tune(a.knob, 0)
    for v in values:
        cols = list(zip(*res[v]))
        print("knob %d = %d: " % (a.knob, v) + "  ".join("%s med %.3f min %.3f" % (n, statistics.median(c), min(c))
                                                          for n, c in zip(("fwd", "mac", "inv", "step"), cols)), flush=True)
    print("results identical across values: True" if a.tol == 0.0 else "max abs difference across values: %.3e" % worst)


if __name__ == "__main__":
    main()
