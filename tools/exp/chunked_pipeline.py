#!/usr/bin/env python3
"""Experiment: the bench workload run as a sequence of job chunks that share ONE small plan, so
that the block spectra Z/W of a chunk (23.9 MB per job each) are rewritten in place and can stay in
the 256 MiB Infinity Cache between the three kernels.  Prints ms per step for several chunk sizes
beside the single-launch form.  Same entry point (wp_conv_run_mix_dev), same bits on the bus.

    python tools/exp/chunked_pipeline.py [--tracks 256] [--steps 5]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--chunks", default="512,128,64,32,16,8,4")
    a = ap.parse_args()
    import numpy as np
    import torch
    from dsp_amd import capi
    if os.environ.get("WPDSP_EXP_LIB"):  # a variant build of the library (e.g. plain instead of non-temporal spectra accesses)
        capi.LIB_PATH = os.environ["WPDSP_EXP_LIB"]
    from dsp_amd import device as dev
    F, M = 2880000, 96000
    njobs = 2 * a.tracks
    irs = torch.from_numpy(np.stack([dev.synth_ir(M, 500000), dev.synth_ir(M, 500001)])).cuda()
    ir = dev.DeviceIR(irs, F)
    x = torch.empty((njobs, F), dtype=torch.float32, device="cuda")
    for j in range(njobs):
        dev.synth_noise(x[j], 1000 + j)
    stride = (F + M + 3) & ~3
    out = torch.empty((njobs, stride), dtype=torch.float32, device="cuda")
    vol = [1.0 / 256] * a.tracks
    ref_bus = None
    for cj in [int(c) for c in a.chunks.split(",")]:
        cj = min(cj, njobs)
        conv = dev.BatchedConvolver(ir, [j % 2 for j in range(cj)])
        bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")

        def step():
            bus.zero_()
            for j0 in range(0, njobs, cj):
                conv.run_and_mix(x[j0:j0 + cj], out[j0:j0 + cj], [bus[0], bus[1]], vol[j0 // 2:(j0 + cj) // 2])
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        if ref_bus is None:
            ref_bus = bus.clone()
        same = bool(torch.equal(bus.view(torch.int32), ref_bus.view(torch.int32)))
        print("jobs per chunk %4d  launches/step %4d  workspace %7.1f MB  tchunk %4d  gather %d  %8.3f ms/step  bus identical: %s"
              % (cj, 3 * (njobs // cj), conv.workspace_bytes / 1e6, capi.lib().wp_conv_info(conv.handle, 4),
                 capi.lib().wp_conv_info(conv.handle, 5), dt * 1e3, same), flush=True)
        conv.close()


if __name__ == "__main__":
    main()
