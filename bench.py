#!/usr/bin/env python3
"""bench.py — whole-job throughput of the reverb+mix hot path on MI355X.

Workload (the batched-reverb configuration the headline metric is quoted on):
256 independent stereo tracks x 60 s at 48 kHz (F = 2 880 000 frames), each
channel convolved with a 2 s IR (M = 96 000; one `go` of L = F per channel-job,
lib/convolve.js:570-583), then all 256 stereo results summed into one stereo
master bus (src/mix-buffers.js tuple form, vol = 1/256).  One "step" = that
whole pass: wp_conv_run_dev (every job's full L+M result is written) followed by
wp_mix_dev.  The fused entry point wp_conv_mix_run_dev is measured too and
reported in an extra field.  Inputs are synthetic integer-hash noise already resident in HBM
when the timed region starts.  With --gpus N each rank owns 256 tracks (weak
scaling) and the rank-local partial buses are combined by one RCCL reduce.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (≈6300 GB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tracks", type=int, default=256, help="stereo tracks per GPU")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--ir-seconds", type=float, default=2.0)
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--path", choices=["onepass", "twopass", "fused"], default="onepass",
                    help="onepass: wp_conv_run_mix_dev (every job's L+M result written AND the bus folded in the same "
                         "kernel); twopass: wp_conv_run_dev then wp_mix_dev; fused: wp_conv_mix_run_dev (bus only, "
                         "per-track results never written).  All three give the same bus, bit for bit.")
    ap.add_argument("--no-fused-extra", action="store_true", help="skip the extra fused-path measurement")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only for rehearsing "
                    "the multi-rank flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tracks", type=int, default=20, help="stereo tracks in the bounded CPU sample")
    return ap.parse_args()


def cpu_baseline(F, M, tracks):
    """The oracle (CPU port of the reference algorithm, one thread) on a bounded
    sample of the same workload: `tracks` stereo tracks of the full length, IR
    spectra cached as the reference caches them (lib/impulse-convolve-stereo.js:9-17),
    then the same mix.  Checker code, timed here only as the reported baseline."""
    import numpy as np
    from oracle import oracle as o
    o.build()
    irs = [o.synth_ir(M, 500000), o.synth_ir(M, 500001)]
    kern = [o.fft_process_kernel(F, h) for h in irs]
    gos = [o.fft_convolution(F, k, M) for k in kern]
    xs = [[o.synth_noise(F, 1000 + 2 * t + ch) for ch in range(2)] for t in range(tracks)]
    t0 = time.perf_counter()
    outs = [[gos[ch](xs[t][ch]) for ch in range(2)] for t in range(tracks)]
    bus = [np.zeros(F, np.float32), np.zeros(F, np.float32)]
    o.mix_buffers(bus, *[([y[0][:F], y[1][:F]], 1, 1.0 / 256, 0) for y in outs])
    dt = time.perf_counter() - t0
    samples = tracks * 2 * (F + M)
    res = {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%d of 256 stereo tracks, full length (L=%d, M=%d): go() per channel + mixBuffers, "
                     "IR spectra cached; %.1f s of CPU time" % (tracks, F, M, dt)}
    # the same port on every core this process may use: independent channel-jobs on a thread pool
    # (the C calls release the interpreter lock), then the mix on one thread as before
    from concurrent.futures import ThreadPoolExecutor
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    jobs = [(t % tracks, ch) for t in range(2 * cores) for ch in range(2)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        ys = list(pool.map(lambda tc: gos[tc[1]](xs[tc[0]][tc[1]]), jobs))
    bus = [np.zeros(F, np.float32), np.zeros(F, np.float32)]
    o.mix_buffers(bus, *[([ys[2 * t][:F], ys[2 * t + 1][:F]], 1, 1.0 / 256, 0) for t in range(2 * cores)])
    dtm = time.perf_counter() - t0
    res["all_cores"] = {"value": len(jobs) * (F + M) / dtm / 1e6, "unit": "Msamples/s", "cores": cores,
                        "sample": "%d stereo tracks on %d threads; %.1f s wall" % (2 * cores, cores, dtm)}
    return res


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from dsp_amd import device as dev

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        if a.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            except TypeError:  # older signature without device_id
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(a.backend)

    F = int(round(a.seconds * a.rate))
    M = int(round(a.ir_seconds * a.rate))
    tracks = a.tracks
    njobs = 2 * tracks
    out_len = F + M
    stride_out = (out_len + 3) & ~3

    # ---- resident inputs ----
    ir_host = torch.from_numpy(__import__("numpy").stack([dev.synth_ir(M, 500000), dev.synth_ir(M, 500001)]))
    ir = dev.DeviceIR(ir_host.cuda().contiguous(), F)
    conv = dev.BatchedConvolver(ir, [j % 2 for j in range(njobs)])
    x = torch.empty((njobs, F), dtype=torch.float32, device="cuda")
    t_lo, _ = dev.shard_range(tracks * world, rank, world)
    for j in range(njobs):
        dev.synth_noise(x[j], 1000 + 2 * (t_lo + j // 2) + (j % 2))
    bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
    vol = [1.0 / 256] * tracks
    fused = a.path == "fused"
    onepass = a.path == "onepass"
    if fused:
        kern_ms = {"k_fft_fwd": 0.0, "k_mac": 0.0, "k_fft_inv_mix": 0.0}
    elif onepass:
        out = torch.empty((njobs, stride_out), dtype=torch.float32, device="cuda")
        kern_ms = {"k_fft_fwd": 0.0, "k_mac": 0.0, "k_fft_inv_mix_w": 0.0}
    else:
        out = torch.empty((njobs, stride_out), dtype=torch.float32, device="cuda")
        mix = dev.DeviceMix([{"ch": [out[2 * t], out[2 * t + 1]], "rl": F, "sl": F, "vol": vol[t], "o": 0}
                             for t in range(tracks)])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        kern_ms = {"k_fft_fwd": 0.0, "k_mac": 0.0, "k_fft_inv": 0.0, "k_mix": 0.0}
    conv.profile(True)

    def step(timed):
        bus.zero_()
        if fused:
            conv.run_mix(x, [bus[0], bus[1]], vol)
        elif onepass:
            conv.run_and_mix(x, out, [bus[0], bus[1]], vol)
        else:
            conv.run(x, out)
            ev[0].record()
            mix.run([bus[0], bus[1]])
            ev[1].record()
        if world > 1:
            dev.reduce_master_bus(bus, dist, dst=0)
        if timed:
            f, m, i = conv.profile_read()
            kern_ms["k_fft_fwd"] += f
            kern_ms["k_mac"] += m
            if fused:
                kern_ms["k_fft_inv_mix"] += i
            elif onepass:
                kern_ms["k_fft_inv_mix_w"] += i
            else:
                ev[1].synchronize()
                kern_ms["k_fft_inv"] += i
                kern_ms["k_mix"] += ev[0].elapsed_time(ev[1])

    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # secondary measurement, outside the headline's timed region: the fused reverb+mix path
    fused_extra = None
    if not fused and not a.no_fused_extra:
        def fstep():
            bus.zero_()
            conv.run_mix(x, [bus[0], bus[1]], vol)
            if world > 1:
                dev.reduce_master_bus(bus, dist, dst=0)
        fstep()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        f0 = time.perf_counter()
        for _ in range(a.steps):
            fstep()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        fdt = time.perf_counter() - f0
        if world > 1:
            ft = torch.tensor([fdt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(ft, op=dist.ReduceOp.MAX)
            fdt = float(ft.item())
        fused_extra = {"ms_per_step": fdt / a.steps * 1e3,
                       "bus_input_Msamples_per_s": world * njobs * F * a.steps / fdt / 1e6,
                       "note": "wp_conv_mix_run_dev: identical bus, per-track results and the M-sample tails beyond the "
                               "bus are not produced, so it is reported beside the headline, not as it"}

    if rank == 0:
        K = a.steps
        # samples the timed path produced: every job's L+M result (unfused), or the F frames per job that reach the bus (fused)
        total_samples = world * njobs * (F if fused else out_len) * K
        B = capi_info(conv, 2)
        T = capi_info(conv, 1)
        # blocks each kernel touches: the fused path only needs the blocks that reach the bus (tl = F)
        Tout = (F + B - 1) // B if fused else T
        Tin = min(Tout, (F + B - 1) // B + 1)
        # algorithmic bytes per launch: every operand of the kernel read or written exactly once
        alg = {
            "k_fft_fwd": njobs * (4 * F + 8 * B * Tin),
            "k_mac": njobs * (8 * B * Tin + 8 * B * Tout),
            "k_fft_inv": njobs * (8 * B * T + 4 * out_len),
            "k_fft_inv_mix": njobs * 8 * B * Tout + 2 * 2 * 4 * F,
            "k_fft_inv_mix_w": njobs * (8 * B * T + 4 * out_len) + 2 * 2 * 4 * F,
            "k_mix": F * (4 * 2 * tracks + 2 * 4 * 2),
        }
        avg = {k: v / K for k, v in kern_ms.items()}
        dom = max(avg, key=avg.get)
        ach = alg[dom] / (avg[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC bytes per launch from the committed rocprofv3 passes
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # counters are per launch of ONE workload: quote them only for that workload
                same = tj.get("workload") == {"tracks_per_gpu": tracks, "frames": F, "ir_len": M, "path": a.path}
                traffic = tj.get(dom) if same else None
            except Exception:
                traffic = None
        res = {
            "metric": "output Msamples/s (whole node) for stereo IR reverb+mix",
            "value": total_samples / dt / 1e6, "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": a.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batched reverb+mix: %d stereo tracks/GPU x %.0f s @ %d Hz x %.0f s stereo IR, "
                                   "one go() per channel-job then mixBuffers into a stereo bus" % (tracks, a.seconds, a.rate, a.ir_seconds),
                       "tracks_per_gpu": tracks, "frames": F, "ir_len": M, "partition": B, "blocks": T,
                       "path": {"fused": "wp_conv_mix_run_dev (bus only)", "onepass": "wp_conv_run_mix_dev (all go() results + bus in one pass)",
                                "twopass": "wp_conv_run_dev + wp_mix_dev"}[a.path],
                       "parallelism": "tracks sharded over %d GPU(s), RCCL reduce of the bus" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": avg,
                         "kernel_GBs": {k: alg[k] / (avg[k] * 1e-3) / 1e9 for k in avg if avg[k] > 0},
                         "pipeline_compulsory_GBs": (njobs * 4 * F + (16 * F if fused else njobs * 4 * out_len)) / (dt / K) / 1e9},
        }
        if fused_extra is not None:
            res["fused_path"] = fused_extra
        if world == 1 and not a.no_cpu_baseline and a.cpu_tracks > 0:
            res["cpu_baseline"] = cpu_baseline(F, M, a.cpu_tracks)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def capi_info(conv, which):
    from dsp_amd import capi
    return int(capi.lib().wp_conv_info(conv.handle, which))


if __name__ == "__main__":
    main()
