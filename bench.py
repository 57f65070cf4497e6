#!/usr/bin/env python3
"""bench.py — whole-job throughput of the reverb+mix hot path on MI355X.

Workload (`--config c3`, the batched-reverb configuration the headline metric is
quoted on): 256 independent stereo tracks x 60 s at 48 kHz (F = 2 880 000 frames),
each channel convolved with a 2 s IR (M = 96 000; one `go` of L = F per
channel-job, lib/convolve.js:570-583), then all 256 stereo results summed into
one stereo master bus (src/mix-buffers.js tuple form, vol = 1/256).  One "step" =
that whole pass through wp_conv_run_mix_dev (every job's full L+M result is
written and the bus folded in the same pass).  `--config c5` is the per-GPU share
of the 2048-track render: the same tracks with a 4 s impulse response of its own
per channel-job.  Inputs are synthetic integer-hash noise already resident in HBM
when the timed region starts.  With --gpus N each rank owns 256 tracks (weak
scaling) and the rank-local partial buses are combined by one RCCL reduce
(wp_bus_reduce, called through the C ABI on a side stream with two bus buffers,
so the reduce of step i overlaps the transforms of step i+1).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

After the timed region the results are checked: job 0 against the samples the
reference JavaScript produced for that job (tests/golden), a bus slice against the
in-order fold of the written results.  Prints ONE JSON line on rank 0.
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
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # (the first three passes run 5-30 % slower: first touch of the workspace, clocks)
    ap.add_argument("--config", choices=["c3", "c5"], default="c3",
                    help="c3: 2 s stereo IR shared by all tracks (the configuration the metric is quoted on); "
                         "c5: per-GPU share of the 2048-track render, a 4 s IR per channel-job")
    ap.add_argument("--tracks", type=int, default=256, help="stereo tracks per GPU")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--ir-seconds", type=float, default=None, help="default: 2 (c3) or 4 (c5)")
    ap.add_argument("--rate", type=int, default=48000)
    ap.add_argument("--path", choices=["onepass", "twopass", "fused"], default="onepass",
                    help="onepass: wp_conv_run_mix_dev (every job's L+M result written AND the bus folded in the same "
                         "kernel); twopass: wp_conv_run_dev then wp_mix_dev; fused: wp_conv_mix_run_dev (bus only, "
                         "per-track results never written).  All three give the same bus, bit for bit.")
    ap.add_argument("--no-fused-extra", action="store_true", help="skip the extra fused-path measurement")
    ap.add_argument("--no-partitioned-extra", action="store_true", help="skip the extra measurement of the partitioned form")
    ap.add_argument("--no-c5-extra", action="store_true", help="with --gpus > 1: skip the extra 4 s-IR measurement")
    ap.add_argument("--reduce", choices=["capi", "torch"], default="capi",
                    help="capi: wp_bus_reduce (RCCL through the C ABI; rendezvous over gloo); torch: torch.distributed's "
                         "own process-group reduce on the --backend given")
    ap.add_argument("--backend", default="nccl", help="with --reduce torch: process-group backend (nccl = RCCL; gloo only "
                    "for rehearsing the multi-rank flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--lib", default=None, help="a variant build of libwpdsp.so to measure instead of the in-tree one")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="wp_tune(knob, value) before the plans are made (measurements of the library's alternatives)")
    ap.add_argument("--cpu-tracks", type=int, default=20, help="stereo tracks in the bounded CPU sample")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
    res = {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
           "sample": "%d of 256 stereo tracks, full length (L=%d, M=%d): go() per channel + mixBuffers, "
                     "IR spectra cached; %.1f s of CPU time (C restatement of the reference's JavaScript, one thread)"
                     % (tracks, F, M, dt)}
    # the same port on every core this process may use: independent channel-jobs on a thread pool
    # (the C calls release the interpreter lock), every job a different track, then the mix on one thread
    from concurrent.futures import ThreadPoolExecutor
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    ntr = max(tracks, cores)
    while len(xs) < ntr:
        t = len(xs)
        xs.append([o.synth_noise(F, 1000 + 2 * t + ch) for ch in range(2)])
    jobs = [(t, ch) for t in range(ntr) for ch in range(2)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        ys = list(pool.map(lambda tc: gos[tc[1]](xs[tc[0]][tc[1]]), jobs))
    bus = [np.zeros(F, np.float32), np.zeros(F, np.float32)]
    o.mix_buffers(bus, *[([ys[2 * t][:F], ys[2 * t + 1][:F]], 1, 1.0 / 256, 0) for t in range(ntr)])
    dtm = time.perf_counter() - t0
    res["all_cores"] = {"value": len(jobs) * (F + M) / dtm / 1e6, "unit": "Msamples/s", "cores": cores,
                        "sample": "%d different stereo tracks on %d threads; %.1f s wall" % (ntr, cores, dtm)}
    js = js_baseline(F, M)
    if js is not None:
        res["node"] = js
    return res


def js_baseline(F, M):
    """The same algorithm restated in JavaScript (oracle/oracle_go.mjs, bit-identical to the reference's
    results on the committed fixtures) under this host's node, one thread: the closest stand-in for the
    reference's own Node.js path, which cannot travel to this machine."""
    import shutil
    import subprocess
    node = shutil.which("node")
    script = os.path.join(ROOT, "oracle", "oracle_go.mjs")
    if not node or not os.path.exists(script):
        return None
    try:
        out = subprocess.run([node, "--no-warnings", script, "bench", str(F), str(M), "4"], capture_output=True, text=True,
                             timeout=240)
        if out.returncode != 0:
            return {"error": (out.stderr or out.stdout)[-300:]}
        r = json.loads(out.stdout.strip().splitlines()[-1])
        r.update({"unit": "Msamples/s", "cores": 1, "kind": "port", "cpu": cpu_model()})
        return r
    except Exception as e:  # the baseline is a reported extra, never a reason to lose the bench line
        return {"error": repr(e)[:300]}


class Workload:
    """Device-resident inputs and plans of one configuration on this rank."""

    def __init__(self, a, config, rank, world, torch, dev):
        import numpy as np
        self.config = config
        self.F = F = int(round(a.seconds * a.rate))
        ir_seconds = a.ir_seconds if a.ir_seconds is not None else (2.0 if config == "c3" else 4.0)
        self.ir_seconds = ir_seconds
        self.M = M = int(round(ir_seconds * a.rate))
        self.tracks = tracks = a.tracks
        self.njobs = njobs = 2 * tracks
        self.out_len = F + M
        self.stride_out = (self.out_len + 3) & ~3
        self.t_lo, _ = dev.shard_range(tracks * world, rank, world)
        if config == "c3":   # one stereo IR shared by every track
            ir_host = np.stack([dev.synth_ir(M, 500000), dev.synth_ir(M, 500001)])
            job_ch = [j % 2 for j in range(njobs)]
        else:                # an impulse response per channel-job: track t, channel c -> seed 500000 + 2t + c
            ir_host = np.stack([dev.synth_ir(M, 500000 + 2 * (self.t_lo + j // 2) + (j % 2)) for j in range(njobs)])
            job_ch = list(range(njobs))
        self.ir = dev.DeviceIR(torch.from_numpy(ir_host).cuda().contiguous(), F)
        del ir_host
        self.conv = dev.BatchedConvolver(self.ir, job_ch)
        self.x = torch.empty((njobs, F), dtype=torch.float32, device="cuda")
        for j in range(njobs):
            dev.synth_noise(self.x[j], 1000 + 2 * (self.t_lo + j // 2) + (j % 2))
        self.vol = [1.0 / 256] * tracks
        self.out = None

    def alloc_out(self, torch):
        if self.out is None:
            self.out = torch.empty((self.njobs, self.stride_out), dtype=torch.float32, device="cuda")
        return self.out

    def close(self):
        self.conv.close()
        self.ir.close()
        self.x = self.out = None


def verify(w, rank, torch, no_oracle=False):
    """One untimed pass of the headline entry point, checked: job 0 of global track 0 against what the
    reference JavaScript produced for it (committed fixture), job 5 and the last job of this rank against
    the CPU oracle run on the spot, and a slice of the rank-local bus against the in-order f64 fold of the
    written results (src/mix-buffers.js:49-52)."""
    import numpy as np
    out = w.alloc_out(torch)
    bus = torch.zeros((2, w.F), dtype=torch.float32, device="cuda")
    w.conv.run_and_mix(w.x, out, [bus[0], bus[1]], w.vol)
    torch.cuda.synchronize()
    res = {}
    gpath = os.path.join(ROOT, "tests", "golden", "g5_config_sized.json" if w.config == "c3" else "g7_c5_jobs.json")
    if rank == 0 and w.t_lo == 0 and os.path.exists(gpath):
        g = json.load(open(gpath))
        g = g["c3_job0"] if w.config == "c3" else dict(g["jobs"][0], L=g["L"], M=g["M"])
        if (g["L"], g["M"]) == (w.F, w.M):
            y0 = out[0, :w.out_len].cpu().numpy().astype(np.float64)
            bits = lambda b: np.array(b, dtype=np.uint32).view(np.float32).astype(np.float64)  # noqa: E731
            err = max(float(np.abs(y0[::g["stride"]] - bits(g["samples"])).max()),
                      float(np.abs(y0[:64] - bits(g["head"])).max()), float(np.abs(y0[-64:] - bits(g["tail"])).max()))
            res["job0_max_abs_err_vs_reference"] = err
            res["job0_ok"] = bool(err <= 1e-5)
        else:
            res["job0_ok"] = None  # non-default sizes: no fixture
    # further jobs against the oracle computed on the spot (CPU restatement of lib/convolve.js:564-586; checker only):
    # an odd job inside a group of eight of the row pass, and the last job of the batch
    if not no_oracle:
        try:
            from oracle import oracle as o
            o.build()
            worst = 0.0
            checked = []
            for j in sorted(set(k for k in (5, w.njobs - 1) if 0 <= k < w.njobs)):
                track, ch = w.t_lo + j // 2, j % 2
                h = o.synth_ir(w.M, 500000 + ch if w.config == "c3" else 500000 + 2 * track + ch)
                want = o.fft_convolution(w.F, o.fft_process_kernel(w.F, h), w.M)(o.synth_noise(w.F, 1000 + 2 * track + ch))
                got = out[j, :w.out_len].cpu().numpy().astype(np.float64)
                worst = max(worst, float(np.abs(got - want).max()))
                checked.append(j)
            res["oracle_jobs"] = checked
            res["oracle_jobs_max_abs_err"] = worst
            res["oracle_jobs_ok"] = bool(worst <= 1e-5)
        except Exception as e:  # a missing checker is reported, it does not pass
            res["oracle_jobs_ok"] = None
            res["oracle_jobs_error"] = repr(e)[:200]
    sl = slice(1234567 % max(1, w.F - 4096), 1234567 % max(1, w.F - 4096) + min(4096, w.F))
    o_ = out[:, sl].cpu().numpy()
    ok = True
    for ch in range(2):
        acc = np.zeros(o_.shape[1], np.float32)
        for t in range(w.tracks):
            acc = (acc.astype(np.float64) + o_[2 * t + ch].astype(np.float64) * w.vol[t]).astype(np.float32)
        ok = ok and bool(np.array_equal(acc, bus[ch, sl].cpu().numpy()))
    res["bus_slice_bit_exact"] = ok
    res["ok"] = bool(ok and res.get("job0_ok") is not False and res.get("oracle_jobs_ok") is not False)
    return res


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from dsp_amd import capi
    if a.lib:
        capi.LIB_PATH = os.path.abspath(a.lib)
    from dsp_amd import device as dev

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    comm = None
    reduce_kind = "none (one rank)"
    if world > 1:
        if a.reduce == "capi":
            # torch.distributed (gloo, CPU tensors) only carries the rendezvous, the barriers and the
            # max-over-ranks of the timing; the data-path collective is wp_bus_reduce
            dist.init_process_group("gloo")
            # RCCL prints a version banner on stdout at communicator creation: keep stdout for the one JSON line
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            why = ""
            try:
                # agree on readiness BEFORE the collective: wp_comm_create waits for every rank, so a rank that
                # cannot open RCCL (or has no GPU) must be known to all of them first
                rc = capi.lib().wp_comm_available()
                if rc != 0:
                    why = "wp_comm_available: %s" % capi.lib().wp_last_error().decode()
                flag = torch.tensor([1.0 if rc == 0 else 0.0], dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if flag.item() == 1.0:
                    try:
                        comm = dev.Comm(rank, world, dev.gloo_rendezvous(dist))
                    except Exception as e:
                        why = "%s: %s" % (type(e).__name__, e)
                elif not why:
                    why = "another rank cannot open RCCL"
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            flag = torch.tensor([0.0 if comm is None else 1.0], dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() != 1.0:
                # no fallback: the collective the product owns could not be set up, so there is nothing to time
                print("[bench] rank %d: no communicator (%s)" % (rank, why or "another rank failed"), file=sys.stderr, flush=True)
                if comm is not None:
                    comm.close()
                if rank == 0:
                    print(json.dumps({"metric": "output Msamples/s (whole node) for stereo IR reverb+mix", "value": None,
                                      "unit": "Msamples/s", "n_gpus": world, "reduce_fallback": True,
                                      "error": "wp_comm_create (--reduce capi) failed on at least one rank: %s"
                                               % (why[:300] or "see the other ranks' stderr")}), flush=True)
                dist.destroy_process_group()
                raise SystemExit(3)
            reduce_kind = ("wp_bus_reduce: ncclReduce(sum, f32) through the C ABI (%s), side stream, two bus buffers"
                           % os.path.basename(comm.library))
        elif a.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            except TypeError:  # older signature without device_id
                dist.init_process_group("nccl")
            reduce_kind = "torch.distributed.reduce (nccl)"
        else:
            dist.init_process_group(a.backend)
            reduce_kind = "torch.distributed.reduce (%s)" % a.backend
    cpu_group_tensor = world > 1 and (a.reduce == "capi" or a.backend != "nccl")

    def max_over_ranks(dt):
        if world == 1:
            return dt
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if cpu_group_tensor else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def time_steps(step, warmup, steps, after_warmup=None):
        for _ in range(warmup):
            step(False)
        fence()
        if after_warmup is not None:
            after_warmup()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        fence()
        return max_over_ranks(time.perf_counter() - t0)

    for kv in a.tune:
        knob, value = kv.split("=")
        capi.check(capi.lib().wp_tune(int(knob), int(value)), "wp_tune")
    w = Workload(a, a.config, rank, world, torch, dev)
    F, M, tracks, njobs, out_len = w.F, w.M, w.tracks, w.njobs, w.out_len
    conv, x, vol = w.conv, w.x, w.vol
    # two bus buffers: the reduce of one step runs on a side stream while the next step computes
    buses = [torch.zeros((2, F), dtype=torch.float32, device="cuda") for _ in range(2 if world > 1 else 1)]
    side = torch.cuda.Stream() if comm is not None else None
    ready = [torch.cuda.Event() for _ in buses]
    reduced = [None for _ in buses]
    fused = a.path == "fused"
    onepass = a.path == "onepass"
    out = None if fused else w.alloc_out(torch)
    # which form of the convolution the library picked for these sizes: the partitioned one (forward
    # transforms, multiply-accumulate over IR partitions, inverse transforms) or the long-frame one
    # (column pass, row pass with the spectral product, inverse column pass)
    lf = int(capi.lib().wp_conv_info(conv.handle, 6)) == 1
    k1n, k2n, k3n = ("k_lf_fwd", "k_lf_mid", "k_lf_inv") if lf else ("k_fft_fwd", "k_mac", "k_fft_inv")
    if fused:
        kern_ms = {k1n: 0.0, k2n: 0.0, k3n + "_mix": 0.0}
    elif onepass:
        kern_ms = {k1n: 0.0, k2n: 0.0, k3n + "_mix_w": 0.0}
    else:
        mix = dev.DeviceMix([{"ch": [out[2 * t], out[2 * t + 1]], "rl": F, "sl": F, "vol": vol[t], "o": 0}
                             for t in range(tracks)])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        kern_ms = {k1n: 0.0, k2n: 0.0, k3n: 0.0, "k_mix": 0.0}
    conv.profile(True)
    counter = [0]

    def reduce_bus(k):
        bus = buses[k]
        if world == 1:
            return
        if comm is not None:
            ready[k].record()
            side.wait_event(ready[k])
            comm.reduce(bus, dst=0, stream=side)
            e = torch.cuda.Event()
            e.record(side)
            reduced[k] = e
        else:
            dev.reduce_master_bus(bus, dist, dst=0)

    def step(timed, which="main"):
        k = counter[0] % len(buses)
        counter[0] += 1
        bus = buses[k]
        if reduced[k] is not None:  # this buffer's previous reduce must have finished before it is zeroed
            torch.cuda.current_stream().wait_event(reduced[k])
        bus.zero_()
        if which == "fused" or fused:
            conv.run_mix(x, [bus[0], bus[1]], vol)
        elif onepass:
            conv.run_and_mix(x, out, [bus[0], bus[1]], vol)
        else:
            conv.run(x, out)
            ev[0].record()
            mix.run([bus[0], bus[1]])
            ev[1].record()
        reduce_bus(k)
        if timed and which == "main" and not fused and not onepass:
            ev[1].synchronize()  # (two-pass path only: the mix kernel between torch events)
            kern_ms["k_mix"] += ev[0].elapsed_time(ev[1])

    # the library's events of the K timed steps are added up after the timed region (wp_conv_profile_sum): no step
    # waits for the one before it
    dt = time_steps(step, a.warmup, a.steps, after_warmup=conv.profile_sum)
    (f, m, i), nruns_prof = conv.profile_sum()
    assert nruns_prof == a.steps, "profiled runs %d != timed steps %d" % (nruns_prof, a.steps)
    kern_ms[k1n] += f
    kern_ms[k2n] += m
    kern_ms[(k3n + "_mix") if fused else (k3n + "_mix_w") if onepass else k3n] += i

    # secondary measurement, outside the headline's timed region: the fused reverb+mix path
    fused_extra = None
    if not fused and not a.no_fused_extra:
        fdt = time_steps(lambda timed: step(timed, "fused"), 1, a.steps)
        fused_extra = {"ms_per_step": fdt / a.steps * 1e3,
                       "bus_input_Msamples_per_s": world * njobs * F * a.steps / fdt / 1e6,
                       "note": "wp_conv_mix_run_dev: identical bus, per-track results and the M-sample tails beyond the "
                               "bus are not produced, so it is reported beside the headline, not as it"}

    # secondary measurement: the same workload forced onto the partitioned form (forward transforms, multiply-accumulate
    # over the impulse response's partitions, inverse transforms: the algorithm the metric's name describes), so that it
    # has a number from the same run beside the form the library picks
    part_extra = None
    if onepass and lf and not a.no_partitioned_extra:
        import numpy as np
        capi.check(capi.lib().wp_tune(5, 1), "wp_tune")  # WP_TUNE_CONV_FORM: partitioned
        try:
            if a.config == "c3":
                irp = dev.DeviceIR(torch.from_numpy(np.stack([dev.synth_ir(M, 500000), dev.synth_ir(M, 500001)])).cuda().contiguous(), F)
                convp = dev.BatchedConvolver(irp, [j % 2 for j in range(njobs)])
            else:
                irp = dev.DeviceIR(torch.from_numpy(np.stack([dev.synth_ir(M, 500000 + 2 * (w.t_lo + j // 2) + (j % 2))
                                                              for j in range(njobs)])).cuda().contiguous(), F)
                convp = dev.BatchedConvolver(irp, list(range(njobs)))
        finally:
            capi.lib().wp_tune(5, 0)
        convp.profile(True)
        pk = [0.0, 0.0, 0.0]

        def stepp(timed):
            buses[0].zero_()
            convp.run_and_mix(x, out, [buses[0][0], buses[0][1]], vol)
        pdt = time_steps(stepp, 1, a.steps, after_warmup=convp.profile_sum)
        pk = list(convp.profile_sum()[0])
        Bp, Tp = int(capi.lib().wp_conv_info(convp.handle, 2)), int(capi.lib().wp_conv_info(convp.handle, 1))
        Tinp = min(Tp, (F + Bp - 1) // Bp + 1)
        mac_bytes = njobs * 8 * Bp * (Tinp + Tp)
        part_extra = {"ms_per_step": pdt / a.steps * 1e3, "value": world * njobs * out_len * a.steps / pdt / 1e6, "unit": "Msamples/s",
                      "partition": Bp, "blocks": Tp,
                      "kernel_ms": {"k_fft_fwd": pk[0] / a.steps, "k_mac": pk[1] / a.steps, "k_fft_inv_mix_w": pk[2] / a.steps},
                      "k_mac_GBs": mac_bytes / (pk[1] / a.steps * 1e-3) / 1e9 if pk[1] > 0 else None,
                      "note": "wp_tune(WP_TUNE_CONV_FORM, 1): uniformly partitioned overlap-save, same entry point and results "
                              "within 1e-5; reported beside the headline, which runs the form the library picks"}
        convp.close()
        irp.close()
        del convp, irp
        torch.cuda.empty_cache()

    verified = None
    if not a.no_verify and not fused:
        v = verify(w, rank, torch)
        if world > 1:  # every rank checked its own bus slice: all must hold
            flag = torch.tensor([1.0 if v["ok"] else 0.0], dtype=torch.float64, device="cpu" if cpu_group_tensor else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            v["all_ranks_ok"] = bool(flag.item() == 1.0)
            v["ok"] = v["ok"] and v["all_ranks_ok"]
        verified = v

    B = int(capi.lib().wp_conv_info(conv.handle, 2))
    T = int(capi.lib().wp_conv_info(conv.handle, 1))
    P = int(capi.lib().wp_conv_info(conv.handle, 3))
    # long-frame form: runs of equal frames (frame length N2 samples = N2/2 complex points, frames, advance, first sample)
    nruns = int(capi.lib().wp_conv_info(conv.handle, 10)) if lf else 0
    runs = [tuple(int(capi.lib().wp_conv_info(conv.handle, 16 + 4 * i + k)) for k in range(4)) for i in range(nruns)]

    # secondary measurement for multi-GPU runs: the 2048-track render's per-GPU share (4 s IR per job)
    c5_extra = None
    if world > 1 and a.config == "c3" and not a.no_c5_extra and a.path == "onepass":
        w.close()
        del conv, x, out
        torch.cuda.empty_cache()
        w5 = Workload(a, "c5", rank, world, torch, dev)
        out5 = w5.alloc_out(torch)

        def step5(timed):
            k = counter[0] % len(buses)
            counter[0] += 1
            bus = buses[k]
            if reduced[k] is not None:
                torch.cuda.current_stream().wait_event(reduced[k])
            bus.zero_()
            w5.conv.run_and_mix(w5.x, out5, [bus[0], bus[1]], w5.vol)
            reduce_bus(k)
        dt5 = time_steps(step5, 1, a.steps)
        c5_extra = {"workload": "%d tracks over %d GPUs, 4 s impulse response per channel-job, 60 s, RCCL reduce of the bus"
                                % (tracks * world, world),
                    "ms_per_step": dt5 / a.steps * 1e3, "value": world * njobs * w5.out_len * a.steps / dt5 / 1e6,
                    "unit": "Msamples/s", "partition": int(capi.lib().wp_conv_info(w5.conv.handle, 2))}
        if not a.no_verify:
            v5 = verify(w5, rank, torch)
            flag = torch.tensor([1.0 if v5["ok"] else 0.0], dtype=torch.float64, device="cpu" if cpu_group_tensor else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            v5["all_ranks_ok"] = bool(flag.item() == 1.0)
            v5["ok"] = v5["ok"] and v5["all_ranks_ok"]
            c5_extra["verified"] = v5
        w5.close()

    # a failed check (on any rank: `ok` already holds the minimum over ranks) voids the line and the exit status
    bad = []
    if verified is not None and not verified["ok"]:
        bad.append("headline workload")
    if c5_extra is not None and "verified" in c5_extra and not c5_extra["verified"]["ok"]:
        bad.append("render_4s_ir")
    if rank == 0:
        K = a.steps
        # samples the timed path produced: every job's L+M result (unfused), or the F frames per job that reach the bus (fused)
        total_samples = world * njobs * (F if fused else out_len) * K
        # blocks each kernel touches: the fused path only needs the blocks that reach the bus (tl = F)
        Tout = (F + B - 1) // B if fused else T
        Tin = min(Tout, (F + B - 1) // B + 1)
        # algorithmic bytes per launch: every operand of the kernel read or written exactly once
        # long-frame form: frames of N2 samples = Nc complex points, advancing by hop
        # (with several runs of frames a kernel is launched once per run: bytes and times are per step, summed over the runs)
        pts_all = sum(n2 // 2 * nf for n2, nf, hop, base in runs)
        pts_run = sum(n2 // 2 * min(nf, max(0, -(-(F - base) // hop))) for n2, nf, hop, base in runs) if fused else pts_all
        alg = {
            "k_lf_fwd": njobs * (4 * F + 8 * pts_run),
            # (+ 8 bytes per frame point when every job has a kernel spectrum of its own to read)
            "k_lf_mid": njobs * (16 + (8 if a.config == "c5" else 0)) * pts_run,
            "k_lf_inv": njobs * (8 * pts_all + 4 * out_len),
            "k_lf_inv_mix": njobs * 8 * pts_run + 2 * 2 * 4 * F,
            "k_lf_inv_mix_w": njobs * (8 * pts_all + 4 * out_len) + 2 * 2 * 4 * F,
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
        config_form = ("long-frame: " + " + ".join("%dx%d" % (nf, n2) for n2, nf, hop, base in runs)) if lf else "partitioned: %d" % B
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC bytes per launch from the committed rocprofv3 passes
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                tj = tj.get(a.config, tj)  # one entry per profiled configuration
                # counters are per launch of ONE workload and one cover by frames: quote them only for those
                same = tj.get("workload") == {"tracks_per_gpu": tracks, "frames": F, "ir_len": M, "path": a.path}
                same = same and tj.get("form", config_form) == config_form
                traffic = tj.get(dom) if same else None
            except Exception:
                traffic = None
        ir_desc = "%.0f s stereo IR" % w.ir_seconds if a.config == "c3" else "%.0f s impulse response per channel-job" % w.ir_seconds
        res = {
            "metric": "output Msamples/s (whole node) for stereo IR reverb+mix",
            "value": total_samples / dt / 1e6, "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": a.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batched reverb+mix: %d stereo tracks/GPU x %.0f s @ %d Hz x %s, "
                                   "one go() per channel-job then mixBuffers into a stereo bus" % (tracks, a.seconds, a.rate, ir_desc),
                       "name": a.config, "tracks_per_gpu": tracks, "frames": F, "ir_len": M, "form": ("long-frame overlap-save, per job: " + " + ".join("%d frame(s) of %d samples advancing by %d" % (nf, n2, hop) for n2, nf, hop, base in runs)) if lf
                       else ("partitioned overlap-save: %d partitions of %d, %d blocks per job" % (P, B, T)),
                       "partition": B, "partitions": P, "blocks": T,
                       "path": {"fused": "wp_conv_mix_run_dev (bus only)", "onepass": "wp_conv_run_mix_dev (all go() results + bus in one pass)",
                                "twopass": "wp_conv_run_dev + wp_mix_dev"}[a.path],
                       "parallelism": "tracks sharded over %d GPU(s); bus reduce: %s" % (world, reduce_kind)},
            "verified": verified,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "launches_per_step": max(1, nruns), "kernel_ms": avg,
                         "kernel_GBs": {k: alg[k] / (avg[k] * 1e-3) / 1e9 for k in avg if avg[k] > 0},
                         "pipeline_compulsory_GBs": (njobs * 4 * F + (16 * F if fused else njobs * 4 * out_len)) / (dt / K) / 1e9},
        }
        if fused_extra is not None:
            res["fused_path"] = fused_extra
        if part_extra is not None:
            res["partitioned_path"] = part_extra
        if c5_extra is not None:
            res["render_4s_ir"] = c5_extra
        if world == 1 and not a.no_cpu_baseline and a.cpu_tracks > 0:
            res["cpu_baseline"] = cpu_baseline(F, M, a.cpu_tracks)
        if bad:
            # a build whose results differ from the reference's has no throughput to report
            res["measured_value"] = res["value"]
            res["value"] = None
            res["error"] = "verification failed: " + "; ".join(bad)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        if comm is not None:
            torch.cuda.synchronize()
            comm.close()
        dist.destroy_process_group()
    if bad:
        raise SystemExit(4)


if __name__ == "__main__":
    main()
