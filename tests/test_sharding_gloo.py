"""The N>1 path rehearsed with two gloo ranks: the tracks are sharded
(dsp_amd.device.shard_range), each rank builds its partial master bus, and ONE
reduce combines them (reduce_master_bus).  Without a GPU the per-rank buses come
from the oracle; the `gpu` variant runs the same flow with both ranks on the one
MI355X of the test box, each rank computing its bus with the HIP path
(BatchedConvolver.run_and_mix) and the reduce staged through gloo, because RCCL
refuses two ranks on one device (the RCCL reduce itself: test_gpu_device_api.py,
one rank).  Either way the sharded + reduced bus must equal the single-process
oracle mix to <= 1e-5 of full scale (the accumulation order differs, so
bit-exactness is not expected)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tracks, L, M, q, use_gpu=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dsp_amd import device
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = device.shard_range(tracks, rank, world)
    if use_gpu:
        # the product path: device-resident noise, batched go() per channel-job and the mix in one pass
        torch.cuda.set_device(0)
        irs = torch.from_numpy(np.stack([device.synth_ir(M, 500000), device.synth_ir(M, 500001)])).cuda()
        ir = device.DeviceIR(irs, L)
        n = hi - lo
        conv = device.BatchedConvolver(ir, [j % 2 for j in range(2 * n)])
        x = torch.empty((2 * n, L), dtype=torch.float32, device="cuda")
        for j in range(2 * n):
            device.synth_noise(x[j], 1000 + 2 * (lo + j // 2) + (j % 2))
        out = torch.empty((2 * n, L + M), dtype=torch.float32, device="cuda")
        tb = torch.zeros((2, L), dtype=torch.float32, device="cuda")
        conv.run_and_mix(x, out, [tb[0], tb[1]], [1.0 / tracks] * n)
        torch.cuda.synchronize()
    else:
        from oracle import oracle as o
        irs = [o.synth_ir(M, 500000), o.synth_ir(M, 500001)]
        bus = [np.zeros(L, np.float32), np.zeros(L, np.float32)]
        srcs = []
        for t in range(lo, hi):
            y = [o.fft_convolution(L, o.fft_process_kernel(L, irs[ch]), M)(o.synth_noise(L, 1000 + 2 * t + ch)) for ch in range(2)]
            srcs.append(([y[0][:L], y[1][:L]], 1, 1.0 / tracks, 0))
        o.mix_buffers(bus, *srcs)
        tb = torch.from_numpy(np.stack(bus))
    device.reduce_master_bus(tb, dist, dst=0)
    if rank == 0:
        q.put(tb.cpu().numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_is_a_partition():
    from dsp_amd import device
    for n in (0, 1, 7, 256, 2048):
        for w in (1, 2, 3, 8):
            spans = [device.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_rank_sharded_mix_matches_single_process(oracle):
    _two_ranks(oracle, False)


@pytest.mark.gpu
def test_two_rank_sharded_mix_hip_path(oracle):
    _two_ranks(oracle, True)


def _two_ranks(oracle, use_gpu):
    import torch.multiprocessing as mp
    tracks, L, M, world = 5, 3000, 700, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tracks, L, M, q, use_gpu)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = oracle
    irs = [o.synth_ir(M, 500000), o.synth_ir(M, 500001)]
    bus = [np.zeros(L, np.float32), np.zeros(L, np.float32)]
    srcs = []
    for t in range(tracks):
        y = [o.fft_convolution(L, o.fft_process_kernel(L, irs[ch]), M)(o.synth_noise(L, 1000 + 2 * t + ch)) for ch in range(2)]
        srcs.append(([y[0][:L], y[1][:L]], 1, 1.0 / tracks, 0))
    o.mix_buffers(bus, *srcs)
    err = float(np.abs(got - np.stack(bus)).max())
    assert err <= 1e-5, err


def _bench_two_ranks(extra, timeout=420):
    """bench.py's own multi-rank control flow as a fresh child: two ranks launched the way the driver
    launches them, both on the one GPU of the test box."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tracks", "4", "--seconds", "4",
           "--steps", "2", "--warmup", "1", "--cpu-tracks", "0"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


@pytest.mark.gpu
def test_bench_two_rank_flow_with_gloo_reduce():
    """Track sharding, two bus buffers, barrier + max-over-ranks timing, the verification on every rank
    (bus slice bit-exact, jobs against the oracle) and the extra 4 s-IR measurement, with the reduce carried
    by gloo because RCCL refuses two ranks on one device."""
    out, line = _bench_two_ranks(["--reduce", "torch", "--backend", "gloo"])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert line is not None and line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    v = line["verified"]
    assert v["ok"] and v["all_ranks_ok"] and v["bus_slice_bit_exact"] and v["oracle_jobs_ok"], v
    v5 = line["render_4s_ir"]["verified"]
    assert v5["ok"] and v5["all_ranks_ok"] and v5["oracle_jobs_ok"], v5
    assert "reduce_fallback" not in line


@pytest.mark.gpu
def test_bench_refuses_to_time_without_its_own_collective():
    """--reduce capi with both ranks on one GPU: RCCL rejects the communicator (duplicate device), every rank
    learns it, and the bench prints a line with value null and exits non-zero instead of timing another
    library's collective."""
    out, line = _bench_two_ranks(["--reduce", "capi"], timeout=300)
    assert out.returncode != 0
    assert line is not None and line["value"] is None and line["reduce_fallback"] is True, out.stdout[-2000:]
