"""The N>1 path on CPU: two gloo ranks shard the tracks, each builds its partial
master bus, and ONE reduce combines them (dsp_amd.device.shard_range /
reduce_master_bus — the same code the GPU ranks run over RCCL).  The per-rank
buses come from the oracle here (no GPU in this test); the check is that the
sharded + reduced bus equals the single-process mix to <= 1e-5 of full scale
(the accumulation order differs, so bit-exactness is not expected)."""
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


def _worker(rank, world, port, tracks, L, M, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dsp_amd import device
    from oracle import oracle as o
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = device.shard_range(tracks, rank, world)
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
        q.put(tb.numpy().copy())
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
    import torch.multiprocessing as mp
    tracks, L, M, world = 5, 3000, 700, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tracks, L, M, q)) for r in range(world)]
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
