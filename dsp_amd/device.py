"""Device-resident batched reverb + master-bus mix for torch tensors.

This is the plumbing layer the benchmark and the multi-GPU path use: torch owns
device memory, streams and the process group; every sample is produced by
libwpdsp.so through the C ABI (``wp_conv_run_dev``, ``wp_mix_dev``).

Sharding (one process per GPU): tracks are independent (each `c.src`/`c.render`
in src/context.js:145-162 renders into its own buffer), so rank r owns a
contiguous range of tracks, convolves and sums them into a rank-local partial
bus, and the partial buses are combined with ONE collective — an RCCL
``reduce(SUM)`` onto rank 0 (or all-reduce).  That collective is the only
exchange step on the path.
"""
import ctypes as C

import numpy as np

from . import capi


def _stream_ptr(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceIR:
    """wp_ir built from an IR tensor already on the device: shape [nch, M]."""

    def __init__(self, ir, length):
        import torch
        assert ir.is_cuda and ir.dtype == torch.float32 and ir.dim() == 2 and ir.is_contiguous()
        capi.init()
        self.nch, self.M = int(ir.shape[0]), int(ir.shape[1])
        self.L = int(length)
        self.handle = capi.check_handle(
            capi.lib().wp_ir_create_dev(C.c_void_p(ir.data_ptr()), self.nch, self.M, self.M, self.L, _stream_ptr(torch)),
            "wp_ir_create_dev")

    def info(self, which):
        return capi.lib().wp_ir_info(self.handle, which)

    def close(self):
        if getattr(self, "handle", None):
            capi.lib().wp_ir_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchedConvolver:
    """njobs channel-jobs of one `go` each (lib/convolve.js:570-583), one launch sequence."""

    def __init__(self, ir, job_ch):
        self.ir = ir
        self.njobs = len(job_ch)
        arr = (C.c_int32 * self.njobs)(*[int(c) for c in job_ch])
        self.handle = capi.check_handle(capi.lib().wp_conv_create(ir.handle, self.njobs, arr), "wp_conv_create")
        self.out_len = ir.L + ir.M

    @property
    def workspace_bytes(self):
        return capi.lib().wp_conv_workspace_bytes(self.handle)

    def run(self, x, out, in_len=None, prev=None, prev_len=0):
        """x: [njobs, in_stride] float32 cuda; out: [njobs, out_stride >= L+M]."""
        import torch
        assert x.is_cuda and out.is_cuda and x.dtype == torch.float32 and out.dtype == torch.float32
        assert x.shape[0] == self.njobs and out.shape[0] == self.njobs and out.shape[1] >= self.out_len
        assert x.stride(1) == 1 and out.stride(1) == 1
        n = int(x.shape[1]) if in_len is None else int(in_len)
        capi.check(capi.lib().wp_conv_run_dev(
            self.handle, C.c_void_p(x.data_ptr()), x.stride(0), n, C.c_void_p(out.data_ptr()), out.stride(0),
            C.c_void_p(prev.data_ptr()) if prev is not None else None,
            prev.stride(0) if prev is not None else 0, int(prev_len), _stream_ptr(torch)), "wp_conv_run_dev")
        return out

    def run_mix(self, x, target, vol, in_len=None):
        """Fused reverb + mix: jobs (2s, 2s+1) are the stereo source s; target is
        [bus_left, bus_right] (1-D cuda tensors of equal length <= L+M), updated in
        place like mixBuffers does; vol: one gain per source."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[0] == self.njobs and x.stride(1) == 1
        assert len(target) == 2 and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in target)
        n = int(x.shape[1]) if in_len is None else int(in_len)
        tp = (C.c_void_p * 2)(target[0].data_ptr(), target[1].data_ptr())
        g = (C.c_double * (self.njobs // 2))(*[float(v) for v in vol])
        capi.check(capi.lib().wp_conv_mix_run_dev(self.handle, C.c_void_p(x.data_ptr()), x.stride(0), n, tp,
                                                  int(target[0].shape[0]), g, _stream_ptr(torch)), "wp_conv_mix_run_dev")
        return target

    def run_and_mix(self, x, out, target, vol, in_len=None):
        """go() results for every job into `out` AND the master bus into `target`, one pass."""
        import torch
        assert x.is_cuda and out.is_cuda and x.shape[0] == self.njobs and out.shape[0] == self.njobs
        assert out.shape[1] >= self.out_len and x.stride(1) == 1 and out.stride(1) == 1
        n = int(x.shape[1]) if in_len is None else int(in_len)
        tp = (C.c_void_p * 2)(target[0].data_ptr(), target[1].data_ptr())
        g = (C.c_double * (self.njobs // 2))(*[float(v) for v in vol])
        capi.check(capi.lib().wp_conv_run_mix_dev(self.handle, C.c_void_p(x.data_ptr()), x.stride(0), n,
                                                  C.c_void_p(out.data_ptr()), out.stride(0), tp, int(target[0].shape[0]), g,
                                                  _stream_ptr(torch)), "wp_conv_run_mix_dev")
        return out, target

    def profile(self, enable=True):
        capi.check(capi.lib().wp_conv_profile(self.handle, 1 if enable else 0), "wp_conv_profile")

    def profile_read(self):
        """(fwd_ms, mac_ms, inv_ms) of the last profiled run (synchronises)."""
        ms = (C.c_float * 3)()
        capi.check(capi.lib().wp_conv_profile_read(self.handle, ms), "wp_conv_profile_read")
        return tuple(float(v) for v in ms)

    def profile_sum(self):
        """((fwd_ms, mac_ms, inv_ms) added up, runs) over every profiled run since the last call of this
        (synchronises once; the runs themselves are queued without a wait between them)."""
        ms = (C.c_double * 3)()
        runs = C.c_int(0)
        capi.check(capi.lib().wp_conv_profile_sum(self.handle, ms, C.byref(runs)), "wp_conv_profile_sum")
        return tuple(float(v) for v in ms), int(runs.value)

    def close(self):
        if getattr(self, "handle", None):
            capi.lib().wp_conv_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceMix:
    """mixBuffers (src/mix-buffers.js) with every buffer on the device.  The
    source table is uploaded once and reused by every call."""

    def __init__(self, sources):
        """sources: list of dicts {ch: [tensor, (tensor)], rl, sl, vol, o} (tuple-form fields decoded)."""
        import torch
        self._keep = sources
        n = len(sources)
        host = (capi.WpSrc * max(n, 1))()
        for i, s in enumerate(sources):
            chans = s["ch"]
            host[i].nch = len(chans)
            for c in range(min(2, len(chans))):
                assert chans[c].is_cuda and chans[c].dtype == torch.float32 and chans[c].stride(-1) == 1
                host[i].ch[c] = chans[c].data_ptr()
            rl = int(s.get("rl", chans[0].shape[-1]))
            host[i].rl = rl
            host[i].sl = int(s.get("sl", rl))
            host[i].vol = float(s.get("vol", 1.0))
            host[i].o = int(s.get("o", 0))
            host[i].vol_vec = None
        raw = np.frombuffer(host, dtype=np.uint8, count=C.sizeof(capi.WpSrc) * n).copy()
        self.table = torch.from_numpy(raw).cuda()
        self.n = n

    def run(self, target):
        import torch
        tch = len(target)
        tp = (C.c_void_p * tch)(*[t.data_ptr() for t in target])
        capi.check(capi.lib().wp_mix_dev(tp, tch, int(target[0].shape[-1]), C.c_void_p(self.table.data_ptr()), self.n, 1,
                                         _stream_ptr(torch)), "wp_mix_dev")
        return target


def synth_noise(out, seed, start=0):
    """Fill a float32 cuda tensor (1-D, contiguous) with the integer-hash test noise."""
    import torch
    assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
    capi.check(capi.lib().wp_synth_noise_dev(C.c_void_p(out.data_ptr()), out.numel(), seed & 0xFFFFFFFF, int(start),
                                             _stream_ptr(torch)), "wp_synth_noise_dev")
    return out


def synth_ir(M, seed):
    out = np.empty(M, dtype=np.float32)
    capi.check(capi.lib().wp_synth_ir(out.ctypes.data_as(capi.f32p), M, seed & 0xFFFFFFFF), "wp_synth_ir")
    return out


def shard_range(n_items, rank, world):
    """Contiguous, balanced range of items owned by `rank` (tracks, stems, IR jobs)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Comm:
    """One rank of the track-sharded render: wp_comm (an RCCL communicator created through the
    C ABI).  `rendezvous` carries the 128-byte id from rank 0 to the others: a callable
    bytes -> bytes that returns rank 0's argument on every rank (e.g. a gloo broadcast)."""

    def __init__(self, rank, world, rendezvous=None):
        capi.init()
        L = capi.lib()
        idb = (C.c_ubyte * 128)()
        if world > 1 and rendezvous is None:
            raise ValueError("more than one rank needs a rendezvous callable to pass the id")
        why = None
        if rank == 0:
            try:
                capi.check(L.wp_comm_unique_id(idb), "wp_comm_unique_id")
            except Exception as e:  # the other ranks are waiting in the rendezvous: release them with an all-zero id
                why = e
                idb = (C.c_ubyte * 128)()
        raw = bytes(idb)
        if world > 1:
            raw = rendezvous(raw)
        if why is not None:
            raise why
        if not any(raw):
            raise RuntimeError("rank 0 could not make a communicator id")
        idb = (C.c_ubyte * 128).from_buffer_copy(raw)
        self.rank, self.world = rank, world
        self.handle = capi.check_handle(L.wp_comm_create(idb, world, rank), "wp_comm_create")

    @property
    def library(self):
        return capi.lib().wp_comm_library().decode()

    def reduce(self, bus, dst=0, out=None, stream=None):
        """Sum the rank-local buses onto rank `dst` (in place unless `out` is given), one message."""
        import torch
        assert bus.is_cuda and bus.dtype == torch.float32 and bus.is_contiguous()
        recv = bus if out is None else out
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else _stream_ptr(torch)
        capi.check(capi.lib().wp_bus_reduce(self.handle, C.c_void_p(bus.data_ptr()), C.c_void_p(recv.data_ptr()),
                                            bus.numel(), dst, sp), "wp_bus_reduce")
        return recv

    def allreduce(self, bus, out=None, stream=None):
        import torch
        assert bus.is_cuda and bus.dtype == torch.float32 and bus.is_contiguous()
        recv = bus if out is None else out
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else _stream_ptr(torch)
        capi.check(capi.lib().wp_bus_allreduce(self.handle, C.c_void_p(bus.data_ptr()), C.c_void_p(recv.data_ptr()),
                                               bus.numel(), sp), "wp_bus_allreduce")
        return recv

    def close(self):
        if getattr(self, "handle", None):
            capi.lib().wp_comm_destroy(self.handle)
            self.handle = None


def gloo_rendezvous(dist, src=0):
    """bytes -> bytes broadcast over an initialised torch.distributed group with CPU tensors."""
    def bcast(raw):
        import torch
        t = torch.tensor(list(raw), dtype=torch.uint8)
        dist.broadcast(t, src=src)
        return bytes(t.tolist())
    return bcast


def reduce_master_bus(bus, dist, dst=0, comm=None, group=None):
    """The path's one exchange step: sum the rank-local partial buses onto rank
    `dst`.  With `comm` (a Comm) the reduce is wp_bus_reduce, RCCL called through the C ABI;
    otherwise torch.distributed carries it (gloo for CPU tensors in the CPU rehearsal of the
    multi-rank flow; `group` selects a process group other than the default one).  `bus` is a
    single tensor [channels, frames]; one message per call."""
    if comm is not None:
        comm.reduce(bus, dst=dst)
        return bus
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if bus.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal without RCCL: stage through the host
            host = bus.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            bus.copy_(host)
        else:
            dist.reduce(bus, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return bus
