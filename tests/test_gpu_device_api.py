"""GPU tests of the device-resident entry points (wp_conv_run_dev, wp_mix_dev,
wp_reverb_block_dev, wp_synth_noise_dev) through dsp_amd.device, of the Node
shims, and size-independent properties at benchmark sizes."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch


def _node():
    node = shutil.which("node")
    if not node:
        pytest.skip("node is not installed on this box")
    if not os.path.exists(os.path.join(ROOT, "dsp_amd", "napi", "wpdsp.node")):
        pytest.skip("the N-API addon is not built (make -C dsp_amd/napi)")
    return node


def test_node_shims_parity_on_gpu():
    out = subprocess.run([_node(), "--no-warnings", os.path.join(ROOT, "tests", "node", "test_shims.mjs")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.strip().startswith("ok "), out.stdout


def test_node_addon_in_worker_threads():
    """Context-aware addon: two worker_threads use the shims concurrently with a kernel
    object cloned from the main thread; results equal the main thread's bit for bit."""
    out = subprocess.run([_node(), "--no-warnings", os.path.join(ROOT, "tests", "node", "test_workers.mjs")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ok workers 2" in out.stdout, out.stdout


def test_node_device_buffers_match_host_path():
    """DeviceBuffer in place of Float32Array[] (c.buffer, mixBuffers targets/sources, go input):
    buffer semantics, go, every mixBuffers source form, mono/stereo block reverb with resume from a
    cloned cached tail, and worker-thread rendered stems -- bit-identical to the host-array path."""
    out = subprocess.run([_node(), "--no-warnings", os.path.join(ROOT, "tests", "node", "test_device.mjs")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    # RCCL prints a version banner on stdout when the one-rank communicator is created
    assert out.stdout.strip().splitlines()[-1].startswith("ok device "), out.stdout


def test_synth_noise_device_matches_oracle(torch, oracle):
    from dsp_amd import device
    for n, seed, start in [(1000, 7, 0), (4097, 1001, 96000), (12345, 2 ** 31 + 5, 2 ** 31 - 100)]:
        t = torch.empty(n, dtype=torch.float32, device="cuda")
        device.synth_noise(t, seed, start)
        assert np.array_equal(t.cpu().numpy().view(np.uint32), oracle.synth_noise(n, seed, start).view(np.uint32))


@pytest.mark.parametrize("L,M,njobs", [(5000, 1200, 6), (20000, 9000, 5), (70000, 30000, 3), (4096, 4096, 4), (100, 50000, 2),
                                        (3000, 110000, 2), (1, 1, 1), (700, 7400, 3), (20000, 210000, 2)])
@pytest.mark.parametrize("mac", ["stream", "gather"])
def test_batched_convolver_vs_oracle(torch, oracle, L, M, njobs, mac):
    """Ragged sizes (L, M not multiples of the partition), several jobs, IR channel map;
    (3000, 110000) needs 27 partitions = two accumulating MAC passes; (1, 1) is the
    smallest legal FFT; (20000, 210000) uses the 8192-point plan with two passes; (700, 7400) is the small-partition plan with 8 partitions.
    Each case runs with both forms of the spectral multiply-accumulate (the streaming
    delay-line kernel and the small-batch gather kernel), which the library otherwise
    picks by batch size."""
    from dsp_amd import capi, device
    irs = np.stack([oracle.synth_ir(M, 40), oracle.synth_ir(M, 41)])
    ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    job_ch = [(j * 3) % 2 for j in range(njobs)]
    capi.check(capi.lib().wp_tune(2, {"stream": 1, "gather": 2}[mac]), "wp_tune")  # WP_TUNE_MAC_FORM
    try:
        conv = device.BatchedConvolver(ir, job_ch)
    finally:
        capi.lib().wp_tune(2, 0)
    assert capi.lib().wp_conv_info(conv.handle, 5) == (1 if mac == "gather" else 0)
    x = np.stack([oracle.synth_noise(L, 50 + j) for j in range(njobs)])
    xd = torch.from_numpy(x).cuda()
    stride = L + M + 5  # odd stride: exercises the unaligned store path
    out = torch.full((njobs, stride), 7.0, dtype=torch.float32, device="cuda")
    conv.run(xd, out)
    got = out.cpu().numpy()
    for j in range(njobs):
        want = oracle.fft_convolution(L, oracle.fft_process_kernel(L, irs[job_ch[j]]), M)(x[j])
        err = float(np.abs(got[j, :L + M].astype(np.float64) - want).max())
        assert err <= TOL, (j, err)
        assert (got[j, L + M:] == 7.0).all(), "wrote past L+M"
    # shorter input than L (zero padded like input.set into the padded array, convolve.js:589-590)
    n_in = L // 3
    conv.run(xd, out, in_len=n_in)
    want = oracle.fft_convolution(L, oracle.fft_process_kernel(L, irs[job_ch[0]]), M)(x[0][:n_in])
    assert float(np.abs(out[0, :L + M].cpu().numpy().astype(np.float64) - want).max()) <= TOL


def test_per_track_impulse_responses(torch, oracle):
    """Every job with an impulse response of its own (one IR handle with as many channels as
    jobs): results per job against the oracle, and the one-pass reverb+mix against the oracle's
    mix of those results."""
    from dsp_amd import device
    L, M, tracks = 9000, 5200, 5
    njobs = 2 * tracks
    irs = np.stack([oracle.synth_ir(M, 300 + j) for j in range(njobs)])
    ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    conv = device.BatchedConvolver(ir, list(range(njobs)))
    x = np.stack([oracle.synth_noise(L, 400 + j) for j in range(njobs)])
    xd = torch.from_numpy(x).cuda()
    out = torch.empty((njobs, L + M), dtype=torch.float32, device="cuda")
    bus = torch.zeros((2, L), dtype=torch.float32, device="cuda")
    vol = [0.5 / (s + 1) for s in range(tracks)]
    conv.run_and_mix(xd, out, [bus[0], bus[1]], vol)
    got = out.cpu().numpy()
    want = [oracle.fft_convolution(L, oracle.fft_process_kernel(L, irs[j]), M)(x[j]) for j in range(njobs)]
    for j in range(njobs):
        assert float(np.abs(got[j].astype(np.float64) - want[j]).max()) <= TOL, j
    want_bus = [np.zeros(L, np.float32), np.zeros(L, np.float32)]
    oracle.mix_buffers(want_bus, *[[[want[2 * s], want[2 * s + 1]], 1, vol[s], 0] for s in range(tracks)])
    for ch in range(2):
        assert float(np.abs(bus[ch].cpu().numpy().astype(np.float64) - want_bus[ch]).max()) <= TOL


def test_reverb_blocks_device_resident(torch, oracle):
    """wp_reverb_block_dev over several blocks == the oracle's impulse_reverb_stereo."""
    from dsp_amd import capi, device
    L, M, offset, blocks = 6000, 4500, 37, 4
    irs = [oracle.synth_ir(M, 60), oracle.synth_ir(M, 61)]
    ir = device.DeviceIR(torch.from_numpy(np.stack(irs)).cuda(), L)
    job_ch = (C.c_int32 * 2)(0, 1)
    rv = capi.check_handle(capi.lib().wp_reverb_create(ir.handle, 2, job_ch, offset), "wp_reverb_create")
    store = {}
    c = oracle.StubContext([oracle.synth_noise(L, 70), oracle.synth_noise(L, 71)], 0, {"u": irs}, store)
    fn = oracle.impulse_reverb_stereo(c, "u", offset=offset)
    try:
        for b in range(blocks):
            xs = [oracle.synth_noise(L, 70, b * L), oracle.synth_noise(L, 71, b * L)]
            want = fn(oracle.StubContext(xs, b * L, {"u": irs}, store))
            xd = torch.from_numpy(np.stack(xs)).cuda()
            curr, stride, v = C.c_void_p(), C.c_int64(), capi.WpReverbView()
            capi.check(capi.lib().wp_reverb_block_dev(rv, C.c_void_p(xd.data_ptr()), xd.stride(0), L, C.byref(curr),
                                                      C.byref(stride), C.byref(v), None), "wp_reverb_block_dev")
            torch.cuda.synchronize()
            assert (v.ret_off, v.ret_len) == (offset, L) and v.curr_len == L + M
            host = np.empty((2, stride.value), np.float32)
            import torch as _t
            buf = _t.empty((2, stride.value), dtype=_t.float32, device="cuda")
            # copy out of the library-owned buffer with a device-to-device memcpy through torch
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(buf.data_ptr()), curr, C.c_size_t(host.nbytes), 3)
            host = buf.cpu().numpy()
            for ch in range(2):
                got = host[ch, v.ret_off:v.ret_off + v.ret_len]
                assert float(np.abs(got.astype(np.float64) - want[ch]).max()) <= TOL, (b, ch)
    finally:
        capi.lib().wp_reverb_destroy(rv)


def test_ir_spectra_are_shared_between_handles(torch, oracle):
    """Two handles built from the same samples (what two worker threads do with one cached kernel
    object) hold one copy of the spectra; each stays usable after the other is destroyed, and a
    different IR of the same shape is not confused with it."""
    from dsp_amd import capi
    lib = capi.lib()
    f32p = C.POINTER(C.c_float)
    L, M = 3000, 2000
    irs = [oracle.synth_ir(M, 5), oracle.synth_ir(M, 6)]
    other = [irs[0].copy(), irs[1].copy()]
    other[1][M // 2] = np.nextafter(other[1][M // 2], np.float32(1))  # one ulp apart

    def make(chs):
        return capi.check_handle(lib.wp_ir_create((f32p * 2)(*[a.ctypes.data_as(f32p) for a in chs]), 2, M, L), "wp_ir_create")

    def go(h, ch, x):
        out = np.empty(L + M, np.float32)
        capi.check(lib.wp_convolve(h, ch, x.ctypes.data_as(f32p), x.size, out.ctypes.data_as(f32p)), "wp_convolve")
        return out

    x = oracle.synth_noise(L, 9)
    a, b, c = make(irs), make(irs), make(other)
    assert lib.wp_ir_info(a, 8) == 2 and lib.wp_ir_info(b, 8) == 2 and lib.wp_ir_info(c, 8) == 1
    ya = go(a, 1, x)
    assert np.array_equal(ya.view(np.uint32), go(b, 1, x).view(np.uint32))
    assert not np.array_equal(ya.view(np.uint32), go(c, 1, x).view(np.uint32))
    lib.wp_ir_destroy(a)
    assert lib.wp_ir_info(b, 8) == 1
    assert np.array_equal(ya.view(np.uint32), go(b, 1, x).view(np.uint32))
    want = oracle.fft_convolution(L, oracle.fft_process_kernel(L, irs[1]), M)(x)
    assert float(np.abs(ya.astype(np.float64) - want).max()) <= TOL
    lib.wp_ir_destroy(b)
    lib.wp_ir_destroy(c)


def test_device_buffers_that_miss_the_pool_are_carved_apart(torch):
    """A block loop whose host releases nothing for a while asks for a fresh result buffer per block
    (lib/convolve.js:575,582 allocate per call): buffers the pool cannot serve are carved from slabs.  Forty
    live buffers of 1.5 MB (more than one 64 MiB slab's worth) must not overlap, must come zero-filled, and
    must come back from the pool once released; a buffer too large for the slabs still works."""
    from dsp_amd import capi
    lib = capi.lib()
    f32p = C.POINTER(C.c_float)
    frames, n = 190016, 48  # (a size no other test uses: the pool holds none of it)
    bufs = [capi.check_handle(lib.wp_buf_create(2, frames), "wp_buf_create") for _ in range(n)]
    spans = sorted((lib.wp_buf_channel(b, 0), lib.wp_buf_info(b, 4)) for b in bufs)
    for (p0, b0), (p1, _) in zip(spans, spans[1:]):
        assert p0 + b0 <= p1, "buffers overlap"
    pat = np.arange(16, dtype=np.float32) + 1
    for i, b in enumerate(bufs):
        out = np.empty(16, np.float32)
        capi.check(lib.wp_buf_read(b, 1, frames - 16, out.ctypes.data_as(f32p), 16), "wp_buf_read")
        assert not out.any(), "a fresh buffer is zero-filled"
        capi.check(lib.wp_buf_write(b, 1, frames - 16, (pat * (i + 1)).ctypes.data_as(f32p), 16), "wp_buf_write")
    for i, b in enumerate(bufs):  # nobody wrote into a neighbour
        out = np.empty(16, np.float32)
        capi.check(lib.wp_buf_read(b, 1, frames - 16, out.ctypes.data_as(f32p), 16), "wp_buf_read")
        assert np.array_equal(out, pat * (i + 1))
    first = {lib.wp_buf_channel(b, 0) for b in bufs}
    for b in bufs:
        lib.wp_buf_release(b)
    again = [capi.check_handle(lib.wp_buf_create(2, frames), "wp_buf_create") for _ in range(n)]
    assert {lib.wp_buf_channel(b, 0) for b in again} == first, "released buffers come back from the pool"
    for b in again:
        out = np.empty(16, np.float32)
        capi.check(lib.wp_buf_read(b, 1, frames - 16, out.ctypes.data_as(f32p), 16), "wp_buf_read")
        assert not out.any(), "a recycled buffer is zero-filled again"
        lib.wp_buf_release(b)
    big = capi.check_handle(lib.wp_buf_create(2, 3000000), "wp_buf_create")  # 24 MB: its own allocation
    assert lib.wp_buf_info(big, 4) >= 24000000
    lib.wp_buf_release(big)


def test_device_buffers_c_abi_reverb_to_mix_chain(torch, oracle):
    """wp_buf + wp_reverb_block_to + wp_mix_dev: a stereo bar goes upload -> block reverb (result in
    a fresh wp_buf per block, carried tail read from the previous one) -> mixed into a device bus,
    compared with the oracle's impulse_reverb_stereo + mix_buffers.  Also ids, reference counts,
    fill/copy/read/write and their error codes."""
    from dsp_amd import capi
    lib = capi.lib()
    L, M, offset, blocks = 5000, 3100, 11, 3
    f32p = C.POINTER(C.c_float)
    irs = [oracle.synth_ir(M, 80), oracle.synth_ir(M, 81)]
    irp = (f32p * 2)(*[a.ctypes.data_as(f32p) for a in irs])
    ir = capi.check_handle(lib.wp_ir_create(irp, 2, M, L), "wp_ir_create")
    rv = capi.check_handle(lib.wp_reverb_create(ir, 2, (C.c_int32 * 2)(0, 1), offset), "wp_reverb_create")

    def write(buf, ch, arr, off=0):
        return lib.wp_buf_write(buf, ch, off, arr.ctypes.data_as(f32p), arr.size)

    def read(buf, ch, n, off=0):
        out = np.empty(n, np.float32)
        capi.check(lib.wp_buf_read(buf, ch, off, out.ctypes.data_as(f32p), n), "wp_buf_read")
        return out

    bus = capi.check_handle(lib.wp_buf_create(2, L), "wp_buf_create")
    bid = lib.wp_buf_id(bus)
    assert bid > 0 and lib.wp_buf_info(bus, 0) == 2 and lib.wp_buf_info(bus, 1) == L and lib.wp_buf_info(bus, 2) % 64 == 0
    assert not read(bus, 1, L).any()                                  # fresh buffers are zero
    again = capi.check_handle(lib.wp_buf_open(bid), "wp_buf_open")
    assert again == bus and lib.wp_buf_open(bid + 12345) is None
    lib.wp_buf_release(again)
    assert write(bus, 0, np.ones(8, np.float32), L - 7) == -2         # RangeError like TypedArray.set
    assert lib.wp_buf_fill(bus, 2, 0, 1, C.c_float(0), None) == -1    # no such channel
    capi.check(lib.wp_buf_fill(bus, 0, 10, 5, C.c_float(0.25), None), "wp_buf_fill")
    capi.check(lib.wp_buf_copy(bus, 1, 100, bus, 0, 8, 10, None), "wp_buf_copy")
    assert read(bus, 1, 10, 100).tolist() == [0, 0] + [0.25] * 5 + [0, 0, 0]
    capi.check(lib.wp_buf_fill(bus, 0, 0, L, C.c_float(0), None), "zero")
    capi.check(lib.wp_buf_fill(bus, 1, 0, L, C.c_float(0), None), "zero")

    store, want_bus = {}, [np.zeros(L, np.float32), np.zeros(L, np.float32)]
    fn = oracle.impulse_reverb_stereo(oracle.StubContext([np.zeros(L, np.float32)] * 2, 0, {"u": irs}, store), "u", offset=offset)
    inb = capi.check_handle(lib.wp_buf_create(2, L), "wp_buf_create")
    keep = []
    try:
        for b in range(blocks):
            xs = [oracle.synth_noise(L, 90, b * L), oracle.synth_noise(L, 91, b * L)]
            want = fn(oracle.StubContext(xs, b * L, {"u": irs}, store))
            oracle.mix_buffers(want_bus, [list(want), 1, 0.5, 0])
            for ch in range(2):
                capi.check(write(inb, ch, xs[ch]), "wp_buf_write")
            curr = capi.check_handle(lib.wp_buf_create(2, L + M), "wp_buf_create")
            keep.append(curr)                                         # the tail is a view into it
            v = capi.WpReverbView()
            capi.check(lib.wp_reverb_block_to(rv, lib.wp_buf_channel(inb, 0), lib.wp_buf_info(inb, 2), L,
                                              lib.wp_buf_channel(curr, 0), lib.wp_buf_info(curr, 2), C.byref(v), None),
                       "wp_reverb_block_to")
            src = capi.WpSrc()
            src.ch[0] = lib.wp_buf_channel(curr, 0) + 4 * v.ret_off
            src.ch[1] = lib.wp_buf_channel(curr, 1) + 4 * v.ret_off
            src.nch, src.rl, src.sl, src.vol, src.o = 2, v.ret_len, v.ret_len, 0.5, 0
            tgt = (C.c_void_p * 2)(lib.wp_buf_channel(bus, 0), lib.wp_buf_channel(bus, 1))
            capi.check(lib.wp_mix_dev(tgt, 2, L, C.byref(src), 1, 0, None), "wp_mix_dev")
            for ch in range(2):
                got = read(curr, ch, v.ret_len, v.ret_off)
                assert float(np.abs(got.astype(np.float64) - want[ch]).max()) <= TOL, (b, ch)
            if len(keep) > 2:
                lib.wp_buf_release(keep.pop(0))                       # recycled through the pool
        for ch in range(2):
            assert float(np.abs(read(bus, ch, L).astype(np.float64) - want_bus[ch]).max()) <= TOL
        assert lib.wp_reverb_tail_len(rv) == M - offset
    finally:
        lib.wp_reverb_destroy(rv)
        lib.wp_ir_destroy(ir)
        for b_ in keep + [inb, bus]:
            lib.wp_buf_release(b_)
    assert lib.wp_buf_open(bid) is None                               # gone with its last reference


def test_mix_dev_many_stems_bit_exact(torch, oracle):
    """mix-reduce with the stems resident on the device (the 4096-stem master
    reduce, scaled to what the oracle sums in seconds): plain fast path, then a
    looping/offset mix (modular path)."""
    from dsp_amd import device
    S, F = 96, 50000
    stems = np.stack([oracle.synth_noise(F, 3000 + s) for s in range(2 * S)]).reshape(S, 2, F)
    sd = torch.from_numpy(stems).cuda()
    for variant in ("contiguous", "modular"):
        if variant == "contiguous":
            desc = [{"ch": [sd[s, 0], sd[s, 1]], "vol": 1.0 / 4096} for s in range(S)]
            srcs = [([stems[s, 0], stems[s, 1]], 1, 1.0 / 4096, 0) for s in range(S)]
        else:
            desc = [{"ch": [sd[s, 0], sd[s, 1]], "sl": int(F * .75), "vol": 1.3, "o": 1000 * s} for s in range(S)]
            srcs = [([stems[s, 0], stems[s, 1]], .75, 1.3, 1000 * s) for s in range(S)]
        bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
        device.DeviceMix(desc).run([bus[0], bus[1]])
        want = [np.zeros(F, np.float32), np.zeros(F, np.float32)]
        oracle.mix_buffers(want, *srcs)
        got = bus.cpu().numpy()
        for ch in range(2):
            assert_bits_equal(got[ch], want[ch].view(np.uint32), "%s ch%d" % (variant, ch))


@pytest.mark.parametrize("tch", [2, 1])
def test_mix_dev_long_target_bit_exact(torch, oracle, tch):
    """Targets of 2 Mi frames and more take the 4-rows-per-workgroup tiles: whole tiles plus a
    ragged tail, stereo and mono sources, odd source count, an unaligned source view, constant
    gains -- bit-exact against the oracle."""
    from dsp_amd import device
    F = 512 * 4096 + 4096 + 1234
    rl = F + 7
    stems = [oracle.synth_noise(rl, 7000 + i) for i in range(9)]
    sd = [torch.from_numpy(a).cuda() for a in stems]
    desc = [{"ch": [sd[0][:F], sd[1][:F]], "vol": 1.0 / 3},
            {"ch": [sd[2][:F]], "vol": 0.7},
            {"ch": [sd[3][3:3 + F], sd[4][3:3 + F]], "vol": 1.0},           # 12 bytes off a 16-byte boundary
            {"ch": [sd[5], sd[6]], "sl": rl, "o": 5, "vol": -1.25},          # contiguous run starting at frame 5
            {"ch": [sd[7][:F], sd[8][:F]], "vol": 0.5}]
    srcs = [([stems[0][:F], stems[1][:F]], 1, 1.0 / 3, 0), ([stems[2][:F]], 1, 0.7, 0),
            [stems[3][3:3 + F], stems[4][3:3 + F]], ([stems[5], stems[6]], 1, -1.25, 5),
            ([stems[7][:F], stems[8][:F]], 1, 0.5, 0)]
    init = [oracle.synth_noise(F, 7100 + c) for c in range(tch)]
    bus = torch.from_numpy(np.stack(init)).cuda()
    device.DeviceMix(desc).run([bus[c] for c in range(tch)])
    want = [a.copy() for a in init]
    oracle.mix_buffers(want, *srcs)
    got = bus.cpu().numpy()
    for ch in range(tch):
        assert_bits_equal(got[ch], want[ch].view(np.uint32), "long target ch%d" % ch)


def test_mix_full_size_properties(torch):
    """At the master-reduce size (60 s at 48 kHz, 64 stems here to bound memory):
    mixing k copies of one coarse-grained stem with vol=1/k (k a power of two)
    reproduces the stem exactly; a stereo -> mono fold-down matches its formula."""
    from dsp_amd import device
    F, k = 2880000, 64
    stem = torch.empty((2, F), dtype=torch.float32, device="cuda")
    device.synth_noise(stem[0], 11)
    device.synth_noise(stem[1], 12)
    stem = torch.round(stem * 1024) / 1024  # 11-bit samples: every partial sum j/k * x is exact in f32
    bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
    device.DeviceMix([{"ch": [stem[0], stem[1]], "vol": 1.0 / k} for _ in range(k)]).run([bus[0], bus[1]])
    assert torch.equal(bus, stem)
    # mono fold-down: ((L+R)/2)*1 of a stereo stem into a mono bus, twice -> L+R
    mono = torch.zeros((1, F), dtype=torch.float32, device="cuda")
    device.DeviceMix([{"ch": [stem[0], stem[1]]}, {"ch": [stem[0], stem[1]]}]).run([mono[0]])
    ref = ((stem[0].double() + stem[1].double()) / 2).float()
    ref = (ref.double() + (stem[0].double() + stem[1].double()) / 2).float()
    assert torch.equal(mono[0], ref)


@pytest.mark.parametrize("L,M,tracks,tl,part", [(9000, 5000, 5, 9000, 0), (20000, 3000, 3, 23000, 0), (4096, 4096, 2, 4097, 0),
                                                 (70000, 20000, 4, 65536, 0),
                                                 (30000, 100000, 3, 30000, 0),    # 8192-point plan, 13 partitions
                                                 (20000, 192000, 2, 50000, 0),    # 4 s IR: 8192-point plan, 24 partitions
                                                 (3000, 2000, 3, 3000, 0),        # 1024-point plan
                                                 (9000, 5000, 2, 9001, 2048)])    # 2048-point plan
def test_fused_reverb_mix_is_bit_identical_to_unfused(torch, L, M, tracks, tl, part):
    """wp_conv_mix_run_dev == wp_conv_run_dev followed by wp_mix_dev, bit for bit
    (same samples, same f64 fold, same source order), for ragged sizes, a
    non-zero starting bus and tl shorter/longer than L, with every partition size
    the library has a plan for."""
    from dsp_amd import capi, device
    njobs = 2 * tracks
    irs = torch.from_numpy(np.stack([device.synth_ir(M, 80), device.synth_ir(M, 81)])).cuda()
    capi.check(capi.lib().wp_tune(0, part), "wp_tune")  # WP_TUNE_PARTITION
    try:
        ir = device.DeviceIR(irs, L)
    finally:
        capi.lib().wp_tune(0, 0)
    want_B = part or (1024 if L + M <= 8192 else (8192 if M > 24 * 4096 else 4096))
    assert ir.info(3) == want_B
    conv = device.BatchedConvolver(ir, [j % 2 for j in range(njobs)])
    x = torch.empty((njobs, L), dtype=torch.float32, device="cuda")
    for j in range(njobs):
        device.synth_noise(x[j], 500 + j)
    vol = [1.3 / (1 + t) for t in range(tracks)]
    out = torch.empty((njobs, L + M + (L + M) % 2), dtype=torch.float32, device="cuda")
    conv.run(x, out)
    bus_a = torch.empty((2, tl), dtype=torch.float32, device="cuda")
    device.synth_noise(bus_a[0], 9)
    device.synth_noise(bus_a[1], 10)
    bus_b = bus_a.clone()
    device.DeviceMix([{"ch": [out[2 * t], out[2 * t + 1]], "rl": L + M, "sl": L + M, "vol": vol[t]} for t in range(tracks)]).run([bus_a[0], bus_a[1]])
    conv.run_mix(x, [bus_b[0], bus_b[1]], vol)
    torch.cuda.synchronize()
    assert torch.equal(bus_a.view(torch.int32), bus_b.view(torch.int32))
    # one-pass variant that also writes the per-track results: both outputs identical
    bus_c = bus_a.clone()
    device.synth_noise(bus_c[0], 9)
    device.synth_noise(bus_c[1], 10)
    out_c = torch.full_like(out, 3.0)
    conv.run_and_mix(x, out_c, [bus_c[0], bus_c[1]], vol)
    torch.cuda.synchronize()
    assert torch.equal(bus_a.view(torch.int32), bus_c.view(torch.int32))
    assert torch.equal(out[:, :L + M].view(torch.int32), out_c[:, :L + M].view(torch.int32))
    assert (out_c[:, L + M:] == 3.0).all()
    # the plan remembers the gain table it uploaded last: different gains on the same plan take effect
    vol2 = [-0.25 * (1 + t) for t in range(tracks)]
    bus_d = torch.zeros((2, tl), dtype=torch.float32, device="cuda")
    bus_e = torch.zeros((2, tl), dtype=torch.float32, device="cuda")
    device.DeviceMix([{"ch": [out[2 * t], out[2 * t + 1]], "rl": L + M, "sl": L + M, "vol": vol2[t]} for t in range(tracks)]).run([bus_d[0], bus_d[1]])
    conv.run_mix(x, [bus_e[0], bus_e[1]], vol2)
    torch.cuda.synchronize()
    assert torch.equal(bus_d.view(torch.int32), bus_e.view(torch.int32))


@pytest.mark.parametrize("L,M,tracks,per_track_ir", [(30000, 100000, 3, False), (20000, 192000, 2, False),
                                                      (30000, 100000, 2, True), (3000, 2000, 3, False)])
def test_fused_paths_vs_oracle_all_plans(torch, oracle, L, M, tracks, per_track_ir):
    """The fused inverse-transform + mix kernels of the 8192-point plan (what a 4 s impulse
    response selects: M = 192 000 gives 24 partitions of 8192) and of the 1024-point plan,
    against the oracle: every job's go() result, the bus of wp_conv_run_mix_dev and the bus of
    wp_conv_mix_run_dev, with a shared stereo IR and with an IR per job."""
    from dsp_amd import device
    njobs = 2 * tracks
    nir = njobs if per_track_ir else 2
    irs = np.stack([oracle.synth_ir(M, 600 + j) for j in range(nir)])
    ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    assert ir.info(3) == (8192 if M > 24 * 4096 else 1024)
    job_ch = list(range(njobs)) if per_track_ir else [j % 2 for j in range(njobs)]
    conv = device.BatchedConvolver(ir, job_ch)
    x = np.stack([oracle.synth_noise(L, 700 + j) for j in range(njobs)])
    xd = torch.from_numpy(x).cuda()
    out = torch.full((njobs, L + M + 2), 9.0, dtype=torch.float32, device="cuda")
    tl = L + M // 2
    bus = torch.zeros((2, tl), dtype=torch.float32, device="cuda")
    bus2 = torch.zeros((2, tl), dtype=torch.float32, device="cuda")
    vol = [0.75 / (s + 1) for s in range(tracks)]
    conv.run_and_mix(xd, out, [bus[0], bus[1]], vol)
    conv.run_mix(xd, [bus2[0], bus2[1]], vol)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    kern = [oracle.fft_process_kernel(L, irs[c]) for c in range(nir)]
    want = [oracle.fft_convolution(L, kern[job_ch[j]], M)(x[j]) for j in range(njobs)]
    for j in range(njobs):
        assert float(np.abs(got[j, :L + M].astype(np.float64) - want[j]).max()) <= TOL, j
        assert (got[j, L + M:] == 9.0).all()
    want_bus = [np.zeros(tl, np.float32), np.zeros(tl, np.float32)]
    oracle.mix_buffers(want_bus, *[[[want[2 * s], want[2 * s + 1]], 1, vol[s], 0] for s in range(tracks)])
    for ch in range(2):
        assert float(np.abs(bus[ch].cpu().numpy().astype(np.float64) - want_bus[ch]).max()) <= TOL
    assert torch.equal(bus.view(torch.int32), bus2.view(torch.int32))


@pytest.mark.parametrize("L,M,tracks,cap,lim,runs", [
    (600000, 5000, 2, 0, (18, 20), [(262144, 3)]),                                   # three frames of 2^18
    (1150001, 96000, 1, 1, (18, 20), [(524288, 3)]),                                 # one frame size only: 2^19; odd result length
    (1300000, 96000, 1, 0, (18, 20), [(1048576, 1), (262144, 3)]),                   # a long frame, the rest in short ones
    (1440000, 96000, 1, 0, (18, 20), [(1048576, 1), (524288, 1), (262144, 1)]),      # three frame sizes
    (900000, 300000, 1, 0, (18, 20), [(1048576, 2)]),                                # M > 2^18: only 2^20 holds the kernel
    (600000, 5000, 2, 0, (0, 0), [(131072, 5)]),                                     # frames of 2^17 samples (32 rows)
    (1300000, 96000, 1, 0, (0, 0), [(1048576, 1), (524288, 1), (131072, 1)]),        # the tail in one frame of 2^17
    (2880000, 96000, 1, 0, (0, 0), [(1048576, 3), (262144, 1)]),                     # the 60 s render's cover with a shared stereo response
    (2880000, 96000, 1, 0, (17, 21), [(2097152, 1), (1048576, 1), (131072, 1)]),     # its cover with the fewest frame samples: 2^21 (rows of 4096 points) + 2^20 + 2^17
    (1000000, 800000, 1, 0, (0, 0), [(2097152, 2)]),                                 # M > 3/4 * 2^20: only 2^21 holds the kernel
])
def test_long_frame_form_vs_oracle(torch, oracle, L, M, tracks, cap, lim, runs):
    """The long-frame form of the convolution (whole impulse response per frame, four-step transform)
    against the oracle: plain results with an odd output stride and a short input, the one-pass
    results + bus, the bus-only entry point, and the bus against wp_mix_dev of the written results
    bit for bit.  A result is covered by up to three runs of equal frames (2^21 ... 2^17 samples);
    `cap` limits the number of runs (WP_TUNE_LF_RUNS), `lim` the frame sizes (2^lim[0] ... 2^lim[1])."""
    from dsp_amd import capi, device
    lib = capi.lib()
    njobs = 2 * tracks
    irs = np.stack([oracle.synth_ir(M, 800 + c) for c in range(2)])
    capi.check(lib.wp_tune(5, 2), "wp_tune")  # WP_TUNE_CONV_FORM: long-frame where the sizes allow
    capi.check(lib.wp_tune(6, cap), "wp_tune")  # WP_TUNE_LF_RUNS
    capi.check(lib.wp_tune(10, lim[0]), "wp_tune")  # WP_TUNE_LF_MIN_FRAME
    capi.check(lib.wp_tune(9, lim[1]), "wp_tune")  # WP_TUNE_LF_MAX_FRAME
    capi.check(lib.wp_tune(11, 1 if lim == (17, 21) else 0), "wp_tune")  # WP_TUNE_LF_COST: 1 = bytes only
    try:
        cover = (C.c_int64 * 12)()
        assert lib.wp_lf_cover(L, M, 2, cover) == len(runs)
        ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    finally:
        lib.wp_tune(5, 0)
        lib.wp_tune(6, 0)
        lib.wp_tune(9, 0)
        lib.wp_tune(10, 0)
        lib.wp_tune(11, 0)
    assert ir.info(9) == runs[0][0] and ir.info(10) == len(runs) and ir.info(11) == sum(f * n for f, n in runs)
    job_ch = [(j + 1) % 2 for j in range(njobs)]
    conv = device.BatchedConvolver(ir, job_ch)
    assert lib.wp_conv_info(conv.handle, 6) == 1 and lib.wp_conv_info(conv.handle, 10) == len(runs)
    base = 0
    for i, (frame, nfr) in enumerate(runs):
        got_run = [lib.wp_conv_info(conv.handle, 16 + 4 * i + k) for k in range(4)]
        assert got_run[:2] == [frame, nfr] and got_run[3] == base, (i, got_run)
        base += nfr * got_run[2]
    assert base >= L + M
    x = np.stack([oracle.synth_noise(L, 900 + j) for j in range(njobs)])
    xd = torch.from_numpy(x).cuda()
    kern = [oracle.fft_process_kernel(L, irs[c]) for c in range(2)]
    want = [oracle.fft_convolution(L, kern[job_ch[j]], M)(x[j]) for j in range(njobs)]
    out = torch.full((njobs, L + M + 5), 7.0, dtype=torch.float32, device="cuda")  # odd stride: unaligned rows
    conv.run(xd, out)
    got = out.cpu().numpy()
    for j in range(njobs):
        err = float(np.abs(got[j, :L + M].astype(np.float64) - want[j]).max())
        assert err <= TOL, (j, err)
        assert (got[j, L + M:] == 7.0).all(), "wrote past L+M"
    n_in = L // 3 + 1
    conv.run(xd, out, in_len=n_in)
    w0 = oracle.fft_convolution(L, kern[job_ch[0]], M)(x[0][:n_in])
    assert float(np.abs(out[0, :L + M].cpu().numpy().astype(np.float64) - w0).max()) <= TOL
    # one pass with the bus, and bus only
    out2 = torch.full((njobs, L + M + 2), 3.0, dtype=torch.float32, device="cuda")
    tl = L + M // 2
    bus = torch.empty((2, tl), dtype=torch.float32, device="cuda")
    device.synth_noise(bus[0], 5)
    device.synth_noise(bus[1], 6)
    bus0 = bus.clone()
    bus2 = bus.clone()
    bus3 = bus.clone()
    vol = [0.75 / (s + 1) for s in range(tracks)]
    conv.run_and_mix(xd, out2, [bus[0], bus[1]], vol)
    conv.run_mix(xd, [bus2[0], bus2[1]], vol)
    torch.cuda.synchronize()
    got2 = out2.cpu().numpy()
    for j in range(njobs):
        assert float(np.abs(got2[j, :L + M].astype(np.float64) - want[j]).max()) <= TOL, j
        assert (got2[j, L + M:] == 3.0).all()
    assert torch.equal(bus.view(torch.int32), bus2.view(torch.int32))
    device.DeviceMix([{"ch": [out2[2 * t], out2[2 * t + 1]], "rl": L + M, "sl": L + M, "vol": vol[t]} for t in range(tracks)]).run([bus3[0], bus3[1]])
    assert torch.equal(bus.view(torch.int32), bus3.view(torch.int32))
    want_bus = [bus0[0].cpu().numpy().copy(), bus0[1].cpu().numpy().copy()]
    oracle.mix_buffers(want_bus, *[[[want[2 * s], want[2 * s + 1]], 1, vol[s], 0] for s in range(tracks)])
    for ch in range(2):
        assert float(np.abs(bus[ch].cpu().numpy().astype(np.float64) - want_bus[ch]).max()) <= TOL
    # gains that are powers of two take the one-instruction fold: same bits as wp_mix_dev's binary64 arithmetic
    vol2 = [(-1.0 if s % 2 else 1.0) / (1 << (3 * s + 1)) for s in range(tracks)]
    b4, b5, b6 = bus0.clone(), bus0.clone(), bus0.clone()
    conv.run_and_mix(xd, out2, [b4[0], b4[1]], vol2)
    conv.run_mix(xd, [b5[0], b5[1]], vol2)
    device.DeviceMix([{"ch": [out2[2 * t], out2[2 * t + 1]], "rl": L + M, "sl": L + M, "vol": vol2[t]} for t in range(tracks)]).run([b6[0], b6[1]])
    assert torch.equal(b4.view(torch.int32), b6.view(torch.int32)) and torch.equal(b5.view(torch.int32), b6.view(torch.int32))


@pytest.mark.parametrize("own_ir", [False, True])
@pytest.mark.parametrize("L,M,frame0", [(300000, 3000, None), (1000000, 96000, 1048576), (2100000, 96000, 2097152)])
def test_long_frame_batch_row_pass_vs_oracle(torch, oracle, own_ir, L, M, frame0):
    """Batches of 16 jobs or more take the row pass that walks eight jobs per workgroup: 18 jobs (two full
    groups and a group of two) with a shared stereo impulse response (kernel spectra stored as product
    coefficients) and with an impulse response per job (plain spectra), every job against the oracle, the
    bus of the one-pass entry point against the oracle's mix of its own results, and the batch against
    one-job calls of the same handle bit for bit.  The second size is covered by a frame of 2^20 samples
    (256 rows of 2048 points) and a frame of 2^18, the third by a frame of 2^21 samples (256 rows of 4096 points: the
    largest frame of the 60 s render) and a shorter one."""
    from dsp_amd import capi, device
    lib = capi.lib()
    njobs = 18
    nir = njobs if own_ir else 2
    irs = np.stack([oracle.synth_ir(M, 700 + c) for c in range(nir)])
    capi.check(lib.wp_tune(5, 2), "wp_tune")  # long-frame form
    capi.check(lib.wp_tune(11, 1), "wp_tune")  # cover with the fewest frame samples, whatever the row length
    try:
        ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    finally:
        lib.wp_tune(5, 0)
        lib.wp_tune(11, 0)
    assert ir.info(10) >= 1
    job_ch = list(range(njobs)) if own_ir else [j % 2 for j in range(njobs)]
    conv = device.BatchedConvolver(ir, job_ch)
    assert lib.wp_conv_info(conv.handle, 6) == 1
    if frame0 is not None:
        assert lib.wp_conv_info(conv.handle, 16) == frame0, "run 0's frame length"
    x = np.stack([oracle.synth_noise(L, 1200 + j) for j in range(njobs)])
    xd = torch.from_numpy(x).cuda()
    out = torch.empty((njobs, L + M), dtype=torch.float32, device="cuda")
    conv.run(xd, out)
    got = out.cpu().numpy()
    kern = [oracle.fft_process_kernel(L, irs[c]) for c in range(nir)]
    want = [oracle.fft_convolution(L, kern[job_ch[j]], M)(x[j]) for j in range(njobs)]
    worst = 0.0
    for j in range(njobs):
        err = float(np.abs(got[j].astype(np.float64) - want[j]).max())
        worst = max(worst, err)
        assert err <= TOL, (j, err)
    for j in (0, 9, 17):  # the one-job plan runs the row pass with one job per workgroup: same arithmetic
        one = device.BatchedConvolver(ir, [job_ch[j]])
        o1 = torch.empty((1, L + M), dtype=torch.float32, device="cuda")
        one.run(xd[j:j + 1], o1)
        assert torch.equal(o1[0].view(torch.int32), out[j].view(torch.int32)), j
    # the one-pass entry point (results + bus): same samples as the plain run, bus = the oracle's mixBuffers
    # of the oracle's results (src/mix-buffers.js:49-52), gains that are not powers of two
    tracks = njobs // 2
    vol = [0.9 / (s + 2) for s in range(tracks)]
    out2 = torch.empty((njobs, L + M), dtype=torch.float32, device="cuda")
    bus = torch.zeros((2, L), dtype=torch.float32, device="cuda")
    conv.run_and_mix(xd, out2, [bus[0], bus[1]], vol)
    torch.cuda.synchronize()
    assert torch.equal(out2.view(torch.int32), out.view(torch.int32))
    want_bus = [np.zeros(L, np.float32), np.zeros(L, np.float32)]
    oracle.mix_buffers(want_bus, *[[[want[2 * s][:L], want[2 * s + 1][:L]], 1, vol[s], 0] for s in range(tracks)])
    for ch in range(2):
        assert float(np.abs(bus[ch].cpu().numpy().astype(np.float64) - want_bus[ch]).max()) <= TOL, ch
    # the later runs of frames fold the bus inside their inverse pass or through the mix kernel over their range
    # (WP_TUNE_LF_SPLIT 1 / 2; the library picks per run): the same fold in the same order, the same bits
    for how in (1, 2):
        capi.check(lib.wp_tune(12, how), "wp_tune")
        try:
            b2 = torch.zeros((2, L), dtype=torch.float32, device="cuda")
            o2 = torch.empty((njobs, L + M), dtype=torch.float32, device="cuda")
            conv.run_and_mix(xd, o2, [b2[0], b2[1]], vol)
            torch.cuda.synchronize()
        finally:
            lib.wp_tune(12, 0)
        assert torch.equal(b2.view(torch.int32), bus.view(torch.int32)), how
        assert torch.equal(o2.view(torch.int32), out.view(torch.int32)), how
    # bus-fold workgroups of 256-row frames over tiles of 16 columns only / of 32 only (WP_TUNE_LF_WIDE 1 / 2; the
    # library schedules a split): a column's transform does not depend on its tile, so again the same bits
    for how in (1, 2):
        capi.check(lib.wp_tune(13, how), "wp_tune")
        try:
            b2 = torch.zeros((2, L), dtype=torch.float32, device="cuda")
            o2 = torch.empty((njobs, L + M), dtype=torch.float32, device="cuda")
            conv.run_and_mix(xd, o2, [b2[0], b2[1]], vol)
            torch.cuda.synchronize()
        finally:
            lib.wp_tune(13, 0)
        assert torch.equal(b2.view(torch.int32), bus.view(torch.int32)), ("tile width", how)
        assert torch.equal(o2.view(torch.int32), out.view(torch.int32)), ("tile width", how)


def test_profile_sum_counts_every_run(torch, oracle):
    """wp_conv_profile_sum: the three pass durations added up over runs queued back to back, more of them than the
    library keeps event sets for, and a fresh sum after every call; wp_conv_profile_read still returns the last run."""
    from dsp_amd import capi, device
    L, M, njobs = 20000, 3000, 4
    irs = np.stack([oracle.synth_ir(M, 770 + c) for c in range(2)])
    ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    conv = device.BatchedConvolver(ir, [0, 1, 0, 1])
    x = torch.from_numpy(np.stack([oracle.synth_noise(L, 1400 + j) for j in range(njobs)])).cuda()
    out = torch.empty((njobs, L + M), dtype=torch.float32, device="cuda")
    conv.profile(True)
    for _ in range(3):
        conv.run(x, out)
    ms, runs = conv.profile_sum()
    assert runs == 3 and all(v > 0 for v in ms)
    for _ in range(75):
        conv.run(x, out)
    last = conv.profile_read()
    ms, runs = conv.profile_sum()
    assert runs == 75 and all(v > 0 for v in ms) and all(v > 0 for v in last)
    assert sum(last) < sum(ms)
    assert conv.profile_sum() == ((0.0, 0.0, 0.0), 0)
    want = oracle.fft_convolution(L, oracle.fft_process_kernel(L, irs[1]), M)(x[3].cpu().numpy())
    assert float(np.abs(out[3].cpu().numpy().astype(np.float64) - want).max()) <= TOL


def test_long_frame_wide_and_narrow_chains_of_one_run(torch, oracle):
    """Three frames of 2^20 samples per job (the 60 s render's first run of frames): on 256 CUs the bus-fold pass
    takes tiles of 32 columns for the first two frames and of 16 columns, in a launch that starts at frame 2, for
    the third.  Every job against the oracle, the bus over all three frames against the oracle's mixBuffers of
    its results (src/mix-buffers.js:49-52), and bit for bit against the pass on narrow tiles only."""
    from dsp_amd import capi, device
    lib = capi.lib()
    L, M, njobs = 2300000, 96000, 16
    irs = np.stack([oracle.synth_ir(M, 760 + c) for c in range(2)])
    for knob, val in ((5, 2), (9, 20), (6, 1)):  # long-frame form, frames of at most 2^20 samples, one run of frames
        capi.check(lib.wp_tune(knob, val), "wp_tune")
    try:
        ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    finally:
        for knob in (5, 9, 6):
            lib.wp_tune(knob, 0)
    job_ch = [j % 2 for j in range(njobs)]
    conv = device.BatchedConvolver(ir, job_ch)
    assert lib.wp_conv_info(conv.handle, 6) == 1 and lib.wp_conv_info(conv.handle, 10) == 1
    assert lib.wp_conv_info(conv.handle, 16) == 1048576 and lib.wp_conv_info(conv.handle, 17) == 3
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    wide = lib.wp_lf_wide_frames(2048, 3, cus)
    x = np.stack([oracle.synth_noise(L, 1300 + j) for j in range(njobs)])
    xd = torch.from_numpy(x).cuda()
    tracks = njobs // 2
    vol = [0.8 / (s + 1.5) for s in range(tracks)]
    out = torch.empty((njobs, L + M), dtype=torch.float32, device="cuda")
    bus = torch.zeros((2, L), dtype=torch.float32, device="cuda")
    conv.run_and_mix(xd, out, [bus[0], bus[1]], vol)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    kern = [oracle.fft_process_kernel(L, irs[c]) for c in range(2)]
    want = [oracle.fft_convolution(L, kern[job_ch[j]], M)(x[j]) for j in range(njobs)]
    for j in range(njobs):
        assert float(np.abs(got[j].astype(np.float64) - want[j]).max()) <= TOL, j
    want_bus = [np.zeros(L, np.float32), np.zeros(L, np.float32)]
    oracle.mix_buffers(want_bus, *[[[want[2 * s][:L], want[2 * s + 1][:L]], 1, vol[s], 0] for s in range(tracks)])
    for ch in range(2):
        assert float(np.abs(bus[ch].cpu().numpy().astype(np.float64) - want_bus[ch]).max()) <= TOL, ch
    for how in (1, 2):  # narrow tiles only, wide tiles only
        capi.check(lib.wp_tune(13, how), "wp_tune")
        try:
            b2 = torch.zeros((2, L), dtype=torch.float32, device="cuda")
            o2 = torch.empty((njobs, L + M), dtype=torch.float32, device="cuda")
            conv.run_and_mix(xd, o2, [b2[0], b2[1]], vol)
            torch.cuda.synchronize()
        finally:
            lib.wp_tune(13, 0)
        assert torch.equal(b2.view(torch.int32), bus.view(torch.int32)), (how, wide)
        assert torch.equal(o2.view(torch.int32), out.view(torch.int32)), (how, wide)
    assert 0 <= wide <= 3 and (cus != 256 or wide == 2)


def test_long_frame_edge_cases(torch, oracle):
    """Long-frame plans at the edges the partitioned form is tested at: no input at all (zeros), a carried tail
    added to the head of the results (lib/impulse-reverb.js:14-16), a bus shorter than the first run of frames
    (later frames are skipped by the bus-only entry point, not by the one that writes results), and a bus of
    length zero."""
    from dsp_amd import capi, device
    lib = capi.lib()
    L, M, njobs = 700000, 4000, 2
    irs = np.stack([oracle.synth_ir(M, 810 + c) for c in range(2)])
    capi.check(lib.wp_tune(5, 2), "wp_tune")
    try:
        ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    finally:
        lib.wp_tune(5, 0)
    conv = device.BatchedConvolver(ir, [0, 1])
    assert lib.wp_conv_info(conv.handle, 6) == 1
    x = np.stack([oracle.synth_noise(L, 910 + j) for j in range(njobs)])
    xd = torch.from_numpy(x).cuda()
    want = [oracle.fft_convolution(L, oracle.fft_process_kernel(L, irs[j]), M)(x[j]) for j in range(njobs)]
    out = torch.full((njobs, L + M), 5.0, dtype=torch.float32, device="cuda")
    conv.run(xd, out, in_len=0)
    assert float(out.abs().max()) == 0.0
    prev = torch.from_numpy(np.stack([oracle.synth_noise(M, 930 + j) for j in range(njobs)])).cuda()
    conv.run(xd, out, prev=prev, prev_len=M - 7)
    got = out.cpu().numpy()
    for j in range(njobs):
        w = want[j].astype(np.float64).copy()
        w[:M - 7] = (want[j][:M - 7].astype(np.float32) + prev[j, :M - 7].cpu().numpy()).astype(np.float64)
        assert float(np.abs(got[j].astype(np.float64) - w).max()) <= 2 * TOL, j
    hop0 = lib.wp_conv_info(conv.handle, 18)
    for tl in (hop0 // 3, 0):
        bus = torch.full((2, max(tl, 1)), 0.25, dtype=torch.float32, device="cuda")
        bus2 = bus.clone()
        out2 = torch.full((njobs, L + M), 9.0, dtype=torch.float32, device="cuda")
        tp = (C.c_void_p * 2)(bus[0].data_ptr(), bus[1].data_ptr())
        tp2 = (C.c_void_p * 2)(bus2[0].data_ptr(), bus2[1].data_ptr())
        vol = (C.c_double * 1)(0.5)
        capi.check(lib.wp_conv_mix_run_dev(conv.handle, C.c_void_p(xd.data_ptr()), xd.stride(0), L, tp, tl, vol, None), "wp_conv_mix_run_dev")
        capi.check(lib.wp_conv_run_mix_dev(conv.handle, C.c_void_p(xd.data_ptr()), xd.stride(0), L, C.c_void_p(out2.data_ptr()),
                                           out2.stride(0), tp2, tl, vol, None), "wp_conv_run_mix_dev")
        torch.cuda.synchronize()
        assert torch.equal(bus.view(torch.int32), bus2.view(torch.int32))
        g2 = out2.cpu().numpy()
        for j in range(njobs):
            assert float(np.abs(g2[j].astype(np.float64) - want[j]).max()) <= TOL, (tl, j)
        if tl:
            wb = np.float32(0.25) + (want[0][:tl].astype(np.float32) * np.float32(0.5))
            assert float(np.abs(bus[0, :tl].cpu().numpy().astype(np.float64) - wb.astype(np.float64)).max()) <= TOL
        else:
            assert float((bus - 0.25).abs().max()) == 0.0


def test_mix_4096_aliased_stems_exact(torch):
    """The master reduce at its full source count: 4096 stereo sources (all views of one
    60 s stem, so the test holds 46 MB instead of 94 GB) with vol = 1/4096.  Samples with 11
    significant bits make every partial sum j/4096 * x exact in f32, so the bus must equal the
    stem bit for bit; a second pass over a ragged, unaligned range takes the general path."""
    from dsp_amd import device
    F, S = 2880000, 4096
    stem = torch.empty((2, F), dtype=torch.float32, device="cuda")
    device.synth_noise(stem[0], 21)
    device.synth_noise(stem[1], 22)
    stem = torch.round(stem * 1024) / 1024
    bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
    device.DeviceMix([{"ch": [stem[0], stem[1]], "vol": 1.0 / S} for _ in range(S)]).run([bus[0], bus[1]])
    assert torch.equal(bus, stem)
    # looping form: sl = 1000 frames of the stem repeated, offset 7: (x + 7) % 1000
    n = 100003
    bus = torch.zeros((2, n), dtype=torch.float32, device="cuda")
    device.DeviceMix([{"ch": [stem[0], stem[1]], "rl": F, "sl": 1000, "o": 7, "vol": 1.0 / S} for _ in range(S)]).run([bus[0], bus[1]])
    idx = (torch.arange(n, device="cuda") + 7) % 1000
    assert torch.equal(bus, stem[:, idx])


def test_single_job_calls_alternate_channels_across_streams(torch, oracle):
    """wp_convolve_dev on one handle, alternating IR channels and alternating between two
    non-blocking streams with no host synchronisation in between: the channel travels as a
    launch argument and a call on a new stream waits for the handle's previous call, so every
    result equals the one computed alone."""
    from dsp_amd import capi
    lib = capi.lib()
    L, M = 50000, 30000
    f32p = C.POINTER(C.c_float)
    irs = [oracle.synth_ir(M, 5), oracle.synth_ir(M, 6)]
    h = capi.check_handle(lib.wp_ir_create((f32p * 2)(*[a.ctypes.data_as(f32p) for a in irs]), 2, M, L), "wp_ir_create")
    xs = [torch.from_numpy(oracle.synth_noise(L, 30 + i)).cuda() for i in range(2)]
    ref = []
    for ch in range(2):
        o = torch.empty(L + M, dtype=torch.float32, device="cuda")
        capi.check(lib.wp_convolve_dev(h, ch, C.c_void_p(xs[ch].data_ptr()), L, C.c_void_p(o.data_ptr()), None))
        torch.cuda.synchronize()
        ref.append(o)
        want = oracle.fft_convolution(L, oracle.fft_process_kernel(L, irs[ch]), M)(xs[ch].cpu().numpy())
        assert float(np.abs(o.cpu().numpy().astype(np.float64) - want).max()) <= TOL
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty(L + M, dtype=torch.float32, device="cuda") for _ in range(12)]
    torch.cuda.synchronize()
    for i in range(12):
        ch = i % 2
        st = streams[(i // 3) % 2]
        capi.check(lib.wp_convolve_dev(h, ch, C.c_void_p(xs[ch].data_ptr()), L, C.c_void_p(outs[i].data_ptr()),
                                       C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize()
    for i in range(12):
        assert torch.equal(outs[i].view(torch.int32), ref[i % 2].view(torch.int32)), i
    lib.wp_ir_destroy(h)


def test_bus_reduce_one_rank_through_c_abi(torch):
    """wp_comm_* / wp_bus_reduce with a single rank: RCCL is opened, a communicator of one rank
    is created from a unique id, and the reduce returns the bus unchanged, bit for bit, in place
    and into another buffer; the all-reduce likewise.  (More ranks need more GPUs than this box
    has: RCCL refuses two ranks on one device.)"""
    from dsp_amd import capi, device
    comm = device.Comm(0, 1)
    try:
        assert "rccl" in comm.library
        assert capi.lib().wp_comm_info(comm.handle, 0) == 1 and capi.lib().wp_comm_info(comm.handle, 1) == 0
        F = 2880000
        bus = torch.empty((2, F), dtype=torch.float32, device="cuda")
        device.synth_noise(bus[0], 3)
        device.synth_noise(bus[1], 4)
        keep = bus.clone()
        device.reduce_master_bus(bus, None, dst=0, comm=comm)
        torch.cuda.synchronize()
        assert torch.equal(bus.view(torch.int32), keep.view(torch.int32))
        out = torch.zeros_like(bus)
        comm.reduce(bus, out=out)
        comm.allreduce(bus)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        out2 = torch.zeros_like(bus)
        comm.allreduce(bus, out=out2, stream=side)
        torch.cuda.synchronize()
        for t in (out, out2, bus):
            assert torch.equal(t.view(torch.int32), keep.view(torch.int32))
        assert capi.lib().wp_bus_reduce(comm.handle, C.c_void_p(bus.data_ptr()), None, 8, 0, None) == capi.WP_ERR_INVALID
        assert capi.lib().wp_bus_reduce(comm.handle, C.c_void_p(bus.data_ptr()), C.c_void_p(bus.data_ptr()), 8, 1, None) == capi.WP_ERR_INVALID
    finally:
        comm.close()


def test_reverb_tail_roundtrip_and_error_paths(torch, oracle):
    from dsp_amd import capi, convolve, mix
    lib = capi.lib()
    L, M = 500, 300
    ir = convolve.IR([oracle.synth_ir(M, 5), oracle.synth_ir(M, 6)], L)
    job_ch = (C.c_int32 * 2)(0, 1)
    rv = capi.check_handle(lib.wp_reverb_create(ir.handle, 2, job_ch, 10), "wp_reverb_create")
    try:
        xs = [oracle.synth_noise(L, 1), oracle.synth_noise(L, 2)]
        outs = [np.empty(L + M, np.float32), np.empty(L + M, np.float32)]
        ip = (capi.f32p * 2)(*[a.ctypes.data_as(capi.f32p) for a in xs])
        op = (capi.f32p * 2)(*[a.ctypes.data_as(capi.f32p) for a in outs])
        v = capi.WpReverbView()
        capi.check(lib.wp_reverb_block(rv, ip, L, op, C.byref(v)))
        # the tail the library carries is the view the reference would store (impulse-reverb-stereo.js:23-25)
        assert lib.wp_reverb_tail_len(rv) == v.prev_len == M - 10
        tails = [np.empty(v.prev_len, np.float32), np.empty(v.prev_len, np.float32)]
        tp = (capi.f32p * 2)(*[a.ctypes.data_as(capi.f32p) for a in tails])
        capi.check(lib.wp_reverb_get_tail(rv, tp))
        for ch in range(2):
            assert np.array_equal(tails[ch], outs[ch][v.prev_off:v.prev_off + v.prev_len])
        # set_tail then one block == adding that tail to a plain convolution
        capi.check(lib.wp_reverb_set_tail(rv, tp, v.prev_len))
        capi.check(lib.wp_reverb_block(rv, ip, L, op, C.byref(v)))
        for ch in range(2):
            plain = np.empty(L + M, np.float32)
            capi.check(lib.wp_convolve(ir.handle, ch, xs[ch].ctypes.data_as(capi.f32p), L, plain.ctypes.data_as(capi.f32p)))
            want = plain.copy()
            want[:len(tails[ch])] += tails[ch]
            assert np.array_equal(outs[ch], want)
    finally:
        lib.wp_reverb_destroy(rv)
    # an input longer than L is fine while the reference's FFT would not wrap ...
    k = convolve.fft_process_kernel(600, oracle.synth_ir(200, 7))     # N = 1024
    go = convolve.fft_convolution(600, k, 200)
    x = oracle.synth_noise(800, 3)
    want = oracle.fft_convolution(600, oracle.fft_process_kernel(600, oracle.synth_ir(200, 7)), 200)(x)
    assert float(np.abs(go(x).astype(np.float64) - want).max()) <= TOL
    ogo = oracle.fft_convolution(600, oracle.fft_process_kernel(600, oracle.synth_ir(200, 7)), 200)
    for n in (1000, 826, 1024, 999):                                  # ... n + 200 - 1 > 1024: the head carries the wrapped tail
        xw = oracle.synth_noise(n, 3)
        assert float(np.abs(go(xw).astype(np.float64) - ogo(xw)).max()) <= TOL, n
    assert float(np.abs(go(x).astype(np.float64) - want).max()) <= TOL  # and the ordinary plan still answers afterwards
    # the batched entry point still refuses such an input instead of mis-computing it
    plan = capi.check_handle(lib.wp_conv_create(go.ir.handle, 1, (C.c_int32 * 1)(0)), "wp_conv_create")
    xin = torch.zeros(1000, dtype=torch.float32, device="cuda")
    yout = torch.zeros(800, dtype=torch.float32, device="cuda")
    assert lib.wp_conv_run_dev(plan, C.c_void_p(xin.data_ptr()), 0, 1000, C.c_void_p(yout.data_ptr()), 0, None, 0, 0, None) == -6
    lib.wp_conv_destroy(plan)
    with pytest.raises(IndexError):
        go(oracle.synth_noise(1025, 3))
    bad = (C.c_int32 * 1)(2)
    assert not lib.wp_conv_create(ir.handle, 1, bad) and "IR channel" in capi.last_error()
    # layouts the reference ignores stay untouched (mix-buffers.js:18-44, 47-69)
    t3 = [np.ones(8, np.float32) for _ in range(3)]
    mix.mix_buffers(t3, [np.ones(8, np.float32)])
    assert all((t == 1).all() for t in t3)
    t2 = [np.ones(8, np.float32), np.ones(8, np.float32)]
    mix.mix_buffers(t2, [np.ones(8, np.float32)] * 3)
    assert all((t == 1).all() for t in t2)


def test_bench_path_full_length_against_reference_golden(torch):
    """The benchmark's own call sequence (device-resident noise, batched jobs, one-pass
    results + bus) at the full 60 s x 2 s-IR job size: job 0 is the (seed 1000, IR seed
    500000) channel-job whose output the reference JavaScript produced for the fixtures."""
    from conftest import from_bits, load_golden
    from dsp_amd import device
    g = load_golden("g5_config_sized.json")["c3_job0"]
    F, M, tracks = g["L"], g["M"], 4
    njobs = 2 * tracks
    irs = torch.from_numpy(np.stack([device.synth_ir(M, 500000), device.synth_ir(M, 500001)])).cuda()
    ir = device.DeviceIR(irs, F)
    conv = device.BatchedConvolver(ir, [j % 2 for j in range(njobs)])
    x = torch.empty((njobs, F), dtype=torch.float32, device="cuda")
    for j in range(njobs):
        device.synth_noise(x[j], 1000 + j)
    out = torch.empty((njobs, F + M), dtype=torch.float32, device="cuda")
    bus = torch.zeros((2, F), dtype=torch.float32, device="cuda")
    vol = [1.0 / 256] * tracks
    conv.run_and_mix(x, out, [bus[0], bus[1]], vol)
    y0 = out[0].cpu().numpy()
    want = from_bits(g["samples"])
    err = float(np.abs(y0[::g["stride"]].astype(np.float64) - want).max())
    assert err <= TOL, err
    assert float(np.abs(y0[:64].astype(np.float64) - from_bits(g["head"])).max()) <= TOL
    assert float(np.abs(y0[-64:].astype(np.float64) - from_bits(g["tail"])).max()) <= TOL
    # the bus is the in-order f64 fold of the written results (checked on the host for a slice)
    o = out[:, :F].cpu().numpy()
    sl = slice(1234567, 1234567 + 4096)
    for ch in range(2):
        acc = np.zeros(4096, np.float32)
        for t in range(tracks):
            acc = (acc.astype(np.float64) + o[2 * t + ch, sl].astype(np.float64) * vol[t]).astype(np.float32)
        assert np.array_equal(acc, bus[ch, sl].cpu().numpy())


def test_two_host_threads_drive_two_plans_of_one_ir(torch, oracle):
    """What two Node workers do after receiving the same kernel object: two host threads, each with a plan
    (wp_conv) of its own made from ONE wp_ir, each on a stream of its own, run at the same time (ctypes
    releases the interpreter lock during the calls).  Every result equals the one computed alone."""
    import threading
    from dsp_amd import device
    L, M = 40000, 9000
    irs = np.stack([oracle.synth_ir(M, 31), oracle.synth_ir(M, 32)])
    ir = device.DeviceIR(torch.from_numpy(irs).cuda(), L)
    x = torch.from_numpy(np.stack([oracle.synth_noise(L, 40 + j) for j in range(4)])).cuda()
    alone = torch.empty((4, L + M), dtype=torch.float32, device="cuda")
    device.BatchedConvolver(ir, [0, 1, 0, 1]).run(x, alone)
    torch.cuda.synchronize()
    want = oracle.fft_convolution(L, oracle.fft_process_kernel(L, irs[1]), M)(x[3].cpu().numpy())
    assert float(np.abs(alone[3].cpu().numpy().astype(np.float64) - want).max()) <= TOL
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            conv = device.BatchedConvolver(ir, [0, 1])
            out = torch.empty((2, L + M), dtype=torch.float32, device="cuda")
            with torch.cuda.stream(stream):
                for _ in range(20):
                    out.zero_()
                    conv.run(x[2 * k:2 * k + 2], out)
                stream.synchronize()
            if not torch.equal(out.view(torch.int32), alone[2 * k:2 * k + 2].view(torch.int32)):
                errors.append("thread %d: results differ from the single-threaded run" % k)
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors


def test_host_pointer_calls_with_page_locked_arrays(torch, oracle):
    """wp_reverb_block and wp_mix with every host array from wp_host_alloc (transfers queued on the calling
    thread's stream, one wait) give the bits the same calls give with pageable arrays (blocking copies)."""
    from dsp_amd import capi
    lib = capi.lib()
    L, M = 12000, 5000
    f32p = C.POINTER(C.c_float)
    irs = [oracle.synth_ir(M, 91), oracle.synth_ir(M, 92)]
    h = capi.check_handle(lib.wp_ir_create((f32p * 2)(*[a.ctypes.data_as(f32p) for a in irs]), 2, M, L), "wp_ir_create")
    job_ch = (C.c_int32 * 2)(0, 1)

    def locked(n):
        p = lib.wp_host_alloc(4 * n)
        assert p, lib.wp_last_error()
        return p, np.ctypeslib.as_array(C.cast(C.c_void_p(p), f32p), shape=(n,))

    blocks = []
    try:
        results = {}
        for kind in ("pageable", "locked"):
            rv = capi.check_handle(lib.wp_reverb_create(h, 2, job_ch, 3), "wp_reverb_create")
            if kind == "locked":
                mem = [locked(L) for _ in range(2)] + [locked(L + M) for _ in range(2)] + [locked(L) for _ in range(2)]
                blocks += [p for p, _ in mem]
                xin, out, bus = [a for _, a in mem[:2]], [a for _, a in mem[2:4]], [a for _, a in mem[4:]]
            else:
                xin, out, bus = [np.empty(L, np.float32) for _ in range(2)], [np.empty(L + M, np.float32) for _ in range(2)], \
                    [np.empty(L, np.float32) for _ in range(2)]
            got = []
            v = capi.WpReverbView()
            for b in range(3):
                for ch in range(2):
                    xin[ch][:] = oracle.synth_noise(L, 95 + ch, b * L)
                    bus[ch][:] = 0.125
                capi.check(lib.wp_reverb_block(rv, (f32p * 2)(*[a.ctypes.data_as(f32p) for a in xin]), L,
                                               (f32p * 2)(*[a.ctypes.data_as(f32p) for a in out]), C.byref(v)), "wp_reverb_block")
                srcs = (capi.WpSrc * 2)()
                for i, chans in enumerate((xin, [o[v.ret_off:v.ret_off + v.ret_len] for o in out])):
                    srcs[i].nch = 2
                    srcs[i].ch[0], srcs[i].ch[1] = chans[0].ctypes.data, chans[1].ctypes.data
                    srcs[i].rl = srcs[i].sl = len(chans[0])
                    srcs[i].vol, srcs[i].o = 0.5, 0
                capi.check(lib.wp_mix((f32p * 2)(*[a.ctypes.data_as(f32p) for a in bus]), 2, L, srcs, 2), "wp_mix")
                got.append([o.copy() for o in out] + [t.copy() for t in bus])
            lib.wp_reverb_destroy(rv)
            results[kind] = got
        for a, b in zip(results["pageable"], results["locked"]):
            for x, y in zip(a, b):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    finally:
        for p in blocks:
            lib.wp_host_free(C.c_void_p(p))
        lib.wp_ir_destroy(h)
