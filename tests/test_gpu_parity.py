"""GPU parity: the HIP path, called through the C ABI (dsp_amd.* are ctypes
callers of libwpdsp.so), against (a) the golden vectors recorded from the
reference JavaScript and (b) the CPU oracle on the same seeded inputs.

Bar: bit-exact for integer/index work and for the mix (f64 multiply/add, f32
store per source, source order); max abs error <= 1e-5 of full scale for
convolution samples (full scale = 1.0; the tiny vectors whose values exceed 1
are scaled by their peak).
"""
import numpy as np
import pytest

from conftest import assert_bits_equal, from_bits, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _close(actual, expected, what, tol=TOL):
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, "%s: length %s != %s" % (what, a.shape, e.shape)
    fs = max(1.0, float(np.abs(e).max()) if e.size else 1.0)
    err = float(np.abs(a - e).max()) if e.size else 0.0
    assert err <= tol * fs, "%s: max abs err %.3e > %.1e of full scale %.3g" % (what, err, tol, fs)
    return err


@pytest.fixture(scope="module")
def wp():
    from dsp_amd import capi, convolve, impulse, mix
    assert capi.init() >= 1

    class NS:
        pass
    ns = NS()
    ns.capi, ns.convolve, ns.impulse, ns.mix = capi, convolve, impulse, mix
    return ns


def test_g1_tiny_go_vectors(wp):
    g = load_golden("g1_convolve_tiny.json")
    for c in g["cases"]:
        x, h = from_bits(c["x"]), from_bits(c["h"])
        k = wp.convolve.fft_process_kernel(c["L"], h)
        assert len(k) == c["kernel_len"]
        y = wp.convolve.fft_convolution(c["L"], k, len(h))(x)
        assert len(y) == c["out_len"] and y.dtype == np.float32
        _close(y, from_bits(c["out"]), "go L=%d M=%d" % (c["L"], c["M"]))
        d = wp.convolve.direct_convolution(x, h)
        assert_bits_equal(d, c["direct"], "direct L=%d" % c["L"])  # the reference's own accumulation chain per sample
        # the kernel object the reference itself produced (its interleaved spectrum) is accepted as it is
        ref_kernel = from_bits(c["kernel"])
        y2 = wp.convolve.fft_convolution(c["L"], ref_kernel, len(h))(x)
        _close(y2, from_bits(c["out"]), "go with the reference's own kernel, L=%d" % c["L"])
        # and compat=True returns that layout, equal to the reference's array to its rounding noise
        ck = wp.convolve.fft_process_kernel(c["L"], h, compat=True)
        assert ck.dtype == np.float32 and len(ck) == c["kernel_len"]
        assert float(np.abs(ck.astype(np.float64) - ref_kernel).max()) <= 2e-6 * max(1.0, float(np.abs(ref_kernel).max()))


def test_reference_layout_kernels_at_config_size(wp, oracle):
    """The reference-layout kernel at the C1 size (N = 131 072): the spectrum the oracle computes the
    reference's way is accepted by fft_convolution, compat=True reproduces it to rounding noise, and
    direct_convolution matches the oracle's bit for bit on a case with thousands of terms per sample."""
    L, M = 44100, 22050
    h, x = oracle.synth_ir(M, 500000), oracle.synth_noise(L, 1000)
    ref_kernel = oracle.fft_process_kernel(L, h)
    want = oracle.fft_convolution(L, ref_kernel, M)(x)
    _close(wp.convolve.fft_convolution(L, ref_kernel, M)(x), want, "go with an oracle-made reference-layout kernel")
    ck = wp.convolve.fft_process_kernel(L, h, compat=True)
    assert len(ck) == len(ref_kernel) == 2 * 131072
    assert float(np.abs(ck.astype(np.float64) - ref_kernel).max()) <= 2e-6 * float(np.abs(ref_kernel).max())
    xs, hs = oracle.synth_noise(5000, 3), oracle.synth_ir(3000, 4)
    assert_bits_equal(wp.convolve.direct_convolution(xs, hs), oracle.direct_convolution(xs, hs).view(np.uint32), "direct 5000 x 3000")


def test_integer_helpers_on_the_gpu_box(wp):
    """The bit-exact integer contract (nextPowerOfTwo, beat rate and bpm getter, ring index, (rl*length)|0)
    against the reference's vectors, run on the GPU box as well as in the CPU suite."""
    from test_boundary_cpu import test_integer_helpers_match_reference_vectors, test_reverb_view_arithmetic_matches_reference_blocks
    test_integer_helpers_match_reference_vectors()
    test_reverb_view_arithmetic_matches_reference_blocks()


def test_error_behaviour(wp):
    k = wp.convolve.fft_process_kernel(4, np.ones(4, np.float32))
    go = wp.convolve.fft_convolution(4, k, 4)
    with pytest.raises(IndexError):        # RangeError in the reference (convolve.js:590)
        go(np.zeros(9, np.float32))
    with pytest.raises(ValueError):        # 'FFT size must be a power of two and bigger than 1'
        wp.convolve.fft_process_kernel(0, np.zeros(1, np.float32))
    assert len(go(np.zeros(0, np.float32))) == 8   # empty input: L+M zeros
    assert not go(np.zeros(0, np.float32)).any()
    y = go(np.array([1, 0], np.float32))           # ragged: shorter than L
    _close(y, [1, 1, 1, 1, 0, 0, 0, 0], "short input")


def test_g3_reverb_block_sequences(wp, oracle):
    g = load_golden("g3_reverb_blocks.json")
    for run in g["runs"]:
        p = dict(run["params"])
        url = p.pop("url")
        store, log = {}, []
        make = wp.impulse.impulse_reverb_stereo if run["kind"] == "stereo" else wp.impulse.impulse_reverb
        irs = {url: [from_bits(b).copy() for b in run["ir"]]}
        ctx = lambda blk: oracle.StubContext([from_bits(b).copy() for b in blk["input"]], blk["n"], irs, store, log)
        c0 = ctx(run["blocks"][0])
        fn = make(c0, url, **p)
        for i, blk in enumerate(run["blocks"]):
            c = c0 if i == 0 else ctx(blk)
            ret = fn(c)
            ret = [ret] if run["kind"] == "mono" else ret
            what = "%s %s block %d" % (run["kind"], run["params"], i)
            for ch, exp in enumerate(blk["ret"]):
                # index work is bit-exact: view length, byte offset, backing buffer size
                assert len(ret[ch]) == exp["length"], what
                base = ret[ch].base if ret[ch].base is not None else ret[ch]
                off = ret[ch].__array_interface__["data"][0] - base.__array_interface__["data"][0]
                assert off == exp["byte_offset"], what
                assert base.nbytes == exp["buffer_bytes"], what
                _close(ret[ch], from_bits(exp["data"]), what)
                _close(base, from_bits(exp["full"]), what + " full buffer")
            prev = store[log[-1][1]]
            prev = [prev] if run["kind"] == "mono" else prev
            for ch, exp in enumerate(blk["prev"]):
                assert len(prev[ch]) == exp["length"], what + " prev length"
                _close(prev[ch], from_bits(exp["data"]), what + " prev")
        r = run["resumed"]
        cR = oracle.StubContext([from_bits(b).copy() for b in r["input"]], r["n"], irs, store, log)
        ret = make(cR, url, **p)(cR)
        ret = [ret] if run["kind"] == "mono" else ret
        for ch, exp in enumerate(r["ret"]):
            _close(ret[ch], from_bits(exp["data"]), "resumed from cached tail")
        assert [[e[0], e[1], e[2]] for e in log] == [[e[0], e[1], e[2]] for e in run["log"]]


def _sources(case):
    out = []
    for s in case["sources"]:
        buf = [from_bits(b).copy() for b in s["buffer"]]
        if not s["tuple"]:
            out.append(buf)
            continue
        vol = s.get("volume")
        if isinstance(vol, dict):
            vol = {"ramp": lambda x: 0.25 + x / 8, "alt": lambda x: -1.3 if (x & 1) else 0.7}[vol["fn"]]
        out.append((buf, s.get("length"), vol, s.get("offset")))
    return out


def test_g4_mix_buffers_bit_exact(wp):
    g = load_golden("g4_mix_buffers.json")
    for case in g["cases"]:
        target = [from_bits(b).copy() for b in case["target_in"]]
        ret = wp.mix.mix_buffers(target, *_sources(case))
        assert ret is target
        for ch, exp in enumerate(case["target_out"]):
            assert_bits_equal(target[ch], exp, case["name"] + " ch%d" % ch)


def test_mix_source_is_the_target_itself(wp, oracle):
    """mixBuffers(t, t): every index is read before it is written in the reference (same length, no
    offset), so the target doubles; the host-array entry point stages the target apart from its sources."""
    t = [oracle.synth_noise(1000, 77), oracle.synth_noise(1000, 78)]
    want = [(a.astype(np.float64) + a.astype(np.float64) * 1.0).astype(np.float32) for a in t]
    wp.mix.mix_buffers(t, t)
    for ch in range(2):
        assert np.array_equal(t[ch], want[ch])
    m = [oracle.synth_noise(64, 79)]
    want = [(m[0].astype(np.float64) + m[0].astype(np.float64) * 0.5).astype(np.float32)]
    wp.mix.mix_buffers(m, (m, 1, 0.5, 0))
    assert np.array_equal(m[0], want[0])


def test_mix_vs_oracle_random_shapes(wp, oracle):
    rng = np.random.default_rng(5)
    for trial in range(12):
        tch, tl = int(rng.integers(1, 3)), int(rng.integers(1, 5000))
        target = [oracle.synth_noise(tl, 900 + trial * 2 + c) for c in range(tch)]
        sources = []
        for s in range(int(rng.integers(1, 9))):
            sch, rl = int(rng.integers(1, 3)), int(rng.integers(1, 7000))
            buf = [oracle.synth_noise(rl, 2000 + trial * 100 + s * 2 + c) for c in range(sch)]
            kind = int(rng.integers(0, 3))
            if kind == 0:
                sources.append(buf)
            else:
                sources.append((buf, float(rng.choice([1, .5, .75, 2.25, 1 / 3])), float(rng.choice([1, 1.3, 1 / 64, -0.2])),
                                int(rng.choice([0, 3, 96000, 2 ** 33 + 11, -2]))))
        a = [t.copy() for t in target]
        b = [t.copy() for t in target]
        wp.mix.mix_buffers(a, *sources)
        oracle.mix_buffers(b, *sources)
        for ch in range(tch):
            assert_bits_equal(a[ch], b[ch].view(np.uint32), "trial %d ch %d" % (trial, ch))


def test_g5_c1_mono_impulse_convolve(wp, oracle):
    g = load_golden("g5_config_sized.json")["c1"]
    ir = oracle.synth_ir(g["M"], g["seed_ir"])
    x = oracle.synth_noise(g["L"], g["seed_x"])
    store, log = {}, []
    c = oracle.StubContext([x], 0, {"ir:c1": [ir]}, store, log)
    go = wp.impulse.impulse_convolve(c, "ir:c1", -1)
    y = go(x)
    assert len(y) == g["out_len"]
    assert len(store["kernel:ir:c1:44100:-1"]) == g["kernel_len"]
    err = _close(y[::g["stride"]], from_bits(g["samples"]), "c1 strided samples vs reference JS")
    st = g["stats"]
    assert abs(float(y.astype(np.float64).sum()) - st["sum"]) < 1e-3
    assert abs(float(np.abs(y).max()) - st["max_abs"]) < TOL
    full = oracle.fft_convolution(g["L"], oracle.fft_process_kernel(g["L"], ir), g["M"])(x)
    _close(y, full, "c1 every sample vs oracle")
    assert [[e[0], e[1]] for e in log] == [[e[0], e[1]] for e in g["cache_log"]]
    print("c1 max abs err vs reference: %.3e" % err)


@pytest.mark.parametrize("offset", [0, 1000])
def test_g5_c2_stereo_reverb_four_bars(wp, oracle, offset):
    g = load_golden("g5_config_sized.json")["c2_offset%d" % offset]
    L, M = g["L"], g["M"]
    irs = {"ir:c2": [oracle.synth_ir(M, s) for s in g["seeds_ir"]]}
    store, log = {}, []
    mk = lambda n: oracle.StubContext([oracle.synth_noise(L, s, n) for s in g["seeds_x"]], n, irs, store, log)
    c0 = mk(0)
    fn = wp.impulse.impulse_reverb_stereo(c0, "ir:c2", offset=offset, id="c2")
    for b, bar in enumerate(g["bars"]):
        c = c0 if b == 0 else mk(b * L)
        r = fn(c)
        assert [len(r[0]), len(r[1])] == bar["ret_len"]
        for ch in range(2):
            base = r[ch].base
            assert r[ch].__array_interface__["data"][0] - base.__array_interface__["data"][0] == bar["byte_offset"][ch]
            assert base.nbytes == bar["buffer_bytes"]
            _close(r[ch][::bar["stride"]], from_bits(bar["samples"][ch]), "c2 bar %d ch %d" % (b, ch))
            assert abs(float(np.abs(r[ch]).max()) - bar["stats"][ch]["max_abs"]) < TOL
    assert [e[1] for e in log if e[0] == "set"] == g["keys"]


def test_g5_c3_single_job_full_length(wp, oracle):
    g = load_golden("g5_config_sized.json")["c3_job0"]
    ir = oracle.synth_ir(g["M"], g["seed_ir"])
    x = oracle.synth_noise(g["L"], g["seed_x"])
    k = wp.convolve.fft_process_kernel(g["L"], ir)
    y = wp.convolve.fft_convolution(g["L"], k, g["M"])(x)
    assert len(y) == g["out_len"]
    _close(y[:64], from_bits(g["head"]), "head")
    _close(y[-64:], from_bits(g["tail"]), "tail")
    err = _close(y[::g["stride"]], from_bits(g["samples"]), "strided")
    print("c3 job max abs err vs reference: %.3e" % err)


def test_convolution_properties_full_size(wp, oracle):
    """Size-independent checks at the 60 s x 2 s-IR job size: linearity, shift,
    and an impulse input reproducing the IR."""
    L, M = 2880000, 96000
    ir = oracle.synth_ir(M, 500001)
    k = wp.convolve.fft_process_kernel(L, ir)
    go = wp.convolve.fft_convolution(L, k, M)
    d = np.zeros(L, np.float32)
    d[0] = 1.0
    y = go(d)
    _close(y[:M], ir, "delta -> IR", tol=1e-6)
    assert float(np.abs(y[M:]).max()) <= 1e-6
    d[:] = 0
    d[L - 1] = 0.5
    y = go(d)
    _close(y[L - 1:L - 1 + M], 0.5 * ir, "delayed delta", tol=1e-6)
    assert float(np.abs(y[:L - 1]).max()) <= 1e-6
    a, b = oracle.synth_noise(L, 31), oracle.synth_noise(L, 32)
    ya, yb, yab = go(a), go(b), go((a * 0.5 + b * 0.25).astype(np.float32))
    _close(yab, 0.5 * ya.astype(np.float64) + 0.25 * yb, "linearity", tol=2e-6)


@pytest.mark.parametrize("form", ["long-frame", "partitioned"])
def test_g7_render_jobs_with_4s_impulse_responses(wp, oracle, form):
    """Two channel-jobs of the 2048-track render (L = 2 880 000, M = 192 000, one impulse response per
    job) through the batched one-pass entry point, against what the reference JavaScript produced for
    them.  Once in the form the library picks for these sizes (long-frame: frames that hold the whole
    4 s response) and once forced to the partitioned form (the 8192-point plan, 24 partitions)."""
    import torch
    from dsp_amd import capi, device
    lib = capi.lib()
    g = load_golden("g7_c5_jobs.json")
    L, M = g["L"], g["M"]
    jobs = g["jobs"]
    irs = torch.from_numpy(np.stack([device.synth_ir(M, j["seed_ir"]) for j in jobs])).cuda()
    capi.check(lib.wp_tune(5, 0 if form == "long-frame" else 1), "wp_tune")  # WP_TUNE_CONV_FORM
    try:
        ir = device.DeviceIR(irs, L)
    finally:
        lib.wp_tune(5, 0)
    assert ir.info(3) == 8192 and ir.info(4) == 24  # the partition plan every handle carries
    conv = device.BatchedConvolver(ir, [0, 1])
    if form == "long-frame":
        assert lib.wp_conv_info(conv.handle, 6) == 1
        frames = [(int(lib.wp_conv_info(conv.handle, 16 + 4 * i)), int(lib.wp_conv_info(conv.handle, 17 + 4 * i)))
                  for i in range(int(lib.wp_conv_info(conv.handle, 10)))]
        assert frames[0][0] >= 1048576 and all(f >= 2 * M for f, _ in frames), frames  # every frame holds the whole response
    else:
        assert lib.wp_conv_info(conv.handle, 6) == 0 and lib.wp_conv_info(conv.handle, 2) == 8192
    x = torch.empty((2, L), dtype=torch.float32, device="cuda")
    for i, j in enumerate(jobs):
        device.synth_noise(x[i], j["seed_x"])
    out = torch.empty((2, L + M), dtype=torch.float32, device="cuda")
    bus = torch.zeros((2, L), dtype=torch.float32, device="cuda")
    conv.run_and_mix(x, out, [bus[0], bus[1]], [0.5])
    got = out.cpu().numpy()
    for i, j in enumerate(jobs):
        y = got[i]
        _close(y[:64], from_bits(j["head"]), "head")
        _close(y[-64:], from_bits(j["tail"]), "tail")
        _close(y[j["mid"]["at"]:j["mid"]["at"] + 256], from_bits(j["mid"]["data"]), "mid")
        err = _close(y[::j["stride"]], from_bits(j["samples"]), "strided")
        print("4 s IR job %d max abs err vs reference: %.3e" % (i, err))
    # the bus is the f64 fold of the two written results with vol = 0.5 (mix-buffers.js:49-52)
    hb = bus.cpu().numpy()
    for ch in range(2):
        want = (np.float64(0) + got[ch, :L].astype(np.float64) * 0.5).astype(np.float32)
        assert np.array_equal(hb[ch], want)


def test_g8_mix_negative_offsets_bit_exact(wp):
    g = load_golden("g8_mix_negative_offsets.json")
    for case in g["cases"]:
        target = [from_bits(b).copy() for b in case["target_in"]]
        wp.mix.mix_buffers(target, *_sources(case))
        for ch, exp in enumerate(case["target_out"]):
            assert_bits_equal(target[ch], exp, case["name"] + " ch%d" % ch)


def test_g11_mix_sources_that_share_memory_with_the_target_bit_exact(wp):
    """Sources that are the target itself, views into it, or views that reach beyond it: the host-array entry
    point keeps the sharing on the device (one staged piece per group of overlapping arrays) and such sources
    are folded frame by frame in the reference's order; everything the call may touch is compared."""
    from conftest import shared_memory_mix_case
    g = load_golden("g11_mix_shared_memory.json")
    for case in g["cases"]:
        bases, target, sources = shared_memory_mix_case(case)
        ret = wp.mix.mix_buffers(target, *sources)
        assert ret is target
        for i, exp in enumerate(case["bases_out"]):
            assert_bits_equal(bases[i], exp, case["name"] + " base %d" % i)


def test_g9_render_buffer_conventions_host_and_device(wp):
    """The buffer conventions of src/render.js:27-48 reproduced by the mix kernel, with host
    arrays (wp_mix) and with device-resident buffers (wp_buf + wp_mix_dev, the route a bar
    takes when it stays on the GPU)."""
    import ctypes as C
    from test_oracle_golden import render_conventions_via_mix
    g = load_golden("g9_render_conventions.json")
    for case, target in render_conventions_via_mix(wp.mix.mix_buffers, g):
        for ch, exp in enumerate(case["out"]):
            assert_bits_equal(target[ch], exp, "host: %s fn into %d channel(s), ch%d" % (case["fn"], case["channels"], ch))
    lib, capi = wp.capi.lib(), wp.capi
    n = g["frames"]
    f32p = C.POINTER(C.c_float)
    chans = {"mono": [from_bits(g["mono"])], "stereo": [from_bits(g["left"]), from_bits(g["right"])]}
    for case in g["cases"]:
        src = chans[case["fn"]]
        sb = capi.check_handle(lib.wp_buf_create(len(src), n), "wp_buf_create")
        tb = capi.check_handle(lib.wp_buf_create(case["channels"], n), "wp_buf_create")
        try:
            for c, a in enumerate(src):
                capi.check(lib.wp_buf_write(sb, c, 0, a.ctypes.data_as(f32p), n), "wp_buf_write")
            d = capi.WpSrc()
            for c in range(len(src)):
                d.ch[c] = lib.wp_buf_channel(sb, c)
            d.nch, d.rl, d.sl, d.vol, d.o = len(src), n, n, 1.0, 0
            tgt = (C.c_void_p * case["channels"])(*[lib.wp_buf_channel(tb, c) for c in range(case["channels"])])
            capi.check(lib.wp_mix_dev(tgt, case["channels"], n, C.byref(d), 1, 0, None), "wp_mix_dev")
            for ch, exp in enumerate(case["out"]):
                got = np.empty(n, np.float32)
                capi.check(lib.wp_buf_read(tb, ch, 0, got.ctypes.data_as(f32p), n), "wp_buf_read")
                assert_bits_equal(got, exp, "device: %s fn into %d channel(s), ch%d" % (case["fn"], case["channels"], ch))
        finally:
            lib.wp_buf_release(sb)
            lib.wp_buf_release(tb)
