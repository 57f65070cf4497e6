"""Pins the CPU oracle (oracle/wp_oracle.c + oracle/oracle.py) bit-for-bit to
vectors recorded from the reference's own JavaScript (oracle/gen_golden.mjs)."""
import numpy as np
import pytest

from conftest import assert_bits_equal, from_bits, load_golden


def test_g1_tiny_convolutions(oracle):
    g = load_golden("g1_convolve_tiny.json")
    for c in g["cases"]:
        x, h = from_bits(c["x"]), from_bits(c["h"])
        k = oracle.fft_process_kernel(c["L"], h)
        assert len(k) == c["kernel_len"]
        assert_bits_equal(k, c["kernel"], "kernel L=%d" % c["L"])
        y = oracle.fft_convolution(c["L"], k, len(h))(x)
        assert len(y) == c["out_len"] == c["L"] + c["M"]
        assert c["out_byte_offset"] == 0 and c["out_buffer_bytes"] == 4 * (c["L"] + c["M"])
        assert_bits_equal(y, c["out"], "go L=%d M=%d" % (c["L"], c["M"]))
        assert_bits_equal(oracle.direct_convolution(x, h), c["direct"], "direct")
    assert g["errors"]["input_too_long"] == "RangeError"
    with pytest.raises(IndexError):
        oracle.fft_convolution(4, np.zeros(16, np.float32), 4)(np.zeros(9, np.float32))
    assert g["errors"]["size_one"][1] == "FFT size must be a power of two and bigger than 1"
    with pytest.raises(ValueError):
        oracle.fft_process_kernel(0, np.zeros(1, np.float32))


def test_g2_next_pot_and_twiddle_tables(oracle):
    g = load_golden("g2_tables.json")
    for n, N in g["next_pot_via_kernel_len"]:
        assert oracle.next_pot(n) == N
    assert oracle.next_pot(0) == 1
    for t in g["delta_spectra"]:
        N = t["N"]
        spec = oracle.fft_process_kernel(N - 2, np.array([0, 1], np.float32))
        assert len(spec) == 2 * N
        assert_bits_equal(spec[:len(t["head"])], t["head"], "N=%d head" % N)
        assert oracle.checksum(spec) == t["checksum"], "twiddle-column checksum N=%d" % N


def _ctx(oracle, run, blk, store, log):
    bufs = [from_bits(b).copy() for b in blk["input"]]
    irs = {run["params"]["url"]: [from_bits(b).copy() for b in run["ir"]]}
    return oracle.StubContext(bufs, blk["n"], irs, store, log)


def test_g3_reverb_block_sequences(oracle):
    g = load_golden("g3_reverb_blocks.json")
    for run in g["runs"]:
        p = dict(run["params"])
        url = p.pop("url")
        store, log = {}, []
        make = oracle.impulse_reverb_stereo if run["kind"] == "stereo" else oracle.impulse_reverb
        # IR channels are shared between sample() calls like the reference's (truncation mutates
        # only the per-call array)
        irs = {url: [from_bits(b).copy() for b in run["ir"]]}
        c0 = oracle.StubContext([from_bits(b).copy() for b in run["blocks"][0]["input"]], 0, irs, store, log)
        fn = make(c0, url, **p)
        for i, blk in enumerate(run["blocks"]):
            c = c0 if i == 0 else oracle.StubContext([from_bits(b).copy() for b in blk["input"]], blk["n"], irs, store, log)
            ret = fn(c)
            if run["kind"] == "mono":
                ret = [ret]
            what = "%s %s block %d" % (run["kind"], run["params"], i)
            for ch, exp in enumerate(blk["ret"]):
                assert len(ret[ch]) == exp["length"], what
                base = ret[ch].base if ret[ch].base is not None else ret[ch]
                off = (ret[ch].__array_interface__["data"][0] - base.__array_interface__["data"][0])
                assert off == exp["byte_offset"], what
                assert base.nbytes == exp["buffer_bytes"], what
                assert_bits_equal(ret[ch], exp["data"], what)
                assert_bits_equal(base, exp["full"], what + " full buffer")
            key = log[-1][1]
            prev = store[key]
            prev = [prev] if run["kind"] == "mono" else prev
            for ch, exp in enumerate(blk["prev"]):
                assert len(prev[ch]) == exp["length"], what + " prev len"
                assert_bits_equal(prev[ch], exp["data"], what + " prev")
        # resume from the stored tail with a fresh closure
        r = run["resumed"]
        cR = oracle.StubContext([from_bits(b).copy() for b in r["input"]], r["n"], irs, store, log)
        fn2 = make(cR, url, **p)
        ret = fn2(cR)
        ret = [ret] if run["kind"] == "mono" else ret
        for ch, exp in enumerate(r["ret"]):
            assert_bits_equal(ret[ch], exp["data"], "resumed")
        # same cache protocol: same keys in the same order, same hit/miss and ttl
        assert [[e[0], e[1]] for e in log] == [[e[0], e[1]] for e in run["log"]]
        for mine, ref in zip(log, run["log"]):
            assert mine[2] == ref[2], (mine, ref)


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


def test_g4_mix_buffers_all_branches(oracle):
    g = load_golden("g4_mix_buffers.json")
    assert len(g["cases"]) > 60
    for case in g["cases"]:
        target = [from_bits(b).copy() for b in case["target_in"]]
        ret = oracle.mix_buffers(target, *_sources(case))
        assert ret is target and case["returns_target"]
        for ch, exp in enumerate(case["target_out"]):
            assert_bits_equal(target[ch], exp, case["name"] + " ch%d" % ch)


def test_g4_reference_test_expectation(oracle):
    # test/mix.js:264-286 shape: 2-sample child folds into a 4-sample parent by wrapping
    t = [np.array([1, 2, 3, 4], np.float32)]
    oracle.mix_buffers(t, [np.array([1, 2], np.float32)])
    assert t[0].tolist() == [2, 4, 4, 6]


def test_g5_c1_mono_impulse_convolve(oracle):
    g = load_golden("g5_config_sized.json")["c1"]
    ir = oracle.synth_ir(g["M"], g["seed_ir"])
    x = oracle.synth_noise(g["L"], g["seed_x"])
    assert oracle.checksum(ir) == g["ir_checksum"], "synthetic IR differs from the JS generator"
    assert oracle.checksum(x) == g["x_checksum"]
    store, log = {}, []
    c = oracle.StubContext([x], 0, {"ir:c1": [ir]}, store, log)
    go = oracle.impulse_convolve(c, "ir:c1", -1)
    y = go(x)
    assert len(y) == g["out_len"] == g["L"] + g["M"]
    k = store["kernel:ir:c1:44100:-1"]
    assert len(k) == g["kernel_len"] == 2 * 131072
    assert oracle.checksum(k) == g["kernel_checksum"]
    assert oracle.checksum(y) == g["out_checksum"]
    assert_bits_equal(y[::g["stride"]], g["samples"], "c1 strided samples")
    assert [[e[0], e[1]] for e in log] == [[e[0], e[1]] for e in g["cache_log"]]


@pytest.mark.parametrize("offset", [0, 1000])
def test_g5_c2_stereo_reverb_four_bars(oracle, offset):
    g = load_golden("g5_config_sized.json")["c2_offset%d" % offset]
    L, M = g["L"], g["M"]
    irs = {"ir:c2": [oracle.synth_ir(M, s) for s in g["seeds_ir"]]}
    store, log = {}, []
    mk = lambda n: oracle.StubContext([oracle.synth_noise(L, s, n) for s in g["seeds_x"]], n, irs, store, log)
    c0 = mk(0)
    fn = oracle.impulse_reverb_stereo(c0, "ir:c2", offset=offset, id="c2")
    for b, bar in enumerate(g["bars"]):
        c = c0 if b == 0 else mk(b * L)
        r = fn(c)
        assert [len(r[0]), len(r[1])] == bar["ret_len"]
        for ch in range(2):
            assert_bits_equal(r[ch][::bar["stride"]], bar["samples"][ch], "c2 bar %d ch %d" % (b, ch))
    assert [e[1] for e in log if e[0] == "set"] == g["keys"]


def test_g5_c3_single_job(oracle):
    g = load_golden("g5_config_sized.json")["c3_job0"]
    ir = oracle.synth_ir(g["M"], g["seed_ir"])
    x = oracle.synth_noise(g["L"], g["seed_x"])
    k = oracle.fft_process_kernel(g["L"], ir)
    y = oracle.fft_convolution(g["L"], k, g["M"])(x)
    assert len(y) == g["out_len"]
    assert_bits_equal(y[:64], g["head"], "head")
    assert_bits_equal(y[-64:], g["tail"], "tail")
    assert_bits_equal(y[::g["stride"]], g["samples"], "strided")


def test_g6_ring_buffer_and_loop_integers(oracle):
    g = load_golden("g6_ring_and_loop.json")
    buf = np.zeros(3, np.float32)
    ring = oracle.RingBuffer(buf)
    for st in g["ring3"]:
        ring[0] = st["write"]
        assert buf.tolist() == st["buffer"]
        for k, v in st["reads"].items():
            assert ring[int(k)] == v, (st["write"], k)
    buf = np.zeros(5, np.float32)
    ring = oracle.RingBuffer(buf)
    for st in g["ring5"]:
        ring[st["key"]] = st["value"]
        assert buf.tolist() == st["buffer"], st
    for b in g["beat"]:
        assert oracle.beat_rate(b["sr"], b["bpm"]) == b["beat_rate"]
        assert oracle.buffer_size(b["sr"], b["bpm"]) == b["buffer_size"]
    # the configuration the stereo-reverb benchmark is quoted on: 48 kHz, 120 BPM
    assert oracle.beat_rate(48000, 120) == 24000 and oracle.buffer_size(48000, 120) == 96000


def test_g7_render_jobs_with_4s_impulse_responses(oracle):
    """Channel-jobs of the 2048-track render (L = 2 880 000, M = 192 000, an IR per job)."""
    g = load_golden("g7_c5_jobs.json")
    for job in g["jobs"]:
        ir = oracle.synth_ir(g["M"], job["seed_ir"])
        x = oracle.synth_noise(g["L"], job["seed_x"])
        k = oracle.fft_process_kernel(g["L"], ir)
        assert len(k) == job["kernel_len"]
        y = oracle.fft_convolution(g["L"], k, g["M"])(x)
        assert len(y) == job["out_len"] == g["L"] + g["M"]
        assert_bits_equal(y[:64], job["head"], "head")
        assert_bits_equal(y[-64:], job["tail"], "tail")
        assert_bits_equal(y[job["mid"]["at"]:job["mid"]["at"] + 256], job["mid"]["data"], "mid")
        assert_bits_equal(y[::job["stride"]], job["samples"], "strided")


def test_g8_mix_negative_offsets(oracle):
    g = load_golden("g8_mix_negative_offsets.json")
    assert len(g["cases"]) == 20
    for case in g["cases"]:
        target = [from_bits(b).copy() for b in case["target_in"]]
        oracle.mix_buffers(target, *_sources(case))
        for ch, exp in enumerate(case["target_out"]):
            assert_bits_equal(target[ch], exp, case["name"] + " ch%d" % ch)
    # the -0 index is hit: some outputs of the first case are finite beside NaNs
    first = from_bits(g["cases"][0]["target_out"][0])
    assert np.isnan(first).any() and np.isfinite(first).any()


def test_g11_mix_sources_that_share_memory_with_the_target(oracle):
    """mixBuffers(t, a, t), feedback offsets, views into and beyond the target: the reference reads such a
    source sample by sample as the loop updates the target (src/mix-buffers.js:4, 20-67)."""
    from conftest import shared_memory_mix_case
    g = load_golden("g11_mix_shared_memory.json")
    assert len(g["cases"]) == 15
    for case in g["cases"]:
        bases, target, sources = shared_memory_mix_case(case)
        oracle.mix_buffers(target, *sources)
        for i, exp in enumerate(case["bases_out"]):
            assert_bits_equal(bases[i], exp, case["name"] + " base %d" % i)


def render_conventions_via_mix(mix_buffers, g):
    """src/render.js:27-48 as mixBuffers arithmetic on a zeroed buffer: a mono function's
    samples into 2 channels are (s/2)*1 on both (render: /numOfChannels, then buffer[1].set(buffer[0])),
    a stereo function's into 1 channel ((l+r)/2)*1, equal layouts are copies."""
    n = g["frames"]
    mono, left, right = from_bits(g["mono"]), from_bits(g["left"]), from_bits(g["right"])
    out = []
    for case in g["cases"]:
        target = [np.zeros(n, np.float32) for _ in range(case["channels"])]
        mix_buffers(target, [mono] if case["fn"] == "mono" else [left, right])
        assert case["n_after"] == n
        out.append((case, target))
    return out


def test_g9_render_buffer_conventions(oracle):
    g = load_golden("g9_render_conventions.json")
    assert g["errors"]["nan"][0] == "TypeError" and g["errors"]["three_channels"][0] == "RangeError"
    for case, target in render_conventions_via_mix(oracle.mix_buffers, g):
        for ch, exp in enumerate(case["out"]):
            assert_bits_equal(target[ch], exp, "%s fn into %d channel(s), ch%d" % (case["fn"], case["channels"], ch))


def test_g10_loop_node_bpm_getter(oracle):
    g = load_golden("g10_loop_bpm.json")
    for r in g["rows"]:
        br0 = oracle.beat_rate(r["sr"], r["bpm"])
        back = oracle.bpm_from_beat_rate(r["sr"], br0)
        assert back == r["bpm_back"], r
        assert oracle.beat_rate(r["sr"], back) == r["beat_rate"] and 4 * r["beat_rate"] == r["buffer_size"]
    for t in g["ties"]:
        assert oracle.bpm_from_beat_rate(t["sr"], t["beat_rate"]) == t["bpm_back"], t
    for b in load_golden("g6_ring_and_loop.json")["beat"]:
        assert oracle.bpm_from_beat_rate(b["sr"], b["beat_rate"]) == b["bpm_back"]


def test_javascript_restatement_is_pinned_to_the_same_fixtures():
    """oracle/oracle_go.mjs (the Node.js CPU baseline bench.py times) reproduces the reference-generated
    fixtures g1, g4, g8 and the C1-sized run bit for bit."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    node = shutil.which("node")
    if not node:
        pytest.skip("node is not installed")
    out = subprocess.run([node, "--no-warnings", os.path.join(ROOT, "oracle", "oracle_go.mjs"), "check"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
