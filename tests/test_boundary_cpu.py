"""CPU-side checks of the drop-in boundary (no GPU, no compute calls):
the C-ABI library loads and exports every symbol include/wpdsp.h declares, the
integer helpers match the golden vectors, the product path fails loudly without
a GPU, the FFT stage code passes its host emulation, the Node addon + shims load."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _header_functions():
    text = open(os.path.join(ROOT, "include", "wpdsp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from dsp_amd import capi
    lib = capi.lib()
    declared = _header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libwpdsp.so lacks %s" % name
    # and the Python binding covers the same set
    assert sorted(capi.PROTOTYPES) == declared
    assert C.sizeof(capi.WpSrc) == 64 and C.sizeof(capi.WpReverbView) == 40


def test_integer_helpers_match_reference_vectors():
    from dsp_amd import capi
    lib = capi.lib()
    g2 = load_golden("g2_tables.json")
    for n, N in g2["next_pot_via_kernel_len"]:
        assert lib.wp_next_pot(n) == N
    assert lib.wp_next_pot(0) == 1
    assert lib.wp_fft_length(44100, 22050) == 131072
    assert lib.wp_fft_length(2880000, 96000) == 4194304
    assert lib.wp_fft_length(1 << 30, 1) == capi.WP_ERR_FFT_SIZE
    g6 = load_golden("g6_ring_and_loop.json")
    for b in g6["beat"]:
        assert lib.wp_beat_rate(b["sr"], b["bpm"]) == b["beat_rate"]
    # the bpm getter (src/loop-node.js:9-16), including toFixed(6) ties
    for b in g6["beat"]:
        assert lib.wp_bpm_from_beat_rate(b["sr"], b["beat_rate"]) == b["bpm_back"]
    g10 = load_golden("g10_loop_bpm.json")
    for r in g10["rows"]:
        br0 = lib.wp_beat_rate(r["sr"], r["bpm"])
        back = lib.wp_bpm_from_beat_rate(r["sr"], br0)
        assert back == r["bpm_back"], r
        assert lib.wp_beat_rate(r["sr"], back) == r["beat_rate"]
    for t in g10["ties"]:
        assert lib.wp_bpm_from_beat_rate(t["sr"], t["beat_rate"]) == t["bpm_back"], t
    # no element / no block length where the reference yields NaN or Infinity
    assert lib.wp_ring_pos(0, 3, 0) == -1 and lib.wp_ring_advance(0, 0) == -1 and lib.wp_beat_rate(48000, 0) == -1
    # ring buffer replay (test/ring-buffer.js:14-39)
    buf, needle = [0.0] * 3, 0
    for st in g6["ring3"]:
        buf[needle] = st["write"]
        needle = lib.wp_ring_advance(needle, 3)
        assert buf == st["buffer"]
        for k, v in st["reads"].items():
            assert buf[lib.wp_ring_pos(needle, int(k), 3)] == v
    # (rl*length)|0
    assert lib.wp_mix_sl(8, 0.75) == 6 and lib.wp_mix_sl(6, 2.5) == 15
    assert lib.wp_mix_sl(8, float("nan")) == 0 and lib.wp_mix_sl(1 << 31, 1.0) == -(1 << 31)


def test_reverb_view_arithmetic_matches_reference_blocks():
    """Index work is bit-exact: the subarray() shapes of every golden reverb block."""
    from dsp_amd import capi
    lib = capi.lib()
    g = load_golden("g3_reverb_blocks.json")
    n = 0
    for run in g["runs"]:
        M = len(run["ir"][0]) if run["params"].get("length", -1) < 0 else min(len(run["ir"][0]), run["params"]["length"])
        off = run["params"].get("offset", 0)
        for blk in run["blocks"]:
            v = capi.WpReverbView()
            assert lib.wp_reverb_view_for(run["L"], M, off, run["L"], C.byref(v)) == 0
            assert v.curr_len * 4 == blk["ret"][0]["buffer_bytes"]
            assert v.ret_len == blk["ret"][0]["length"] and v.ret_off * 4 == blk["ret"][0]["byte_offset"]
            assert v.prev_len == blk["prev"][0]["length"] and v.prev_off * 4 == blk["prev"][0]["byte_offset"]
            n += 1
    assert n > 40


def test_no_cpu_fallback_without_gpu():
    from dsp_amd import capi, convolve, mix
    if capi.lib().wp_init() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="HIP device"):
        convolve.fft_process_kernel(8, np.ones(3, np.float32))
    with pytest.raises(RuntimeError, match="HIP device"):
        mix.mix_buffers([np.zeros(4, np.float32)], [np.ones(4, np.float32)])


def test_synthetic_ir_matches_oracle_generator(oracle):
    from dsp_amd import device
    for M, seed in [(5, 21), (300, 16), (22050, 500000)]:
        assert np.array_equal(device.synth_ir(M, seed).view(np.uint32), oracle.synth_ir(M, seed).view(np.uint32))


def test_fft_stage_code_host_emulation(tmp_path):
    exe = str(tmp_path / "fft_emu")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "host_emu", "fft_emu.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "FAIL" not in out.stdout


def test_node_addon_and_shims_load():
    node = shutil.which("node")
    addon = os.path.join(ROOT, "dsp_amd", "napi", "wpdsp.node")
    if not node or not os.path.exists(addon):
        pytest.skip("node or the built addon is not available")
    out = subprocess.run([node, "--no-warnings", os.path.join(ROOT, "tests", "node", "test_load.mjs")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok load")


def test_node_sample_file_ingestion():
    """RIFF/WAVE reader that stands in for the browser's decodeAudioData in front of c.sample(url)
    (src/sample-service.js:12-37): PCM 8/16/24/32, float 32/64, extensible header, odd chunks,
    per-path cache.  Scaling follows the Web Audio rule (divide by 2^(bits-1)); the reference holds
    no audio fixture, so the expected values are hand-computed: parity unpinned."""
    node = shutil.which("node")
    if not node:
        pytest.skip("node is not available")
    out = subprocess.run([node, "--no-warnings", os.path.join(ROOT, "tests", "node", "test_sample_file.mjs")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok sample-file")


def test_power_of_two_gain_fold_is_one_rounding():
    """The kernels fold a sample into the bus with ONE binary32 fused multiply-add when the gain is +-2^k,
    instead of the reference's binary64 multiply, binary64 add and binary32 store
    (src/mix-buffers.js:49-52).  The two are the same number: checked here with exact rational arithmetic
    on operands chosen to sit on and around rounding ties, far-apart exponents and subnormals."""
    from fractions import Fraction

    def round_f32(fr):  # exact value -> nearest binary32, ties to even (what fmaf returns)
        if fr == 0:
            return 0.0
        sign = -1 if fr < 0 else 1
        a = abs(fr)
        e = -149
        while a / Fraction(2) ** e >= 2 ** 24:
            e += 1
        q = a / Fraction(2) ** e
        n = q.numerator // q.denominator
        rem = q - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
            n += 1
        return float(sign * n * Fraction(2) ** e) if e + 24 <= 128 else sign * float("inf")

    rng = np.random.default_rng(5)
    cases = []
    for _ in range(6000):
        ea, eb = int(rng.integers(-140, 60)), int(rng.integers(-140, 60))
        if rng.random() < 0.6:
            eb = ea - int(rng.integers(18, 36))  # around the 24..29-bit gap where double rounding could bite
        a = np.float32(np.ldexp(np.float32(rng.integers(1 << 23, 1 << 24)), ea - 23)) * (1 if rng.random() < 0.5 else -1)
        m = int(rng.integers(1 << 23, 1 << 24))
        if rng.random() < 0.5:
            m = (1 << 23) | (1 << int(rng.integers(0, 23)))  # few bits set: exact ties after alignment
        y = np.float32(np.ldexp(np.float32(m), eb - 23)) * (1 if rng.random() < 0.5 else -1)
        k = int(rng.integers(-12, 4))
        cases.append((a, y, np.ldexp(1.0, k) * (1 if rng.random() < 0.8 else -1)))
    cases += [(np.float32(0.0), np.float32(1e-40), 0.5), (np.float32(-0.0), np.float32(0.0), 1.0), (np.float32(1.0), np.float32(2.0 ** -24), 1.0),
              (np.float32(1.0), np.float32(-(2.0 ** -25)), 1.0), (np.float32(1e-38), np.float32(3e-39), 0.25)]
    with np.errstate(all="ignore"):
        for a, y, vol in cases:
            if not (np.isfinite(a) and np.isfinite(y)):
                continue
            ref = np.float32(np.float64(a) + np.float64(y) * np.float64(vol))
            one = np.float32(round_f32(Fraction(float(a)) + Fraction(float(y)) * Fraction(vol)))
            assert ref.tobytes() == one.tobytes() or (ref == 0 and one == 0), (a, y, vol, ref, one)


def test_long_frame_cover_search_is_host_only_and_consistent():
    """wp_lf_cover: the runs of equal frames that cover a result of L+M samples (no GPU needed).  Every run's
    frames hold the whole impulse response, the runs' output ranges follow each other without gaps and reach
    L+M, and the two bench workloads get the covers the design quotes."""
    import ctypes as C
    from dsp_amd import capi
    lib = capi.lib()

    def cover(L, M, nch):
        r = (C.c_int64 * 12)()
        n = lib.wp_lf_cover(L, M, nch, r)
        assert 0 <= n <= 3
        return [tuple(r[4 * i + k] for k in range(4)) for i in range(n)]

    assert [(f, n) for f, n, _, _ in cover(2880000, 96000, 2)] == [(1048576, 3), (262144, 1)]
    assert [(f, n) for f, n, _, _ in cover(2880000, 192000, 512)] == [(2097152, 1), (1048576, 1), (524288, 1)]
    assert cover(44100, 22050, 1) == [] and cover(96000, 96000, 2) == []  # short inputs: partitioned form
    lib.wp_tune(5, 2)  # long-frame wherever a cover exists
    try:
        for L, M, nch in [(600000, 5000, 2), (1150001, 96000, 2), (2880000, 96000, 2), (2880000, 192000, 512), (1000000, 800000, 2),
                          (123457, 31, 1), (5000000, 700001, 3)]:
            runs = cover(L, M, nch)
            assert runs, (L, M)
            base = 0
            for frame, nfr, hop, first in runs:
                assert frame in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21) and nfr >= 1
                assert first == base and hop >= frame // 4 and frame - hop >= M - 1 and (frame - hop) % 64 == 0
                base += nfr * hop
            assert base >= L + M
            assert [f for f, _, _, _ in runs] == sorted((f for f, _, _, _ in runs), reverse=True)
    finally:
        lib.wp_tune(5, 0)


def test_wide_chain_schedule_is_host_logic():
    """wp_lf_wide_frames (no GPU needed): how many frames of a run's bus-fold pass take the wide tiles.  A wide
    chain costs 1.5 narrow ones and has a CU to itself; the split with the shortest schedule wins, ties go wide."""
    from dsp_amd import capi
    lib = capi.lib()
    # the 60 s render with a 2 s response: three frames of 2^20 samples on 256 CUs -> two wide (256 chains, one
    # round) and one narrow (256 chains); all three wide would leave half the chip idle in a second round
    assert lib.wp_lf_wide_frames(2048, 3, 256) == 2
    assert lib.wp_lf_wide_frames(2048, 2, 256) == 2
    assert lib.wp_lf_wide_frames(2048, 1, 256) == 0      # 128 wide chains on 256 CUs: 1.5 against 1.0 for 256 narrow ones
    assert lib.wp_lf_wide_frames(4096, 1, 256) == 1      # one frame of 2^21 samples: 256 wide chains, one round
    assert lib.wp_lf_wide_frames(2048, 0, 256) == 0
    assert lib.wp_lf_wide_frames(2048, 4, 256) == 4      # 512 wide chains: two full rounds (3.0) against 4.0
    capi.check(lib.wp_tune(13, 1), "wp_tune")
    try:
        assert lib.wp_lf_wide_frames(2048, 3, 256) == 0
        capi.check(lib.wp_tune(13, 2), "wp_tune")
        assert lib.wp_lf_wide_frames(2048, 3, 256) == 3
    finally:
        lib.wp_tune(13, 0)
    assert lib.wp_lf_wide_frames(1024, 3, 256) == capi.WP_ERR_INVALID
