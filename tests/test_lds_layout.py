"""The FFT kernels exchange data between stages through LDS at index ``i ^ ((i >> 4) & 15)``
(dsp_amd/csrc/wp_fft.h: lds_pad).  This checks, for every stage plan, that each exchange
pattern is free of bank conflicts under the banking the hardware applies per instruction:
ds_write_b64 is serviced in groups of 16 lanes over 32 four-byte banks, ds_read_b64 in groups
of 32 lanes over 64 banks.  The plans are restated here (PT points per thread, radix per stage)
and must be kept in step with FftPlan<B>."""
import re
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANS = {1024: (16, [16, 16, 4]), 2048: (16, [16, 16, 8]), 4096: (16, [16, 16, 16]), 8192: (16, [16, 8, 8, 8])}


def lds_pad(i):
    return i ^ ((i >> 4) & 15)


def exchange_patterns(B, PT, radices):
    """(kind, [element index per thread]) for every wave-wide LDS instruction of one transform:
    stage_store of stage s < last writes (j-k)*R + k + t*NS, stage_load of stage s > 0 reads
    j + t*B/R, with j = tid + u*NT and k = j & (NS-1)."""
    NT = B // PT
    ns = 1
    for s, R in enumerate(radices):
        for u in range(PT // R):
            for t in range(R):
                if s > 0:
                    yield "load", [(tid + u * NT) + t * (B // R) for tid in range(NT)]
                if s < len(radices) - 1:
                    idx = []
                    for tid in range(NT):
                        j = tid + u * NT
                        k = j & (ns - 1)
                        idx.append((j - k) * R + k + t * ns)
                    yield "store", idx
        ns *= R


def lds_cycles(kind, idx):
    group, banks = (32, 64) if kind == "load" else (16, 32)
    total = 0
    for g in range(0, len(idx), group):
        hits = {}
        for i in idx[g:g + group]:
            dword = 2 * lds_pad(i)  # 8-byte elements
            for b in (dword % banks, (dword + 1) % banks):
                hits[b] = hits.get(b, 0) + 1
        total += max(hits.values())
    return total


def test_plans_match_the_header():
    src = open(os.path.join(ROOT, "dsp_amd", "csrc", "wp_fft.h")).read()
    assert "return i ^ ((i >> 4) & 15);" in src
    for B, (PT, radices) in PLANS.items():
        m = re.search(r"struct FftPlan<%d> \{\s*static constexpr int PT = (\d+), NSTAGE = (\d+);" % B, src)
        assert m and int(m.group(1)) == PT and int(m.group(2)) == len(radices), B
        prod = 1
        for r in radices:
            prod *= r
        assert prod == B


@pytest.mark.parametrize("B", sorted(PLANS))
def test_exchange_patterns_are_conflict_free(B):
    PT, radices = PLANS[B]
    for kind, idx in exchange_patterns(B, PT, radices):
        ideal = len(idx) // (32 if kind == "load" else 16)
        assert lds_cycles(kind, idx) == ideal, (B, kind)


def test_one_pad_per_sixteen_would_conflict():
    """The layout this replaced (i + i/16) leaves every exchange load two-way conflicted."""
    global lds_pad
    keep = lds_pad
    try:
        lds_pad = lambda i: i + (i >> 4)  # noqa: E731
        PT, radices = PLANS[4096]
        loads = [idx for kind, idx in exchange_patterns(4096, PT, radices) if kind == "load"]
        assert all(lds_cycles("load", idx) == 2 * (len(idx) // 32) for idx in loads)
    finally:
        lds_pad = keep


# ---------------------------------------------------------------------------
# The batched row pass of the long-frame form (k_lf_mid_jobs) keeps its rows at i + i/32 (lds_pad32)
# and lets lane l take butterfly pad32_lane_butterfly(l).  Its two transforms per job go through LDS
# at every stage (the last forward store and the first inverse load included, around the spectral
# product), and the product reads and rewrites bins tid + 256*q and their mirrored partners.
def lds_pad32(i):
    return i + (i >> 5)


def lane_butterfly(lane):
    return (lane & ~31) | ((lane & 15) << 1) | (((lane >> 3) ^ (lane >> 4)) & 1)


def row_pass_patterns(B, PT, radices, jmap):
    NT = B // PT
    ns = 1
    for s, R in enumerate(radices):
        for u in range(PT // R):
            for t in range(R):
                yield "load", s, [(jmap(tid) + u * NT) + t * (B // R) for tid in range(NT)]
                idx = []
                for tid in range(NT):
                    j = jmap(tid) + u * NT
                    k = j & (ns - 1)
                    idx.append((j - k) * R + k + t * ns)
                yield "store", s, idx
        ns *= R
    WGT = 2 * NT  # the product: both rows' threads, bins tid + WGT*q and C-1-(tid + WGT*q)
    for q in range(B // WGT):
        for kind in ("load", "store"):
            yield kind, "product", [tid + WGT * q for tid in range(WGT)]
            yield kind, "product", [B - 1 - (tid + WGT * q) for tid in range(WGT)]


def pad32_cycles(kind, idx):
    group, slots = (32, 32) if kind == "load" else (16, 16)  # eight-byte slots
    total = 0
    for g in range(0, len(idx), group):
        hits = {}
        for i in idx[g:g + group]:
            hits[lds_pad32(i) % slots] = hits.get(lds_pad32(i) % slots, 0) + 1
        total += max(hits.values())
    return total


def test_pad32_helpers_match_the_header():
    src = open(os.path.join(ROOT, "dsp_amd", "csrc", "wp_fft.h")).read()
    assert "WP_HD int lds_pad32(int i) { return i + (i >> 5); }" in src
    assert "return (lane & ~31) | ((lane & 15) << 1) | (((lane >> 3) ^ (lane >> 4)) & 1);" in src
    kern = open(os.path.join(ROOT, "dsp_amd", "csrc", "wp_longframe_kernels.hip")).read()
    assert "pad32_lane_butterfly(tid & (NT - 1))" in kern


@pytest.mark.parametrize("B", [2048, 4096])
def test_row_pass_exchanges_are_conflict_free_on_the_padded_layout(B):
    PT, radices = PLANS[B]
    NT = B // PT
    assert sorted(lane_butterfly(t) for t in range(NT)) == list(range(NT))
    # a wave still covers 64 consecutive butterflies: its global loads and stores stay whole 512-byte pieces
    for w in range(0, NT, 64):
        assert sorted(lane_butterfly(t) for t in range(w, w + 64)) == list(range(w, w + 64))
    for kind, s, idx in row_pass_patterns(B, PT, radices, lane_butterfly):
        ideal = len(idx) // (32 if kind == "load" else 16)
        assert pad32_cycles(kind, idx) == ideal, (B, kind, s)


def test_row_pass_with_lane_equal_butterfly_would_conflict():
    """What the permutation removes: with butterfly = lane every stage-0 store is served in two passes."""
    PT, radices = PLANS[2048]
    extra = {}
    for kind, s, idx in row_pass_patterns(2048, PT, radices, lambda t: t):
        ideal = len(idx) // (32 if kind == "load" else 16)
        extra[(kind, s)] = extra.get((kind, s), 0) + pad32_cycles(kind, idx) - ideal
    assert extra[("store", 0)] == 128 and sum(extra.values()) == 128
