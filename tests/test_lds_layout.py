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
