import json
import os
import sys

import numpy as np
import pytest

# torch bundles its own HIP runtime; load it before libwpdsp.so pulls in the system one, so a
# process that uses both (device-API tests, bench.py) ends up with a single runtime whatever
# order the test modules run in.
try:
    import torch  # noqa: F401
except Exception:  # torch is optional for the oracle / boundary tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def from_bits(b):
    """uint32 bit patterns -> float32 array."""
    return np.array(b, dtype=np.uint32).view(np.float32)


def to_bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(actual, expected_bits, what=""):
    """Bit-exact f32 comparison; any NaN payload counts as equal to any NaN."""
    a = np.ascontiguousarray(actual, dtype=np.float32)
    e = from_bits(expected_bits)
    assert a.shape == e.shape, "%s: length %s != %s" % (what, a.shape, e.shape)
    an, en = np.isnan(a), np.isnan(e)
    assert np.array_equal(an, en), "%s: NaN positions differ" % what
    ab, eb = a.view(np.uint32), e.view(np.uint32)
    bad = (ab != eb) & ~an
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        raise AssertionError("%s: %d of %d samples differ, first at %d: %r (0x%08x) != %r (0x%08x)" % (
            what, int(bad.sum()), a.size, i, float(a[i]), int(ab[i]), float(e[i]), int(eb[i])))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


def shared_memory_mix_case(case):
    """A g11 case rebuilt with the same memory sharing: (base arrays, target channel views, sources) where
    every channel is a numpy view [begin, end) into one of the bases."""
    vol_fns = {"ramp": lambda x: 0.25 + x / 8, "alt": lambda x: -1.3 if (x & 1) else 0.7}
    bases = [from_bits(b).copy() for b in case["bases_in"]]
    view = lambda spec: bases[spec[0]][spec[1]:spec[2]]  # noqa: E731
    target = [view(t) for t in case["target"]]
    sources = []
    for s in case["sources"]:
        ch = [view(c) for c in s["ch"]]
        if "length" not in s:
            sources.append(ch)
            continue
        vol = s.get("volume")
        if isinstance(vol, dict):
            vol = vol_fns[vol["fn"]]
        sources.append((ch, s["length"], vol, s.get("offset")))
    return bases, target, sources
