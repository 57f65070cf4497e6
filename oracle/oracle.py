"""CPU oracle for the wavepot/dsp convolution + mix hot path (TEST INFRASTRUCTURE).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.  It is the checker, never the thing shipped or measured as
the product: the product path is ``dsp_amd`` over ``libwpdsp.so``.

Parity status: pinned — every function below is compared bit-for-bit with
vectors produced by the reference's own JavaScript (oracle/gen_golden.mjs ->
tests/golden/*.json, checked in tests/test_oracle_golden.py).

The sample arithmetic lives in oracle/wp_oracle.c (f64 operators, f32 stores);
this file restates the *wrapper* logic of the reference in Python:

* ``fft_process_kernel`` / ``fft_convolution``  lib/convolve.js:553-586
* ``impulse_convolve[_stereo]``                 lib/impulse-convolve.js:3-20,
                                                lib/impulse-convolve-stereo.js:3-27
* ``impulse_reverb[_stereo]``                   lib/impulse-reverb.js:3-22,
                                                lib/impulse-reverb-stereo.js:3-31
* ``mix_buffers``                               src/mix-buffers.js:1-73
* ``RingBuffer``                                lib/ring-buffer.js:1-59
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libwporacle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def build(force=False):
    """Compile oracle/wp_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "wp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.wpo_next_pot.restype = C.c_int32
        L.wpo_next_pot.argtypes = [C.c_int32]
        L.wpo_fft_new.restype = C.c_void_p
        L.wpo_fft_new.argtypes = [C.c_int32]
        L.wpo_fft_free.argtypes = [C.c_void_p]
        L.wpo_fft_table.restype = _f32p
        L.wpo_fft_table.argtypes = [C.c_void_p]
        L.wpo_fft_transform.argtypes = [C.c_void_p, _f32p, _f32p]
        L.wpo_fft_inverse.argtypes = [C.c_void_p, _f32p, _f32p]
        L.wpo_fft_process_kernel.restype = C.c_int64
        L.wpo_fft_process_kernel.argtypes = [C.c_int32, _f32p, C.c_int32, _f32p]
        L.wpo_fft_convolution_go.restype = C.c_int
        L.wpo_fft_convolution_go.argtypes = [C.c_void_p, C.c_int32, _f32p, C.c_int32, _f32p, C.c_int64, _f32p]
        L.wpo_convolve.restype = C.c_int
        L.wpo_convolve.argtypes = [_f32p, C.c_int64, C.c_int32, _f32p, C.c_int32, _f32p]
        L.wpo_direct_convolution.argtypes = [_f32p, C.c_int64, _f32p, C.c_int64, _f32p]
        L.wpo_add_tail.argtypes = [_f32p, _f32p, C.c_int64]
        L.wpo_mix_source.restype = C.c_int
        L.wpo_mix_source.argtypes = [C.POINTER(_f32p), C.c_int, C.c_int64, C.POINTER(_f32p), C.c_int,
                                     C.c_int64, C.c_int64, C.c_double, _f64p, C.c_int64]
        L.wpo_mix_sl.restype = C.c_int32
        L.wpo_mix_sl.argtypes = [C.c_int64, C.c_double]
        L.wpo_ring_pos.restype = C.c_int64
        L.wpo_ring_pos.argtypes = [C.c_int64, C.c_int64, C.c_int64]
        L.wpo_ring_advance.restype = C.c_int64
        L.wpo_ring_advance.argtypes = [C.c_int64, C.c_int64]
        L.wpo_beat_rate.restype = C.c_int64
        L.wpo_beat_rate.argtypes = [C.c_double, C.c_double]
        L.wpo_synth_noise.argtypes = [_f32p, C.c_int64, C.c_uint32, C.c_int64]
        L.wpo_synth_ir.argtypes = [_f32p, C.c_int64, C.c_uint32]
        L.wpo_checksum.restype = C.c_uint32
        L.wpo_checksum.argtypes = [_f32p, C.c_int64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_f32p)


def _c32(a):
    a = np.asarray(a, dtype=np.float32)
    return a if a.flags["C_CONTIGUOUS"] else np.ascontiguousarray(a)


# ---------------------------------------------------------------- synthetic inputs
def synth_noise(n, seed, start=0):
    out = np.empty(n, dtype=np.float32)
    lib().wpo_synth_noise(_p(out), n, seed & 0xFFFFFFFF, start)
    return out


def synth_ir(M, seed):
    out = np.empty(M, dtype=np.float32)
    lib().wpo_synth_ir(_p(out), M, seed & 0xFFFFFFFF)
    return out


def checksum(a):
    """Order-sensitive 32-bit checksum of f32 bit patterns: h = h*31 + bits."""
    a = _c32(a)
    return int(lib().wpo_checksum(_p(a), len(a)))


# ---------------------------------------------------------------- convolve.js
def next_pot(n):
    return lib().wpo_next_pot(int(n))


class FFT:
    """lib/convolve.js:4-45 state + transform/inverseTransform (:86-122)."""

    def __init__(self, size):
        self.size = int(size)
        self._h = lib().wpo_fft_new(self.size)
        if not self._h:
            raise ValueError("FFT size must be a power of two and bigger than 1")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().wpo_fft_free(self._h)
            self._h = None

    @property
    def table(self):
        return np.ctypeslib.as_array(lib().wpo_fft_table(self._h), shape=(2 * self.size,)).copy()

    def transform(self, data):
        data = _c32(data)
        out = np.empty(2 * self.size, dtype=np.float32)
        lib().wpo_fft_transform(self._h, _p(out), _p(data))
        return out

    def inverse_transform(self, data):
        data = _c32(data)
        out = np.empty(2 * self.size, dtype=np.float32)
        lib().wpo_fft_inverse(self._h, _p(out), _p(data))
        return out


def direct_convolution(x, h):
    x, h = _c32(x), _c32(h)
    out = np.empty(len(x) + len(h) - 1, dtype=np.float32)
    lib().wpo_direct_convolution(_p(x), len(x), _p(h), len(h), _p(out))
    return out


def fft_process_kernel(length, kernel):
    """lib/convolve.js:553-562 -> interleaved spectrum, 2*nextPOT(length+M) floats."""
    kernel = _c32(kernel)
    N = next_pot(length + len(kernel))
    if N <= 1:
        raise ValueError("FFT size must be a power of two and bigger than 1")
    out = np.empty(2 * N, dtype=np.float32)
    rc = lib().wpo_fft_process_kernel(length, _p(kernel), len(kernel), _p(out))
    if rc < 0:
        raise RuntimeError("fft_process_kernel failed")
    return out


def fft_convolution(length, fft_kernel, kernel_length):
    """lib/convolve.js:564-586 -> go(input) returning length+kernel_length floats."""
    fft_kernel = _c32(fft_kernel)
    fft = FFT(next_pot(length + kernel_length))

    def go(x):
        x = _c32(x)
        out = np.empty(length + kernel_length, dtype=np.float32)
        rc = lib().wpo_fft_convolution_go(fft._h, length, _p(fft_kernel), kernel_length, _p(x), len(x), _p(out))
        if rc < 0:
            raise IndexError("offset is out of bounds")  # RangeError from input.set (:590)
        return out

    return go


# ---------------------------------------------------------------- wrappers
def js_subarray(a, begin=0, end=None):
    """TypedArray.prototype.subarray index clamping (negative = from the end)."""
    n = len(a)
    b = int(begin)
    b = max(n + b, 0) if b < 0 else min(b, n)
    e = n if end is None else int(end)
    e = max(n + e, 0) if e < 0 else min(e, n)
    return a[b:max(b, e)]


class StubContext:
    """The surface of src/context.js the wrappers touch: buffer, n, sample/get/set
    (context.js:112-130 -> global-service.js:10-19), synchronous here."""

    def __init__(self, buffer, n, irs, store, log=None):
        self.buffer = buffer
        self.n = n
        self._irs = irs
        self._store = store
        self.log = log if log is not None else []

    def sample(self, url):
        return list(self._irs[url])

    def get(self, key):
        v = self._store.get(key, False)
        self.log.append(["get", key, v is not False])
        return v

    def set(self, key, value, ttl=None):
        self._store[key] = value
        self.log.append(["set", key, ttl])


def impulse_convolve(c, url, length):
    """lib/impulse-convolve.js:3-20"""
    impulse = c.sample(url)
    if length > -1:
        impulse[0] = js_subarray(impulse[0], 0, length)
    L = len(c.buffer[0])
    key = "kernel:%s:%d:%d" % (url, L, length)
    kernel = c.get(key)
    if kernel is False:
        kernel = fft_process_kernel(L, impulse[0])
        c.set(key, kernel)
    return fft_convolution(L, kernel, len(impulse[0]))


def impulse_convolve_stereo(c, url, length):
    """lib/impulse-convolve-stereo.js:3-27 (both go()s use impulse[0].length)"""
    impulse = c.sample(url)
    if length > -1:
        impulse[0] = js_subarray(impulse[0], 0, length)
        impulse[1] = js_subarray(impulse[1], 0, length)
    L = len(c.buffer[0])
    key = "impulse-convolve-stereo:kernel:%s:%d:%d" % (url, L, length)
    kernel = c.get(key)
    if kernel is False:
        kernel = [fft_process_kernel(L, impulse[0]), fft_process_kernel(L, impulse[1])]
        c.set(key, kernel)
    return [fft_convolution(L, kernel[0], len(impulse[0])),
            fft_convolution(L, kernel[1], len(impulse[0]))]


def _add_tail(curr, prev, n):
    pc = np.ascontiguousarray(prev[:n])
    lib().wpo_add_tail(_p(curr), _p(pc), n)


def _num(x):
    """JS number -> string as string concatenation renders integers."""
    return str(int(x)) if float(x).is_integer() else repr(float(x))


def impulse_reverb(c, url, offset=0, length=-1, id=""):
    """lib/impulse-reverb.js:3-22"""
    reverb = impulse_convolve(c, url, length)
    got = c.get("prev:" + id + url + _num(c.n - len(c.buffer[0])))
    state = {"prev": got if got is not False else np.zeros(0, dtype=np.float32)}

    def block(c):
        ln = len(c.buffer[0])
        curr = reverb(c.buffer[0])
        prev = state["prev"]
        _add_tail(curr, prev, len(prev))
        tail = (len(curr) - offset) - ln
        state["prev"] = js_subarray(curr, -tail)
        c.set("prev:" + id + url + _num(c.n), state["prev"], 5000)
        return js_subarray(curr, offset, offset + ln)

    return block


def impulse_reverb_stereo(c, url, offset=0, length=-1, id=""):
    """lib/impulse-reverb-stereo.js:3-31"""
    reverb = impulse_convolve_stereo(c, url, length)
    got = c.get("impulse-reverb-stereo:prev:" + id + url + _num(c.n - len(c.buffer[0])))
    prev = got if got is not False else [np.zeros(0, dtype=np.float32), np.zeros(0, dtype=np.float32)]

    def block(c):
        ln = len(c.buffer[0])
        curr = [reverb[0](c.buffer[0]), reverb[1](c.buffer[1])]
        n = len(prev[0])
        _add_tail(curr[0], prev[0], n)
        _add_tail(curr[1], prev[1], n)
        tail = (len(curr[0]) - offset) - ln
        prev[0] = js_subarray(curr[0], -tail)
        prev[1] = js_subarray(curr[1], -tail)
        c.set("impulse-reverb-stereo:prev:" + id + url + _num(c.n), prev, 5000)
        return [js_subarray(curr[0], offset, offset + ln), js_subarray(curr[1], offset, offset + ln)]

    return block


# ---------------------------------------------------------------- mix-buffers.js
def mix_buffers(target, *sources):
    """src/mix-buffers.js:1-73.  target: list of 1-2 float32 arrays (mutated);
    source: list of 1-2 arrays, or a tuple/list ``(buffer, length, volume, offset)``
    whose first element is itself a list (the reference's Array.isArray test, :6).
    ``volume`` may be a callable x -> gain."""
    L = lib()
    tl = len(target[0])
    tptr = (_f32p * len(target))(*[_p(t) for t in target])
    for source in sources:
        vol, o = 1.0, 0
        if isinstance(source[0], (list, tuple)):
            buf = source[0]
            rl = len(buf[0])
            length = source[1] if len(source) > 1 and source[1] is not None else float("nan")
            sl = L.wpo_mix_sl(rl, float(length))
            if len(source) > 2 and source[2] is not None:
                vol = source[2]
            if len(source) > 3 and source[3] is not None:
                o = source[3]
        else:
            buf = source
            sl = rl = len(buf[0])
        bufs = [_c32(b) for b in buf]
        sptr = (_f32p * len(bufs))(*[_p(b) for b in bufs])
        vec = None
        if callable(vol):
            vec = np.array([vol(x) for x in range(tl)], dtype=np.float64)
            vol = 1.0
        L.wpo_mix_source(tptr, len(target), tl, sptr, len(bufs), rl, sl, float(vol),
                         vec.ctypes.data_as(_f64p) if vec is not None else None, int(o))
    return target


# ---------------------------------------------------------------- ring buffer / loop ints
class RingBuffer:
    """lib/ring-buffer.js:1-59 index rule over a numpy array."""

    def __init__(self, target):
        self.target = target
        self.needle = 0

    def __getitem__(self, key):
        return self.target[lib().wpo_ring_pos(self.needle, key, len(self.target))]

    def __setitem__(self, key, value):
        n = len(self.target)
        if key == 0:
            pos = self.needle
            self.needle = lib().wpo_ring_advance(self.needle, n)
        else:
            pos = lib().wpo_ring_pos(self.needle, key, n)
        self.target[pos] = value


def beat_rate(sample_rate, bpm):
    """src/loop-node.js:145-147"""
    return lib().wpo_beat_rate(float(sample_rate), float(bpm))


def buffer_size(sample_rate, bpm):
    """src/loop-node.js:48-50"""
    return beat_rate(sample_rate, bpm) * 4


def bpm_from_beat_rate(sample_rate, beat_rate_):
    """src/loop-node.js:9-16 the bpm getter: parseFloat((60 * (sampleRate / beatRate)).toFixed(6)).
    Number.prototype.toFixed rounds the exact binary value to six decimals, ties to the larger
    magnitude; Decimal(float) is exact, so ROUND_HALF_UP reproduces it."""
    from decimal import ROUND_HALF_UP, Decimal
    x = 60.0 * (float(sample_rate) / float(beat_rate_))
    return float(Decimal(x).quantize(Decimal("0.000001"), rounding=ROUND_HALF_UP))
