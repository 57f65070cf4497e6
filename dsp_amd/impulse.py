"""Host-side mirror of the four impulse wrappers over the C ABI:

  impulse_convolve(c, url, length)          lib/impulse-convolve.js:3-20
  impulse_convolve_stereo(c, url, length)   lib/impulse-convolve-stereo.js:3-27
  impulse_reverb(c, url, offset, length, id)          lib/impulse-reverb.js:3-22
  impulse_reverb_stereo(c, url, offset, length, id)   lib/impulse-reverb-stereo.js:3-31

`c` is anything with the surface of src/context.js the reference uses:
``c.buffer`` (list of 1-2 float32 arrays), ``c.n``, ``c.sample(url)``,
``c.get(key)`` (value or False), ``c.set(key, value, ttl=None)``.  Cache keys,
hit/miss protocol, the 5000 ms tail ttl and the shapes of every returned view
are the reference's.
"""
import ctypes as C

import numpy as np

from . import capi
from .convolve import IR, KernelSpectrum, _c32, _fp, fft_convolution


def _js_subarray(a, begin=0, end=None):
    n = len(a)
    b = int(begin)
    b = max(n + b, 0) if b < 0 else min(b, n)
    e = n if end is None else int(end)
    e = max(n + e, 0) if e < 0 else min(e, n)
    return a[b:max(b, e)]


def _num(x):
    return str(int(x)) if float(x).is_integer() else repr(float(x))


def impulse_convolve(c, url, length):
    impulse = c.sample(url)
    if length > -1:
        impulse[0] = _js_subarray(impulse[0], 0, length)
    L = len(c.buffer[0])
    key = "kernel:%s:%d:%d" % (url, L, length)
    kernel = c.get(key)
    if kernel is False:
        kernel = KernelSpectrum(L, impulse[0])
        kernel.ir()
        c.set(key, kernel)
    return fft_convolution(L, kernel, len(impulse[0]))


class _StereoKernel(list):
    """The 2-array impulse-convolve-stereo.js:13-16 caches, plus the shared
    two-channel device handle."""
    _ir = None

    def ir(self):
        if self._ir is None:
            self._ir = IR([self[0].ir_samples, self[1].ir_samples], self[0].block_length)
        return self._ir


def impulse_convolve_stereo(c, url, length):
    impulse = c.sample(url)
    if length > -1:
        impulse[0] = _js_subarray(impulse[0], 0, length)
        impulse[1] = _js_subarray(impulse[1], 0, length)
    L = len(c.buffer[0])
    key = "impulse-convolve-stereo:kernel:%s:%d:%d" % (url, L, length)
    kernel = c.get(key)
    if kernel is False:
        kernel = _StereoKernel([KernelSpectrum(L, impulse[0]), KernelSpectrum(L, impulse[1])])
        kernel.ir()
        c.set(key, kernel)
    ir = kernel.ir()
    out_len = L + len(impulse[0])  # both channels use impulse[0].length (:23-24)

    def make(ch):
        def go(x):
            x = _c32(x)
            out = np.empty(out_len, dtype=np.float32)
            capi.check(capi.lib().wp_convolve(ir.handle, ch, _fp(x), len(x), _fp(out)), "go")
            return out
        return go

    fns = [make(0), make(1)]
    fns_ir = ir
    fns[0].ir = fns[1].ir = fns_ir
    return fns


class _Reverb:
    """wp_reverb handle: block convolution + carried tail on the device."""

    def __init__(self, ir, nch, offset):
        self.ir = ir
        self.nch = nch
        job_ch = (C.c_int32 * nch)(*range(nch))
        self.handle = capi.check_handle(capi.lib().wp_reverb_create(ir.handle, nch, job_ch, int(offset)),
                                        "wp_reverb_create")
        self.curr_len = ir.L + ir.M

    def set_tail(self, prev):
        prev = [_c32(p) for p in prev]
        ptrs = (capi.f32p * self.nch)(*[_fp(p) for p in prev])
        capi.check(capi.lib().wp_reverb_set_tail(self.handle, ptrs, len(prev[0])), "wp_reverb_set_tail")

    def block(self, bufs):
        bufs = [_c32(b) for b in bufs]
        outs = [np.empty(self.curr_len, dtype=np.float32) for _ in range(self.nch)]
        ip = (capi.f32p * self.nch)(*[_fp(b) for b in bufs])
        op = (capi.f32p * self.nch)(*[_fp(o) for o in outs])
        v = capi.WpReverbView()
        capi.check(capi.lib().wp_reverb_block(self.handle, ip, len(bufs[0]), op, C.byref(v)), "wp_reverb_block")
        return outs, v

    def __del__(self):
        try:
            if self.handle:
                capi.lib().wp_reverb_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def impulse_reverb(c, url, offset=0, length=-1, id=""):
    go = impulse_convolve(c, url, length)
    rv = _Reverb(go.ir, 1, offset)  # shares the IR handle the convolver built
    got = c.get("prev:" + id + url + _num(c.n - len(c.buffer[0])))
    if got is not False and len(got):
        rv.set_tail([got])

    def block(c):
        outs, v = rv.block([c.buffer[0]])
        curr = outs[0]
        prev = curr[v.prev_off:v.prev_off + v.prev_len]
        c.set("prev:" + id + url + _num(c.n), prev, 5000)
        return curr[v.ret_off:v.ret_off + v.ret_len]

    return block


def impulse_reverb_stereo(c, url, offset=0, length=-1, id=""):
    gos = impulse_convolve_stereo(c, url, length)
    rv = _Reverb(gos[0].ir, 2, offset)
    got = c.get("impulse-reverb-stereo:prev:" + id + url + _num(c.n - len(c.buffer[0])))
    prev = got if got is not False else [np.zeros(0, np.float32), np.zeros(0, np.float32)]
    if got is not False and len(got[0]):
        rv.set_tail(got)

    def block(c):
        outs, v = rv.block([c.buffer[0], c.buffer[1]])
        prev[0] = outs[0][v.prev_off:v.prev_off + v.prev_len]
        prev[1] = outs[1][v.prev_off:v.prev_off + v.prev_len]
        c.set("impulse-reverb-stereo:prev:" + id + url + _num(c.n), prev, 5000)
        return [outs[0][v.ret_off:v.ret_off + v.ret_len], outs[1][v.ret_off:v.ret_off + v.ret_len]]

    return block
