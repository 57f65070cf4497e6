"""Host-side mirror of src/mix-buffers.js:1-73 over the C ABI (wp_mix).

``mix_buffers(target, *sources)``: target is a list of 1-2 float32 arrays,
updated in place and returned.  A source is a list of 1-2 arrays, or the tuple
form ``(buffer, length, volume, offset)`` whose first element is itself a list
(the reference's ``Array.isArray(source[0])`` test, :6); ``volume`` may be a
callable ``x -> gain`` which is evaluated on the host into a per-frame gain
vector (:21) — the summation itself always runs on the GPU.
"""
import ctypes as C

import numpy as np

from . import capi


def decode_source(source, tl):
    """src/mix-buffers.js:5-14 -> (channels, rl, sl, vol, vol_vec, o)"""
    vol, o, vec = 1.0, 0, None
    if isinstance(source[0], (list, tuple)):
        buf = source[0]
        rl = len(buf[0])
        length = source[1] if len(source) > 1 and source[1] is not None else float("nan")
        sl = capi.lib().wp_mix_sl(rl, float(length))
        if len(source) > 2 and source[2] is not None:
            vol = source[2]
        if len(source) > 3 and source[3] is not None:
            o = source[3]
    else:
        buf = source
        sl = rl = len(buf[0])
    if callable(vol):
        vec = np.array([vol(x) for x in range(tl)], dtype=np.float64)
        vol = 1.0
    return buf, rl, sl, float(vol), vec, int(o)


def mix_buffers(target, *sources):
    L = capi.lib()
    capi.init()
    tl = len(target[0])
    for t in target:
        if t.dtype != np.float32 or not t.flags["C_CONTIGUOUS"]:
            raise TypeError("target channels must be contiguous float32 arrays")
        if len(t) < tl:
            raise TypeError("target channels must all hold len(target[0]) frames")
    keep = []
    srcs = (capi.WpSrc * max(len(sources), 1))()
    for i, source in enumerate(sources):
        buf, rl, sl, vol, vec, o = decode_source(source, tl)
        chans = [np.ascontiguousarray(b, dtype=np.float32) for b in buf]
        if len(chans) in (1, 2) and any(len(c) < rl for c in chans):
            # rl frames are read from every channel (NaN past a shorter one in the reference)
            raise TypeError("source channels must all hold len(source[0]) frames")
        keep.append((chans, vec))
        s = srcs[i]
        s.nch = len(chans)
        for c in range(min(len(chans), 2)):
            s.ch[c] = chans[c].ctypes.data
        s.rl, s.sl, s.vol, s.o = rl, sl, vol, o
        s.vol_vec = vec.ctypes.data if vec is not None else None
    tp = (capi.f32p * len(target))(*[t.ctypes.data_as(capi.f32p) for t in target])
    capi.check(L.wp_mix(tp, len(target), tl, srcs, len(sources)), "mixBuffers")
    return target
