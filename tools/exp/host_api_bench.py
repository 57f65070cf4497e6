#!/usr/bin/env python3
"""wp_reverb_block (host-pointer entry point) per bar by kind of host memory; hipHostRegister probes."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dsp_amd import capi, device

lib = capi.lib()
capi.init()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
L = M = 96000
f32p = C.POINTER(C.c_float)
irs = [device.synth_ir(M, 500000), device.synth_ir(M, 500001)]
h = capi.check_handle(lib.wp_ir_create((f32p * 2)(*[a.ctypes.data_as(f32p) for a in irs]), 2, M, L), "wp_ir_create")
job_ch = (C.c_int32 * 2)(0, 1)
rv = capi.check_handle(lib.wp_reverb_create(h, 2, job_ch, 0), "wp_reverb_create")
lib.wp_host_alloc.restype = C.c_void_p


def as_array(ptr, n):
    return np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), f32p), shape=(n,))


def bars(ins, outs, reps=200):
    v = capi.WpReverbView()
    ip = (f32p * 2)(*[C.cast(C.c_void_p(p), f32p) for p in ins])
    op = (f32p * 2)(*[C.cast(C.c_void_p(p), f32p) for p in outs])
    for _ in range(10):
        capi.check(lib.wp_reverb_block(rv, ip, L, op, C.byref(v)), "wp_reverb_block")
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.wp_reverb_block(rv, ip, L, op, C.byref(v))
    return (time.perf_counter() - t0) / reps * 1e6


res = {}
pin = [np.random.rand(L).astype(np.float32) for _ in range(2)]
pout = [np.zeros(L + M, np.float32) for _ in range(2)]
res["pageable_in_pageable_out"] = bars([a.ctypes.data for a in pin], [a.ctypes.data for a in pout])
hin = [lib.wp_host_alloc(C.c_size_t(4 * L)) for _ in range(2)]
hout = [lib.wp_host_alloc(C.c_size_t(4 * (L + M))) for _ in range(2)]
for p, a in zip(hin, pin):
    as_array(p, L)[:] = a
res["pinned_in_pinned_out"] = bars(hin, hout)
res["pageable_in_pinned_out"] = bars([a.ctypes.data for a in pin], hout)
res["pinned_in_pageable_out"] = bars(hin, [a.ctypes.data for a in pout])
# hipHostRegister probes: whole pages, flags
buf = np.zeros(L + 4096, np.float32)
base = (buf.ctypes.data + 4095) & ~4095
for name, flags in (("default", 0), ("portable", 1), ("mapped", 2)):
    rc = hip.hipHostRegister(C.c_void_p(base), C.c_size_t(90 * 4096), C.c_uint(flags))
    res["hipHostRegister_aligned_" + name] = rc
    if rc == 0:
        hip.hipHostUnregister(C.c_void_p(base))
    else:
        hip.hipGetLastError()
import mmap
mm = mmap.mmap(-1, 96 * 4096)
mbase = C.addressof(C.c_char.from_buffer(mm))
rc = hip.hipHostRegister(C.c_void_p(mbase), C.c_size_t(96 * 4096), C.c_uint(0))
res["hipHostRegister_mmap"] = rc
print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}))
