#!/usr/bin/env python3
"""Host<->device transfer cost of one stereo bar (2 x 96 000 floats each way) by kind of host memory and
call: pageable / page-locked by wp_host_pin / wp_host_alloc, blocking hipMemcpy against hipMemcpyAsync on a
stream + one wait.  Prints one JSON line (microseconds per pair of 384 KB transfers)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # loads the HIP runtime the library binds to
from dsp_amd import capi

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
lib = capi.lib()
capi.init()
N = 96000
BYTES = 4 * N
dev = torch.empty(4 * N, dtype=torch.float32, device="cuda")
dptr = dev.data_ptr()
stream = C.c_void_p()
assert hip.hipStreamCreate(C.byref(stream)) == 0
H2D, D2H = 1, 2


def bench(host_ptrs, use_async, reps=300):
    def once():
        for i, p in enumerate(host_ptrs):
            if use_async:
                hip.hipMemcpyAsync(C.c_void_p(dptr + i * BYTES), C.c_void_p(p), C.c_size_t(BYTES), H2D, stream)
            else:
                hip.hipMemcpy(C.c_void_p(dptr + i * BYTES), C.c_void_p(p), C.c_size_t(BYTES), H2D)
        for i, p in enumerate(host_ptrs):
            if use_async:
                hip.hipMemcpyAsync(C.c_void_p(p), C.c_void_p(dptr + (2 + i) * BYTES), C.c_size_t(BYTES), D2H, stream)
            else:
                hip.hipMemcpy(C.c_void_p(p), C.c_void_p(dptr + (2 + i) * BYTES), C.c_size_t(BYTES), D2H)
        if use_async:
            hip.hipStreamSynchronize(stream)
    for _ in range(20):
        once()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    return (time.perf_counter() - t0) / reps * 1e6


res = {}
pageable = [np.zeros(N, np.float32) for _ in range(2)]
res["pageable_blocking"] = bench([a.ctypes.data for a in pageable], False)
res["pageable_async"] = bench([a.ctypes.data for a in pageable], True)
pinned = [np.zeros(N, np.float32) for _ in range(2)]
rcs = [lib.wp_host_pin(C.c_void_p(a.ctypes.data), C.c_size_t(BYTES)) for a in pinned]
res["wp_host_pin_rc"] = rcs
if not any(rcs):
    res["registered_blocking"] = bench([a.ctypes.data for a in pinned], False)
    res["registered_async"] = bench([a.ctypes.data for a in pinned], True)
    for a in pinned:
        lib.wp_host_unpin(C.c_void_p(a.ctypes.data))
else:
    res["wp_host_pin_error"] = lib.wp_last_error().decode()
lib.wp_host_alloc.restype = C.c_void_p
blocks = [lib.wp_host_alloc(C.c_size_t(BYTES)) for _ in range(2)]
res["wp_host_alloc_ok"] = all(blocks)
if all(blocks):
    res["hostalloc_blocking"] = bench(blocks, False)
    res["hostalloc_async"] = bench(blocks, True)
t0 = time.perf_counter()
for _ in range(20):
    a = np.zeros(N, np.float32)
    lib.wp_host_pin(C.c_void_p(a.ctypes.data), C.c_size_t(BYTES))
    lib.wp_host_unpin(C.c_void_p(a.ctypes.data))
res["pin_unpin_384KB_us"] = (time.perf_counter() - t0) / 20 * 1e6
print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}))
