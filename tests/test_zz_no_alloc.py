"""wp_convolve_dev_noalloc (block spectra in the library's static device workspace, no
allocation in the call) against wp_convolve_dev, through the C ABI: two calls in a row,
the second with a smaller problem and a shorter input, bit for bit; the oracle bounds the
small one; sizes beyond the workspace are refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_noalloc_convolve_matches_allocating_entry_point(oracle):
    import torch
    from dsp_amd import capi
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    lib = capi.lib()
    capi.init()
    f32p = C.POINTER(C.c_float)

    def make_ir(L, M, seed):
        irs = [oracle.synth_ir(M, seed), oracle.synth_ir(M, seed + 1)]
        h = capi.check_handle(lib.wp_ir_create((f32p * 2)(*[a.ctypes.data_as(f32p) for a in irs]), 2, M, L), "wp_ir_create")
        return h, irs

    def both(h, ch, x, n_in, out_len):
        xd = torch.from_numpy(x).cuda()
        a = torch.full((out_len + 3,), 5.0, dtype=torch.float32, device="cuda")
        b = torch.full((out_len + 3,), 5.0, dtype=torch.float32, device="cuda")
        capi.check(lib.wp_convolve_dev(h, ch, C.c_void_p(xd.data_ptr()), n_in, C.c_void_p(a.data_ptr()), None), "wp_convolve_dev")
        capi.check(lib.wp_convolve_dev_noalloc(h, ch, C.c_void_p(xd.data_ptr()), n_in, C.c_void_p(b.data_ptr()), None),
                   "wp_convolve_dev_noalloc")
        torch.cuda.synchronize()
        assert (b[out_len:] == 5.0).all(), "wrote past L+M"
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "static-workspace result differs from wp_convolve_dev"
        return b[:out_len].cpu().numpy()

    # first call: one bar of 96 000 frames with a 2 s IR (B = 4096, 47 blocks, streaming/gather MAC as the library picks)
    L1, M1 = 96000, 96000
    h1, _ = make_ir(L1, M1, 500000)
    x1 = oracle.synth_noise(L1, 1000)
    y1 = both(h1, 1, x1, L1, L1 + M1)
    assert np.isfinite(y1).all() and float(np.abs(y1).max()) > 1e-3
    # second call, smaller problem (B = 1024) and an input shorter than L: nothing of the first
    # call's spectra may leak into it
    L2, M2 = 5000, 1200
    h2, irs2 = make_ir(L2, M2, 40)
    x2 = oracle.synth_noise(L2, 50)
    n_in = 3100
    y2 = both(h2, 0, x2, n_in, L2 + M2)
    want = oracle.fft_convolution(L2, oracle.fft_process_kernel(L2, irs2[0]), M2)(x2[:n_in])
    assert float(np.abs(y2.astype(np.float64) - want).max()) <= TOL
    # and the first problem again after the small one, other channel
    both(h1, 0, x1, L1, L1 + M1)
    # an empty input gives zeros through the memset path
    y0 = both(h2, 1, x2, 0, L2 + M2)
    assert not y0.any()
    lib.wp_ir_destroy(h1)
    lib.wp_ir_destroy(h2)
    # beyond the workspace: refused with a clear message, nothing launched
    L3, M3 = 600000, 1000
    h3, _ = make_ir(L3, M3, 7)
    xd = torch.zeros(L3, dtype=torch.float32, device="cuda")
    od = torch.zeros(L3 + M3, dtype=torch.float32, device="cuda")
    rc = lib.wp_convolve_dev_noalloc(h3, 0, C.c_void_p(xd.data_ptr()), L3, C.c_void_p(od.data_ptr()), None)
    assert rc == capi.WP_ERR_RANGE and "static workspace" in capi.last_error()
    lib.wp_ir_destroy(h3)
