"""dsp_amd — MI355X (gfx950) implementation of wavepot/dsp's convolution-reverb
and mix-buffers hot path.

The product is ``lib/libwpdsp.so`` (hand-written HIP kernels behind the C ABI in
``include/wpdsp.h``).  JavaScript callers reach it through the N-API addon and
the shims in ``dsp_amd/js``; this Python package is the same thin layer for
Python callers, tests and the benchmark:

* ``dsp_amd.capi``      ctypes binding of every C-ABI entry point
* ``dsp_amd.convolve``  fft_process_kernel / fft_convolution / direct_convolution
* ``dsp_amd.impulse``   impulse_convolve[_stereo], impulse_reverb[_stereo]
* ``dsp_amd.mix``       mix_buffers
* ``dsp_amd.device``    device-resident batched reverb + mix for torch tensors,
                        sharded over ranks with an RCCL reduce of the master bus

There is no CPU fallback: without the built library or without a GPU the
compute entry points raise.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
