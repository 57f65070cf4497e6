"""Host-side mirror of lib/convolve.js:597-599 over the C ABI.

  fft_process_kernel(length, kernel)            lib/convolve.js:553-562
  fft_convolution(length, fft_kernel, kernel_length) -> go(input)   :564-586
  direct_convolution(input, kernel)             :541-551

`fft_process_kernel` returns a ``KernelSpectrum``: the reference returns the
2N-float interleaved spectrum; here the spectra live on the GPU as partitioned
half-spectra, so the object carries the raw IR samples (what a cache needs in
order to rebuild the device state elsewhere) and reports the length the
reference's array would have.  With ``compat=True`` it returns the reference's own
array instead (2N float32, the interleaved N-point spectrum), and
`fft_convolution` accepts such an array -- from here or from the reference --
by transforming it back to the impulse response on the GPU.
"""
import ctypes as C

import numpy as np

from . import capi


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(capi.f32p)


class IR:
    """Owner of a ``wp_ir`` handle (device-resident partition spectra)."""

    def __init__(self, channels, length):
        chans = [_c32(c) for c in channels]
        M = len(chans[0])
        for c in chans:
            if len(c) != M:
                raise ValueError("IR channels must have equal length")
        ptrs = (capi.f32p * len(chans))(*[_fp(c) for c in chans])
        L = capi.lib()
        capi.init()
        self.handle = capi.check_handle(L.wp_ir_create(ptrs, len(chans), M, int(length)), "wp_ir_create")
        self.M, self.L, self.nch = M, int(length), len(chans)

    def info(self, which):
        return capi.lib().wp_ir_info(self.handle, which)

    @property
    def partition_size(self):
        return self.info(3)

    @property
    def partitions(self):
        return self.info(4)

    def close(self):
        if getattr(self, "handle", None):
            capi.lib().wp_ir_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KernelSpectrum:
    """What fftProcessKernel hands back (lib/convolve.js:553-562)."""

    def __init__(self, length, kernel):
        self.ir_samples = _c32(kernel).copy()
        self.block_length = int(length)
        n = capi.check(capi.lib().wp_fft_length(self.block_length, len(self.ir_samples)), "fftProcessKernel")
        if n <= 1:
            raise ValueError("FFT size must be a power of two and bigger than 1")
        self.length = 2 * n  # floats in the reference's interleaved spectrum
        self._ir = None

    def __len__(self):
        return self.length

    def ir(self):
        if self._ir is None:
            self._ir = IR([self.ir_samples], self.block_length)
        return self._ir


def fft_process_kernel(length, kernel, compat=False):
    k = KernelSpectrum(length, kernel)
    if compat:  # the reference's layout: interleaved spectrum of the kernel zero-padded to N points
        h = _c32(kernel)
        out = np.empty(k.length, dtype=np.float32)
        capi.check(capi.lib().wp_kernel_spectrum(_fp(h), len(h), int(length), _fp(out)), "fftProcessKernel")
        return out
    k.ir()  # build the device spectra now, like the reference does its FFT here
    return k


def _kernel_from_reference_spectrum(length, spectrum, kernel_length):
    """A Float32Array(2N) in the reference's layout (lib/convolve.js:575-579 multiplies by whatever
    array of that length it is given) -> KernelSpectrum, through one inverse transform on the GPU."""
    spec = _c32(spectrum)
    n = capi.check(capi.lib().wp_fft_length(int(length), int(kernel_length)), "fftConvolution")
    if len(spec) != 2 * n:
        raise ValueError("a reference-layout kernel for length=%d, kernel_length=%d has %d floats, not %d"
                         % (length, kernel_length, 2 * n, len(spec)))
    h = np.empty(int(kernel_length), dtype=np.float32)
    capi.check(capi.lib().wp_kernel_from_spectrum(_fp(spec), n, int(kernel_length), _fp(h)), "fftConvolution")
    return KernelSpectrum(length, h)


def fft_convolution(length, fft_kernel, kernel_length):
    """-> go(input): Float32 array of exactly length + kernel_length samples."""
    if isinstance(fft_kernel, np.ndarray):
        fft_kernel = _kernel_from_reference_spectrum(length, fft_kernel, kernel_length)
    if not isinstance(fft_kernel, KernelSpectrum):
        raise TypeError("fft_kernel must come from fft_process_kernel or be a reference-layout spectrum")
    if fft_kernel.block_length != int(length) or len(fft_kernel.ir_samples) != int(kernel_length):
        raise ValueError("fft_kernel was prepared for length=%d kernel_length=%d" %
                         (fft_kernel.block_length, len(fft_kernel.ir_samples)))
    ir = fft_kernel.ir()
    out_len = int(length) + int(kernel_length)

    def go(x):
        x = _c32(x)
        out = np.empty(out_len, dtype=np.float32)
        capi.check(capi.lib().wp_convolve(ir.handle, 0, _fp(x), len(x), _fp(out)), "go")
        return out

    go.ir = ir  # the device handle, for callers that build a reverb on the same IR
    return go


def direct_convolution(x, h):
    """lib/convolve.js:541-551: n+m-1 samples, bit-identical to the reference (the GPU
    kernel keeps the reference's accumulation chain per output sample)."""
    x, h = _c32(x), _c32(h)
    out = np.zeros(max(len(x) + len(h) - 1, 0), dtype=np.float32)
    capi.check(capi.lib().wp_direct_convolution(_fp(x), len(x), _fp(h), len(h), _fp(out)), "directConvolution")
    return out
