// Drop-in for lib/convolve.js (exports at :597-599): same names, same argument
// meaning, same Float32Array results; the transforms run on the GPU.
//
// Kernel object.  The reference's fftProcessKernel (:553-562) returns the
// interleaved spectrum, a Float32Array(2N) over a SharedArrayBuffer, which
// callers cache with c.set and hand to other worker threads
// (lib/impulse-convolve.js:9-13).  Here the spectra live on the device as
// partitioned half-size spectra, so the returned array keeps the reference's
// type, length and SharedArrayBuffer backing (it survives structured clone and
// `kernel.length` reads the same) but carries a small header plus the raw
// impulse response; any thread that receives it can (re)build the device state.
//   u32[0] = 0x57504b31 ('WPK1')   u32[1] = M   u32[2] = L
//   f32[3 .. 3+M) = impulse response samples
// (M + 3 <= 2N for every legal size but L = 0, M = 2, where M and L share one word:
//   u32[0] = 0x57504b32 ('WPK2')   u32[1] = M << 16 | L   f32[2 .. 2+M) = samples)
// so the array always has the reference's length 2N.
//
// Reference-layout kernels.  fftProcessKernel(length, kernel, shared, true) returns what the
// reference returns, the interleaved N-point spectrum (computed on the GPU), and
// fftConvolution accepts any Float32Array(2N) without the header as such a spectrum --
// the reference multiplies by whatever array it is given (:575-579) -- by transforming it
// back to the impulse response once, on the GPU.
import native from '../native.js'
import { DeviceBuffer, isDeviceChannel, liveChannel } from '../src/device-buffer.js'

const MAGIC = 0x57504b31
const MAGIC_PACKED = 0x57504b32
const HEADER = 3
const handles = new WeakMap() // kernel backing buffer -> device IR handle (per thread)

const magicOf = k => (k instanceof Float32Array && k.length >= 2) ? new Uint32Array(k.buffer, k.byteOffset, 1)[0] : 0
const isKernel = k => magicOf(k) === MAGIC || magicOf(k) === MAGIC_PACKED

export const kernelInfo = k => {
  if (!isKernel(k)) throw new TypeError('fftKernel must come from fftProcessKernel')
  if (magicOf(k) === MAGIC_PACKED) {
    const w = new Uint32Array(k.buffer, k.byteOffset, 2)[1]
    return { M: w >>> 16, L: w & 0xffff, ir: k.subarray(2, 2 + (w >>> 16)) }
  }
  const u = new Uint32Array(k.buffer, k.byteOffset, HEADER)
  return { M: u[1], L: u[2], ir: k.subarray(HEADER, HEADER + u[1]) }
}

// the header + impulse response written into `out` (a Float32Array of 2N floats)
const packKernel = (out, length, kernel) => {
  const M = kernel.length
  if (M + HEADER <= out.length) {
    const u = new Uint32Array(out.buffer, out.byteOffset, HEADER)
    u[0] = MAGIC; u[1] = M; u[2] = length
    out.set(kernel, HEADER)
  } else {
    const u = new Uint32Array(out.buffer, out.byteOffset, 2)
    u[0] = MAGIC_PACKED; u[1] = (M << 16) | length
    out.set(kernel, 2)
  }
  return out
}

// device IR handle for one or two kernel objects (mono or stereo pair)
export const irHandle = (...kernels) => {
  const first = kernels[0]
  let perBuffer = handles.get(first.buffer)
  if (!perBuffer) handles.set(first.buffer, perBuffer = {})
  const key = kernels.length + ':' + kernels.map(k => k.byteOffset).join(',')
  if (!perBuffer[key]) {
    const infos = kernels.map(kernelInfo)
    perBuffer[key] = native.irCreate(infos.map(i => i.ir), infos[0].L)
  }
  return perBuffer[key]
}

function fftProcessKernel (length, kernel, shared = true, compat = false) {
  const M = kernel.length
  const N = native.fftLength(length, M) // throws 'FFT size must be a power of two and bigger than 1'
  if (N <= 1) throw new Error('FFT size must be a power of two and bigger than 1')
  const n = 2 * N
  const buffer = shared
    ? new SharedArrayBuffer(n * Float32Array.BYTES_PER_ELEMENT)
    : new ArrayBuffer(n * Float32Array.BYTES_PER_ELEMENT)
  const out = new Float32Array(buffer)
  if (compat) return native.kernelSpectrum(kernel instanceof Float32Array ? kernel : Float32Array.from(kernel), length, out)
  packKernel(out, length, kernel)
  irHandle(out) // build the device spectra now, where the reference does its FFT
  return out
}

// a kernel in the reference's layout (no header): one inverse transform on the GPU gives the
// impulse response back; the converted kernel is remembered per array
const converted = new WeakMap()
const fromReferenceLayout = (length, spectrum, kernelLength) => {
  let k = converted.get(spectrum)
  if (k) return k
  if (!(spectrum instanceof Float32Array)) throw new TypeError('fftKernel must be a Float32Array')
  const N = native.fftLength(length, kernelLength)
  if (spectrum.length !== 2 * N) {
    throw new TypeError('a kernel for length=' + length + ', kernelLength=' + kernelLength + ' has ' + 2 * N + ' floats, not ' + spectrum.length)
  }
  const ir = native.kernelFromSpectrum(spectrum, kernelLength)
  k = packKernel(new Float32Array(new SharedArrayBuffer(8 * N)), length, ir)
  converted.set(spectrum, k)
  return k
}

function fftConvolution (length, fftKernel, kernelLength) {
  if (!isKernel(fftKernel)) fftKernel = fromReferenceLayout(length, fftKernel, kernelLength)
  const info = kernelInfo(fftKernel)
  if (info.L !== length || info.M !== kernelLength) {
    throw new Error('fftKernel was prepared for length=' + info.L + ', kernelLength=' + info.M)
  }
  const ir = irHandle(fftKernel)
  function go (input) {
    if (isDeviceChannel(input)) { // input and result stay in device memory
      const out = DeviceBuffer.alloc(1, length + kernelLength)[0]
      native.convolveDev(ir, 0, liveChannel(input), out)
      return out
    }
    return native.convolve(ir, 0, input) // new Float32Array(length + kernelLength)
  }
  go.kernel = fftKernel
  return go
}

function directConvolution (input, kernel) {
  // :541-551, O(L*M): one GPU thread per output sample keeps the reference's chain of
  // f64 multiply, f64 add, f32 store, so the L+M-1 samples are the reference's bit for bit
  const f = a => a instanceof Float32Array ? a : Float32Array.from(a)
  return native.directConvolution(f(input), f(kernel))
}

export default { directConvolution, fftProcessKernel, fftConvolution }
