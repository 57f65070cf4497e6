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
//   u32[0] = 0x57504b31 ('WPK1')   u32[1] = M   u32[2] = L   u32[3] = 0
//   f32[4 .. 4+M) = impulse response samples
import native from '../native.js'
import { DeviceBuffer, isDeviceChannel, liveChannel } from '../src/device-buffer.js'

const MAGIC = 0x57504b31
const HEADER = 4
const handles = new WeakMap() // kernel backing buffer -> device IR handle (per thread)

const isKernel = k => {
  if (!(k instanceof Float32Array) || k.length < HEADER) return false
  return new Uint32Array(k.buffer, k.byteOffset, HEADER)[0] === MAGIC
}

export const kernelInfo = k => {
  if (!isKernel(k)) throw new TypeError('fftKernel must come from fftProcessKernel')
  const u = new Uint32Array(k.buffer, k.byteOffset, HEADER)
  return { M: u[1], L: u[2], ir: k.subarray(HEADER, HEADER + u[1]) }
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

function fftProcessKernel (length, kernel, shared = true) {
  const M = kernel.length
  const N = native.fftLength(length, M) // throws 'FFT size must be a power of two and bigger than 1'
  if (N <= 1) throw new Error('FFT size must be a power of two and bigger than 1')
  const n = Math.max(2 * N, M + HEADER)
  const buffer = shared
    ? new SharedArrayBuffer(n * Float32Array.BYTES_PER_ELEMENT)
    : new ArrayBuffer(n * Float32Array.BYTES_PER_ELEMENT)
  const out = new Float32Array(buffer)
  const u = new Uint32Array(buffer, 0, HEADER)
  u[0] = MAGIC; u[1] = M; u[2] = length; u[3] = 0
  out.set(kernel, HEADER)
  irHandle(out) // build the device spectra now, where the reference does its FFT
  return out
}

function fftConvolution (length, fftKernel, kernelLength) {
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
  // O(L*M) in the reference (:541-551); the FFT route gives the same L+M-1
  // samples to rounding and is what runs here.
  const k = fftProcessKernel(input.length, kernel, false)
  return fftConvolution(input.length, k, kernel.length)(input).slice(0, input.length + kernel.length - 1)
}

export default { directConvolution, fftProcessKernel, fftConvolution }
