// Channel-count-generic implementation behind the four drop-in wrapper modules
// (impulse-convolve.js, impulse-convolve-stereo.js, impulse-reverb.js,
// impulse-reverb-stereo.js).  Behaviour follows the reference wrappers:
//   lib/impulse-convolve.js:3-20, lib/impulse-convolve-stereo.js:3-27  (IR fetch,
//     optional truncation that mutates the sample array, kernel cache protocol)
//   lib/impulse-reverb.js:3-22, lib/impulse-reverb-stereo.js:3-31      (block
//     convolution, carried tail, 5000 ms tail ttl, returned subarray views)
// including their quirks: the stereo pair is built with impulse[0].length for
// both channels, the stereo tail is ONE two-element array mutated in place, the
// mono tail is a fresh view per block.
import convolve, { irHandle } from './convolve.js'
import native from '../native.js'
import { DeviceBuffer, isDeviceChannel, liveChannel, keepAlive } from '../src/device-buffer.js'

const KEYS = {
  1: { kernel: 'kernel:', prev: 'prev:' },
  2: { kernel: 'impulse-convolve-stereo:kernel:', prev: 'impulse-reverb-stereo:prev:' }
}

// -> go functions (one per channel), each carrying its kernel object as go.kernel
async function convolvers (c, url, length, nch) {
  const impulse = await c.sample(url)
  if (length > -1) {
    for (let ch = 0; ch < nch; ch++) impulse[ch] = impulse[ch].subarray(0, length)
  }
  const blockLength = c.buffer[0].length
  const key = KEYS[nch].kernel + url + ':' + blockLength + ':' + length
  let cached = await c.get(key)
  if (cached === false) {
    const made = []
    for (let ch = 0; ch < nch; ch++) made.push(convolve.fftProcessKernel(blockLength, impulse[ch]))
    cached = nch === 1 ? made[0] : made
    await c.set(key, cached)
  }
  const kernels = nch === 1 ? [cached] : cached
  return kernels.map(k => convolve.fftConvolution(blockLength, k, impulse[0].length))
}

export const impulseConvolve = async (c, url, length) => (await convolvers(c, url, length, 1))[0]
export const impulseConvolveStereo = (c, url, length) => convolvers(c, url, length, 2)

// When c.buffer is a DeviceBuffer the block, its result, the carried tail and the
// returned views all stay in device memory (same view shapes, same cache keys).
async function reverb (c, params, nch) {
  const url = params.url
  const offset = params.offset === undefined ? 0 : params.offset
  const length = params.length === undefined ? -1 : params.length
  const id = params.id === undefined ? '' : params.id
  const gos = await convolvers(c, url, length, nch)
  // block convolution + `curr[i] += prev[i]` run on the GPU in one launch sequence
  const ir = irHandle(...gos.map(g => g.kernel))
  const state = native.reverbCreate(ir, nch, offset)
  const resultLength = native.irInfo(ir, 5) // L + M
  const prevKey = n => KEYS[nch].prev + id + url + n
  const stored = await c.get(prevKey(c.n - c.buffer[0].length))
  const empty = () => new Float32Array()
  let prev = stored || (nch === 1 ? empty() : [empty(), empty()])
  const tails = nch === 1 ? [prev] : [prev[0], prev[1]]
  if (tails[0].length) {
    if (isDeviceChannel(tails[0])) native.reverbSetTailDev(state, tails.map(liveChannel))
    else native.reverbSetTail(state, tails)
  }
  const view = (a, off, len) => a.subarray(off, off + len)
  return c => {
    let curr, r
    if (isDeviceChannel(c.buffer[0])) {
      curr = DeviceBuffer.alloc(nch, resultLength)
      r = native.reverbBlockTo(state, c.buffer.slice(0, nch).map(liveChannel), curr)
    } else {
      r = native.reverbBlock(state, c.buffer.slice(0, nch))
      curr = r.curr
    }
    if (nch === 1) {
      prev = view(curr[0], r.prevOff, r.prevLen)
    } else {
      prev[0] = view(curr[0], r.prevOff, r.prevLen)
      prev[1] = view(curr[1], r.prevOff, r.prevLen)
    }
    c.set(prevKey(c.n), prev, 5000)
    if (isDeviceChannel(curr[0])) keepAlive(state, Array.from(curr), 5000)
    const out = Array.from(curr, a => view(a, r.retOff, r.retLen))
    return nch === 1 ? out[0] : out
  }
}

export const impulseReverb = (c, params = c) => reverb(c, params, 1)
export const impulseReverbStereo = (c, params = c) => reverb(c, params, 2)
