// Drop-in for lib/impulse-reverb-stereo.js:3-31: both channels in one launch
// sequence, tail carried on the device and mirrored in the returned arrays.
import impulseConvolve from './impulse-convolve-stereo.js'
import { irHandle } from './convolve.js'
import native from '../native.js'

export default async (c, { url, offset = 0, length = -1, id = '' }=c) => {
  const reverb = await impulseConvolve(c, url, length)
  const rv = native.reverbCreate(irHandle(reverb[0].kernel, reverb[1].kernel), 2, offset)
  let prev = (await c.get('impulse-reverb-stereo:prev:'+id+url+(c.n-c.buffer[0].length)))
    ||[new Float32Array(),new Float32Array()]
  if (prev[0].length) native.reverbSetTail(rv, [prev[0], prev[1]])
  return c => {
    const r = native.reverbBlock(rv, [c.buffer[0], c.buffer[1]])
    prev[0] = r.curr[0].subarray(r.prevOff, r.prevOff + r.prevLen)
    prev[1] = r.curr[1].subarray(r.prevOff, r.prevOff + r.prevLen)
    c.set('impulse-reverb-stereo:prev:'+id+url+c.n, prev, 5000)
    return [
      r.curr[0].subarray(r.retOff, r.retOff + r.retLen),
      r.curr[1].subarray(r.retOff, r.retOff + r.retLen),
    ]
  }
}
