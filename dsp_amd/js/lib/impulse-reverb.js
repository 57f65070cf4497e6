// Drop-in for lib/impulse-reverb.js:3-22.  The block convolution and the
// `curr[i] += prev[i]` tail add (:14-16) run on the GPU in one launch sequence;
// the view arithmetic (:17-20) comes back as integers and is applied with
// subarray so that length, byteOffset and backing-buffer size match.
import impulseConvolve from './impulse-convolve.js'
import { irHandle } from './convolve.js'
import native from '../native.js'

export default async (c, { url, offset = 0, length = -1, id = '' }=c) => {
  const reverb = await impulseConvolve(c, url, length)
  const rv = native.reverbCreate(irHandle(reverb.kernel), 1, offset)
  let prev = (await c.get('prev:'+id+url+(c.n-c.buffer[0].length)))||new Float32Array()
  if (prev.length) native.reverbSetTail(rv, [prev])
  return c => {
    const r = native.reverbBlock(rv, [c.buffer[0]])
    const curr = r.curr[0]
    prev = curr.subarray(r.prevOff, r.prevOff + r.prevLen)
    c.set('prev:'+id+url+c.n, prev, 5000)
    return curr.subarray(r.retOff, r.retOff + r.retLen)
  }
}
