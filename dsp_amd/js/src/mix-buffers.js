// Drop-in for src/mix-buffers.js:1-73: same call shape (plain sources and the
// [buffer, length, volume, offset] tuple form, constant or per-frame-function
// volume), target updated in place and returned.  The decode below is the
// reference's (:5-14); the summation runs on the GPU with the reference's
// arithmetic (f64 multiply and add, f32 store, sources in order), so results
// are bit-identical.
import native from '../native.js'

const def = (v, d) => (v !== undefined && v !== null) ? v : d

export default (target, ...sources) => {
  const tl = target[0].length
  const mixable = target.length === 1 || target.length === 2
  const descs = sources.map(source => {
    let sl, rl, _vol = 1, o = 0
    if (Array.isArray(source[0])) { // [buffer,length,volume,offset]
      sl = (source[0][0].length * source[1])|0 // specified length
      _vol = def(source[2], 1)
      o = def(source[3], 0) // offset
      source = source[0] // actual buffer
      rl = source[0].length // real length
    } else {
      sl = rl = source[0].length
    }
    let vol = 1, volVec
    const used = mixable && (source.length === 1 || source.length === 2)
    if (typeof _vol === 'function') {
      if (used) { // the reference calls _vol(x) once per frame, in order (:21)
        volVec = new Float64Array(tl)
        for (let x = 0; x < tl; x++) volVec[x] = _vol(x)
      }
    } else {
      vol = +_vol
    }
    o = +o
    if (!Number.isInteger(o)) { // a fractional or NaN index reads undefined: every sample NaN
      sl = 0
      o = 0
    }
    return { ch: Array.from(source), rl, sl, vol, volVec, o }
  })
  native.mix(target, descs)
  return target
}
