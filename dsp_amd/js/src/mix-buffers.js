// Drop-in for src/mix-buffers.js:1-73: same call shape (plain sources and the
// [buffer, length, volume, offset] tuple form, constant or per-frame-function
// volume), target updated in place and returned.  Each source is decoded to the
// numbers the reference derives (:5-14: sl, rl, volume, offset) and handed to
// the GPU mix, which applies the reference's arithmetic (f64 multiply and add,
// f32 store, sources in order), so results are bit-identical.
import native from '../native.js'
import { DeviceBuffer, isDeviceChannel, liveChannel } from './device-buffer.js'

const present = v => v !== undefined && v !== null
const layoutOk = chans => chans.length === 1 || chans.length === 2

function describe (source, frames, targetOk) {
  const tuple = Array.isArray(source[0])
  const chans = tuple ? source[0] : source
  const rl = chans[0].length
  const d = { ch: Array.from(chans), rl, sl: rl, vol: 1, volVec: undefined, o: 0 }
  if (tuple) {
    d.sl = (rl * source[1]) | 0
    const gain = present(source[2]) ? source[2] : 1
    if (typeof gain === 'function') {
      // the reference evaluates the callback once per frame, in order (:21), and
      // only for layouts it mixes at all
      if (targetOk && layoutOk(chans)) {
        d.volVec = Float64Array.from({ length: frames }, (_, x) => gain(x))
      }
    } else {
      d.vol = +gain
    }
    const off = present(source[3]) ? +source[3] : 0
    if (Number.isInteger(off)) d.o = off
    else d.sl = 0 // a fractional or NaN index reads `undefined`: every sample becomes NaN
  }
  return d
}

// channels of one source as live device channels (host arrays are uploaded: the
// worker's copy-back, src/mix-worker-thread.js:23-24, becomes this one transfer)
const onDevice = chans => isDeviceChannel(chans[0])
  ? chans.map(liveChannel)
  : Array.from(DeviceBuffer.upload(chans))

export default (target, ...sources) => {
  const frames = target[0].length
  const targetOk = layoutOk(target)
  const decoded = sources.map(s => describe(s, frames, targetOk))
  if (isDeviceChannel(target[0])) {
    // device-resident bus: nothing crosses PCIe except host sources on their way in
    if (targetOk) {
      const mixable = decoded.filter(d => layoutOk(d.ch))
      for (const d of mixable) d.ch = onDevice(d.ch)
      native.mixDev(Array.from(target, liveChannel), mixable)
    }
  } else {
    native.mix(target, decoded)
  }
  return target
}
