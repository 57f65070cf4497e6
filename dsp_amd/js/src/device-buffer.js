// Device-resident counterparts of the reference's sample buffers.
//
// In the reference a buffer is an Array of Float32Array channels over
// SharedArrayBuffers (src/buffer-service.js:8-27); workers render into their own
// copy and copy back with `buffer.set(...)` (src/mix-worker-thread.js:14-16,23-24),
// Context.zero fills with 0 (src/context.js:140-143) and render.js duplicates
// mono into stereo with `buffer[1].set(buffer[0])` (src/render.js:45).
//
// DeviceBuffer keeps that shape -- it IS an Array whose elements have `.length`,
// `.subarray`, `.set`, `.fill`, `.slice` -- but the samples live in HBM.  Passing one
// where the drop-in modules expect a buffer (c.buffer, mixBuffers targets and
// sources, go(input)) keeps a whole bar on the GPU: the copy-back becomes one
// upload and nothing comes back until `.toFloat32Array()`.
//
// A buffer is visible to every thread of the process: after structured clone
// (postMessage) a DeviceBuffer arrives as a plain array of plain objects;
// `DeviceBuffer.revive(value)` turns it back into live channels.
import native from '../native.js'

const clampIndex = (i, n) => {
  i = Math.trunc(+i) || 0 // NaN -> 0, like ToIntegerOrInfinity
  if (i < 0) return Math.max(n + i, 0)
  return Math.min(i, n)
}

// key of the native handle: a symbol-keyed property is an ordinary fast property for the
// constructor, and a structured clone (which copies own string-keyed properties only)
// carries just the numbers
const HANDLE = Symbol('wpdsp handle')

export class DeviceChannel {
  constructor (h, wpbuf, ch, off, length) {
    this[HANDLE] = h // native handle (keeps the device memory alive)
    this.wpbuf = wpbuf // process-wide buffer id
    this.ch = ch
    this.off = off
    this.length = length
  }

  get h () { return this[HANDLE] }

  get byteOffset () { return this.off * 4 }
  get byteLength () { return this.length * 4 }

  // TypedArray.prototype.subarray: relative indices, clamped, shares memory
  subarray (begin = 0, end = this.length) {
    const b = clampIndex(begin, this.length)
    const e = clampIndex(end, this.length)
    return new DeviceChannel(this.h, this.wpbuf, this.ch, this.off + b, Math.max(e - b, 0))
  }

  // TypedArray.prototype.set: host array -> upload, device channel -> device copy
  set (source, offset = 0) {
    offset = Math.trunc(+offset) || 0
    if (offset < 0 || offset + source.length > this.length) throw new RangeError('offset is out of bounds')
    if (isDeviceChannel(source)) {
      native.bufCopy(this, liveChannel(source), offset)
    } else {
      const f32 = source instanceof Float32Array ? source : Float32Array.from(source)
      native.bufWrite(this.h, this.ch, this.off + offset, f32)
    }
  }

  fill (value, start = 0, end = this.length) {
    native.bufFill(this, +value, clampIndex(start, this.length), clampIndex(end, this.length))
    return this
  }

  // TypedArray.prototype.slice: a copy in fresh device memory
  slice (begin = 0, end = this.length) {
    const view = this.subarray(begin, end)
    const copy = DeviceBuffer.alloc(1, view.length)[0]
    if (view.length) native.bufCopy(copy, view, 0)
    return copy
  }

  toFloat32Array () {
    return native.bufRead(this.h, this.ch, this.off, this.length)
  }
}

export const isDeviceChannel = x =>
  x !== null && typeof x === 'object' && typeof x.wpbuf === 'number' && typeof x.ch === 'number'

// a DeviceChannel for `x`, which may be a structured clone of one (no prototype, no
// handle): the clone's id is opened again, which takes a reference of its own that
// the garbage collector gives back with the returned channel
export const liveChannel = x => {
  if (x instanceof DeviceChannel) return x
  if (!isDeviceChannel(x)) throw new TypeError('expected a device channel')
  return new DeviceChannel(native.bufOpen(x.wpbuf), x.wpbuf, x.ch, x.off, x.length)
}

export class DeviceBuffer extends Array {
  // results of map/slice/filter are plain arrays, as they are for the reference's buffers
  static get [Symbol.species] () { return Array }

  // channels x frames of zeros in device memory
  static alloc (channels, frames) {
    const h = native.bufCreate(channels, frames)
    const id = native.bufInfo(h, -1)
    const b = new DeviceBuffer()
    for (let c = 0; c < channels; c++) b.push(new DeviceChannel(h, id, c, 0, frames))
    return b
  }

  // upload an array of Float32Array channels
  static upload (channels) {
    const b = DeviceBuffer.alloc(channels.length, channels[0].length)
    channels.forEach((a, c) => b[c].set(a))
    return b
  }

  // live buffer from a structured clone (or from anything already live)
  static revive (value) {
    const b = new DeviceBuffer()
    let h = null
    for (const c of value) {
      if (c instanceof DeviceChannel) { b.push(c); continue }
      // channels of one buffer share one opened handle
      if (h === null || h.id !== c.wpbuf) h = { id: c.wpbuf, handle: liveChannel(c).h }
      b.push(new DeviceChannel(h.handle, c.wpbuf, c.ch, c.off, c.length))
    }
    return b
  }

  download () {
    return Array.prototype.map.call(this, c => c.toFloat32Array())
  }
}

// Hold a reference to the memory behind `channels` for `ttl` milliseconds: a value
// handed to a cache that stores clones (c.set(id, value, ttl), src/global-service.js:15-19)
// must stay openable for as long as the cache may return it.  `owner` groups the
// entries of one producer, of which only the newest `depth` are held (a block loop
// running faster than real time must not pin one buffer per block for the whole ttl).
// Expired entries are dropped on the next call; no timer keeps the event loop alive.
const kept = []
export const keepAlive = (owner, channels, ttl, depth = 2) => {
  const now = Date.now()
  let mine = 0
  for (let i = kept.length - 1; i >= 0; i--) {
    const k = kept[i]
    if (k.until < now || (k.owner === owner && ++mine >= depth)) kept.splice(i, 1)
  }
  kept.push({ owner, until: now + ttl, handles: channels.map(c => c.h) })
}

export const isDeviceBuffer = b => Array.isArray(b) && b.length > 0 && isDeviceChannel(b[0])

export default DeviceBuffer
