// Sample ingestion outside the browser.  The reference's sample service fetches a
// URL, decodes it with the browser's AudioContext.decodeAudioData and keeps the
// channels as shared Float32Arrays in a Map keyed by URL
// (src/sample-service.js:12-37); every impulse response reaches the convolution
// through it (`c.sample(url)`, lib/impulse-convolve.js:4).  Node has no
// decodeAudioData, so this module reads RIFF/WAVE files: PCM 8/16/24/32-bit and
// IEEE float 32/64-bit, any channel count, WAVE_FORMAT_EXTENSIBLE included.
//
// Integer PCM is scaled the way the Web Audio specification scales it for
// decodeAudioData (divide by 2^(bits-1); 8-bit is unsigned with a 128 bias).  No
// resampling is done: the file's rate is returned with the channels.  The
// reference holds no audio fixture, so the scaling is checked against
// hand-computed vectors only (tests/node/test_sample_file.mjs).
import fs from 'fs'

const shared = n => new Float32Array(new SharedArrayBuffer(n * Float32Array.BYTES_PER_ELEMENT))
const tag = (view, at) => String.fromCharCode(view.getUint8(at), view.getUint8(at + 1), view.getUint8(at + 2), view.getUint8(at + 3))

// ArrayBuffer | Buffer | Uint8Array -> { sampleRate, channels: [Float32Array (shared), ...] }
export function decodeWav (bytes) {
  const u8 = bytes instanceof ArrayBuffer ? new Uint8Array(bytes) : new Uint8Array(bytes.buffer, bytes.byteOffset, bytes.byteLength)
  const view = new DataView(u8.buffer, u8.byteOffset, u8.byteLength)
  if (u8.byteLength < 12 || tag(view, 0) !== 'RIFF' || tag(view, 8) !== 'WAVE') throw new Error('not a RIFF/WAVE file')
  let fmt = null, data = null
  for (let at = 12; at + 8 <= u8.byteLength;) {
    const id = tag(view, at), size = view.getUint32(at + 4, true)
    if (id === 'fmt ') fmt = { at: at + 8, size }
    else if (id === 'data') { data = { at: at + 8, size: Math.min(size, u8.byteLength - at - 8) }; break }
    at += 8 + size + (size & 1) // chunks are word aligned
  }
  if (!fmt || !data || fmt.size < 16) throw new Error('WAVE file without fmt/data chunks')
  let format = view.getUint16(fmt.at, true)
  const nch = view.getUint16(fmt.at + 2, true)
  const sampleRate = view.getUint32(fmt.at + 4, true)
  const blockAlign = view.getUint16(fmt.at + 12, true)
  const bits = view.getUint16(fmt.at + 14, true)
  if (format === 0xfffe && fmt.size >= 26) format = view.getUint16(fmt.at + 24, true) // sub-format GUID's first word
  const bytesPer = bits >> 3
  if (nch < 1 || bytesPer < 1 || blockAlign < nch * bytesPer) throw new Error('unsupported WAVE layout')
  const frames = Math.floor(data.size / blockAlign)
  const channels = Array.from({ length: nch }, () => shared(frames))
  let read
  if (format === 3 && bits === 32) read = p => view.getFloat32(p, true)
  else if (format === 3 && bits === 64) read = p => view.getFloat64(p, true)
  else if (format === 1 && bits === 8) read = p => (view.getUint8(p) - 128) / 128
  else if (format === 1 && bits === 16) read = p => view.getInt16(p, true) / 32768
  else if (format === 1 && bits === 24) read = p => ((view.getInt8(p + 2) << 16) | (view.getUint8(p + 1) << 8) | view.getUint8(p)) / 8388608
  else if (format === 1 && bits === 32) read = p => view.getInt32(p, true) / 2147483648
  else throw new Error('unsupported WAVE sample format ' + format + '/' + bits)
  for (let f = 0, p = data.at; f < frames; f++, p += blockAlign) {
    for (let c = 0; c < nch; c++) channels[c][f] = read(p + c * bytesPer)
  }
  return { sampleRate, channels }
}

// channels -> RIFF/WAVE bytes, IEEE float 32 (lossless for rendered stems and IRs)
export function encodeWavF32 (channels, sampleRate) {
  const nch = channels.length, frames = channels[0].length
  const out = Buffer.alloc(44 + frames * nch * 4)
  out.write('RIFF', 0); out.writeUInt32LE(36 + frames * nch * 4, 4); out.write('WAVE', 8)
  out.write('fmt ', 12); out.writeUInt32LE(16, 16); out.writeUInt16LE(3, 20); out.writeUInt16LE(nch, 22)
  out.writeUInt32LE(sampleRate, 24); out.writeUInt32LE(sampleRate * nch * 4, 28); out.writeUInt16LE(nch * 4, 32); out.writeUInt16LE(32, 34)
  out.write('data', 36); out.writeUInt32LE(frames * nch * 4, 40)
  for (let f = 0, p = 44; f < frames; f++) for (let c = 0; c < nch; c++, p += 4) out.writeFloatLE(channels[c][f], p)
  return out
}

// fetchSample for files: one decode per path, the same shared arrays for every caller
// (src/sample-service.js:19-36)
const samples = new Map()
export async function fetchSample (path) {
  let sample = samples.get(path)
  if (!sample) {
    sample = decodeWav(await fs.promises.readFile(path)).channels
    samples.set(path, sample)
  }
  return sample
}

export default { decodeWav, encodeWavF32, fetchSample }
