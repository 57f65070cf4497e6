// WAVE ingestion (dsp_amd/js/src/sample-file.js): hand-computed PCM vectors, float
// round trip, odd chunks, extensible header, the per-path cache.  No GPU needed.
import fs from 'fs'
import os from 'os'
import path from 'path'
import { decodeWav, encodeWavF32, fetchSample } from '../../dsp_amd/js/src/sample-file.js'

let checks = 0
const fail = m => { throw new Error(m) }
const eq = (a, b, what) => { if (a.length !== b.length) fail(what + ': length'); for (let i = 0; i < a.length; i++) if (!Object.is(a[i], Math.fround(b[i]))) fail(what + ': [' + i + '] ' + a[i] + ' != ' + b[i]); checks++ }
const wav = (format, bits, nch, rate, payload, extra = []) => {
  const fmtSize = format === 0xfffe ? 40 : 16
  const head = Buffer.alloc(12 + 8 + fmtSize)
  head.write('RIFF', 0); head.write('WAVE', 8); head.write('fmt ', 12); head.writeUInt32LE(fmtSize, 16)
  head.writeUInt16LE(format, 20); head.writeUInt16LE(nch, 22); head.writeUInt32LE(rate, 24)
  head.writeUInt32LE(rate * nch * bits / 8, 28); head.writeUInt16LE(nch * bits / 8, 32); head.writeUInt16LE(bits, 34)
  if (format === 0xfffe) { head.writeUInt16LE(22, 36); head.writeUInt16LE(bits, 38); head.writeUInt32LE(3, 40); head.writeUInt16LE(3, 44) }
  const dh = Buffer.alloc(8); dh.write('data', 0); dh.writeUInt32LE(payload.length, 4)
  const all = Buffer.concat([head, ...extra, dh, payload])
  all.writeUInt32LE(all.length - 8, 4)
  return all
}

// 16-bit stereo, with an odd-sized LIST chunk (padded) before the data
{
  const p = Buffer.alloc(12); [0, 32767, -32768, 16384, 1, -1].forEach((v, i) => p.writeInt16LE(v, 2 * i))
  const list = Buffer.concat([Buffer.from('LIST'), Buffer.from([3, 0, 0, 0]), Buffer.from('abc'), Buffer.from([0])])
  const d = decodeWav(wav(1, 16, 2, 44100, p, [list]))
  if (d.sampleRate !== 44100 || d.channels.length !== 2) fail('header')
  if (!(d.channels[0].buffer instanceof SharedArrayBuffer)) fail('channels must be shared')
  eq(d.channels[0], [0, -1, 1 / 32768], 'pcm16 L'); eq(d.channels[1], [32767 / 32768, 0.5, -1 / 32768], 'pcm16 R')
}
// 8-bit unsigned, 24-bit, 32-bit int
eq(decodeWav(wav(1, 8, 1, 8000, Buffer.from([128, 0, 255, 192]))).channels[0], [0, -1, 127 / 128, 0.5], 'pcm8')
eq(decodeWav(wav(1, 24, 1, 48000, Buffer.from([0, 0, 0x40, 0xff, 0xff, 0xff, 0, 0, 0x80, 0xff, 0xff, 0x7f]))).channels[0], [0.5, -1 / 8388608, -1, 8388607 / 8388608], 'pcm24')
{ const p = Buffer.alloc(8); p.writeInt32LE(-2147483648, 0); p.writeInt32LE(1073741824, 4); eq(decodeWav(wav(1, 32, 1, 48000, p)).channels[0], [-1, 0.5], 'pcm32') }
// float64 and extensible-float32
{ const p = Buffer.alloc(16); p.writeDoubleLE(0.1, 0); p.writeDoubleLE(-2.5, 8); eq(decodeWav(wav(3, 64, 1, 96000, p)).channels[0], [0.1, -2.5], 'f64') }
{ const p = Buffer.alloc(8); p.writeFloatLE(0.3, 0); p.writeFloatLE(-0.7, 4); eq(decodeWav(wav(0xfffe, 32, 2, 48000, p)).channels[1], [-0.7], 'extensible f32') }
// float32 round trip is bit-exact, including -0, subnormals, NaN payloads are not required
{
  const ch = [Float32Array.from([0, -0, 1e-42, 0.1, -1, 3.4e38]), Float32Array.from([1, 2, 3, 4, 5, 6])]
  const d = decodeWav(encodeWavF32(ch, 48000))
  eq(d.channels[0], ch[0], 'f32 round trip L'); eq(d.channels[1], ch[1], 'f32 round trip R')
  const file = path.join(os.tmpdir(), 'wpdsp-test-' + process.pid + '.wav')
  fs.writeFileSync(file, encodeWavF32(ch, 48000))
  const run = async () => {
    const a = await fetchSample(file), b = await fetchSample(file)
    if (a !== b) fail('fetchSample must cache per path')
    eq(a[1], ch[1], 'fetchSample')
    fs.unlinkSync(file)
    for (const bad of [Buffer.from('nope'), wav(1, 12, 1, 8000, Buffer.alloc(4)), wav(7, 8, 1, 8000, Buffer.alloc(4))]) {
      try { decodeWav(bad); fail('accepted a bad file') } catch (e) { if (!/WAVE/.test(e.message)) throw e; checks++ }
    }
    console.log('ok sample-file ' + checks + ' checks')
  }
  run().catch(e => { console.error(e); process.exit(1) })
}
