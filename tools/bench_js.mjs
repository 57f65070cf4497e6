// End-to-end timing THROUGH the JavaScript boundary (node -> N-API -> libwpdsp.so):
// the stereo block reverb + mix of one bar at 48 kHz / 120 BPM (L = 96 000 frames,
// 2 s stereo IR), the way src/context.js:132-138,164-166 drives it, measured
//   host   : c.buffer and the bus are Float32Arrays (every bar crosses PCIe twice)
//   device : c.buffer and the bus are DeviceBuffers (nothing leaves the GPU)
// and a single mono go() of L = 44 100, M = 22 050.  Prints one JSON line.
//   node --no-warnings tools/bench_js.mjs [bars]
import impulseConvolve from '../dsp_amd/js/lib/impulse-convolve.js'
import impulseReverbStereo from '../dsp_amd/js/lib/impulse-reverb-stereo.js'
import mixBuffers from '../dsp_amd/js/src/mix-buffers.js'
import { DeviceBuffer } from '../dsp_amd/js/src/device-buffer.js'

const fmix32 = h => { h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13; h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16; return h >>> 0 }
const noise = (n, seed, start = 0) => { const k = Math.imul(seed, 0x9E3779B9) >>> 0; const o = new Float32Array(n); for (let i = 0; i < n; i++) o[i] = ((fmix32(((start + i) ^ k) >>> 0) >>> 8) - 8388608) / 8388608; return o }
const synthIr = (M, seed) => { const o = noise(M, seed); const r = Math.fround(1 - 6.907755 / M); let e = 1, norm = 0; for (let i = 0; i < M; i++) { o[i] = o[i] * e; norm += Math.abs(o[i]); e = Math.fround(e * r) } for (let i = 0; i < M; i++) o[i] = o[i] / norm; return o }
const now = () => Number(process.hrtime.bigint()) / 1e3 // microseconds

const bars = parseInt(process.argv[2] || '64', 10)
const L = 96000, M = 96000
const ir = [synthIr(M, 500000), synthIr(M, 500001)]
const context = (buffer, n, store) => ({
  buffer, n,
  sample: async () => ir.map(a => a),
  get: async id => store.has(id) ? store.get(id) : false,
  set: async (id, v) => { store.set(id, v) }
})

const run = async device => {
  const wrap = a => device ? DeviceBuffer.upload(a) : a
  const dry = [0, 1, 2, 3].map(b => wrap([noise(L, 1000, b * L), noise(L, 1001, b * L)]))
  const bus = wrap([new Float32Array(L), new Float32Array(L)])
  const store = new Map()
  const c0 = context(dry[0], 0, store)
  const reverb = await impulseReverbStereo(c0, { url: 'ir', id: 'bench' })
  const parts = { reverb: 0, zero: 0, mix: 0 }  // host time spent inside each call (device work queued by a call may end later)
  const bar = (b, timed) => {
    const c = context(dry[b & 3], b * L, store)
    const t1 = timed ? now() : 0
    const wet = reverb(c)
    const t2 = timed ? now() : 0
    bus.forEach(ch => ch.fill(0)) // Context.zero
    const t3 = timed ? now() : 0
    mixBuffers(bus, [c.buffer, 1, 0.5, 0], [wet, 1, 0.5, 0])
    if (timed) { const t4 = now(); parts.reverb += t2 - t1; parts.zero += t3 - t2; parts.mix += t4 - t3 }
  }
  const sync = () => device ? bus[0].subarray(0, 1).toFloat32Array() : bus[0]
  for (let b = 0; b < 4; b++) bar(b, false)
  sync()
  const t0 = now()
  for (let b = 0; b < bars; b++) bar(b, false)
  sync()
  const per = (now() - t0) / bars
  for (let b = 0; b < bars; b++) bar(b, true)
  sync()
  for (const k in parts) parts[k] = +(parts[k] / bars).toFixed(1)
  return { per, parts }
}

const main = async () => {
  const host = await run(false)
  const device = await run(true)
  // one mono go() through impulseConvolve
  const x = noise(44100, 1000), h1 = synthIr(22050, 500000), store = new Map()
  const c = { buffer: [x], n: 0, sample: async () => [h1], get: async id => store.has(id) ? store.get(id) : false, set: async (id, v) => { store.set(id, v) } }
  const go = await impulseConvolve(c, 'ir:c1', -1)
  for (let i = 0; i < 3; i++) go(x)
  const t0 = now()
  for (let i = 0; i < 50; i++) go(x)
  const c1 = (now() - t0) / 50
  console.log(JSON.stringify({
    workload: 'stereo impulse-reverb + mix per bar, L=96000 @48 kHz, 2 s stereo IR, through the JS shims',
    bars,
    us_per_bar_host_arrays: +host.per.toFixed(1),
    us_per_bar_device_buffers: +device.per.toFixed(1),
    host_us_inside_calls_host_arrays: host.parts,
    host_us_inside_calls_device_buffers: device.parts,
    realtime_factor_device: +((L / 48000) * 1e6 / device.per).toFixed(0),
    us_per_go_L44100_M22050_host_arrays: +c1.toFixed(1)
  }))
}
main().catch(e => { console.error(e); process.exit(1) })
