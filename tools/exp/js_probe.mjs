// Where the host time of one device-resident bar goes through the JavaScript boundary: each native call of
// the block reverb timed alone (node --no-warnings tools/exp/js_probe.mjs [iterations]).
import native from '../../dsp_amd/js/native.js'
import convolve, { irHandle } from '../../dsp_amd/js/lib/convolve.js'
import { DeviceBuffer, liveChannel, keepAlive } from '../../dsp_amd/js/src/device-buffer.js'

const now = () => Number(process.hrtime.bigint()) / 1e3
const N = parseInt(process.argv[2] || '2000', 10)
const L = 96000, M = 96000
const ir = [0, 1].map(s => { const a = new Float32Array(M); for (let i = 0; i < M; i++) a[i] = Math.sin(i * (s + 1) * 0.001) / M; return a })
const kernels = ir.map(h => convolve.fftProcessKernel(L, h))
const gos = kernels.map(k => convolve.fftConvolution(L, k, M))
const handle = irHandle(...gos.map(g => g.kernel))
const state = native.reverbCreate(handle, 2, 0)
const dry = DeviceBuffer.upload([new Float32Array(L).fill(0.25), new Float32Array(L).fill(-0.25)])
const fixedOut = DeviceBuffer.alloc(2, L + M)
const sync = () => fixedOut[0].subarray(0, 1).toFloat32Array()
const time = (name, f) => {
  for (let i = 0; i < 50; i++) f()
  sync()
  const t0 = now()
  for (let i = 0; i < N; i++) f()
  const t1 = now()
  sync()
  const t2 = now()
  console.log(JSON.stringify({ call: name, host_us: +((t1 - t0) / N).toFixed(2), with_device_us: +((t2 - t0) / N).toFixed(2) }))
}
const ins = dry.map(liveChannel)
time('DeviceBuffer.alloc(2, L+M)', () => DeviceBuffer.alloc(2, L + M))
time('native.bufCreate only', () => native.bufCreate(2, L + M))
time('reverbBlockTo into one buffer', () => native.reverbBlockTo(state, ins, fixedOut))
time('alloc + reverbBlockTo', () => { const o = DeviceBuffer.alloc(2, L + M); native.reverbBlockTo(state, ins, o) })
time('alloc + reverbBlockTo + views + keepAlive', () => {
  const o = DeviceBuffer.alloc(2, L + M)
  const r = native.reverbBlockTo(state, dry.slice(0, 2).map(liveChannel), o)
  const p = [o[0].subarray(r.prevOff, r.prevOff + r.prevLen), o[1].subarray(r.prevOff, r.prevOff + r.prevLen)]
  keepAlive(state, Array.from(o), 5000)
  return [p, Array.from(o, a => a.subarray(r.retOff, r.retOff + r.retLen))]
})
time('fill(0) of a stereo bus', () => { fixedOut[0].fill(0); fixedOut[1].fill(0) })
