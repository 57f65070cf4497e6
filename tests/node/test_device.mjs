// Device-resident buffers through the JavaScript shims (needs a GPU).  The same
// calls are made twice -- once on Float32Arrays (the path pinned to the reference's
// golden vectors by test_shims.mjs) and once on DeviceBuffers -- and every result
// must agree bit for bit: view shapes, cache keys, samples, NaNs.
import v8 from 'v8'
import { Worker, isMainThread, parentPort, workerData } from 'worker_threads'
import { fileURLToPath } from 'url'
import convolve from '../../dsp_amd/js/lib/convolve.js'
import impulseReverb from '../../dsp_amd/js/lib/impulse-reverb.js'
import impulseReverbStereo from '../../dsp_amd/js/lib/impulse-reverb-stereo.js'
import mixBuffers from '../../dsp_amd/js/src/mix-buffers.js'
import { DeviceBuffer, DeviceChannel, isDeviceChannel } from '../../dsp_amd/js/src/device-buffer.js'
import { uniqueId, createComm, reduceBus } from '../../dsp_amd/js/src/bus-reduce.js'

let checks = 0
const fail = msg => { throw new Error(msg) }
const ramp = (n, k) => { const a = new Float32Array(n); for (let i = 0; i < n; i++) a[i] = ((i * 37 + k * 11) % 101) / 101 - 0.5; return a }
const host = x => isDeviceChannel(x) ? (x instanceof DeviceChannel ? x : DeviceBuffer.revive([x])[0]).toFloat32Array() : x
const sameBits = (a, b, what) => {
  a = host(a); b = host(b)
  if (a.length !== b.length) fail(what + ': length ' + a.length + ' != ' + b.length)
  const ua = new Uint32Array(a.buffer, a.byteOffset, a.length), ub = new Uint32Array(b.buffer, b.byteOffset, b.length)
  for (let i = 0; i < a.length; i++) if (ua[i] !== ub[i] && !(Number.isNaN(a[i]) && Number.isNaN(b[i]))) fail(what + ': sample ' + i + ': ' + a[i] + ' != ' + b[i])
  checks++
}
const clone = x => v8.deserialize(v8.serialize(x)) // what postMessage / c.set do to a value

// the stem a worker renders: any per-sample JavaScript, like src/render.js:53-90
const renderStem = (frames, seed) => [0, 1].map(ch => { const a = new Float32Array(frames); for (let i = 0; i < frames; i++) a[i] = Math.sin((i + 1) * (seed + 1) * 0.001 + ch) * 0.25; return a })

const main = async () => {
  // ---- buffer semantics mirror Float32Array's
  {
    const h = [ramp(1000, 1), ramp(1000, 2)]
    const d = DeviceBuffer.upload(h)
    if (!Array.isArray(d) || d.length !== 2 || d[0].length !== 1000) fail('shape')
    sameBits(d[0], h[0], 'upload/download ch0'); sameBits(d[1], h[1], 'upload/download ch1')
    sameBits(DeviceBuffer.alloc(1, 77)[0], new Float32Array(77), 'fresh buffer is zero')
    for (const [b, e] of [[10, 20], [-5], [-2000, 3], [990, 5000], [30, 10], [undefined, -1]]) {
      const dv = d[0].subarray(b, e), hv = h[0].subarray(b, e)
      if (dv.length !== hv.length || dv.byteOffset !== hv.byteOffset) fail('subarray(' + b + ',' + e + ') shape')
      sameBits(dv, hv, 'subarray')
    }
    d[1].set(d[0]); h[1].set(h[0]) // render.js:45
    sameBits(d[1], h[1], 'channel.set(channel)')
    d[0].set(ramp(10, 9), 5); h[0].set(ramp(10, 9), 5)
    sameBits(d[0], h[0], 'channel.set(host, offset)')
    d[0].subarray(100, 400).set(d[0].subarray(50, 350)); h[0].subarray(100, 400).set(h[0].subarray(50, 350))
    sameBits(d[0], h[0], 'overlapping set')
    try { d[0].set(ramp(10, 1), 995); fail('no RangeError') } catch (e) { if (!(e instanceof RangeError)) throw e; checks++ }
    d[0].fill(0.5, 3, -3); h[0].fill(0.5, 3, -3)
    sameBits(d[0], h[0], 'fill range')
    const ds = d[0].slice(7, 30); ds.fill(1)
    sameBits(d[0], h[0], 'slice is a copy')
    d.forEach(b => b.fill(0)) // context.js:140-143 zero()
    sameBits(d[0], new Float32Array(1000), 'zero'); sameBits(d[1], new Float32Array(1000), 'zero')
    const back = DeviceBuffer.revive(clone(d))
    back[1].fill(2)
    sameBits(d[1], new Float32Array(1000).fill(2), 'clone shares the memory')
  }

  // ---- go(): device input -> device result
  {
    const L = 3000, M = 1700, x = ramp(L, 3)
    const k = convolve.fftProcessKernel(L, ramp(M, 5))
    const go = convolve.fftConvolution(L, k, M)
    const yd = go(DeviceBuffer.upload([x])[0])
    if (!(yd instanceof DeviceChannel) || yd.length !== L + M || yd.byteOffset !== 0) fail('go result shape')
    sameBits(yd, go(x), 'go on device')
  }

  // ---- mixBuffers: device bus, device and host sources, every source form
  {
    const tl = 4000
    const stems = [renderStem(tl, 1), renderStem(1500, 2), [ramp(700, 4)], renderStem(tl, 3)]
    const forms = dev => {
      const S = stems.map(s => dev ? DeviceBuffer.upload(s) : s)
      return [S[0], [S[1], 0.75, 0.5, 123], [S[2], 1, x => 0.25 + (x & 3) / 8, 5], S[2], [S[3], 1, 2, 4294967301], [S[1], 1, 1, -3], [S[0]]]
    }
    for (const tch of [2, 1]) {
      const th = Array.from({ length: tch }, (_, c) => ramp(tl, 20 + c))
      const td = DeviceBuffer.upload(th)
      if (mixBuffers(td, ...forms(true)) !== td) fail('mixBuffers must return target')
      mixBuffers(th, ...forms(false))
      th.forEach((a, c) => sameBits(td[c], a, 'device mix, ' + tch + ' target channels, ch' + c))
      // host stems into a device bus: the copy-back as an upload
      const td2 = DeviceBuffer.upload(Array.from({ length: tch }, (_, c) => ramp(tl, 20 + c)))
      mixBuffers(td2, ...forms(false))
      th.forEach((a, c) => sameBits(td2[c], a, 'host sources into a device bus ch' + c))
    }
  }

  // ---- a source that shares memory with the device bus is read as the loop updates the bus, like the reference's
  // arrays (src/mix-buffers.js:4): the same calls on host arrays (pinned to the reference's vectors) give the same bits
  {
    const host = [ramp(100, 31), ramp(100, 32)], other = [ramp(100, 33), ramp(100, 34)]
    const dev = DeviceBuffer.upload(host.map(a => Float32Array.from(a)))
    mixBuffers(dev, dev, Array.from(DeviceBuffer.upload(other)), [dev, 1, 0.5, 7])
    mixBuffers(host, host, other, [host, 1, 0.5, 7])
    host.forEach((a, c) => sameBits(dev[c], a, 'device bus mixed with itself ch' + c))
    checks++
  }

  // ---- refused rather than undefined: ragged channels
  {
    const bus = DeviceBuffer.alloc(2, 100)
    const ragged = [DeviceBuffer.alloc(1, 100)[0], DeviceBuffer.alloc(1, 50)[0]]
    try { mixBuffers(bus, ragged); fail('ragged device source accepted') } catch (e) { if (!(e instanceof TypeError)) throw e; checks++ }
    try { mixBuffers([new Float32Array(100), new Float32Array(100)], [new Float32Array(100), new Float32Array(50)]); fail('ragged host source accepted') } catch (e) { if (!(e instanceof TypeError)) throw e; checks++ }
  }

  // ---- block reverb with c.buffer on the device
  for (const [make, nch] of [[impulseReverb, 1], [impulseReverbStereo, 2]]) {
    for (const offset of [0, 2, 700, 1202]) {
      const L = 2000, M = 1200, url = 'ir:' + nch
      const ir = Array.from({ length: 2 }, (_, c) => ramp(M, 30 + c))
      const run = async dev => {
        const store = new Map(), log = [], rets = []
        const ctx = (blk, n) => ({
          buffer: dev ? DeviceBuffer.upload(blk) : blk, n,
          sample: async () => ir.map(a => a),
          get: async id => { log.push('get ' + id); return store.has(id) ? clone(store.get(id)) : false },
          set: async (id, v, ttl) => { log.push('set ' + id + ' ' + ttl); store.set(id, clone(v)) }
        })
        const blocks = [0, 1, 2, 3].map(b => Array.from({ length: nch }, (_, c) => ramp(L, 50 + 2 * b + c)))
        const c0 = ctx(blocks[0], 0)
        let fn = await make(c0, { url, offset, id: 'x' })
        for (let b = 0; b < 4; b++) {
          if (b === 2) fn = await make(ctx(blocks[b], b * L), { url, offset, id: 'x' }) // resume from the cached (cloned) tail
          const r = fn(b === 0 ? c0 : ctx(blocks[b], b * L))
          rets.push(nch === 1 ? [r] : r)
        }
        return { rets, log }
      }
      const h = await run(false), d = await run(true)
      if (JSON.stringify(h.log) !== JSON.stringify(d.log)) fail('cache protocol differs')
      h.rets.forEach((r, b) => r.forEach((a, c) => {
        const v = d.rets[b][c]
        if (!isDeviceChannel(v) || v.length !== a.length || v.byteOffset !== a.byteOffset) fail('view shape, block ' + b)
        sameBits(v, a, 'reverb nch=' + nch + ' offset=' + offset + ' block ' + b + ' ch' + c)
      }))
    }
  }

  // ---- worker threads render stems straight into device buffers; the bus never leaves the GPU
  {
    const frames = 6000, nw = 3
    const stemsDev = Array.from({ length: nw }, () => DeviceBuffer.alloc(2, frames))
    await Promise.all(stemsDev.map((buf, seed) => new Promise((resolve, reject) => {
      const w = new Worker(fileURLToPath(import.meta.url), { workerData: { buffer: buf, frames, seed } })
      w.on('message', resolve); w.on('error', reject)
    })))
    const bus = DeviceBuffer.alloc(2, frames), want = [new Float32Array(frames), new Float32Array(frames)]
    mixBuffers(bus, ...stemsDev.map((b, i) => [b, 1, 1 / (i + 1), 0]))
    mixBuffers(want, ...stemsDev.map((_, i) => [renderStem(frames, i), 1, 1 / (i + 1), 0]))
    want.forEach((a, c) => sameBits(bus[c], a, 'bus from worker-rendered device stems ch' + c))
    // the multi-GPU exchange step with the one rank this box has: RCCL through the addon returns the bus unchanged
    const comm = createComm(uniqueId(), 1, 0)
    reduceBus(comm, bus)
    want.forEach((a, c) => sameBits(bus[c], a, 'bus after a one-rank reduce ch' + c))
    reduceBus(comm, bus, { all: true })
    want.forEach((a, c) => sameBits(bus[c], a, 'bus after a one-rank all-reduce ch' + c))
  }
  console.log('ok device ' + checks + ' checks')
}

if (isMainThread) {
  main().catch(e => { console.error(e); process.exit(1) })
} else {
  // the counterpart of src/mix-worker-thread.js:8-27: render into a private buffer,
  // then "copy back" -- here one upload into the shared device buffer
  const { buffer, frames, seed } = workerData
  const target = DeviceBuffer.revive(buffer)
  const mine = renderStem(frames, seed)
  target.forEach((ch, i) => ch.set(mine[i]))
  parentPort.postMessage('done')
}
