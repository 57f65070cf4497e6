// Parity of the JavaScript shims (dsp_amd/js -> N-API addon -> libwpdsp.so)
// against the golden vectors recorded from the reference.  Needs a GPU.
//   node tests/node/test_shims.mjs
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'
import convolve from '../../dsp_amd/js/lib/convolve.js'
import impulseConvolve from '../../dsp_amd/js/lib/impulse-convolve.js'
import impulseReverb from '../../dsp_amd/js/lib/impulse-reverb.js'
import impulseReverbStereo from '../../dsp_amd/js/lib/impulse-reverb-stereo.js'
import mixBuffers from '../../dsp_amd/js/src/mix-buffers.js'

const here = path.dirname(fileURLToPath(import.meta.url))
const golden = name => JSON.parse(fs.readFileSync(path.join(here, '..', 'golden', name), 'utf8'))
const f32 = bits => new Float32Array(new Uint32Array(bits).buffer)
const bitsOf = a => Array.from(new Uint32Array(a.buffer, a.byteOffset, a.length))
let checks = 0
const fail = msg => { throw new Error(msg) }
const close = (a, e, what, tol = 1e-5) => {
  if (a.length !== e.length) fail(what + ': length ' + a.length + ' != ' + e.length)
  let fs_ = 1, err = 0
  for (let i = 0; i < e.length; i++) { fs_ = Math.max(fs_, Math.abs(e[i])); err = Math.max(err, Math.abs(a[i] - e[i])) }
  if (!(err <= tol * fs_)) fail(what + ': max abs err ' + err)
  checks++
}
const same = (a, bits, what) => {
  const e = f32(bits)
  if (a.length !== e.length) fail(what + ': length')
  const ab = bitsOf(a)
  for (let i = 0; i < e.length; i++) {
    if (Number.isNaN(e[i]) ? !Number.isNaN(a[i]) : ab[i] !== bits[i]) fail(what + ': sample ' + i + ' ' + a[i] + ' != ' + e[i])
  }
  checks++
}
const makeContext = (buffer, n, irs, store, log) => ({
  buffer, n,
  sample: async url => irs[url].map(ch => ch),
  get: async id => { const v = store.has(id) ? store.get(id) : false; log.push(['get', id, v !== false]); return v },
  set: async (id, v, ttl) => { store.set(id, v); log.push(['set', id, ttl === undefined ? null : ttl]) }
})

const main = async () => {
  // G1: tiny go() vectors, kernel object shape
  for (const c of golden('g1_convolve_tiny.json').cases) {
    const x = f32(c.x), h = f32(c.h)
    const k = convolve.fftProcessKernel(c.L, h)
    if (!(k instanceof Float32Array)) fail('kernel type')
    if (k.length !== c.kernel_len) fail('kernel length ' + k.length + ' != ' + c.kernel_len)
    if (!(k.buffer instanceof SharedArrayBuffer)) fail('kernel not shared')
    const y = convolve.fftConvolution(c.L, k, h.length)(x)
    if (y.byteOffset !== c.out_byte_offset || y.buffer.byteLength !== c.out_buffer_bytes) fail('result view shape')
    close(y, f32(c.out), 'go L=' + c.L)
    same(convolve.directConvolution(x, h), c.direct, 'direct L=' + c.L)
    // the kernel object the reference itself produced (its interleaved spectrum) is accepted as it is
    const refKernel = f32(c.kernel)
    close(convolve.fftConvolution(c.L, refKernel, h.length)(x), f32(c.out), 'go with the reference\'s own kernel, L=' + c.L)
    // and the compat switch returns that layout: equal to the reference's array to its rounding noise
    const ck = convolve.fftProcessKernel(c.L, h, true, true)
    if (ck.length !== c.kernel_len || !(ck.buffer instanceof SharedArrayBuffer)) fail('compat kernel shape')
    let peak = 0
    for (let i = 0; i < refKernel.length; i++) peak = Math.max(peak, Math.abs(refKernel[i]))
    for (let i = 0; i < refKernel.length; i++) if (Math.abs(ck[i] - refKernel[i]) > 2e-6 * Math.max(1, peak)) fail('compat spectrum differs at ' + i)
    checks++
  }
  // error classes
  try { convolve.fftConvolution(4, convolve.fftProcessKernel(4, new Float32Array(4)), 4)(new Float32Array(9)); fail('no RangeError') } catch (e) { if (!(e instanceof RangeError)) throw e; checks++ }
  try { convolve.fftProcessKernel(0, new Float32Array(1)); fail('no Error') } catch (e) { if (e.message !== 'FFT size must be a power of two and bigger than 1') throw e; checks++ }

  // G3: reverb block sequences: view shapes bit-exact, samples <= 1e-5, cache protocol identical
  for (const run of golden('g3_reverb_blocks.json').runs) {
    const store = new Map(), log = []
    const irs = {}; irs[run.params.url] = run.ir.map(f32)
    const make = run.kind === 'stereo' ? impulseReverbStereo : impulseReverb
    const ctx = blk => makeContext(blk.input.map(f32), blk.n, irs, store, log)
    const c0 = ctx(run.blocks[0])
    const fn = await make(c0, run.params)
    for (let i = 0; i < run.blocks.length; i++) {
      const blk = run.blocks[i]
      let ret = fn(i === 0 ? c0 : ctx(blk))
      if (run.kind === 'mono') ret = [ret]
      blk.ret.forEach((exp, ch) => {
        const what = run.kind + ' ' + JSON.stringify(run.params) + ' block ' + i
        if (ret[ch].length !== exp.length || ret[ch].byteOffset !== exp.byte_offset || ret[ch].buffer.byteLength !== exp.buffer_bytes) fail(what + ': view shape')
        close(ret[ch], f32(exp.data), what)
      })
      let pv = store.get(log[log.length - 1][1])
      if (run.kind === 'mono') pv = [pv]
      blk.prev.forEach((exp, ch) => {
        if (pv[ch].length !== exp.length || pv[ch].byteOffset !== exp.byte_offset) fail('prev view shape')
        close(pv[ch], f32(exp.data), 'prev')
      })
    }
    const cR = makeContext(run.resumed.input.map(f32), run.resumed.n, irs, store, log)
    let r2 = (await make(cR, run.params))(cR)
    if (run.kind === 'mono') r2 = [r2]
    run.resumed.ret.forEach((exp, ch) => close(r2[ch], f32(exp.data), 'resumed'))
    if (JSON.stringify(log) !== JSON.stringify(run.log)) fail('cache protocol differs: ' + JSON.stringify(run.params))
    checks++
  }

  // G4: mixBuffers bit-exact, every branch
  const vols = { ramp: x => 0.25 + x / 8, alt: x => (x & 1) ? -1.3 : 0.7 }
  for (const c of golden('g4_mix_buffers.json').cases) {
    const target = c.target_in.map(f32)
    const sources = c.sources.map(s => {
      const buf = s.buffer.map(f32)
      if (!s.tuple) return buf
      const t = [buf]
      if ('length' in s) t[1] = s.length
      if ('volume' in s) t[2] = (typeof s.volume === 'object') ? vols[s.volume.fn] : s.volume
      if ('offset' in s) t[3] = s.offset
      return t
    })
    const ret = mixBuffers(target, ...sources)
    if (ret !== target) fail('mixBuffers must return target')
    c.target_out.forEach((exp, ch) => same(target[ch], exp, c.name + ' ch' + ch))
  }

  // G11: sources that share memory with the target (the same arrays, views into and beyond them), bit-exact
  // over everything the call may touch
  for (const c of golden('g11_mix_shared_memory.json').cases) {
    const bases = c.bases_in.map(f32)
    const view = spec => bases[spec[0]].subarray(spec[1], spec[2])
    const target = c.target.map(view)
    const sources = c.sources.map(s => {
      const ch = s.ch.map(view)
      if (!('length' in s)) return ch
      return [ch, s.length, (typeof s.volume === 'object') ? vols[s.volume.fn] : s.volume, s.offset]
    })
    if (mixBuffers(target, ...sources) !== target) fail('mixBuffers must return target')
    c.bases_out.forEach((exp, i) => same(bases[i], exp, c.name + ' base ' + i))
    checks++
  }

  // C1 through the wrapper
  {
    const g = golden('g5_config_sized.json').c1
    // regenerate the synthetic inputs (same integer-hash definition as the fixtures' generator)
    const fmix32 = h => { h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13; h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16; return h >>> 0 }
    const noise = (n, seed) => { const k = Math.imul(seed, 0x9E3779B9) >>> 0; const o = new Float32Array(n); for (let i = 0; i < n; i++) o[i] = ((fmix32((i ^ k) >>> 0) >>> 8) - 8388608) / 8388608; return o }
    const synthIr = (M, seed) => { const k = Math.imul(seed, 0x9E3779B9) >>> 0; const r = Math.fround(1 - 6.907755 / M); const o = new Float32Array(M); let e = 1, norm = 0; for (let i = 0; i < M; i++) { o[i] = (((fmix32((i ^ k) >>> 0) >>> 8) - 8388608) / 8388608) * e; norm += Math.abs(o[i]); e = Math.fround(e * r) } for (let i = 0; i < M; i++) o[i] = o[i] / norm; return o }
    const x = noise(g.L, g.seed_x), ir = synthIr(g.M, g.seed_ir)
    const store = new Map(), log = []
    const c = makeContext([x], 0, { 'ir:c1': [ir] }, store, log)
    const y = (await impulseConvolve(c, 'ir:c1', -1))(x)
    const strided = new Float32Array(Math.ceil(y.length / g.stride))
    for (let i = 0, j = 0; i < y.length; i += g.stride, j++) strided[j] = y[i]
    close(strided, f32(g.samples), 'c1 strided')
    if (store.get('kernel:ir:c1:44100:-1').length !== g.kernel_len) fail('c1 kernel length')
  }
  console.log('ok ' + checks + ' checks')
}
main().catch(e => { console.error(e); process.exit(1) })
