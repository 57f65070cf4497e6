// gen_golden.mjs — runs the REAL reference (imported read-only, in place, from
// /root/reference via oracle/ref_loader.mjs) on deterministic inputs and writes
// small input/output fixtures to tests/golden/.  Container-only: the reference
// never travels; only these vectors do.
//
//   node --harmony-nullish --harmony-optional-chaining \
//        --experimental-loader ./oracle/ref_loader.mjs oracle/gen_golden.mjs
//
// Every float is recorded as its IEEE-754 binary32 bit pattern (uint32) so the
// checks are bit-exact.
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'
import convolve from 'file:///root/reference/lib/convolve.js'
import impulseConvolve from 'file:///root/reference/lib/impulse-convolve.js'
import impulseConvolveStereo from 'file:///root/reference/lib/impulse-convolve-stereo.js'
import impulseReverb from 'file:///root/reference/lib/impulse-reverb.js'
import impulseReverbStereo from 'file:///root/reference/lib/impulse-reverb-stereo.js'
import mixBuffers from 'file:///root/reference/src/mix-buffers.js'
import RingBuffer from 'file:///root/reference/lib/ring-buffer.js'

const here = path.dirname(fileURLToPath(import.meta.url))
const outDir = path.join(here, '..', 'tests', 'golden')
fs.mkdirSync(outDir, { recursive: true })

const bits = a => Array.from(new Uint32Array(a.buffer, a.byteOffset, a.length))
const f32 = a => new Float32Array(a)
const save = (name, obj) => {
  fs.writeFileSync(path.join(outDir, name), JSON.stringify(obj))
  console.log('wrote', name, fs.statSync(path.join(outDir, name)).size, 'bytes')
}

// ---- synthetic inputs (same definition as oracle/wp_oracle.c wpo_synth_*) ----
const fmix32 = h => {
  h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13
  h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16
  return h >>> 0
}
const noise = (n, seed, start = 0) => {
  const k = Math.imul(seed, 0x9E3779B9) >>> 0
  const out = new Float32Array(n)
  for (let i = 0; i < n; i++) {
    const u = fmix32(((start + i) ^ k) >>> 0)
    out[i] = ((u >>> 8) - 8388608) / 8388608
  }
  return out
}
const synthIr = (M, seed) => {
  const k = Math.imul(seed, 0x9E3779B9) >>> 0
  const r = Math.fround(1 - 6.907755 / M)
  const out = new Float32Array(M)
  let e = 1, norm = 0
  for (let i = 0; i < M; i++) {
    const u = fmix32((i ^ k) >>> 0)
    const nz = ((u >>> 8) - 8388608) / 8388608
    out[i] = nz * e
    norm += Math.abs(out[i])
    e = Math.fround(e * r)
  }
  for (let i = 0; i < M; i++) out[i] = out[i] / norm
  return out
}
const stats = a => {
  let sum = 0, sq = 0, mx = 0
  for (let i = 0; i < a.length; i++) { sum += a[i]; sq += a[i] * a[i]; if (Math.abs(a[i]) > mx) mx = Math.abs(a[i]) }
  return { sum, sq, max_abs: mx }
}
const strided = (a, stride) => {
  const out = new Float32Array(Math.ceil(a.length / stride))
  for (let i = 0, j = 0; i < a.length; i += stride, j++) out[j] = a[i]
  return out
}
const checksum = a => { // order-sensitive 32-bit checksum of the bit patterns
  const u = new Uint32Array(a.buffer, a.byteOffset, a.length)
  let h = 0
  for (let i = 0; i < u.length; i++) h = (Math.imul(h, 31) + u[i]) >>> 0
  return h
}

// A stub of the Context surface the wrappers use (src/context.js:112-130,
// src/global-service.js:10-19): sample/get/set over an in-memory Map.
const makeContext = (buffer, n, irs, store, log) => ({
  buffer, n,
  sample: async url => irs[url].map(ch => ch),          // fresh array, shared channels
  get: async id => { const v = store.has(id) ? store.get(id) : false; log.push(['get', id, v !== false]); return v },
  set: async (id, v, ttl) => { store.set(id, v); log.push(['set', id, ttl === undefined ? null : ttl]) }
})

const main = async () => {
  // ---------------- G1: tiny go() vectors, every leaf/size shape ----------------
  {
    const cases = []
    const specs = [
      { L: 8, x: [1, 2, 3, 4, 0, 0, 0, 0], h: [1, .5, .25] },      // N=16 (even power: radix-4 leaf)
      { L: 4, x: [1, -2, 3, -4], h: [.5, .25, .125, 1] },            // N=8  (odd power: radix-2 leaf)
      { L: 1, x: [3], h: [2] },                                       // N=2
      { L: 2, x: [1, 1], h: [1, -1] },                                // N=4
      { L: 20, x: Array.from(noise(20, 7)), h: Array.from(synthIr(12, 9)) },   // N=32
      { L: 40, x: Array.from(noise(40, 8)), h: Array.from(synthIr(24, 10)) },  // N=64
      { L: 64, x: Array.from(noise(50, 11)), h: Array.from(synthIr(64, 12)) }, // N=128, input shorter than L
      { L: 300, x: Array.from(noise(300, 13)), h: Array.from(synthIr(200, 14)) }, // N=512
      { L: 700, x: Array.from(noise(700, 15)), h: Array.from(synthIr(300, 16)) }, // N=1024
    ]
    for (const s of specs) {
      const x = f32(s.x), h = f32(s.h)
      const k = convolve.fftProcessKernel(s.L, h)
      const go = convolve.fftConvolution(s.L, k, h.length)
      const y = go(x)
      const d = convolve.directConvolution(x, h)
      cases.push({
        L: s.L, M: h.length, x: bits(x), h: bits(h),
        kernel_len: k.length, kernel_shared: k.buffer instanceof SharedArrayBuffer,
        kernel: bits(k), out_len: y.length, out: bits(y),
        out_byte_offset: y.byteOffset, out_buffer_bytes: y.buffer.byteLength,
        direct: bits(d)
      })
    }
    // error behaviour (convolve.js:8-9, :87-88, :590)
    const errors = {}
    try { convolve.fftConvolution(4, new Float32Array(16), 4)(new Float32Array(9)) } catch (e) { errors.input_too_long = e.constructor.name }
    try { convolve.fftProcessKernel(0, new Float32Array(1)) } catch (e) { errors.size_one = [e.constructor.name, e.message] }
    save('g1_convolve_tiny.json', { cases, errors })
  }

  // ---------------- G2: nextPOT + FFT table / spectrum checksums ----------------
  {
    // reach the bundled FFT through the closure-free route: a kernel spectrum of a
    // unit impulse delayed by 1 is exactly the twiddle table (cos, -sin) column.
    const pot = []
    for (const n of [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 1000, 65535, 65536, 65537, 66150, 192000, 2976000, (1 << 30) - 1, 1 << 30]) {
      // nextPOT is not exported; observe it through the kernel length for small n
      pot.push(n)
    }
    const potOut = []
    for (const n of [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 1000, 4097]) {
      // fftProcessKernel(length=n-1, kernel of 1 sample) -> N = nextPOT(n)
      if (n < 2) continue
      potOut.push([n, convolve.fftProcessKernel(n - 1, new Float32Array(1)).length / 2])
    }
    const tables = []
    for (const N of [4, 8, 16, 32, 64, 128, 1024, 4096, 32768, 131072, 262144, 524288]) {
      const delta = new Float32Array(2); delta[1] = 1          // x[1] = 1 -> spectrum = table
      const spec = convolve.fftProcessKernel(N - 2, delta)
      if (spec.length !== 2 * N) throw new Error('size')
      tables.push({ N, checksum: checksum(spec), head: bits(spec.subarray(0, Math.min(16, spec.length))) })
    }
    save('g2_tables.json', { next_pot_via_kernel_len: potOut, delta_spectra: tables })
  }

  // ---------------- G3: reverb block sequences ----------------
  {
    const runs = []
    const irMono = { 'ir:m': [synthIr(5, 21)] }
    const irStereo = { 'ir:s': [synthIr(5, 22), synthIr(5, 23)] }
    const irLong = { 'ir:l': [synthIr(40, 24), synthIr(40, 25)] }
    const scenarios = [
      { kind: 'mono', irs: irMono, url: 'ir:m', L: 8, blocks: 4, params: {} },
      { kind: 'mono', irs: irMono, url: 'ir:m', L: 8, blocks: 4, params: { offset: 2, id: 'a' } },
      { kind: 'mono', irs: irMono, url: 'ir:m', L: 8, blocks: 3, params: { offset: 5 } },
      { kind: 'mono', irs: irMono, url: 'ir:m', L: 8, blocks: 3, params: { offset: 7 } },
      { kind: 'mono', irs: irMono, url: 'ir:m', L: 8, blocks: 3, params: { length: 3 } },
      { kind: 'mono', irs: irLong, url: 'ir:l', L: 16, blocks: 5, params: { offset: 3, length: 33, id: 'z' } },
      { kind: 'stereo', irs: irStereo, url: 'ir:s', L: 8, blocks: 4, params: {} },
      { kind: 'stereo', irs: irStereo, url: 'ir:s', L: 8, blocks: 4, params: { offset: 2, id: 'b' } },
      { kind: 'stereo', irs: irStereo, url: 'ir:s', L: 8, blocks: 3, params: { offset: 5 } },
      { kind: 'stereo', irs: irStereo, url: 'ir:s', L: 8, blocks: 3, params: { offset: 7, length: 4 } },
      { kind: 'stereo', irs: irLong, url: 'ir:l', L: 16, blocks: 5, params: { offset: 0, id: 'w' } },
      { kind: 'stereo', irs: irLong, url: 'ir:l', L: 24, blocks: 4, params: { offset: 40 } },
    ]
    for (const sc of scenarios) {
      const store = new Map(), log = []
      const nch = sc.kind === 'stereo' ? 2 : 1
      // deep-copy the IR so `length` truncation (impulse[0] = subarray) is visible
      const irs = {}; for (const u in sc.irs) irs[u] = sc.irs[u].map(ch => f32(ch))
      const mk = n => makeContext(Array.from({ length: nch }, (_, ch) => noise(sc.L, 100 + ch, n)), n, irs, store, log)
      const c0 = mk(0)
      const params = Object.assign({ url: sc.url }, sc.params)
      const fn = sc.kind === 'stereo' ? await impulseReverbStereo(c0, params) : await impulseReverb(c0, params)
      const blocks = []
      for (let b = 0; b < sc.blocks; b++) {
        const c = b === 0 ? c0 : mk(b * sc.L)
        const before = log.length
        let ret = fn(c)
        if (nch === 1) ret = [ret]
        blocks.push({
          n: c.n, input: c.buffer.map(bits),
          ret: ret.map(v => ({ length: v.length, byte_offset: v.byteOffset, buffer_bytes: v.buffer.byteLength, data: bits(v), full: bits(new Float32Array(v.buffer)) })),
          set_calls: log.slice(before)
        })
        // what was stored as the carried tail
        const last = log[log.length - 1]
        const pv = store.get(last[1])
        const pvs = nch === 1 ? [pv] : pv
        blocks[blocks.length - 1].prev = pvs.map(v => ({ length: v.length, byte_offset: v.byteOffset, data: bits(v) }))
      }
      // resume: a new closure created at n = blocks*L picks the stored tail up
      const cR = mk(sc.blocks * sc.L)
      const fn2 = sc.kind === 'stereo' ? await impulseReverbStereo(cR, params) : await impulseReverb(cR, params)
      let r2 = fn2(cR); if (nch === 1) r2 = [r2]
      runs.push({
        kind: sc.kind, L: sc.L, params, ir: sc.irs[sc.url].map(bits), blocks,
        resumed: { n: cR.n, input: cR.buffer.map(bits), ret: r2.map(v => ({ length: v.length, byte_offset: v.byteOffset, data: bits(v) })) },
        log
      })
    }
    save('g3_reverb_blocks.json', { runs })
  }

  // ---------------- G4: mixBuffers, all branches ----------------
  {
    const cases = []
    const T = (ch, n, seed) => Array.from({ length: ch }, (_, c) => noise(n, seed + c))
    const run = (name, target, sources, desc) => {
      const tIn = target.map(bits)
      const ret = mixBuffers(target, ...sources)
      cases.push({ name, target_in: tIn, sources: desc, target_out: target.map(bits), returns_target: ret === target })
    }
    const describe = s => {
      if (Array.isArray(s[0])) {
        const o = { buffer: s[0].map(bits) }
        if (s.length > 1 && s[1] !== undefined) o.length = s[1]
        if (s.length > 2 && s[2] !== undefined) o.volume = (typeof s[2] === 'function') ? { fn: s[2].desc } : s[2]
        if (s.length > 3 && s[3] !== undefined) o.offset = s[3]
        o.tuple = true
        return o
      }
      return { buffer: s.map(bits), tuple: false }
    }
    const both = (name, tch, tl, srcs) => run(name, T(tch, tl, 300), srcs, srcs.map(describe))
    const ramp = x => 0.25 + x / 8; ramp.desc = 'ramp'        // vol function: 0.25 + x/8
    const alt = x => (x & 1) ? -1.3 : 0.7; alt.desc = 'alt'    // vol function: x odd ? -1.3 : 0.7
    for (const tch of [1, 2]) for (const sch of [1, 2]) {
      const tag = `${sch}to${tch}`
      both(`plain_${tag}_same_len`, tch, 16, [T(sch, 16, 310)])
      both(`plain_${tag}_short_src_wraps`, tch, 16, [T(sch, 5, 320)])
      both(`plain_${tag}_long_src`, tch, 7, [T(sch, 19, 330)])
      both(`tuple_${tag}_vol_offset`, tch, 16, [[T(sch, 16, 340), 1, 1.3, 5]])
      both(`tuple_${tag}_len_0.75`, tch, 16, [[T(sch, 8, 350), .75, .5, 5]])
      both(`tuple_${tag}_len_2.5_sl_gt_rl`, tch, 16, [[T(sch, 6, 360), 2.5, 0.3, 3]])
      both(`tuple_${tag}_defaults`, tch, 12, [[T(sch, 12, 370), 1]])
      both(`tuple_${tag}_big_offset`, tch, 16, [[T(sch, 11, 380), 1, 0.9, 2 ** 31 + 12345]])
      both(`tuple_${tag}_huge_offset`, tch, 16, [[T(sch, 11, 385), 1, 0.9, 2 ** 40 + 7]])
      both(`tuple_${tag}_neg_offset_nan`, tch, 16, [[T(sch, 8, 390), 1, 1, -3]])
      both(`tuple_${tag}_neg_offset_multiple`, tch, 16, [[T(sch, 4, 395), 1, 1, -8]])
      both(`tuple_${tag}_len_undefined_nan`, tch, 8, [[T(sch, 8, 400)]])
      both(`tuple_${tag}_len_zero_nan`, tch, 8, [[T(sch, 8, 405), 0, 1, 0]])
      both(`tuple_${tag}_volfn_ramp`, tch, 16, [[T(sch, 16, 410), 1, ramp, 2]])
      both(`tuple_${tag}_volfn_alt_wrap`, tch, 16, [[T(sch, 5, 420), 1, alt, 1]])
      both(`multi_${tag}_three_sources_in_order`, tch, 16, [T(sch, 16, 430), [T(sch, 7, 440), 1, 1.3, 96000], [T(sch, 16, 450), .5, 1 / 3, 1]])
    }
    // channel counts the reference ignores
    both('ignored_3ch_source', 2, 8, [T(3, 8, 460)])
    // the one reference test that reaches mixBuffers (test/mix.js:264-286): a
    // 2-sample child buffer folded into a 4-sample parent
    run('ref_test_mix_deep_shape', [f32([1, 2, 3, 4])], [[f32([1, 2])]], [{ buffer: [bits(f32([1, 2]))], tuple: false }])
    // usage evidence: test/fixtures/miss-judged.js:26-33  c.mix(target, [buf,1,1.3,c.n], ...)
    both('usage_miss_judged', 2, 32, [[T(2, 32, 470), 1, 1.3, 192000], [T(2, 32, 480), 1, 1.3, 192000], [T(2, 32, 490), 1, .5, 192000]])
    save('g4_mix_buffers.json', { cases })
  }

  // ---------------- G5: config-sized runs ----------------
  {
    const out = {}
    // C1: mono impulseConvolve, 1 s 44.1 kHz noise x 0.5 s IR
    {
      const L = 44100, M = 22050
      const ir = synthIr(M, 500000)
      const x = noise(L, 1000)
      const store = new Map(), log = []
      const c = makeContext([x], 0, { 'ir:c1': [ir] }, store, log)
      const go = await impulseConvolve(c, 'ir:c1', -1)
      const y = go(x)
      const kernel = store.get('kernel:ir:c1:44100:-1')
      out.c1 = {
        L, M, seed_x: 1000, seed_ir: 500000, out_len: y.length, stride: 16,
        samples: bits(strided(y, 16)), stats: stats(y), kernel_len: kernel.length,
        kernel_checksum: checksum(kernel), out_checksum: checksum(y), cache_log: log,
        ir_checksum: checksum(ir), x_checksum: checksum(x)
      }
    }
    // C2: stereo impulseReverbStereo, 4 bars of 96000 frames, 2 s stereo IR, offset 0 and 1000
    for (const offset of [0, 1000]) {
      const L = 96000, M = 96000
      const irs = { 'ir:c2': [synthIr(M, 500000), synthIr(M, 500001)] }
      const store = new Map(), log = []
      const mk = n => makeContext([noise(L, 1000, n), noise(L, 1001, n)], n, irs, store, log)
      const c0 = mk(0)
      const fn = await impulseReverbStereo(c0, { url: 'ir:c2', offset, id: 'c2' })
      const bars = []
      for (let b = 0; b < 4; b++) {
        const c = b === 0 ? c0 : mk(b * L)
        const r = fn(c)
        bars.push({
          n: c.n, ret_len: [r[0].length, r[1].length], byte_offset: [r[0].byteOffset, r[1].byteOffset],
          buffer_bytes: r[0].buffer.byteLength, stride: 64,
          samples: [bits(strided(r[0], 64)), bits(strided(r[1], 64))],
          stats: [stats(r[0]), stats(r[1])]
        })
      }
      out['c2_offset' + offset] = { L, M, offset, seeds_x: [1000, 1001], seeds_ir: [500000, 500001], bars, keys: log.filter(e => e[0] === 'set').map(e => e[1]) }
    }
    // C3-shaped single channel-job: L = 2 880 000, M = 96 000 (track 0, ch 0)
    {
      const L = 2880000, M = 96000
      const ir = synthIr(M, 500000)
      const x = noise(L, 1000)
      const k = convolve.fftProcessKernel(L, ir)
      const y = convolve.fftConvolution(L, k, M)(x)
      out.c3_job0 = { L, M, seed_x: 1000, seed_ir: 500000, out_len: y.length, stride: 1024, samples: bits(strided(y, 1024)), stats: stats(y), head: bits(y.subarray(0, 64)), tail: bits(y.subarray(y.length - 64)) }
    }
    save('g5_config_sized.json', out)
  }

  // ---------------- G6: ring buffer + loop integers ----------------
  {
    // test/ring-buffer.js:14-39 replayed, plus reads at keys -7..7 after each write
    const buffer = new Float32Array(3)
    const ring = new RingBuffer(buffer)
    const steps = []
    for (const v of [1, 2, 3, 4, 5]) {
      ring[0] = v
      const reads = {}
      for (let k = -7; k <= 7; k++) reads[k] = ring[k]
      steps.push({ write: v, buffer: Array.from(buffer), reads })
    }
    const b5 = new Float32Array(5), r5 = new RingBuffer(b5)
    const steps5 = []
    const script = [[0, 1], [0, 2], [2, 9], [-1, 8], [0, 3], [7, 6], [-9, 5], [0, 4], [0, 7], [0, 11]]
    for (const [k, v] of script) { r5[k] = v; steps5.push({ key: k, value: v, buffer: Array.from(b5) }) }
    // loop integers (src/loop-node.js:145-147, :48-50)
    const beat = []
    for (const [sr, bpm] of [[48000, 120], [44100, 120], [44100, 125], [48000, 125], [44100, 137], [96000, 93.5], [22050, 174]]) {
      const beatRate = Math.round(sr * (60 / bpm))
      beat.push({ sr, bpm, beat_rate: beatRate, buffer_size: beatRate * 4, bpm_back: parseFloat((60 * (sr / beatRate)).toFixed(6)) })
    }
    save('g6_ring_and_loop.json', { ring3: steps, ring5: steps5, beat })
  }
}

main().catch(e => { console.error(e); process.exit(1) })
