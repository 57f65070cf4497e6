// gen_golden2.mjs — second batch of fixtures from the REAL reference (imported
// read-only, in place, from /root/reference via oracle/ref_loader.mjs), written
// beside the first batch (gen_golden.mjs) without touching it.  Container-only:
// the reference never travels; only these vectors do.
//
//   node --harmony-nullish --harmony-optional-chaining \
//        --experimental-loader ./oracle/ref_loader.mjs oracle/gen_golden2.mjs
//
// Every float is recorded as its IEEE-754 binary32 bit pattern (uint32).
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'
import convolve from 'file:///root/reference/lib/convolve.js'
import mixBuffers from 'file:///root/reference/src/mix-buffers.js'
import render from 'file:///root/reference/src/render.js'
import LoopNode from 'file:///root/reference/src/loop-node.js'

const here = path.dirname(fileURLToPath(import.meta.url))
const outDir = path.join(here, '..', 'tests', 'golden')
const bits = a => Array.from(new Uint32Array(a.buffer, a.byteOffset, a.length))
const f32 = a => new Float32Array(a)
const save = (name, obj) => {
  fs.writeFileSync(path.join(outDir, name), JSON.stringify(obj))
  console.log('wrote', name, fs.statSync(path.join(outDir, name)).size, 'bytes')
}
const fmix32 = h => {
  h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13
  h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16
  return h >>> 0
}
const noise = (n, seed, start = 0) => {
  const k = Math.imul(seed, 0x9E3779B9) >>> 0
  const out = new Float32Array(n)
  for (let i = 0; i < n; i++) out[i] = ((fmix32(((start + i) ^ k) >>> 0) >>> 8) - 8388608) / 8388608
  return out
}
const synthIr = (M, seed) => {
  const k = Math.imul(seed, 0x9E3779B9) >>> 0
  const r = Math.fround(1 - 6.907755 / M)
  const out = new Float32Array(M)
  let e = 1, norm = 0
  for (let i = 0; i < M; i++) {
    const nz = ((fmix32((i ^ k) >>> 0) >>> 8) - 8388608) / 8388608
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

const main = async () => {
  // ---- G7: channel-jobs of the 2048-track render: L = 2 880 000, 4 s IR (M = 192 000),
  // a different impulse response per job (track t, channel c: x seed 1000+2t+c, IR seed 500000+2t+c)
  {
    const L = 2880000, M = 192000
    const jobs = []
    for (const [t, c] of [[0, 0], [1, 1]]) {
      const sx = 1000 + 2 * t + c, sir = 500000 + 2 * t + c
      const ir = synthIr(M, sir)
      const x = noise(L, sx)
      const k = convolve.fftProcessKernel(L, ir)
      const y = convolve.fftConvolution(L, k, M)(x)
      jobs.push({ track: t, ch: c, seed_x: sx, seed_ir: sir, out_len: y.length, kernel_len: k.length, stride: 1024,
        samples: bits(strided(y, 1024)), stats: stats(y), head: bits(y.subarray(0, 64)), tail: bits(y.subarray(y.length - 64)),
        mid: { at: 1500000, data: bits(y.subarray(1500000, 1500000 + 256)) } })
    }
    save('g7_c5_jobs.json', { L, M, jobs })
  }

  // ---- G8: mixBuffers, negative offsets with the loop length above the real length
  // ((x+o) % sl % rl can be -0, which names element 0), all four layouts
  {
    const cases = []
    const T = (ch, n, seed) => Array.from({ length: ch }, (_, c) => noise(n, seed + c))
    for (const tch of [1, 2]) for (const sch of [1, 2]) {
      for (const [name, rl, len, vol, o, tl] of [['neg_o_sl_2rl', 5, 2, 1, -5, 16], ['neg_o_sl_2rl_far', 5, 2, 0.5, -20, 24],
        ['neg_o_sl_3rl', 4, 3, 1.3, -7, 20], ['neg_o_sl_lt_rl', 8, 0.5, 1, -6, 16], ['neg_o_sl_2.5rl', 6, 2.5, 0.7, -12, 32]]) {
        const target = T(tch, tl, 900), src = T(sch, rl, 910)
        const tIn = target.map(bits)
        mixBuffers(target, [src, len, vol, o])
        cases.push({ name: `${name}_${sch}to${tch}`, target_in: tIn, sources: [{ buffer: src.map(bits), length: len, volume: vol, offset: o, tuple: true }],
          target_out: target.map(bits) })
      }
    }
    save('g8_mix_negative_offsets.json', { cases })
  }

  // ---- G9: buffer conventions of src/render.js:27-48 (mono function into a stereo buffer:
  // /2 and duplicate; stereo function into a mono buffer: average), with a context stub that
  // has the members render() touches (buffer, n, update, tick, tickBar)
  {
    const n = 24
    const mono = noise(n, 950), left = noise(n, 951), right = noise(n, 952)
    const mk = nch => ({ buffer: Array.from({ length: nch }, () => new Float32Array(n)), n: 0, i: 0,
      update () {}, tick () { this.n++; this.i++ }, tickBar () {} })
    const cases = []
    for (const nch of [1, 2]) {
      let c = mk(nch)
      await render(ctx => mono[ctx.i], c)
      cases.push({ fn: 'mono', channels: nch, n_after: c.n, out: c.buffer.map(bits) })
      c = mk(nch)
      await render(ctx => [left[ctx.i], right[ctx.i]], c)
      cases.push({ fn: 'stereo', channels: nch, n_after: c.n, out: c.buffer.map(bits) })
    }
    const errors = {}
    try { await render(() => NaN, mk(1)) } catch (e) { errors.nan = [e.constructor.name, e.message] }
    try { await render(() => 1, { buffer: [f32(4), f32(4), f32(4)], n: 0, update () {} }) } catch (e) { errors.three_channels = [e.constructor.name, e.message] }
    save('g9_render_conventions.json', { frames: n, mono: bits(mono), left: bits(left), right: bits(right), cases, errors })
  }

  // ---- G10: the bpm getter of src/loop-node.js:9-16 through the class itself, including
  // values whose sixth decimal is a tie in binary
  {
    const rows = []
    for (const [sr, bpm] of [[48000, 120], [44100, 120], [44100, 125], [48000, 125], [44100, 137], [96000, 93.5], [22050, 174],
      [48000, 43.9453125], [48000, 117.1875], [44100, 60.5], [192000, 33.333333], [8000, 999], [48000, 0.7]]) {
      const node = new LoopNode()
      node.context = { sampleRate: sr, currentTime: 0 }
      node._bpm = bpm
      rows.push({ sr, bpm, beat_rate: node.beatRate, bpm_back: node.bpm, buffer_size: node.bufferSize })
    }
    // toFixed(6) ties straight from the expression, for beat rates that make 60*sr/beatRate a multiple of 2^-7
    const ties = []
    for (const [sr, br] of [[48000, 65536], [48000, 32768], [44100, 16384], [48000, 24576], [48000, 3], [44100, 7], [96000, 131072]]) {
      ties.push({ sr, beat_rate: br, bpm_back: parseFloat((60 * (sr / br)).toFixed(6)) })
    }
    save('g10_loop_bpm.json', { rows, ties })
  }
}

main().catch(e => { console.error(e); process.exit(1) })
