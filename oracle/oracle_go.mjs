// oracle_go.mjs — the CPU oracle's convolution and mix restated in JavaScript
// (Node 12 syntax), so that a Node.js CPU baseline can be timed on machines the
// reference cannot travel to.
//
// TEST INFRASTRUCTURE ONLY: nothing in dsp_amd/ imports this file.  It restates
// oracle/wp_oracle.c (which follows lib/convolve.js:4-45,128-311,553-595 and
// src/mix-buffers.js:1-73 of the reference) operation for operation; under V8 the
// Float32Array loads/stores and Number arithmetic give the same f64-operator /
// f32-store sequence, so `check` below must reproduce the reference-generated
// fixtures in tests/golden/ bit for bit.
//
//   node oracle/oracle_go.mjs check            fixtures g1, g4, g5(c1), g8 -> "ok ..."
//   node oracle/oracle_go.mjs bench F M JOBS   one JSON line: Msamples/s of go() + mix on one thread
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'

const here = path.dirname(fileURLToPath(import.meta.url))

// ---- lib/convolve.js:521-530
export const nextPot = n => {
  if (n === 0) return 1
  let s = n - 1
  s |= s >> 1; s |= s >> 2; s |= s >> 4; s |= s >> 8; s |= s >> 16
  return s + 1
}

// ---- lib/convolve.js:4-45: twiddle table (cos, -sin as f32) and base-4 digit reversal
export const makePlan = N => {
  if (N <= 1 || (N & (N - 1)) !== 0) throw new Error('FFT size must be a power of two and bigger than 1')
  const floats = N << 1
  const tw = new Float32Array(floats)
  for (let i = 0; i < floats; i += 2) {
    const ang = Math.PI * i / N
    tw[i] = Math.cos(ang)
    tw[i + 1] = -Math.sin(ang)
  }
  let power = 0
  for (let t = 1; N > t; t <<= 1) power++
  const width = power % 2 === 0 ? power - 1 : power
  const nrev = 1 << (width - 1)          // the half of the table the transform reads
  const rev = new Int32Array(nrev)
  for (let j = 0; j < nrev; j++) {
    let r = 0
    for (let sh = 0; sh < width; sh += 2) {
      const up = width - sh - 2
      if (up >= 0) r |= ((j >> sh) & 3) << up
    }
    rev[j] = r
  }
  return { N, floats, tw, width, rev }
}

// ---- lib/convolve.js:128-311: out-of-place radix-4 transform, radix-2 or radix-4 leaves
const transform = (plan, dst, src, sign) => {
  const total = plan.floats, tw = plan.tw, rev = plan.rev
  let stride = 1 << plan.width
  let span = (total / stride) << 1
  if (span === 4) {
    for (let o = 0, t = 0; o < total; o += span, t++) {
      const p = rev[t]
      const er = src[p], ei = src[p + 1], or_ = src[p + stride], oi = src[p + stride + 1]
      dst[o] = er + or_; dst[o + 1] = ei + oi
      dst[o + 2] = er - or_; dst[o + 3] = ei - oi
    }
  } else {
    for (let o = 0, t = 0; o < total; o += span, t++) {
      const p = rev[t], s2 = stride * 2, s3 = stride * 3
      const ar = src[p], ai = src[p + 1], br = src[p + stride], bi = src[p + stride + 1]
      const cr = src[p + s2], ci = src[p + s2 + 1], dr = src[p + s3], di = src[p + s3 + 1]
      const u0r = ar + cr, u0i = ai + ci, u1r = ar - cr, u1i = ai - ci
      const u2r = br + dr, u2i = bi + di
      const u3r = sign * (br - dr), u3i = sign * (bi - di)
      dst[o] = u0r + u2r; dst[o + 1] = u0i + u2i
      dst[o + 2] = u1r + u3i; dst[o + 3] = u1i - u3r
      dst[o + 4] = u0r - u2r; dst[o + 5] = u0i - u2i
      dst[o + 6] = u1r - u3i; dst[o + 7] = u1i + u3r
    }
  }
  for (stride >>= 2; stride >= 2; stride >>= 2) {
    span = (total / stride) << 1
    const q = span >> 2
    for (let o = 0; o < total; o += span) {
      const lim = o + q
      for (let i = o, k = 0; i < lim; i += 2, k += stride) {
        const a = i, b = a + q, c = b + q, d = c + q
        const ar = dst[a], ai = dst[a + 1], br = dst[b], bi = dst[b + 1]
        const cr = dst[c], ci = dst[c + 1], dr = dst[d], di = dst[d + 1]
        const wbr = tw[k], wbi = sign * tw[k + 1]
        const mbr = br * wbr - bi * wbi, mbi = br * wbi + bi * wbr
        const wcr = tw[2 * k], wci = sign * tw[2 * k + 1]
        const mcr = cr * wcr - ci * wci, mci = cr * wci + ci * wcr
        const wdr = tw[3 * k], wdi = sign * tw[3 * k + 1]
        const mdr = dr * wdr - di * wdi, mdi = dr * wdi + di * wdr
        const u0r = ar + mcr, u0i = ai + mci, u1r = ar - mcr, u1i = ai - mci
        const u2r = mbr + mdr, u2i = mbi + mdi
        const u3r = sign * (mbr - mdr), u3i = sign * (mbi - mdi)
        dst[a] = u0r + u2r; dst[a + 1] = u0i + u2i
        dst[b] = u1r + u3i; dst[b + 1] = u1i - u3r
        dst[c] = u0r - u2r; dst[c + 1] = u0i - u2i
        dst[d] = u1r - u3i; dst[d + 1] = u1i + u3r
      }
    }
  }
}

// ---- lib/convolve.js:588-595: zero-pad, real -> interleaved complex, forward transform
const paddedSpectrum = (plan, data) => {
  if (data.length > plan.N) throw new RangeError('offset is out of bounds')
  const z = new Float32Array(plan.floats)
  for (let i = 0; i < data.length; i++) z[2 * i] = data[i]
  const out = new Float32Array(plan.floats)
  transform(plan, out, z, 1)
  return out
}

// ---- lib/convolve.js:553-562
export const processKernel = (length, kernel) => paddedSpectrum(makePlan(nextPot(length + kernel.length)), kernel)

// ---- lib/convolve.js:564-586
export const convolution = (length, spectrum, kernelLength) => {
  const plan = makePlan(nextPot(length + kernelLength))
  const resultLength = length + kernelLength
  return input => {
    const x = paddedSpectrum(plan, input)
    for (let i = 0; i < plan.floats; i += 2) {
      const re = x[i] * spectrum[i] - x[i + 1] * spectrum[i + 1]
      x[i + 1] = x[i] * spectrum[i + 1] + x[i + 1] * spectrum[i]
      x[i] = re
    }
    const y = new Float32Array(plan.floats)
    transform(plan, y, x, -1)
    for (let i = 0; i < plan.floats; i++) y[i] /= plan.N     // :118-119, a pass of its own
    const out = new Float32Array(resultLength)
    for (let i = 0; i < resultLength && 2 * i < plan.floats; i++) out[i] = y[2 * i]
    return out
  }
}

// ---- lib/convolve.js:541-551
export const direct = (x, h) => {
  const out = new Float32Array(x.length + h.length - 1)
  for (let i = 0; i < x.length; i++) for (let j = 0; j < h.length; j++) out[i + j] += x[i] * h[j]
  return out
}

// ---- src/mix-buffers.js:1-73 (decoded source: channels, rl, sl, vol number or function, o)
export const mix = (target, ...sources) => {
  const tl = target[0].length
  for (const source of sources) {
    let ch, rl, sl, vol = 1, o = 0
    if (Array.isArray(source[0])) {
      ch = source[0]; rl = ch[0].length
      sl = (rl * source[1]) | 0
      if (source[2] !== undefined && source[2] !== null) vol = source[2]
      if (source[3] !== undefined && source[3] !== null) o = source[3]
    } else {
      ch = source; rl = sl = ch[0].length
    }
    const gain = typeof vol === 'function' ? vol : null
    const tc = target.length, sc = ch.length
    if ((tc !== 1 && tc !== 2) || (sc !== 1 && sc !== 2)) continue
    for (let x = 0; x < tl; x++) {
      const g = gain ? gain(x) : vol
      const i = (x + o) % sl % rl
      if (tc === 2 && sc === 2) { target[0][x] += ch[0][i] * g; target[1][x] += ch[1][i] * g }
      else if (tc === 2) { const h = (ch[0][i] / 2) * g; target[0][x] += h; target[1][x] += h }
      else if (sc === 2) target[0][x] += ((ch[0][i] + ch[1][i]) / 2) * g
      else target[0][x] += ch[0][i] * g
    }
  }
  return target
}

// ---- synthetic inputs (same definition as wpo_synth_* in oracle/wp_oracle.c)
const fmix32 = h => {
  h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13
  h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16
  return h >>> 0
}
export const noise = (n, seed, start = 0) => {
  const k = Math.imul(seed, 0x9E3779B9) >>> 0
  const out = new Float32Array(n)
  for (let i = 0; i < n; i++) out[i] = ((fmix32(((start + i) ^ k) >>> 0) >>> 8) - 8388608) / 8388608
  return out
}
export const synthIr = (M, seed) => {
  const k = Math.imul(seed, 0x9E3779B9) >>> 0
  const r = Math.fround(1 - 6.907755 / M)
  const out = new Float32Array(M)
  let e = 1, norm = 0
  for (let i = 0; i < M; i++) {
    out[i] = (((fmix32((i ^ k) >>> 0) >>> 8) - 8388608) / 8388608) * e
    norm += Math.abs(out[i])
    e = Math.fround(e * r)
  }
  for (let i = 0; i < M; i++) out[i] = out[i] / norm
  return out
}

// ---- command line
const fromBits = b => new Float32Array(new Uint32Array(b).buffer)
const sameBits = (a, bitsArr, what) => {
  const u = new Uint32Array(a.buffer, a.byteOffset, a.length)
  if (u.length !== bitsArr.length) throw new Error(what + ': length ' + u.length + ' != ' + bitsArr.length)
  for (let i = 0; i < u.length; i++) {
    if (u[i] !== bitsArr[i] && !(Number.isNaN(a[i]) && Number.isNaN(fromBits([bitsArr[i]])[0]))) {
      throw new Error(what + ': differs at ' + i)
    }
  }
}
const golden = name => JSON.parse(fs.readFileSync(path.join(here, '..', 'tests', 'golden', name), 'utf8'))

const check = () => {
  let n = 0
  for (const c of golden('g1_convolve_tiny.json').cases) {
    const x = fromBits(c.x), h = fromBits(c.h)
    const k = processKernel(c.L, h)
    sameBits(k, c.kernel, 'g1 kernel')
    sameBits(convolution(c.L, k, h.length)(x), c.out, 'g1 go')
    sameBits(direct(x, h), c.direct, 'g1 direct')
    n++
  }
  const fns = { ramp: x => 0.25 + x / 8, alt: x => (x & 1) ? -1.3 : 0.7 }
  for (const file of ['g4_mix_buffers.json', 'g8_mix_negative_offsets.json']) {
    for (const c of golden(file).cases) {
      const target = c.target_in.map(fromBits)
      const sources = c.sources.map(s => {
        const buf = s.buffer.map(fromBits)
        if (!s.tuple) return buf
        const v = s.volume && typeof s.volume === 'object' ? fns[s.volume.fn] : s.volume
        return [buf, s.length, v, s.offset]
      })
      mix(target, ...sources)
      c.target_out.forEach((b, ch) => sameBits(target[ch], b, file + ' ' + c.name))
      n++
    }
  }
  const c1 = golden('g5_config_sized.json').c1
  const y = convolution(c1.L, processKernel(c1.L, synthIr(c1.M, c1.seed_ir)), c1.M)(noise(c1.L, c1.seed_x))
  const strided = new Float32Array(Math.ceil(y.length / c1.stride))
  for (let i = 0, j = 0; i < y.length; i += c1.stride, j++) strided[j] = y[i]
  sameBits(strided, c1.samples, 'g5 c1')
  console.log('ok ' + (n + 1) + ' fixtures bit-exact')
}

const bench = (F, M, jobs) => {
  const irs = [synthIr(M, 500000), synthIr(M, 500001)]
  const gos = irs.map(h => convolution(F, processKernel(F, h), M))     // spectra cached, as the wrappers cache them
  const xs = []
  for (let j = 0; j < jobs; j++) xs.push(noise(F, 1000 + j))
  const t0 = process.hrtime.bigint()
  const ys = xs.map((x, j) => gos[j % 2](x))
  const bus = [new Float32Array(F), new Float32Array(F)]
  const srcs = []
  for (let t = 0; t + 1 < jobs; t += 2) srcs.push([[ys[t].subarray(0, F), ys[t + 1].subarray(0, F)], 1, 1 / 256, 0])
  mix(bus, ...srcs)
  const dt = Number(process.hrtime.bigint() - t0) / 1e9
  console.log(JSON.stringify({ value: jobs * (F + M) / dt / 1e6, node: process.version,
    sample: jobs + ' channel-jobs (L=' + F + ', M=' + M + ') + mixBuffers, IR spectra cached; ' + dt.toFixed(1) + ' s, JavaScript restatement of the reference under node' }))
}

const mode = process.argv[2]
if (mode === 'check') check()
else if (mode === 'bench') bench(parseInt(process.argv[3], 10), parseInt(process.argv[4], 10), parseInt(process.argv[5] || '2', 10))
