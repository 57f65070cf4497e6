// gen_golden3.mjs — third batch of fixtures from the REAL reference (imported read-only, in place, from
// /root/reference via oracle/ref_loader.mjs).  Container-only: the reference never travels; only these
// vectors do.
//
//   node --harmony-nullish --harmony-optional-chaining \
//        --experimental-loader ./oracle/ref_loader.mjs oracle/gen_golden3.mjs
//
// G11: mixBuffers with sources that share memory with the target (the same arrays, views into them, views
// that reach beyond them).  A channel is described by the base array it lives in and its range there, so a
// test can rebuild the same sharing.  Every float is recorded as its IEEE-754 binary32 bit pattern.
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'
import mixBuffers from 'file:///root/reference/src/mix-buffers.js'

const here = path.dirname(fileURLToPath(import.meta.url))
const outDir = path.join(here, '..', 'tests', 'golden')
const bits = a => Array.from(new Uint32Array(a.buffer, a.byteOffset, a.length))
const fmix32 = h => {
  h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13
  h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16
  return h >>> 0
}
const noise = (n, seed) => {
  const k = Math.imul(seed, 0x9E3779B9) >>> 0
  const out = new Float32Array(n)
  for (let i = 0; i < n; i++) out[i] = ((fmix32((i ^ k) >>> 0) >>> 8) - 8388608) / 8388608
  return out
}
const volFns = { ramp: x => 0.25 + x / 8, alt: x => (x & 1) ? -1.3 : 0.7 }

// a case: bases = sizes of the base arrays (filled with noise of seed 1100 + index); target = channel specs
// [base, begin, end]; sources = plain {ch: [specs]} or tuple {ch, length, volume, offset}
const cases = [
  { name: 't_a_t', bases: [16, 16, 16, 16], target: [[0, 0, 16], [1, 0, 16]], sources: [{ ch: [[2, 0, 16], [3, 0, 16]] }, { ch: [[0, 0, 16], [1, 0, 16]] }] },
  { name: 't_t_a_t', bases: [12, 12], target: [[0, 0, 12]], sources: [{ ch: [[0, 0, 12]] }, { ch: [[1, 0, 12]] }, { ch: [[0, 0, 12]] }] },
  { name: 'feedback_neg_offset', bases: [24], target: [[0, 0, 24]], sources: [{ ch: [[0, 0, 24]], length: 1, volume: 0.5, offset: -3 }] },
  { name: 'forward_offset_wraps', bases: [24], target: [[0, 0, 24]], sources: [{ ch: [[0, 0, 24]], length: 1, volume: 0.5, offset: 3 }] },
  { name: 'view_inside_target', bases: [16], target: [[0, 0, 16]], sources: [{ ch: [[0, 2, 10]] }] },
  { name: 'view_inside_target_loop', bases: [32, 32], target: [[0, 0, 32], [1, 0, 32]], sources: [{ ch: [[0, 4, 12], [1, 20, 28]], length: 0.75, volume: 1.3, offset: 5 }] },
  { name: 'channels_swapped', bases: [16, 16], target: [[0, 0, 16], [1, 0, 16]], sources: [{ ch: [[1, 0, 16], [0, 0, 16]] }] },
  { name: 'mono_source_is_left', bases: [16, 16], target: [[0, 0, 16], [1, 0, 16]], sources: [{ ch: [[0, 0, 16]], length: 1, volume: 0.8, offset: 0 }] },
  { name: 'mono_source_is_right_delayed', bases: [16, 16], target: [[0, 0, 16], [1, 0, 16]], sources: [{ ch: [[1, 0, 16]], length: 1, volume: 0.8, offset: 15 }] },
  { name: 'stereo_to_mono_with_self', bases: [20, 20], target: [[0, 0, 20]], sources: [{ ch: [[0, 0, 20], [1, 0, 20]] }, { ch: [[1, 0, 20], [0, 0, 20]], length: 1, volume: 0.5, offset: 7 }] },
  { name: 'volume_function_on_self', bases: [16], target: [[0, 0, 16]], sources: [{ ch: [[0, 0, 16]], length: 1, volume: { fn: 'alt' }, offset: 1 }] },
  { name: 'source_reaches_before_target', bases: [40], target: [[0, 8, 32]], sources: [{ ch: [[0, 0, 24]] }] },
  { name: 'source_reaches_behind_target', bases: [40, 40], target: [[0, 0, 24], [1, 4, 28]], sources: [{ ch: [[0, 10, 40], [1, 0, 30]], length: 0.5, volume: 1, offset: 2 }, { ch: [[1, 6, 30], [0, 2, 26]] }] },
  { name: 'target_channels_in_one_base', bases: [64], target: [[0, 0, 24], [0, 32, 56]], sources: [{ ch: [[0, 16, 48]], length: 1, volume: 0.9, offset: 4 }] },
  { name: 'plain_between_selves', bases: [4096, 4096, 4096], target: [[0, 0, 4096], [1, 0, 4096]], sources: [{ ch: [[2, 0, 4096]] }, { ch: [[0, 0, 4096], [1, 0, 4096]], length: 1, volume: 0.25, offset: 1000 }, { ch: [[2, 0, 4096], [2, 0, 4096]], length: 0.5, volume: -0.5, offset: 77 }, { ch: [[1, 0, 4096]] }] },
]

const out = []
for (const c of cases) {
  const bases = c.bases.map((n, i) => noise(n, 1100 + i))
  const basesIn = bases.map(bits)
  const view = ([b, lo, hi]) => bases[b].subarray(lo, hi)
  const target = c.target.map(view)
  const sources = c.sources.map(s => {
    const ch = s.ch.map(view)
    if (s.length === undefined) return ch
    const vol = (s.volume && s.volume.fn) ? volFns[s.volume.fn] : s.volume
    return [ch, s.length, vol, s.offset]
  })
  const ret = mixBuffers(target, ...sources)
  if (ret !== target) throw new Error('mixBuffers must return its target')
  out.push({ name: c.name, bases_in: basesIn, target: c.target, sources: c.sources, bases_out: bases.map(bits) })
}
fs.writeFileSync(path.join(outDir, 'g11_mix_shared_memory.json'), JSON.stringify({ cases: out }))
console.log('wrote g11_mix_shared_memory.json', out.length, 'cases')
