// The addon is context-aware: load it in two worker_threads at once, run the
// shims there, and compare with the main thread's results (needs a GPU).
// A kernel object made in the main thread is handed to the workers the way
// c.set/c.get would (structured clone of a SharedArrayBuffer-backed Float32Array).
import { Worker, isMainThread, parentPort, workerData } from 'worker_threads'
import { fileURLToPath } from 'url'
import convolve from '../../dsp_amd/js/lib/convolve.js'
import mixBuffers from '../../dsp_amd/js/src/mix-buffers.js'

const ramp = (n, k) => { const a = new Float32Array(n); for (let i = 0; i < n; i++) a[i] = ((i * 37 + k * 11) % 101) / 101 - 0.5; return a }
const work = (kernel, L, M, seed) => {
  const y = convolve.fftConvolution(L, kernel, M)(ramp(L, seed))
  const t = [new Float32Array(L), new Float32Array(L)]
  mixBuffers(t, [[y.subarray(0, L), y.subarray(M, M + L)], 1, 1.3, seed], [ramp(7, seed)])
  return [Array.from(new Uint32Array(y.buffer)), Array.from(new Uint32Array(t[0].buffer)), Array.from(new Uint32Array(t[1].buffer))]
}

if (isMainThread) {
  const L = 3000, M = 1700
  const kernel = convolve.fftProcessKernel(L, ramp(M, 5))
  const want = [1, 2].map(seed => JSON.stringify(work(kernel, L, M, seed)))
  let done = 0
  ;[1, 2].forEach(seed => {
    const w = new Worker(fileURLToPath(import.meta.url), { workerData: { kernel, L, M, seed } })
    w.on('message', got => {
      if (JSON.stringify(got) !== want[seed - 1]) { console.error('worker ' + seed + ' differs'); process.exit(1) }
      if (++done === 2) console.log('ok workers 2')
    })
    w.on('error', e => { console.error(e); process.exit(1) })
  })
} else {
  const { kernel, L, M, seed } = workerData
  parentPort.postMessage(work(kernel, L, M, seed))
}
