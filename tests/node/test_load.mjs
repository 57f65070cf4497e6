// CPU-side check (no GPU needed): the shims import, export the reference's
// names, and fail loudly (no JavaScript fallback) when there is no device.
import convolve from '../../dsp_amd/js/lib/convolve.js'
import impulseConvolve from '../../dsp_amd/js/lib/impulse-convolve.js'
import impulseConvolveStereo from '../../dsp_amd/js/lib/impulse-convolve-stereo.js'
import impulseReverb from '../../dsp_amd/js/lib/impulse-reverb.js'
import impulseReverbStereo from '../../dsp_amd/js/lib/impulse-reverb-stereo.js'
import mixBuffers from '../../dsp_amd/js/src/mix-buffers.js'
import native from '../../dsp_amd/js/native.js'

const names = Object.keys(convolve).sort().join(',')
if (names !== 'directConvolution,fftConvolution,fftProcessKernel') throw new Error('convolve exports: ' + names)
for (const f of [impulseConvolve, impulseConvolveStereo, impulseReverb, impulseReverbStereo, mixBuffers]) {
  if (typeof f !== 'function') throw new Error('default export is not a function')
}
if (native.fftLength(44100, 22050) !== 131072) throw new Error('fftLength')
if (native.beatRate(48000, 120) !== 24000) throw new Error('beatRate')
let gpus = 0
try { gpus = native.init() } catch (e) { gpus = 0 }
if (gpus === 0) {
  let threw = false
  try { mixBuffers([new Float32Array(4)], [new Float32Array(4)]) } catch (e) { threw = /HIP device|no GPU/i.test(e.message) }
  if (!threw) throw new Error('mixBuffers did not fail loudly without a GPU')
  threw = false
  try { convolve.fftProcessKernel(8, new Float32Array(3)) } catch (e) { threw = true }
  if (!threw) throw new Error('fftProcessKernel did not fail loudly without a GPU')
}
console.log('ok load gpus=' + gpus)
