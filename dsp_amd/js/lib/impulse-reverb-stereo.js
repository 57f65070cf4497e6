// Drop-in for lib/impulse-reverb-stereo.js (default export: async (c, params = c) => c => [viewL, viewR]).
export { impulseReverbStereo as default } from './impulse.js'
