// Drop-in for lib/impulse-convolve-stereo.js (default export: async (c, url, length) => [goL, goR]).
export { impulseConvolveStereo as default } from './impulse.js'
