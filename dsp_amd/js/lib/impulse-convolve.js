// Drop-in for lib/impulse-convolve.js (default export: async (c, url, length) => go).
export { impulseConvolve as default } from './impulse.js'
