// Drop-in for lib/impulse-reverb.js (default export: async (c, {url, offset, length, id} = c) => c => view).
export { impulseReverb as default } from './impulse.js'
