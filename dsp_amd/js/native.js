// Loads the N-API addon (dsp_amd/napi/wpdsp.node -> libwpdsp.so).  There is no
// JavaScript fallback: if the addon is missing or no GPU is present the calls
// below throw.
import { createRequire } from 'module'
const require = createRequire(import.meta.url)
const native = require('../napi/wpdsp.node')
export default native
