// Drop-in for lib/impulse-convolve.js:3-20: same cache key and get/set protocol.
import convolve from './convolve.js'

export default async (c, url, length) => {
  const impulse = await c.sample(url)
  if (length > -1) {
    impulse[0] = impulse[0].subarray(0, length)
  }
  const id = 'kernel:' + url + ':' + c.buffer[0].length + ':' + length
  let kernel = await c.get(id)
  if (kernel === false) {
    kernel = convolve.fftProcessKernel(c.buffer[0].length, impulse[0])
    await c.set(id, kernel)
  }
  const reverb = convolve.fftConvolution(c.buffer[0].length, kernel, impulse[0].length)
  return reverb
}
