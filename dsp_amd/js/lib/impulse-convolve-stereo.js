// Drop-in for lib/impulse-convolve-stereo.js:3-27 (both channels are built with
// impulse[0].length, :23-24, like the reference).
import convolve from './convolve.js'

export default async (c, url, length) => {
  const impulse = await c.sample(url)
  if (length > -1) {
    impulse[0] = impulse[0].subarray(0, length)
    impulse[1] = impulse[1].subarray(0, length)
  }
  const id = 'impulse-convolve-stereo:kernel:' + url + ':' + c.buffer[0].length + ':' + length
  let kernel = await c.get(id)
  if (kernel === false) {
    kernel = [
      convolve.fftProcessKernel(c.buffer[0].length, impulse[0]),
      convolve.fftProcessKernel(c.buffer[0].length, impulse[1])
    ]
    await c.set(id, kernel)
  }
  const reverb = [
    convolve.fftConvolution(c.buffer[0].length, kernel[0], impulse[0].length),
    convolve.fftConvolution(c.buffer[0].length, kernel[1], impulse[0].length)
  ]
  return reverb
}
