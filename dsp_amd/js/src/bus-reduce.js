// Multi-GPU master bus: one Node process per GPU (select the device with
// HIP_VISIBLE_DEVICES before the first call), each renders its share of the tracks
// into a DeviceBuffer bus with the drop-in modules, and ONE reduce sums the
// rank-local buses.  The reference has no counterpart beyond the fact that tracks
// are independent buffers (src/context.js:145-162) whose only coupling is their sum
// (src/mix-buffers.js:1-73); the sum across processes is RCCL over xGMI, reached
// through the addon (wp_comm_*, wp_bus_reduce in include/wpdsp.h).
//
//   rank 0:   const id = uniqueId()            // Uint8Array(128): send it to the other ranks
//   all:      const comm = createComm(id, ranks, rank)
//   per bar:  reduceBus(comm, bus)             // rank 0's bus now holds the sum
import native from '../native.js'
import { isDeviceBuffer } from './device-buffer.js'

export const uniqueId = () => native.commUniqueId()

export const createComm = (id, ranks, rank) => {
  if (!(id instanceof Uint8Array) || id.length !== 128) throw new TypeError('id must be the Uint8Array(128) made by uniqueId()')
  return { h: native.commCreate(id, ranks, rank), ranks, rank }
}

// Sum the ranks' buses onto rank `root` (in place); `all: true` leaves the sum on every rank.
// The cross-rank sum adds partial buses in RCCL's order: equal to the sequential mix to rounding.
export const reduceBus = (comm, bus, { root = 0, all = false } = {}) => {
  if (!isDeviceBuffer(bus)) throw new TypeError('reduceBus needs a DeviceBuffer')
  if (bus[0].off !== 0 || bus[0].length !== native.bufInfo(bus[0].h, 1)) throw new TypeError('reduceBus takes a whole buffer, not a view')
  native.busReduce(comm.h, bus[0].h, root, !!all)
  return bus
}

export default reduceBus
