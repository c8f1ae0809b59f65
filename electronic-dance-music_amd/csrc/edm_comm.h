// edm_comm.h -- the collectives of the multi-GPU hill exchange behind one small interface, so that the exchange
// protocol of edm_bias.cpp / apply_hills (counts, packets, padded records, integral slices, delta grids: what replaces
// EDMBias::flush_buffers / update_height, edm_bias.cpp:630-706, :922-931) is ONE code path whatever carries the
// bytes.  Two carriers:
//   RCCL over xGMI (production)        -- ncclAllGather / ncclAllReduce queued on the handle's stream
//   POSIX shared memory (host-staged)   -- ranks on one host without RCCL peer access: device -> host slot,
//                                          barrier, host -> device.  It is what lets the REAL exchange code run with
//                                          two or more ranks on a one-GPU box (tests/test_gpu_two_ranks.py), and a
//                                          debugging aid; the payloads and their order are identical.
// Both deliver the same bits to every rank (the sum of an all-reduce is formed in rank order by the host carrier
// and by RCCL's ring alike on all ranks), which the replicated-grid design relies on.
#pragma once

#include <hip/hip_runtime_api.h>

#include <stddef.h>

namespace edm {

struct Transport {
  virtual ~Transport() {}
  virtual int nranks() const = 0;
  virtual int rank() const = 0;
  virtual const char *name() const = 0;
  // d_recv receives nranks blocks of `bytes` bytes, rank-major; device pointers; ordered on stream s.
  // Returns EDM_HIP_OK or EDM_HIP_ERR_COMM (message in edm_hip_last_error()).
  virtual int all_gather(const void *d_send, void *d_recv, size_t bytes, hipStream_t s) = 0;
  // in-place sum over the ranks of `count` doubles
  virtual int all_reduce_sum(double *d_buf, size_t count, hipStream_t s) = 0;
};

// id_bytes: the 128-byte ncclUniqueId rank 0 created (edm_hip_comm_unique_id)
int make_rccl_transport(const void *id_bytes, int nranks, int rank, Transport **out);
// shm_name: a name unique to this job ("/edm_job_1234"); every rank passes the same one
int make_shm_transport(const char *shm_name, int nranks, int rank, Transport **out);

}  // namespace edm
