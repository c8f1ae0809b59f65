// edm_comm.cpp -- carriers of the hill exchange's collectives (see edm_comm.h).
#include "edm_comm.h"

#include <fcntl.h>
#include <sched.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "edm_internal.h"

namespace edm {

// ---- RCCL over xGMI -------------------------------------------------------------
namespace {
struct RcclTransport : Transport {
  ncclComm_t comm = nullptr;
  int n = 1, r = 0;
  ~RcclTransport() override {
    if (comm) (void)ncclCommDestroy(comm);
  }
  int nranks() const override { return n; }
  int rank() const override { return r; }
  const char *name() const override { return "rccl"; }
  int all_gather(const void *d_send, void *d_recv, size_t bytes, hipStream_t s) override {
    if (ncclAllGather(d_send, d_recv, bytes, ncclChar, comm, s) != ncclSuccess) {
      set_error("ncclAllGather failed");
      return EDM_HIP_ERR_COMM;
    }
    return EDM_HIP_OK;
  }
  int all_reduce_sum(double *d_buf, size_t count, hipStream_t s) override {
    if (ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, comm, s) != ncclSuccess) {
      set_error("ncclAllReduce failed");
      return EDM_HIP_ERR_COMM;
    }
    return EDM_HIP_OK;
  }
};

// ---- host shared memory ------------------------------------------------------------
// One POSIX shared-memory object per job: a header with a sense-reversing barrier and nranks slots of SLOT bytes.
// A collective moves its payload in chunks of at most SLOT bytes per rank: device -> own slot, barrier, every slot ->
// device (all-gather) or the rank-ordered sum of the slots -> device (all-reduce), barrier.  A rank that waits longer
// than TIMEOUT_S at a barrier gives up with EDM_HIP_ERR_COMM instead of hanging its GPU.
struct ShmHeader {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
  int nranks;
  std::atomic<int> magic;   // written last by the creator: the object is ready
  char pad[128 - 4 * sizeof(std::atomic<int>) - sizeof(int)];
};
static const size_t SHM_SLOT = (size_t)4 << 20;
static const double SHM_TIMEOUT_S = 120.0;

struct ShmTransport : Transport {
  std::string shm;
  int n = 1, r = 0;
  char *base = nullptr;
  size_t map_bytes = 0;
  std::vector<double> acc;
  ShmHeader *hdr() const { return reinterpret_cast<ShmHeader *>(base); }
  char *slot(int k) const { return base + sizeof(ShmHeader) + (size_t)k * SHM_SLOT; }
  ~ShmTransport() override {
    if (base) munmap(base, map_bytes);
    if (r == 0 && !shm.empty()) shm_unlink(shm.c_str());
  }
  int nranks() const override { return n; }
  int rank() const override { return r; }
  const char *name() const override { return "host shared memory"; }
  int barrier(const char *what) {
    ShmHeader *h = hdr();
    const int gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
      h->arrived.store(0, std::memory_order_relaxed);
      h->generation.store(gen + 1, std::memory_order_release);
      return EDM_HIP_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (h->generation.load(std::memory_order_acquire) == gen) {
      if ((++spins & 1023) == 0) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > SHM_TIMEOUT_S) {
          set_error(std::string("host transport: a rank did not reach the barrier of ") + what + " within the time limit");
          return EDM_HIP_ERR_COMM;
        }
      }
    }
    return EDM_HIP_OK;
  }
  int all_gather(const void *d_send, void *d_recv, size_t bytes, hipStream_t s) override {
    EDM_HIP_TRY(hipStreamSynchronize(s));
    for (size_t off = 0; off < bytes || off == 0; off += SHM_SLOT) {
      const size_t len = bytes - off < SHM_SLOT ? bytes - off : SHM_SLOT;
      if (len) EDM_HIP_TRY(hipMemcpy(slot(r), static_cast<const char *>(d_send) + off, len, hipMemcpyDeviceToHost));
      int rc = barrier("all_gather (fill)");
      if (rc) return rc;
      for (int k = 0; k < n && len; k++)
        EDM_HIP_TRY(hipMemcpy(static_cast<char *>(d_recv) + (size_t)k * bytes + off, slot(k), len, hipMemcpyHostToDevice));
      rc = barrier("all_gather (drain)");
      if (rc) return rc;
      if (bytes == 0) break;
    }
    return EDM_HIP_OK;
  }
  int all_reduce_sum(double *d_buf, size_t count, hipStream_t s) override {
    EDM_HIP_TRY(hipStreamSynchronize(s));
    const size_t per = SHM_SLOT / sizeof(double);
    acc.resize(per);
    for (size_t off = 0; off < count || off == 0; off += per) {
      const size_t len = count - off < per ? count - off : per;
      if (len) EDM_HIP_TRY(hipMemcpy(slot(r), d_buf + off, len * sizeof(double), hipMemcpyDeviceToHost));
      int rc = barrier("all_reduce (fill)");
      if (rc) return rc;
      for (size_t i = 0; i < len; i++) {   // rank order: the same bits on every rank
        double t = reinterpret_cast<const double *>(slot(0))[i];
        for (int k = 1; k < n; k++) t += reinterpret_cast<const double *>(slot(k))[i];
        acc[i] = t;
      }
      if (len) EDM_HIP_TRY(hipMemcpy(d_buf + off, acc.data(), len * sizeof(double), hipMemcpyHostToDevice));
      rc = barrier("all_reduce (drain)");
      if (rc) return rc;
      if (count == 0) break;
    }
    return EDM_HIP_OK;
  }
};
}  // namespace

int make_rccl_transport(const void *id_bytes, int nranks, int rank, Transport **out) {
  *out = nullptr;
  RcclTransport *t = new RcclTransport;
  t->n = nranks;
  t->r = rank;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  if (ncclCommInitRank(&t->comm, nranks, id, rank) != ncclSuccess) {
    t->comm = nullptr;
    delete t;
    set_error("ncclCommInitRank failed");
    return EDM_HIP_ERR_COMM;
  }
  *out = t;
  return EDM_HIP_OK;
}

int make_shm_transport(const char *shm_name, int nranks, int rank, Transport **out) {
  *out = nullptr;
  if (!shm_name || shm_name[0] != '/' || nranks < 1 || rank < 0 || rank >= nranks) {
    set_error("host transport: the shared-memory name must start with '/' and 0 <= rank < nranks");
    return EDM_HIP_ERR_ARG;
  }
  ShmTransport *t = new ShmTransport;
  t->shm = shm_name;
  t->n = nranks;
  t->r = rank;
  t->map_bytes = sizeof(ShmHeader) + (size_t)nranks * SHM_SLOT;
  // Rank 0 creates the object -- removing whatever a crashed job left behind under the same name first, O_EXCL so that
  // nobody else's object is adopted -- sizes it (a fresh object is zero-filled: the barrier counters start at 0) and
  // writes its magic word LAST; the other ranks open without O_CREAT, retrying until the object exists, has its full
  // size and shows the magic word, and check the rank count the creator recorded.
  static const int SHM_MAGIC = 0x45444d31;   // "EDM1"
  const auto t0 = std::chrono::steady_clock::now();
  auto timed_out = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > SHM_TIMEOUT_S; };
  int fd = -1;
  if (rank == 0) {
    (void)shm_unlink(shm_name);
    fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)t->map_bytes) != 0) {
      if (fd >= 0) close(fd);
      delete t;
      set_error(std::string("host transport: cannot create the shared-memory object ") + shm_name);
      return EDM_HIP_ERR_COMM;
    }
  } else {
    for (;;) {
      fd = shm_open(shm_name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= t->map_bytes) break;
        close(fd);
        fd = -1;
      }
      if (timed_out()) {
        t->shm.clear();
        delete t;
        set_error(std::string("host transport: rank 0 did not create the shared-memory object ") + shm_name + " within the time limit");
        return EDM_HIP_ERR_COMM;
      }
      usleep(200);
    }
  }
  void *p = mmap(nullptr, t->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete t;
    set_error("host transport: mmap failed");
    return EDM_HIP_ERR_COMM;
  }
  t->base = static_cast<char *>(p);
  if (rank == 0) {
    t->hdr()->nranks = nranks;
    t->hdr()->magic.store(SHM_MAGIC, std::memory_order_release);
  } else {
    while (t->hdr()->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
      sched_yield();
      if (timed_out()) {
        delete t;
        set_error("host transport: the shared-memory object never became ready");
        return EDM_HIP_ERR_COMM;
      }
    }
    if (t->hdr()->nranks != nranks) {
      delete t;
      set_error("host transport: the ranks disagree about the rank count");
      return EDM_HIP_ERR_ARG;
    }
  }
  // wait until every rank has attached, so that rank 0's unlink at the end cannot precede an attach
  t->hdr()->attached.fetch_add(1, std::memory_order_acq_rel);
  while (t->hdr()->attached.load(std::memory_order_acquire) < nranks) {
    sched_yield();
    if (timed_out()) {
      delete t;
      set_error("host transport: not every rank attached within the time limit");
      return EDM_HIP_ERR_COMM;
    }
  }
  *out = t;
  return EDM_HIP_OK;
}

}  // namespace edm
