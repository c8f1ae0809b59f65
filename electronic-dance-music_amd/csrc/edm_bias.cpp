// edm_bias.cpp -- the EDMBias controller of libedm_hip.so: configuration, hill heights,
// bias limiting, overflow buffer, HILLS log and CV histogram bookkeeping on the
// host; every data-parallel step (selection, integrals, limiter walk, ordered
// gather, histogram, lookups) runs as a gfx950 kernel on the handle's stream.
// Mirrors lib/edm_bias.cpp of the reference (citations file:line); there is no
// CPU evaluation path for grid values anywhere in this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <map>
#include <string>
#include <vector>

#include "edm_comm.h"
#include "edm_internal.h"

using namespace edm;

#define BIAS_CLAMP 1.0  // edm_bias.h:14

// The per-rank HILLS log (edm_bias.cpp:586-599: one text line per hill event, eight decimals) written by a thread of
// its own: the step hands over the events as numbers -- a line costs ~1 us of printf, a W1 step logs ~125 of them,
// four times what the step's kernels take -- and the writer formats them, in order, and flushes the file after every
// add_hill cycle like the reference's std::endl does.  drain() returns once everything handed over is in the file.
struct HillEvent {
  long long steps;
  int hills_added;
  char type;
  double pos[3], height, added, cum_over_volume;
};
class HillWriter {
 public:
  HillWriter() : fp_(nullptr), dim_(1), stop_(false), busy_(false) {}
  ~HillWriter() { close(); }
  bool open(const char *path, unsigned dim) {
    close();
    fp_ = fopen(path, "w");
    dim_ = dim;
    if (!fp_) return false;
    stop_ = false;
    th_ = std::thread(&HillWriter::run, this);
    return true;
  }
  bool is_open() const { return fp_ != nullptr; }
  void submit(std::vector<HillEvent> &events) {   // (takes the events; leaves the vector empty)
    if (!fp_) {
      events.clear();
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      q_.emplace_back();
      q_.back().swap(events);
    }
    // (no notify: waking a sleeping thread is a system call, ~3 us on the step's critical path every hill step; the
    //  writer looks at the queue every millisecond by itself -- drain() and close() do wake it)
  }
  void drain() {
    if (!fp_) return;
    std::unique_lock<std::mutex> lk(m_);
    if (q_.empty() && !busy_) return;
    cv_.notify_one();
    idle_.wait(lk, [this] { return q_.empty() && !busy_; });
  }
  void close() {
    if (!fp_) return;
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_one();
    if (th_.joinable()) th_.join();
    fclose(fp_);
    fp_ = nullptr;
  }

 private:
  void run() {
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      cv_.wait_for(lk, std::chrono::milliseconds(1), [this] { return stop_ || !q_.empty(); });
      if (q_.empty()) {
        if (stop_) return;
        continue;
      }
      std::vector<HillEvent> batch;
      batch.swap(q_.front());
      q_.pop_front();
      busy_ = true;
      lk.unlock();
      for (size_t i = 0; i < batch.size(); i++) {
        const HillEvent &e = batch[i];
        fprintf(fp_, "%lld %c %d ", e.steps, e.type, e.hills_added);
        for (unsigned int d = 0; d < dim_; d++) fprintf(fp_, "%.8f ", e.pos[d]);
        fprintf(fp_, "%.8f %.8f %.8f\n", e.height, e.added, e.cum_over_volume);
      }
      fflush(fp_);
      lk.lock();
      busy_ = false;
      if (q_.empty()) idle_.notify_all();
    }
  }
  FILE *fp_;
  unsigned dim_;
  std::thread th_;
  std::mutex m_;
  std::condition_variable cv_, idle_;
  std::deque<std::vector<HillEvent> > q_;
  bool stop_, busy_;
};

// step_host's host side: the bias-force delta arrives in pieces (one event each) and is added into the caller's force
// array.  One core adds ~1 MB of delta in ~35 us -- as long as the link needs to deliver it -- so a few helper threads
// share every piece.
struct DeltaAddJob {
  double *h_f = nullptr;
  const double *dl = nullptr;
  long long n = 0, per = 0;
  int dim = 1, f_stride = 1, pieces = 1;
  hipEvent_t *ev = nullptr;
};
static double g_add_trace[64][4];   // development aid (EDM_HIP_TRACE=step_host): per thread, woken / first piece seen / waited / done
static double trace_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int delta_add_pieces(const DeltaAddJob &j, int me, int threads) {
  g_add_trace[me][0] = trace_now_us();
  g_add_trace[me][2] = 0;
  // every thread takes its share of EVERY piece: the copies stay few and large (the link's rate), the add of a piece is
  // spread over the threads
  for (int c = 0; c < j.pieces; c++) {
    const long long p0 = c * j.per, p1 = (p0 + j.per < j.n) ? p0 + j.per : j.n;
    const double tw = trace_now_us();
    hipError_t e = hipEventSynchronize(j.ev[c]);
    g_add_trace[me][2] += trace_now_us() - tw;
    if (c == 0) g_add_trace[me][1] = trace_now_us();
    g_add_trace[me][3] = 0;
    if (e != hipSuccess) return (int)e;
    if (p0 >= p1) continue;
    const long long share = (p1 - p0 + threads - 1) / threads;
    const long long i0 = p0 + me * share, i1 = (i0 + share < p1) ? i0 + share : p1;
    if (i0 >= i1) continue;
    // f[i][d] += delta[i][d]: the delta started from zero, so it holds exactly -dV/ds_d of the masked atoms and
    // (+0.0 or) 0 elsewhere -- the same doubles the reference's `forces[i][j] -= der[j]` subtracts
    if (j.dim == j.f_stride) {
      double *fp = j.h_f + (size_t)i0 * j.dim;
      const double *dp = j.dl + (size_t)i0 * j.dim;
      const size_t m = (size_t)(i1 - i0) * j.dim;
      for (size_t q = 0; q < m; q++) fp[q] += dp[q];
    } else {
      for (long long i = i0; i < i1; i++)
        for (int d = 0; d < j.dim; d++) j.h_f[(size_t)i * j.f_stride + d] += j.dl[(size_t)i * j.dim + d];
    }
  }
  g_add_trace[me][3] = trace_now_us();
  return 0;
}
class DeltaAddPool {
 public:
  DeltaAddPool() : gen_(0), stop_(false), left_(0), err_(0) {}
  ~DeltaAddPool() { close(); }
  int threads() const { return (int)th_.size() + 1; }
  void resize(int total, int device) {
    if (total < 1) total = 1;
    if (total == threads()) return;
    close();
    stop_ = false;
    for (int t = 1; t < total; t++) th_.emplace_back(&DeltaAddPool::run, this, t, total, device);
  }
  int run_job(const DeltaAddJob &j) {   // returns a hipError_t (0 = fine)
    const int total = threads();
    if (total > 1) {
      {
        std::lock_guard<std::mutex> lk(m_);
        job_ = j;
        left_.store(total - 1, std::memory_order_relaxed);
        err_.store(0, std::memory_order_relaxed);
        gen_++;
      }
      cv_.notify_all();
    }
    int rc = delta_add_pieces(j, 0, total);
    if (total > 1) {
      while (left_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
      if (!rc) rc = err_.load(std::memory_order_relaxed);
    }
    return rc;
  }
  void close() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_)
      if (t.joinable()) t.join();
    th_.clear();
  }

 private:
  void run(int me, int total, int device) {
    (void)hipSetDevice(device);
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      const DeltaAddJob j = job_;
      lk.unlock();
      const int rc = delta_add_pieces(j, me, total);
      if (rc) err_.store(rc, std::memory_order_relaxed);
      left_.fetch_sub(1, std::memory_order_release);
      lk.lock();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  unsigned long long gen_;
  bool stop_;
  DeltaAddJob job_;
  std::atomic<int> left_, err_;
};

struct edm_hip_bias {
  // public data members of EDMBias (edm_bias.h:118-157)
  int b_tempering = 0, b_targeting = 0;
  int mpi_rank = 0, mpi_size = 0;
  unsigned int dim = 0;
  double global_tempering = 0, bias_factor = 0, boltzmann_factor = 0, temperature = -1.0;
  double hill_prefactor = 0, bias_per_step = 0, hill_density = -1, cum_bias = 0, total_volume = 0;
  double expected_target = 0;
  int b_outofbounds = 0;
  std::vector<double> bias_dx, bias_sigma, min, max;
  std::vector<int> bper;
  edm_hip_gauss *bias = nullptr;
  edm_hip_grid *hist = nullptr;
  edm_hip_grid *target = nullptr;
  std::string initial_bias_file;
  const int *d_mask = nullptr;
  // private state (edm_bias.h:160-178)
  double temp_hill_cum = -1, temp_hill_prefactor = -1;
  long long est_hill_count = 0;
  int hills_added = 0;
  long long steps = 0;
  std::string hist_output, hills_name;
  // new hills whose log lines are still owed: the batch was released by its header line (every hill added in full) and
  // its positions / per-hill bias are fetched when the next add_hill cycle begins (resolve_deferred_log)
  struct DeferredLog {
    bool active = false;
    long long bound = 0, nh = 0, steps = 0;
    int hills_added_before = 0;
    double height = 0, cum_over_volume = 0;
  } deferred_log;
  HillWriter hills;                  // the HILLS log, written behind the step by a thread of its own
  std::vector<HillEvent> hill_events;   // events of the add_hill cycle in progress
  int hill_log = 1;
  std::vector<double> overflow;  // (dim+1) doubles per record, BIAS_BUFFER_SIZE + 1 records
  size_t overflow_left = 0, overflow_right = 0;
  int b_skip_hill_add = 0;
  // staged per-sample calls (pre_add_hill / add_hill / post_add_hill)
  std::vector<double> staged_x, staged_u;
  bool in_cycle = false;
  // device scratch owned by the controller
  DevBuf<long long> sel;
  long long *h_count = nullptr, *d_count = nullptr;  // host-mapped pinned: the selection kernel writes the count here
  DevBuf<long long> count_dev;                       // ... and here, for kernels consuming a deferred count
  DevBuf<int> sel_stage;                             // per-workgroup ordered lists of the chained selection
  DevBuf<int> sel_scratch;
  DevBuf<double> stage_x, stage_u, stage_h, tail_w;
  DevBuf<double> hx0;
  DevBuf<double> xchg_send, xchg_recv, xchg_all;
  long long xchg_counts[EDM_MAX_RANKS], xchg_est[EDM_MAX_RANKS];  // last synchronous exchange: per-rank hills / est_hill_count
  double *h_flush = nullptr;   // pinned staging of the overflow records handed to a flush
  double *d_flush = nullptr;   // its device-side address (host-mapped: read by launch_fetch_words)
  size_t flush_cap = 0;
  // device uniforms (edm_hip_bias_set_device_rng): add_hill cycles given no uniform array draw from a
  // counter-based stream keyed by (seed, cycle number, sample index)
  bool device_rng = false;
  unsigned long long rng_seed = 0, rng_cycle = 0;
  // device-resident neighbour list (edm_hip_bias_pair_list_upload / _step)
  DevBuf<int> pl_i, pl_j, pl_type, pl_it_idx, pl_jt_idx, pl_it_entry, pl_jt_entry;
  DevBuf<long long> pl_it_off, pl_jt_off;
  // which virtual samples of the uploaded list are live, and how many (static until the list, nlocal or the type
  // pair change): built on the device by the first step that needs them
  bool pl_mask_valid = false;
  int pl_mask_nlocal = -1, pl_mask_itype = 0, pl_mask_jtype = 0;
  long long pl_calls = 0;
  long long pl_npairs = -1, pl_nall = 0;
  DevBuf<double> vs_r;     // virtual add_hill samples of the list (2 per entry)
  DevBuf<int> vs_mask;
  bool force_sync = false;      // redo of a deferred step whose launch bound proved too small
  int debug_force_sync = 0;     // tests: never defer the count (every step takes the synchronous path)
  long long bound_redos = 0;    // steps redone because the accepted count exceeded the deferred launch bound
  PendingForces pending;        // pair forces of a fused step waiting for the launch of the step's selection
  PendingForces *flush_forces = nullptr;   // fix edm step: the pending force kernel an overflow flush may carry (else NULL)
  // staging of the *_host entry points: device copies of the caller's host arrays, a second stream for the copy
  // that runs against the direction of the others, and the event that orders it behind the force kernel
  DevBuf<double> hs_r, hs_f, hs_x, hs_u;
  DevBuf<int> hs_mask;
  double *h_delta = nullptr;       // page-locked landing zone of step_host's force delta
  hipEvent_t delta_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t h_delta_cap = 0;
  // the caller's position / uniform / mask blocks, page-locked in place (step_host)
  enum { REG_X = 0, REG_U = 1, REG_MASK = 2, REG_SLOTS = 3 };
  const void *reg_ptr[REG_SLOTS] = {nullptr, nullptr, nullptr}, *reg_failed_ptr[REG_SLOTS] = {nullptr, nullptr, nullptr};
  size_t reg_bytes[REG_SLOTS] = {0, 0, 0};
  DeltaAddPool add_pool;        // step_host: helper threads of the host-side add
  int host_add_threads = 4;     // edm_hip_bias_set("host_add_threads"): threads adding the delta (the caller's included)
  // fix edm step: called right behind the launch that carries the step's force kernel (PendingForces::on_launched)
  void (*step_hook)(void *) = nullptr;
  void *step_hook_ctx = nullptr;
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_event = nullptr;
  const double *pl_view_x = nullptr;   // pair_list_step: the samples of this add_hill cycle are the virtual samples of the
                                       // uploaded list, their CVs recomputed from these positions ([nall][3])
  int debug_virtual_ranks = 0;  // tests: a one-rank communicator's packet is replicated, emulating that many ranks
  DevBuf<long long> xchg_cnt;
  // fix edm_pair in the reference's order (edm_hip_bias_pair_step_ordered): what the step's hill batch left on the
  // device -- count, heights, sample indices -- and the buffers of the force pass that follows it
  struct LastBatch {
    bool valid = false;
    long long nh = 0, k = 0;
    const double *heights = nullptr, *tail_h1 = nullptr, *tail_h2 = nullptr;
    double h_const = 0;
    const long long *sel = nullptr;
    // multi-GPU: this rank's slice of the rank-major global list (host values, or where they lie on the device)
    long long local_off = 0, local_cnt = -1, local_cap = 0;
    const long long *range_dev = nullptr;
    bool terms_emitted = false;   // the batch's launch stored the hills' stencil terms in ord_terms (rows: ord_terms_rows)
    const LimitResult *res_dev = nullptr;   // provisional entry (force pass queued before the host saw the limiter's result):
                                            // nh is the launch bound, count and split index are read on the device
    // ... or, beside the batch's launch on ord_stream: the limiter's word and the selection's count (OrderedForcesArgs)
    const unsigned long long *wait_flag = nullptr;
    unsigned long long wait_seq = 0;
    const long long *d_nh = nullptr;
  } last_batch;
  // reference-order array step, single rank: the force pass is queued from inside the hill batch's apply (ApplySpec::
  // before_wait) -- behind the batch on the stream, ahead of the host's wait for the limiter
  struct OrderedEarly {
    bool armed = false, done = false;
    long long n = 0;
    const double *d_r = nullptr;
    const int *d_first = nullptr;
    double *d_force = nullptr;
    unsigned long long tag = 0;
    int nblk = 0, rc = 0;
    bool list = false;     // the force pass over the device-resident neighbour list (pair_list_step) instead of arrays
    PairListArgs pl;
  } ord_early;
  bool ord_snap_pending = false;  // ordered_snapshot has been asked for, no launch has made the copy yet
  bool ord_step_active = false;   // between ordered_snapshot and the step's force pass: hill batches may emit their terms
  DevBuf<double> ord_terms;
  long long ord_terms_rows = 0;
  DevBuf<long long> ord_range;
  DevBuf<double> ord_rec0, ord_records;   // OrderedForcesArgs::rec0 (taken before the batch) / ::records
  DevBuf<unsigned short> ord_counts;
  DevBuf<unsigned> ord_dirty;   // OrderedForcesArgs::dirty_hill (kept zero-initialised: step numbers start at 1)
  // the record pass and the force pass of a single-rank reference-order step run on a stream of their own, BESIDE the
  // hill batch's launch: the record pass waits in the kernel for the limiter's word and the emitters' flags
  // (OrderedForcesArgs::wait_flag), not for the launch's end -- the gather tiles' half of it is nothing it needs
  hipStream_t ord_stream = nullptr;
  hipEvent_t ord_done_event = nullptr;
  // (host-array entry: recorded behind the uploads it queued on the object's stream; the second stream's gate wave is
  //  held back until they are through -- its 2 ms are for the step's kernels, not for 29 MB over PCIe ahead of them)
  hipEvent_t ord_up_event = nullptr;
  bool ord_wait_uploads = false;
  DevBuf<unsigned> ord_ready;   // LimitArgs::ord_ready (zero-initialised like ord_dirty)
  DevBuf<int> ord_status;       // OrderedForcesArgs::status: 0 fine, 1 batch refused, 2 the gate wave gave up (kernels of
                                // different streams run one at a time) -- and the latter's host-mapped twin
  int *h_ord_status = nullptr, *d_ord_status = nullptr;
  bool ord_own_stream_off = false;   // ... after which the second stream is not used again by this object
  long long ord_gate_giveups = 0;
  bool ord_on_own_stream = false;   // this step's record / force passes went to ord_stream
  unsigned ord_seq = 0;
  DevBuf<int> ord_first;
  int reference_order = 0;     // edm_hip_bias_set("reference_order"): edm_hip_bias_pair_list_step evaluates its forces in
                               // the reference's order too (edm_hip_bias_pair_step_ordered is that mode's array entry)
  // multi-GPU
  Transport *comm = nullptr;   // RCCL over xGMI, or the host-staged carrier (edm_comm.h)
  int nranks = 1, rank = 0;
  bool split_applied = false;
};

// ---- configuration file (edm_bias.cpp:19-24, :933-979, :986-1095) ---------------
typedef std::map<std::string, std::string> cfg_map;

static bool read_pairs(const char *filename, cfg_map &out) {
  FILE *fp = fopen(filename, "r");
  if (!fp) return false;
  char key[256];
  while (fscanf(fp, "%255s", key) == 1) {
    std::string val;
    int c;
    bool any = false;
    while ((c = fgetc(fp)) != EOF) {
      any = true;
      if (c == '\n') break;
      val.push_back((char)c);
    }
    if (!any) break;                         // getline at EOF fails: the pair is dropped
    out.insert(std::make_pair(std::string(key), val));  // first occurrence wins
  }
  fclose(fp);
  return true;
}
static bool cfg_double(const cfg_map &m, const char *key, double *out) {
  cfg_map::const_iterator it = m.find(key);
  if (it == m.end()) return false;
  *out = atof(it->second.c_str());
  return *out != 0.0;  // a value of exactly 0 is rejected (:937-940)
}
static bool cfg_int(const cfg_map &m, const char *key, int *out) {
  cfg_map::const_iterator it = m.find(key);
  if (it == m.end()) return false;
  *out = atoi(it->second.c_str());
  return true;
}
static bool cfg_array(const cfg_map &m, const char *key, std::vector<double> &out, int len) {
  cfg_map::const_iterator it = m.find(key);
  if (it == m.end()) return false;
  const char *cur = it->second.c_str();
  for (int i = 0; i < len; i++) {
    char *end;
    double t = strtod(cur, &end);
    if (end == cur) break;
    out[(size_t)i] = t;
    cur = end;
  }
  return true;
}
// :1098-1111 -- strips leading blanks/tabs only
static std::string clean_string(const std::string &in, bool append_rank, int rank) {
  std::string r(in);
  size_t k = r.find_first_not_of(" \t");
  if (k != std::string::npos) r = r.substr(k);
  if (append_rank) r += "_" + std::to_string(rank);
  return r;
}

// Grid::expected_bias (grid.h:692-710) on the values read from the target file
static double expected_bias_of(const std::vector<double> &v) {
  double Z = 0, offset = 0, avg = 0;
  for (size_t i = 0; i < v.size(); i++) offset = fmax(offset, v[i]);
  for (size_t i = 0; i < v.size(); i++) Z += exp(-v[i] - offset);
  for (size_t i = 0; i < v.size(); i++) avg += v[i] * exp(-v[i] - offset);
  return avg / Z;
}

static int read_input(edm_hip_bias *b, const char *filename) {
  cfg_map m;
  if (!read_pairs(filename, m)) {
    set_error(std::string("Cannot open input file ") + filename);
    return EDM_HIP_ERR_IO;
  }
  if (!cfg_int(m, "tempering", &b->b_tempering)) {
    set_error("Must specify if tempering is enabled, ex: tempering 1 or tempering 0");
    return EDM_HIP_ERR_IO;
  }
  if (b->b_tempering) {
    if (!cfg_double(m, "bias_factor", &b->bias_factor)) {
      set_error("Could not find key bias_factor");
      return EDM_HIP_ERR_IO;
    }
    cfg_double(m, "global_tempering", &b->global_tempering);
  }
  if (!cfg_double(m, "hill_prefactor", &b->hill_prefactor)) {
    set_error("Could not find key hill_prefactor");
    return EDM_HIP_ERR_IO;
  }
  if (!cfg_double(m, "bias_per_step", &b->bias_per_step)) b->bias_per_step = b->hill_prefactor;
  cfg_double(m, "hill_density", &b->hill_density);
  int tmp = 0;
  if (!cfg_int(m, "dimension", &tmp) || tmp <= 0 || tmp > 3) {
    set_error("Invalid dimesion");
    return EDM_HIP_ERR_IO;
  }
  b->dim = (unsigned int)tmp;
  b->bias_dx.assign(b->dim, 0);
  b->bias_sigma.assign(b->dim, 0);
  b->min.assign(b->dim, 0);
  b->max.assign(b->dim, 0);
  b->bper.assign(b->dim, 0);
  if (!cfg_array(m, "bias_spacing", b->bias_dx, tmp) || !cfg_array(m, "bias_sigma", b->bias_sigma, tmp) ||
      !cfg_array(m, "box_low", b->min, tmp) || !cfg_array(m, "box_high", b->max, tmp)) {
    set_error("Could not find one of bias_spacing, bias_sigma, box_low, box_high");
    return EDM_HIP_ERR_IO;
  }
  cfg_map::const_iterator it = m.find("target_filename");
  if (it != m.end()) {
    b->b_targeting = 1;
    GridFile gf;
    int rc = read_plumed(tmp, clean_string(it->second, false, 0).c_str(), 0, gf);
    if (rc) return rc;
    b->expected_target = expected_bias_of(gf.values);
    // device copy for the nearest-lower lookup of add_hill (:546)
    edm_hip_grid *t = new edm_hip_grid;
    t->g = gf.g;
    t->g.has_deriv = 0;
    t->g.rec = 1;
    t->g.interp = 0;
    EDM_HIP_TRY(hipStreamCreate(&t->stream));
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&t->values), sizeof(double) * gf.values.size()));
    EDM_HIP_TRY(hipMemcpy(t->values, gf.values.data(), sizeof(double) * gf.values.size(), hipMemcpyHostToDevice));
    b->target = t;
  }
  it = m.find("initial_bias_filename");
  if (it != m.end()) b->initial_bias_file = clean_string(it->second, false, 0);
  it = m.find("hills_filename");
  b->hills_name = it != m.end() ? it->second : std::string("HILLS");
  it = m.find("histogram_filename");
  b->hist_output = clean_string(it != m.end() ? it->second : std::string("HIST"), false, 0);
  return EDM_HIP_OK;
}

static void open_hills(edm_hip_bias *b) {
  if (b->hills.is_open()) return;
  b->hills.open(clean_string(b->hills_name, true, b->mpi_rank).c_str(), b->dim);
}

// edm_bias.cpp:586-599 (text) -- the histogram part of output_hill runs on the device
static void log_hill(edm_hip_bias *b, const double *pos, double height, double added, char type) {
  if (!b->hill_log || !b->hills.is_open()) return;
  HillEvent e;
  e.steps = b->steps;
  e.hills_added = b->hills_added;
  e.type = type;
  for (unsigned int d = 0; d < 3; d++) e.pos[d] = d < b->dim ? pos[d] : 0.0;
  e.height = height;
  e.added = added;
  e.cum_over_volume = b->cum_bias / b->total_volume;
  b->hill_events.push_back(e);
}

// the 'h' lines of a batch whose read-back was deferred (edm_bias.cpp:586-599, one per hill, in order)
static int resolve_deferred_log(edm_hip_bias *b) {
  if (!b->deferred_log.active) return EDM_HIP_OK;
  b->deferred_log.active = false;
  std::vector<double> pos, added;
  int rc = apply_hills_fetch_deferred(b->bias, b->deferred_log.bound, b->deferred_log.nh, pos, added);
  if (rc) return rc;
  std::vector<HillEvent> ev((size_t)b->deferred_log.nh);
  for (long long i = 0; i < b->deferred_log.nh; i++) {
    HillEvent &e = ev[(size_t)i];
    e.steps = b->deferred_log.steps;
    e.hills_added = b->deferred_log.hills_added_before + (int)i + 1;
    e.type = 'h';
    for (unsigned int d = 0; d < 3; d++) e.pos[d] = d < b->dim ? pos[(size_t)i * b->dim + d] : 0.0;
    e.height = b->deferred_log.height;
    e.added = added[(size_t)i];
    e.cum_over_volume = b->deferred_log.cum_over_volume;
  }
  // (called between two cycles there is nothing ahead of these lines; called inside the cycle that deferred them -- the
  //  reference-order step does, while its force pass runs -- the cycle's earlier events, an overflow flush's lines, are
  //  still waiting for post_add_hill and go first)
  if (b->hill_events.empty())
    b->hills.submit(ev);
  else
    b->hill_events.insert(b->hill_events.end(), ev.begin(), ev.end());
  return EDM_HIP_OK;
}

extern "C" {

int edm_hip_bias_create(edm_hip_bias **out, const char *input_filename) {
  if (!out || !input_filename) return EDM_HIP_ERR_ARG;
  edm_hip_bias *b = new edm_hip_bias;
  int rc = read_input(b, input_filename);  // the reference ignores the result (:68); keep the handle, report the status
  b->overflow.assign((size_t)(EDM_HIP_BIAS_BUFFER_SIZE + 1) * (b->dim + 1), 0.0);
  *out = b;
  if (rc == EDM_HIP_OK) open_hills(b);
  return rc;
}

int edm_hip_bias_destroy(edm_hip_bias *b) {
  if (!b) return EDM_HIP_OK;
  delete b->comm;
  if (b->copy_stream) {
    (void)hipStreamSynchronize(b->copy_stream);
    (void)hipStreamDestroy(b->copy_stream);
  }
  if (b->copy_event) (void)hipEventDestroy(b->copy_event);
  if (b->ord_stream) {
    (void)hipStreamSynchronize(b->ord_stream);
    (void)hipStreamDestroy(b->ord_stream);
  }
  if (b->ord_done_event) (void)hipEventDestroy(b->ord_done_event);
  if (b->ord_up_event) (void)hipEventDestroy(b->ord_up_event);
  b->ord_ready.release(); b->ord_status.release();
  if (b->h_ord_status) (void)hipHostFree(b->h_ord_status);
  b->hs_r.release(); b->hs_f.release(); b->hs_x.release(); b->hs_u.release(); b->hs_mask.release();
  if (b->h_delta) (void)hipHostFree(b->h_delta);
  for (int c = 0; c < 8; c++)
    if (b->delta_ev[c]) (void)hipEventDestroy(b->delta_ev[c]);
  for (int k = 0; k < edm_hip_bias::REG_SLOTS; k++)
    if (b->reg_ptr[k]) (void)hipHostUnregister(const_cast<void *>(b->reg_ptr[k]));
  (void)resolve_deferred_log(b);
  b->hills.submit(b->hill_events);
  b->hills.close();
  edm_hip_gauss_destroy(b->bias);
  edm_hip_grid_destroy(b->hist);
  edm_hip_grid_destroy(b->target);
  b->sel.release(); b->sel_scratch.release(); b->count_dev.release(); b->sel_stage.release();
  if (b->h_flush) (void)hipHostFree(b->h_flush);
  b->vs_r.release(); b->vs_mask.release(); b->pl_i.release(); b->pl_j.release(); b->pl_type.release();
  b->pl_it_idx.release(); b->pl_jt_idx.release(); b->pl_it_off.release(); b->pl_jt_off.release();
  b->pl_it_entry.release(); b->pl_jt_entry.release();
  if (b->h_count) (void)hipHostFree(b->h_count);
  b->stage_x.release(); b->stage_u.release(); b->stage_h.release(); b->tail_w.release(); b->hx0.release();
  b->ord_terms.release(); b->ord_range.release(); b->ord_rec0.release(); b->ord_records.release(); b->ord_counts.release(); b->ord_dirty.release(); b->ord_first.release();
  delete b;
  return EDM_HIP_OK;
}

int edm_hip_bias_setup(edm_hip_bias *b, double temperature, double boltzmann_constant) {
  b->temperature = temperature;                       // :266
  b->boltzmann_factor = boltzmann_constant * temperature;
  return EDM_HIP_OK;
}

// edm_bias.cpp:98-222
int edm_hip_bias_subdivide(edm_hip_bias *b, const double *sublo, const double *subhi, const double *boxlo,
                           const double *boxhi, const int *b_periodic, const double *skin) {
  if (b->bias != nullptr) return EDM_HIP_OK;          // :121-122
  if (b->temperature < 0) {
    set_error("Must call setup before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  int grid_period[3] = {0, 0, 0};
  double lo[3], hi[3];
  int never_inside = 1;
  for (unsigned int d = 0; d < b->dim; d++) {
    b->bper[d] = 0;
    if (fabs(boxlo[d] - b->min[d]) < 0.000001 && fabs(boxhi[d] - b->max[d]) < 0.000001) b->bper[d] = b_periodic[d];
  }
  for (unsigned int d = 0; d < b->dim; d++) {
    lo[d] = sublo[d];
    hi[d] = subhi[d];
    if (fabs(sublo[d] - b->min[d]) < 0.000001 && fabs(subhi[d] - b->max[d]) < 0.000001) {
      grid_period[d] = b_periodic[d];
      never_inside = 0;
    } else {
      lo[d] -= skin[d];
      hi[d] += skin[d];
    }
    never_inside &= (lo[d] >= b->max[d] || hi[d] <= b->min[d]);
  }
  int rc = edm_hip_gauss_create(&b->bias, (int)b->dim, lo, hi, b->bias_dx.data(), grid_period, 1, b->bias_sigma.data());
  if (rc) return rc;
  rc = edm_hip_grid_create(&b->hist, (int)b->dim, lo, hi, b->bias_sigma.data(), grid_period);  // :163
  if (rc) return rc;
  rc = edm_hip_gauss_set_boundary(b->bias, b->min.data(), b->max.data(), b->bper.data());
  if (rc) return rc;
  if (!b->initial_bias_file.empty()) {
    rc = edm_hip_gauss_add_from_file(b->bias, b->initial_bias_file.c_str(), 1.0, 0.0);
    if (rc) return rc;
  }
  // :175-180 of the MPI build: density and prefactor become per-system quantities
  if (b->nranks > 1 && !b->split_applied && b->hill_density > 0) {
    b->hill_density /= b->nranks;
    b->hill_prefactor /= b->nranks;
    if (b->hill_density == 0) b->hill_density = 1;
    b->split_applied = true;
  }
  if (never_inside) {
    b->b_outofbounds = 1;
    return EDM_HIP_OK;
  }
  double vol = 1;                                      // gaussian_grid.h:437-444
  for (unsigned int d = 0; d < b->dim; d++) vol *= b->max[d] - b->min[d];
  b->total_volume = 0;
  b->total_volume += vol * (b->nranks > 1 ? b->nranks : 1);  // :211-220 (sum over replicas)
  return EDM_HIP_OK;
}

int edm_hip_bias_set_mask(edm_hip_bias *b, const int *d_mask) {
  b->d_mask = d_mask;
  return EDM_HIP_OK;
}

int edm_hip_bias_update_forces(edm_hip_bias *b, long long n, const double *d_x, int x_stride, double *d_f,
                               int f_stride, int apply_mask, double *energy) {
  if (energy) *energy = 0;
  if (b->b_outofbounds) return EDM_HIP_OK;             // :279-280
  if (!b->bias) {
    set_error("update_forces before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  return edm_hip_gauss_update_forces(b->bias, n, d_x, x_stride, d_f, f_stride, b->d_mask, apply_mask, energy);
}

int edm_hip_bias_pair_forces(edm_hip_bias *b, long long n, const double *d_r, double *d_force, double *energy) {
  if (energy) *energy = 0;
  if (b->b_outofbounds) {
    if (n > 0) EDM_HIP_TRY(hipMemset(d_force, 0, sizeof(double) * (size_t)n));
    return EDM_HIP_OK;
  }
  if (!b->bias) {
    set_error("pair_forces before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  return edm_hip_gauss_pair_forces(b->bias, n, d_r, d_force, energy);
}

}  // extern "C"

// ---- overflow buffer (edm_bias.cpp:498-523) -- reproduces the reference's
// "increment before store" on the right end, which leaves slot 0 unwritten and the
// newest record one past what the flush reads.
static int overflow_push(edm_hip_bias *b, const double *pos, double h) {
  const size_t w = b->dim + 1;
  if (b->overflow_right == EDM_HIP_BIAS_BUFFER_SIZE) {
    if (b->overflow_left == 0) {
      set_error("The bias overflow buffer is full. Too many hills. Either increase & recompile, lower hill_density, or lower bias");
      return EDM_HIP_ERR_OVERFLOW;
    }
    b->overflow_left--;
    for (unsigned int d = 0; d < b->dim; d++) b->overflow[b->overflow_left * w + d] = pos[d];
    b->overflow[b->overflow_left * w + b->dim] = h;
  } else {
    b->overflow_right++;
    for (unsigned int d = 0; d < b->dim; d++) b->overflow[b->overflow_right * w + d] = pos[d];
    b->overflow[b->overflow_right * w + b->dim] = h;
  }
  return EDM_HIP_OK;
}

// flush_bias_buffer (edm_bias.cpp:313-380) on the device
static int flush_overflow(edm_hip_bias *b, double max_bias, double *bias_added) {
  *bias_added = 0;
  const size_t w = b->dim + 1;
  const long long n = (long long)b->overflow_right - (long long)b->overflow_left;
  if (n <= 0) {
    if (b->overflow_left == b->overflow_right) b->overflow_left = b->overflow_right = 0;
    return EDM_HIP_OK;
  }
  // positions and heights of the buffered hills: packed [x ... | h ...] in pinned memory, ONE H2D copy
  const size_t nx = (size_t)n * b->dim, ntot = nx + (size_t)n;
  if (b->flush_cap < ntot) {
    if (b->h_flush) (void)hipHostFree(b->h_flush);
    b->h_flush = nullptr;
    b->flush_cap = ntot + ntot / 2 + 64;
    EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_flush), sizeof(double) * b->flush_cap, hipHostMallocMapped));
    EDM_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_flush), b->h_flush, 0));
  }
  for (long long i = 0; i < n; i++) {
    const double *rec = &b->overflow[(b->overflow_left + (size_t)i) * w];
    for (unsigned int d = 0; d < b->dim; d++) b->h_flush[(size_t)i * b->dim + d] = rec[d];
    b->h_flush[nx + (size_t)i] = rec[b->dim];
  }
  EDM_HIP_TRY(b->stage_x.reserve(ntot));
  // (no upload: the preparation kernel reads positions and heights straight from the host-mapped array)
  ApplySpec spec;
  spec.nh = n;
  spec.d_x = b->d_flush;
  spec.x_stride = (int)b->dim;
  spec.d_h = b->stage_x.p + nx;
  spec.h_fetch_src = b->d_flush + nx;
  spec.limited = true;
  spec.flush_mode = 1;
  spec.limit = max_bias;
  spec.cum_in = 0;
  spec.hist_g = &b->hist->g;
  spec.hist_values = b->hist->values;
  const bool log_all = b->hill_log && b->hills.is_open();
  spec.fetch_all = log_all;     // (per-hill bias only for the HILLS log; without one the limiter's header line is enough)
  spec.fetch_heights = false;   // (the heights came from the host's own overflow records)
  spec.forces = b->flush_forces;   // (fix edm step: its pending force kernel can share the preparation's launch)
  ApplyOutcome oc;
  int rc = apply_hills(b->bias, spec, &oc, false);
  if (rc) return rc;
  const int stop = oc.res.stop;              // index of the crossing hill, or n
  const long long nrun = (stop < n) ? stop + 1 : n;
  for (long long i = 0; i < nrun; i++) {
    double *rec = &b->overflow[(b->overflow_left + (size_t)i) * w];
    b->hills_added++;
    if (log_all) log_hill(b, rec, rec[b->dim], oc.added[(size_t)i], 'b');
    if (i == stop) {
      const double h2 = oc.plain_fast ? oc.res.h2_stop : oc.h2[(size_t)i];
      rec[b->dim] = -h2;                      // the remaining part stays buffered (:341)
      if (log_all) log_hill(b, rec, h2, oc.a2[(size_t)i], 'v');
      b->hills_added++;
    }
  }
  if (stop < n)
    b->overflow_left += (size_t)stop;
  else
    b->overflow_left = b->overflow_right;
  if (b->overflow_left == b->overflow_right) b->overflow_left = b->overflow_right = 0;
  *bias_added = oc.res.cum_out;
  return EDM_HIP_OK;
}

// pre_add_hill (edm_bias.cpp:413-442)
static int do_pre_add_hill(edm_hip_bias *b, long long est) {
  if (b->b_outofbounds) return EDM_HIP_OK;
  if (!b->bias) {
    set_error("pre_add_hill before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  {
    int rcd = resolve_deferred_log(b);   // (the previous cycle's log lines, if they were deferred)
    if (rcd) return rcd;
  }
  b->bias->shared_device = b->comm != nullptr && strcmp(b->comm->name(), "rccl") != 0;
  b->est_hill_count = est;
  b->temp_hill_prefactor = b->hill_prefactor;
  if (b->global_tempering > 0)
    if (b->cum_bias / b->total_volume >= b->global_tempering)
      b->temp_hill_prefactor *= exp(-(b->cum_bias / b->total_volume - b->global_tempering) /
                                    (b->global_tempering * (b->bias_factor - 1) * b->boltzmann_factor));
  b->temp_hill_cum = 0;
  b->hills_added = 0;
  double flushed = 0;
  int rc = flush_overflow(b, b->bias_per_step, &flushed);
  if (rc) return rc;
  b->temp_hill_cum += flushed;
  b->b_skip_hill_add = (b->overflow_left == 0 && b->overflow_right == 0) ? 0 : 1;
  return EDM_HIP_OK;
}

// Hill exchange: replaces EDMBias::flush_buffers (edm_bias.cpp:630-706).  The reference
// broadcasts every rank's (position, height) records and each rank replays them; here one RCCL
// all-gather over xGMI collects the records and EVERY rank replays the same rank-major global
// list through the same deterministic kernels, so the replicated grids, limiter decisions and
// overflow buffers stay bit-identical on all GPUs (the reference's per-rank replay order lets
// them drift apart).  Two collectives per hill step: counts (8 B per rank), then the padded
// records; heights are identical on all ranks by construction and are not sent.
static int exchange_hills(edm_hip_bias *b, long long nh_local, const double *d_x, int x_stride,
                          const long long *d_sel, long long *nh_global, const double **d_global) {
  hipStream_t s = b->bias->stream;
  const int N = b->nranks;
  const int dim = (int)b->dim;
  EDM_HIP_TRY(b->xchg_cnt.reserve((size_t)2 * N + 2));
  long long *h_cnt = reinterpret_cast<long long *>(b->bias->h_scalars + 40);  // 2 * N <= 32 slots
  if (N > EDM_MAX_RANKS) {
    set_error("exchange_hills: more than 16 ranks per node are not supported");
    return EDM_HIP_ERR_ARG;
  }
  // (count, est_hill_count) of every rank: the height of an all-samples hill depends on its sender's
  // estimate (edm_bias.cpp:552-556), which travels with the hill in the reference
  h_cnt[0] = nh_local;
  h_cnt[1] = b->est_hill_count;
  EDM_HIP_TRY(hipMemcpyAsync(b->xchg_cnt.p + 2 * N, h_cnt, 2 * sizeof(long long), hipMemcpyHostToDevice, s));
  int rcx = b->comm->all_gather(b->xchg_cnt.p + 2 * N, b->xchg_cnt.p, 2 * sizeof(long long), s);   // hill counts
  if (rcx) return rcx;
  EDM_HIP_TRY(hipMemcpyAsync(h_cnt, b->xchg_cnt.p, sizeof(long long) * (size_t)(2 * N), hipMemcpyDeviceToHost, s));
  EDM_HIP_TRY(hipStreamSynchronize(s));
  long long counts[EDM_MAX_RANKS], total = 0, maxc = 0;
  for (int r = 0; r < N; r++) {
    counts[r] = h_cnt[2 * r];
    b->xchg_counts[r] = counts[r];
    b->xchg_est[r] = h_cnt[2 * r + 1];
    total += counts[r];
    if (counts[r] > maxc) maxc = counts[r];
  }
  *nh_global = total;
  *d_global = nullptr;
  if (total == 0) return EDM_HIP_OK;
  EDM_HIP_TRY(b->xchg_send.reserve((size_t)maxc * dim));
  EDM_HIP_TRY(b->xchg_recv.reserve((size_t)maxc * dim * N));
  EDM_HIP_TRY(b->xchg_all.reserve((size_t)total * dim));
  EDM_HIP_TRY(launch_gather_positions(nh_local, d_x, x_stride, d_sel, dim, b->xchg_send.p, s));
  rcx = b->comm->all_gather(b->xchg_send.p, b->xchg_recv.p, sizeof(double) * (size_t)maxc * dim, s);   // padded hill records
  if (rcx) return rcx;
  long long off = 0;
  for (int r = 0; r < N; r++) {   // rank-major global order
    if (counts[r] > 0)
      EDM_HIP_TRY(hipMemcpyAsync(b->xchg_all.p + (size_t)off * dim, b->xchg_recv.p + (size_t)r * maxc * dim,
                                 sizeof(double) * (size_t)counts[r] * dim, hipMemcpyDeviceToDevice, s));
    off += counts[r];
  }
  *d_global = b->xchg_all.p;
  return EDM_HIP_OK;
}

// the add_hill loop of one cycle (edm_bias.cpp:528-563 and :444-526) over device arrays
static void ordered_snapshot_ride(edm_hip_bias *b, SelectArgs *a);
static int ordered_snapshot_now(edm_hip_bias *b);
static int ordered_forces_enqueue(edm_hip_bias *b);
static int process_new_hills(edm_hip_bias *b, long long n, const double *d_x, int x_stride, const double *d_ru,
                             int apply_mask) {
  if (n <= 0 && !b->comm) return EDM_HIP_OK;  // with a communicator every rank must reach the exchange
  if (b->temp_hill_prefactor < 0) {
    set_error("Must call pre_add_hill before add_hill");
    return EDM_HIP_ERR_STATE;
  }
  if (b->b_skip_hill_add) return EDM_HIP_OK;          // :534-535
  if (b->b_outofbounds) return EDM_HIP_OK;
  if (apply_mask >= 0 && !b->d_mask) {
    set_error("add_hills: apply_mask >= 0 needs set_mask");
    return EDM_HIP_ERR_ARG;
  }
  const bool local_tempering = (b->b_tempering && b->global_tempering < 0);  // :547
  hipStream_t s = b->bias->stream;
  const double *const d_x_in = d_x;   // (the caller's arrays, for the synchronous redo of a deferred step)
  const int x_stride_in = x_stride;
  const int use_thr = !(b->hill_density < 0);
  const double thr = b->hill_density / b->est_hill_count;   // :543
  // one stream per add_hill cycle (the same key for a synchronous redo of this cycle)
  const unsigned long long rng = b->rng_seed + b->rng_cycle * 0x632BE59BD9B4E019ull;
  long long nh = n;
  long long deferred_bound = 0;
  SelectArgs sel_args;
  UnpackArgs unp_args;
  const long long *d_sel = nullptr;
  // Multi-GPU stochastic step without a host round trip: every rank packs its accepted samples into a
  // fixed-size packet [count, positions...], ONE ncclAllGather concatenates the packets, one workgroup
  // unpacks them into the rank-major global list, and the rest of the step runs against a launch bound
  // exactly like the single-GPU deferred step.  Packet size and the decision to take this path depend
  // on replicated state only (hill_density after the per-rank split, rank count), never on rank-local
  // sample counts, so all ranks agree; an overflowing rank makes every rank fall back together.
  bool packed_exchange = false;
  long long pack_bound = 0;
  int pack_ranks = b->nranks;
  if (b->comm && use_thr && !b->b_targeting && !local_tempering && b->nranks <= EDM_MAX_RANKS && !b->force_sync &&
      !b->debug_force_sync) {
    if (b->debug_virtual_ranks > 1 && b->nranks == 1) pack_ranks = b->debug_virtual_ranks;
    pack_bound = (long long)(4.0 * b->hill_density) + 128;
    if (pack_bound < 256) pack_bound = 256;
    packed_exchange = (pack_bound * pack_ranks <= EDM_CHUNK);
  }
  if (packed_exchange) {
    const int dim = (int)b->dim;
    const long long packet = 1 + pack_bound * dim;
    EDM_HIP_TRY(b->xchg_send.reserve((size_t)packet));
    EDM_HIP_TRY(b->xchg_recv.reserve((size_t)packet * pack_ranks));
    EDM_HIP_TRY(b->xchg_all.reserve((size_t)pack_bound * pack_ranks * dim));
    EDM_HIP_TRY(b->count_dev.reserve(2));
    if (!b->h_count) {
      EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_count), 64, hipHostMallocMapped));
      EDM_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_count), b->h_count, 0));
    }
    if (n > 0) {
      if (!d_ru && !b->device_rng) {
        set_error("add_hills: hill_density set but no uniform random numbers given");
        return EDM_HIP_ERR_ARG;
      }
      EDM_HIP_TRY(b->sel_scratch.reserve(select_scratch_ints(n)));
      EDM_HIP_TRY(b->sel_stage.reserve(select_stage_ints(n)));
      memset(&sel_args, 0, sizeof(sel_args));
      sel_args.n = n;
      sel_args.ru = d_ru;
      sel_args.rng = rng;
      sel_args.thr = thr;
      sel_args.use_thr = use_thr;
      sel_args.mask = b->d_mask;
      sel_args.apply_mask = apply_mask;
      sel_args.counts = b->sel_scratch.p;
      sel_args.stage = b->sel_stage.p;
      sel_args.count_host = b->d_count;
      sel_args.count_dev = b->count_dev.p;
      sel_args.ticket = b->bias->d_tickets;
      sel_args.pack = b->xchg_send.p;
      ordered_snapshot_ride(b, &sel_args);
      EDM_HIP_TRY(b->sel.reserve((size_t)pack_bound));
      sel_args.sel = b->sel.p;   // (this rank's accepted sample indices, for the reference-order force pass)
      HillList src;
      memset(&src, 0, sizeof(src));
      src.nh = pack_bound;
      src.x = d_x;
      src.x_stride = x_stride;
      if (b->pl_view_x) {
        src.pl_x = b->pl_view_x;
        src.pl_i = b->pl_i.p;
        src.pl_j = b->pl_j.p;
      }
      int rcs = select_prep_enqueue(b->bias, sel_args, src, &b->pending);
      if (rcs) return rcs;
    } else {
      int rcp = pending_forces_flush(b->bias, &b->pending);
      if (rcp) return rcp;
      EDM_HIP_TRY(hipMemsetAsync(b->xchg_send.p, 0, sizeof(double), s));  // an empty packet
    }
    {
      int rcx = b->comm->all_gather(b->xchg_send.p, b->xchg_recv.p, sizeof(double) * (size_t)packet, s);   // hill packets
      if (rcx) return rcx;
    }
    for (int r = 1; r < pack_ranks && b->nranks == 1; r++)   // (test hook: emulate more ranks with copies)
      EDM_HIP_TRY(hipMemcpyAsync(b->xchg_recv.p + (size_t)r * packet, b->xchg_recv.p, sizeof(double) * (size_t)packet,
                                 hipMemcpyDeviceToDevice, s));
    memset(&unp_args, 0, sizeof(unp_args));
    unp_args.recv = b->xchg_recv.p;
    unp_args.nranks = pack_ranks;
    unp_args.bound = pack_bound;
    unp_args.packet = packet;
    unp_args.all = b->xchg_all.p;
    unp_args.count_dev = b->count_dev.p;
    unp_args.count_host = b->d_count;
    EDM_HIP_TRY(b->ord_range.reserve(2));
    unp_args.rank = (pack_ranks == b->nranks) ? b->rank : 0;
    unp_args.local_range = b->ord_range.p;
    nh = pack_bound * pack_ranks;
    deferred_bound = nh;
    d_x = b->xchg_all.p;
    x_stride = dim;
  } else if (use_thr || apply_mask >= 0) {
    if (use_thr && !d_ru && !b->device_rng) {
      set_error("add_hills: hill_density set but no uniform random numbers given");
      return EDM_HIP_ERR_ARG;
    }
    EDM_HIP_TRY(b->sel.reserve((size_t)n));
    if (!b->h_count) {
      EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_count), 64, hipHostMallocMapped));
      EDM_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_count), b->h_count, 0));
    }
    EDM_HIP_TRY(b->sel_scratch.reserve(select_scratch_ints(n)));
    EDM_HIP_TRY(b->count_dev.reserve(2));
    // Stochastic steps accept a few hundred of millions of samples: queue the whole step against a
    // conservative bound instead of waiting for the count to reach the host (one sync per step saved);
    // should the bound ever be too small the limiter reports it, nothing is applied and the step's hill
    // path is redone below with the exact count.
    long long bound = 0;
    if (use_thr && !b->comm && !b->b_targeting && !local_tempering && !b->force_sync && !b->debug_force_sync) {
      const double expected = thr * (double)n;
      bound = (long long)(4.0 * expected) + 128;
      if (bound < 256) bound = 256;
      if (bound > 2048 || bound >= n) bound = 0;
    }
    deferred_bound = bound;
    if (!bound) {
      int rcp = pending_forces_flush(b->bias, &b->pending);
      if (rcp) return rcp;
      EDM_HIP_TRY(launch_select(n, d_ru, thr, use_thr, b->d_mask, apply_mask, b->sel.p, b->d_count, b->sel_scratch.p, s,
                                b->count_dev.p, rng));
      EDM_HIP_TRY(hipStreamSynchronize(s));
      nh = *b->h_count;
    } else {
      // selection is chained in front of the hill preparation inside apply_hills (one launch)
      EDM_HIP_TRY(b->sel_stage.reserve(select_stage_ints(n)));
      memset(&sel_args, 0, sizeof(sel_args));
      sel_args.n = n;
      sel_args.ru = d_ru;
      sel_args.rng = rng;
      sel_args.thr = thr;
      sel_args.use_thr = use_thr;
      sel_args.mask = b->d_mask;
      sel_args.apply_mask = apply_mask;
      sel_args.counts = b->sel_scratch.p;
      sel_args.stage = b->sel_stage.p;
      sel_args.sel = b->sel.p;
      sel_args.count_host = b->d_count;
      sel_args.count_dev = b->count_dev.p;
      sel_args.ticket = b->bias->d_tickets;
      ordered_snapshot_ride(b, &sel_args);
      nh = bound;
    }
    d_sel = b->sel.p;
  }
  if (!deferred_bound) {   // (no chained selection ahead: the pair forces of a fused step go first)
    int rcp = pending_forces_flush(b->bias, &b->pending);
    if (rcp) return rcp;
  }
  const long long *local_sel = d_sel;   // (before a synchronous exchange replaces the list by the global one)
  const long long local_nh = nh;
  bool rank_heights = false;
  if (b->comm && !packed_exchange) {
    const double *d_all = nullptr;
    long long nh_all = 0;
    rank_heights = (b->hill_density < 0);
    int rc = exchange_hills(b, nh, d_x, x_stride, d_sel, &nh_all, &d_all);
    if (rc) return rc;
    nh = nh_all;
    d_x = d_all;
    x_stride = (int)b->dim;
    d_sel = nullptr;
  }
  if (nh <= 0) return EDM_HIP_OK;
  // height (:537, :552-558); targeting/tempering factors are handled above
  double this_h = b->temp_hill_prefactor;
  if (b->hill_density < 0)
    this_h /= b->est_hill_count;
  else
    this_h /= b->hill_density;
  this_h = fmin(this_h, BIAS_CLAMP * b->bias_per_step);
  const double *d_heights = nullptr;
  if (b->b_targeting && !local_tempering) {
    // per-hill heights: prefactor * exp(target(x) - <target>) / divisor, clamped (:545-558)
    EDM_HIP_TRY(b->stage_h.reserve((size_t)nh));
    const double divisor = (b->hill_density < 0) ? (double)b->est_hill_count : b->hill_density;
    EDM_HIP_TRY(launch_target_heights(b->target->g, b->target->values, nh, d_x, x_stride, d_sel, b->temp_hill_prefactor,
                                      b->expected_target, divisor, BIAS_CLAMP * b->bias_per_step, b->stage_h.p, s));
    d_heights = b->stage_h.p;
  } else if (rank_heights && !local_tempering) {
    // all-samples hills carry the height their sender gave them: prefactor / that rank's estimate
    RankHeights rh;
    memset(&rh, 0, sizeof(rh));
    rh.nranks = b->nranks;
    long long run = 0;
    for (int r = 0; r < b->nranks; r++) {
      rh.offset[r] = run;
      run += b->xchg_counts[r];
      rh.height[r] = fmin(b->temp_hill_prefactor / (double)b->xchg_est[r], BIAS_CLAMP * b->bias_per_step);
    }
    rh.offset[b->nranks] = run;
    EDM_HIP_TRY(b->stage_h.reserve((size_t)nh));
    EDM_HIP_TRY(launch_rank_heights(rh, b->stage_h.p, s));
    d_heights = b->stage_h.p;
  }

  ApplySpec spec;
  spec.nh = nh;
  spec.d_x = d_x;
  spec.x_stride = x_stride;
  if (b->pl_view_x && !packed_exchange) {
    spec.pl_x = b->pl_view_x;
    spec.pl_i = b->pl_i.p;
    spec.pl_j = b->pl_j.p;
  }
  spec.d_sel = d_sel;
  spec.d_h = d_heights;
  spec.h_const = this_h;
  spec.limited = true;
  spec.flush_mode = 0;
  spec.limit = b->bias_per_step;
  spec.cum_in = b->temp_hill_cum;
  if (use_thr && !b->comm) spec.expected_nh = thr * (double)n;
  if (packed_exchange) spec.expected_nh = b->hill_density * pack_ranks;
  spec.hist_g = &b->hist->g;
  spec.hist_values = b->hist->values;
  if (b->comm && !packed_exchange && b->debug_virtual_ranks <= 1) {
    // dense batches on a small grid: this rank applies only its own slice of the global list, the ranks'
    // delta grids are summed (apply_hills decides whether the batch qualifies)
    spec.shard_comm = b->comm;
    spec.shard_counts = b->xchg_counts;
    long long off = 0;
    for (int r = 0; r < b->rank; r++) off += b->xchg_counts[r];
    spec.shard_off = off;
    spec.shard_cnt = b->xchg_counts[b->rank];
  } else if (b->debug_virtual_ranks > 1 && !packed_exchange) {
    spec.shard_virtual = b->debug_virtual_ranks;
  }
  const bool log_all = b->hill_log && b->hills.is_open();
  spec.fetch_all = log_all;
  spec.defer_fetch_ok = log_all;   // (the log lines of a batch the limiter left alone are written behind the step)
  if (local_tempering) {
    // the height of hill i depends on the grid that already holds hills < i (:547-549):
    // strictly ordered application by one workgroup
    spec.ordered = true;
    memset(&spec.op, 0, sizeof(spec.op));
    spec.op.prefactor = b->temp_hill_prefactor;
    spec.op.use_target = b->b_targeting;
    if (b->b_targeting) {
      spec.op.target = b->target->g;
      spec.op.target_values = b->target->values;
      spec.op.expected_target = b->expected_target;
    }
    spec.op.use_tempering = 1;
    spec.op.temper_scale = (b->bias_factor - 1) * b->boltzmann_factor;
    spec.op.divisor = (b->hill_density < 0) ? (double)b->est_hill_count : b->hill_density;
    spec.op.clamp = BIAS_CLAMP * b->bias_per_step;
  }
  // (ranks that share a GPU -- the host-staged carrier exists for exactly that -- never let waiting gather tiles go first)
  b->bias->shared_device = b->comm != nullptr && strcmp(b->comm->name(), "rccl") != 0;
  if (b->ord_step_active && nh <= 2048 && b->dim == 1) {
    // a reference-order step: the batch's launch may store the hills' unit-height stencil terms for the force pass
    const size_t row = (size_t)(2 * b->bias->g.msize[0] + 1) * 2;
    EDM_HIP_TRY(b->ord_terms.reserve((size_t)nh * row));
    EDM_HIP_TRY(b->ord_dirty.reserve_zeroed((size_t)(nh > 4096 ? nh : 4096)));
    b->ord_terms_rows = nh;
    spec.ord_terms = b->ord_terms.p;
    spec.ord_dirty = b->ord_dirty.p;
    spec.ord_seq = ++b->ord_seq;   // (a fresh number per attempt: a step redone with exact counts starts clean)
  }
  ApplyOutcome oc;
  if (deferred_bound) {
    spec.d_nh = b->count_dev.p;
    if (packed_exchange)
      spec.unpack_chain = &unp_args;
    else {
      spec.sel_chain = &sel_args;
      spec.forces = &b->pending;
    }
  }
  {   // (reference-order step whose selection launch did not carry the grid's step-start copy: now, ahead of the hills)
    int rs = ordered_snapshot_now(b);
    if (rs) return rs;
  }
  struct EarlyCtx {
    edm_hip_bias *b;
    double this_h;
    const long long *d_sel;
    long long bound;
    bool packed;
    long long pack_bound;
  } early_ctx{b, this_h, d_sel, nh, packed_exchange, pack_bound};
  b->ord_on_own_stream = false;
  if (b->ord_early.armed && deferred_bound && (packed_exchange || !b->comm)) {
    if (spec.ord_terms && !b->bias->shared_device && !b->ord_own_stream_off) {
      // a rank with the device to itself: record and force pass on their own stream, beside the batch's launch
      if (!b->ord_stream) {
        EDM_HIP_TRY(hipStreamCreateWithFlags(&b->ord_stream, hipStreamNonBlocking));
        EDM_HIP_TRY(hipEventCreateWithFlags(&b->ord_done_event, hipEventDisableTiming));
      }
      EDM_HIP_TRY(b->ord_ready.reserve_zeroed((size_t)2 * (size_t)(nh > 4096 ? nh : 4096)));
      EDM_HIP_TRY(b->ord_status.reserve_zeroed(16));
      if (!b->h_ord_status) {
        EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_ord_status), 64, hipHostMallocMapped));
        EDM_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_ord_status), b->h_ord_status, 0));
        *reinterpret_cast<volatile int *>(b->h_ord_status) = 0;
      }
      spec.ord_ready = b->ord_ready.p;
    }
    spec.before_wait_ctx = &early_ctx;
    spec.before_wait = [](void *ctx, const double *d_base, const double *d_t1, const double *d_t2, const LimitResult *d_res,
                          bool terms_emitted, const unsigned long long *ready_flag, unsigned long long ready_seq,
                          const long long *d_nh) {
      EarlyCtx *c = static_cast<EarlyCtx *>(ctx);
      edm_hip_bias *bb = c->b;
      bb->last_batch = edm_hip_bias::LastBatch();
      bb->last_batch.valid = true;
      bb->last_batch.nh = c->bound;
      bb->last_batch.heights = d_base;
      bb->last_batch.h_const = c->this_h;
      bb->last_batch.tail_h1 = d_t1;
      bb->last_batch.tail_h2 = d_t2;
      bb->last_batch.sel = c->d_sel;
      bb->last_batch.terms_emitted = terms_emitted;
      bb->last_batch.res_dev = d_res;
      if (c->packed) {   // (this rank's slice of the rank-major global list: where k_unpack_prep put it)
        bb->last_batch.sel = bb->sel.p;
        bb->last_batch.range_dev = bb->ord_range.p;
        bb->last_batch.local_cap = c->pack_bound;
      }
      if (terms_emitted && ready_flag && d_nh && bb->ord_stream && bb->ord_ready.p && bb->d_ord_status && !bb->ord_own_stream_off) {
        // beside the batch's launch, behind nothing on the host's side: the record pass waits in the kernel for the
        // limiter's word of THIS batch (an event behind the preparation cost the object's stream ~5 us between the
        // selection and the batch, and the other stream ~10 us until the dependency had resolved)
        if (bb->ord_wait_uploads && bb->ord_up_event) (void)hipStreamWaitEvent(bb->ord_stream, bb->ord_up_event, 0);
        bb->ord_on_own_stream = true;
        bb->last_batch.wait_flag = ready_flag;
        bb->last_batch.wait_seq = ready_seq;
        bb->last_batch.d_nh = d_nh;
      }
      bb->ord_early.rc = ordered_forces_enqueue(bb);
      if (bb->ord_on_own_stream) {
        // whatever comes next on the object's stream (and with it the null stream) is ordered behind the force pass, as it
        // was when the pass ran on that stream: the forces it stores are the call's device output
        if (hipEventRecord(bb->ord_done_event, bb->ord_stream) != hipSuccess ||
            hipStreamWaitEvent(bb->bias->stream, bb->ord_done_event, 0) != hipSuccess) {
          (void)hipGetLastError();
          if (!bb->ord_early.rc) bb->ord_early.rc = EDM_HIP_ERR_HIP;
        }
      }
      bb->ord_early.done = true;
      ht_mark(bb->bias, 9);
    };
  }
  int rc = apply_hills(b->bias, spec, &oc, false);
  if (rc == EDM_APPLY_BOUND_EXCEEDED) {
    // (what the early force pass computed is void: the step's hill path is redone below, and the force pass that follows
    //  it tags its partial sums with a number of its own -- the void pass has written the old one)
    b->ord_early.done = false;
    if (b->ord_early.tag) b->ord_early.tag = ++b->bias->force_seq;
    if (b->ord_on_own_stream) {   // (the void passes on their own stream: out of the way before the redo reuses their buffers)
      EDM_HIP_TRY(hipStreamSynchronize(b->ord_stream));
      b->ord_on_own_stream = false;
    }
    // (practically never) more hills than the launch bound -- on every rank alike, since the count is
    // global: nothing was applied; redo the step's hill path the synchronous way with exact counts
    b->force_sync = true;
    b->bound_redos++;
    rc = process_new_hills(b, n, d_x_in, x_stride_in, d_ru, apply_mask);
    b->force_sync = false;
    return rc;
  }
  if (rc) return rc;
  const LimitResult &res = oc.res;
  b->temp_hill_cum = res.cum_out;
  b->last_batch.valid = true;
  b->last_batch.nh = res.nh;
  b->last_batch.k = res.k;
  b->last_batch.heights = oc.d_base_heights;
  b->last_batch.h_const = this_h;
  b->last_batch.tail_h1 = oc.d_tail_h1;
  b->last_batch.tail_h2 = oc.d_tail_h2;
  b->last_batch.sel = d_sel;
  b->last_batch.local_cnt = -1;
  b->last_batch.range_dev = nullptr;
  b->last_batch.res_dev = nullptr;
  b->last_batch.wait_flag = nullptr;
  b->last_batch.terms_emitted = oc.terms_emitted;
  if (packed_exchange) {
    b->last_batch.sel = b->sel.p;
    b->last_batch.range_dev = b->ord_range.p;
    b->last_batch.local_cap = pack_bound;
  } else if (b->comm) {
    long long off = 0;
    for (int r = 0; r < b->rank; r++) off += b->xchg_counts[r];
    b->last_batch.sel = local_sel;
    b->last_batch.local_off = off;
    b->last_batch.local_cnt = local_nh;
  }

  const long long k = res.k;
  const int ntail = res.n_tail;
  const unsigned int dim = b->dim;
  if (oc.deferred_fetch) {
    // every hill was added in full and the host was released by the limiter's header line: positions and per-hill bias
    // are still on their way -- the 'h' lines are written when the next cycle begins (resolve_deferred_log)
    b->deferred_log.active = true;
    b->deferred_log.bound = oc.deferred_bound;
    b->deferred_log.nh = res.nh;
    b->deferred_log.steps = b->steps;
    b->deferred_log.hills_added_before = b->hills_added;
    b->deferred_log.height = this_h;
    b->deferred_log.cum_over_volume = b->cum_bias / b->total_volume;
    b->hills_added += (int)res.nh;
    return EDM_HIP_OK;
  }
  const long long first = oc.first;
  // base height of hill i (constant, or per hill when a target is set)
  auto height_of = [&](long long i) { return oc.heights.empty() ? this_h : oc.heights[(size_t)(i - first)]; };
  if (log_all) {
    for (long long i = 0; i < k; i++) {
      b->hills_added++;
      log_hill(b, &oc.pos[(size_t)(i - first) * dim], height_of(i), oc.added[(size_t)(i - first)], 'h');
    }
  } else {
    b->hills_added += (int)k;
  }
  if (oc.plain_fast) {   // every hill of the tail was added in full (and there is no log to write): nothing to replay
    b->hills_added += ntail;
    return EDM_HIP_OK;
  }
  // ordered tail: the log lines, and the overflow appends of edm_bias.cpp:498-523
  for (int j = 0; j < ntail; j++) {
    const double *p = &oc.pos[(size_t)(k - first + j) * dim];
    const int fl = oc.flags[(size_t)j];
    if (fl & 1) {
      b->hills_added++;
      log_hill(b, p, height_of(k + j), oc.added.empty() ? 0.0 : oc.added[(size_t)(k - first + j)], 'h');
      if (fl & 2) {
        b->hills_added++;
        log_hill(b, p, oc.h2[(size_t)j], oc.a2[(size_t)j], 'u');
        rc = overflow_push(b, p, -oc.h2[(size_t)j]);   // remainder goes to the buffer (:489)
        if (rc) return rc;
      }
    } else {
      log_hill(b, p, 0, 0, 'h');                       // :493
      rc = overflow_push(b, p, height_of(k + j));
      if (rc) return rc;
    }
  }
  return EDM_HIP_OK;
}

// post_add_hill (edm_bias.cpp:565-583) + update_height (:922-931)
static int do_post_add_hill(edm_hip_bias *b) {
  double step_bias = b->temp_hill_cum;
  if (b->comm) {
    // MPI_Allreduce(temp_hill_cum_) (:925): every rank replays the same global hill list, so the ranks'
    // temp_hill_cum_ are identical by construction and their sum needs no collective
    step_bias *= (double)b->nranks;
  }
  b->cum_bias += step_bias;
  b->temp_hill_cum = -1;
  b->temp_hill_prefactor = -1;
  b->steps++;
  b->rng_cycle++;
  if (!b->hill_events.empty()) b->hills.submit(b->hill_events);   // (formatted, written and flushed by the writer thread)
  return EDM_HIP_OK;
}

extern "C" {

int edm_hip_bias_add_hills(edm_hip_bias *b, long long n, const double *d_x, int x_stride, const double *d_runiform,
                           int apply_mask, long long est_hill_count) {
  int rc = do_pre_add_hill(b, est_hill_count < 0 ? n : est_hill_count);   // :404
  if (rc) return rc;
  if (!b->b_outofbounds) {
    rc = process_new_hills(b, n, d_x, x_stride, d_runiform, apply_mask);
    if (rc) return rc;
  }
  return do_post_add_hill(b);
}

int edm_hip_bias_step(edm_hip_bias *b, long long n, const double *d_x, int x_stride, double *d_f, int f_stride,
                      const double *d_runiform, int apply_mask, long long est_hill_count, double *energy) {
  if (energy) *energy = 0;
  if (!b->bias && !b->b_outofbounds) {
    set_error("step before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  int nblk = 0;
  unsigned long long ftag = 0;
  bool lookup_pending = false;
  if (!b->b_outofbounds) {   // update_forces (:279-280) -- queued, not waited for
    // (tagged partial energy sums: should no polled hill batch follow, the host looks at the slots instead of waiting
    //  for the stream, see edm_hip_gauss_pair_forces)
    if (n > 0 && forces_poll_enabled()) ftag = ++b->bias->force_seq;
    if (n > 0 && b->dim > 1) {
      // 2-D / 3-D: kept pending until the overflow flush queues its preparation -- the two share a launch
      // (launch_lookup_prep); whoever queues anything else first launches it on its own (pending_forces_flush)
      if (apply_mask >= 0 && !b->d_mask) {
        set_error("update_forces: apply_mask >= 0 needs a mask");
        return EDM_HIP_ERR_ARG;
      }
      b->pending = PendingForces();
      b->pending.active = true;
      b->pending.lookup = true;
      b->pending.la = LookupArgs{};
      b->pending.la.n = n;
      b->pending.la.x = d_x;
      b->pending.la.x_stride = x_stride;
      b->pending.la.f = d_f;
      b->pending.la.f_stride = f_stride;
      b->pending.la.mask = b->d_mask;
      b->pending.la.apply_mask = apply_mask;
      b->pending.la.partial_tag = ftag;
      b->pending.on_launched = b->step_hook;
      b->pending.on_launched_ctx = b->step_hook_ctx;
      lookup_pending = true;
    } else {
      int rc = update_forces_enqueue(b->bias, n, d_x, x_stride, d_f, f_stride, b->d_mask, apply_mask, &nblk, ftag);
      if (rc) return rc;
      if (n > 0 && b->step_hook) b->step_hook(b->step_hook_ctx);
    }
  }
  // add_hills behind it on the same stream (:401-411): pre_add_hill, the samples, post_add_hill
  // (the force kernel is ALREADY queued: a polled batch of the overflow flush inside pre_add_hill is behind it on the
  //  stream and shows it complete just as a polled batch of new hills does -- a step whose new hills are skipped,
  //  edm_bias.cpp:534-535, must not fall back to a stream wait for that: 20-30 us of idle GPU per step on W4)
  if (b->bias) b->bias->wait_polled = false;
  b->flush_forces = lookup_pending ? &b->pending : nullptr;
  int rc = do_pre_add_hill(b, est_hill_count < 0 ? n : est_hill_count);
  b->flush_forces = nullptr;
  if (rc) {
    if (lookup_pending) (void)pending_forces_flush(b->bias, &b->pending);
    return rc;
  }
  if (!b->b_outofbounds) {
    const bool flush_polled = b->bias->wait_polled;
    // (no flush this step, or one that could not carry the force kernel: it is still pending, and the new hills'
    //  selection can carry it -- select_prep_enqueue; whatever is left pending after that goes on its own)
    rc = process_new_hills(b, n, d_x, x_stride, d_runiform, apply_mask);
    if (lookup_pending) {
      int rcf = pending_forces_flush(b->bias, &b->pending);
      nblk = b->pending.nblk;
      if (!rc) rc = rcf;
    }
    if (flush_polled) b->bias->wait_polled = true;   // (whatever the new hills did: the forces were seen complete)
    if (rc) return rc;
    double e = 0;
    if (ftag && !b->bias->wait_polled && poll_tagged_partials(b->bias, nblk, ftag, &e)) {
      b->bias->polled_forces++;
    } else {
      if (!b->bias->wait_polled) EDM_HIP_TRY(hipStreamSynchronize(b->bias->stream));
      e = 0;
      for (int k = 0; k < nblk; k++) e += b->bias->h_partials[ftag ? 2 * k : k];
    }
    if (energy) *energy = e;
  }
  return do_post_add_hill(b);
}

int edm_hip_bias_pair_step(edm_hip_bias *b, long long n, const double *d_r, double *d_force, long long n_samples,
                           const double *d_sample_r, const double *d_runiform, long long est_hill_count,
                           double *energy) {
  if (energy) *energy = 0;
  if (!b->bias && !b->b_outofbounds) {
    set_error("pair_step before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  if (b->dim != 1) {
    set_error("pair_step: the pair-distance CV is 1-D (fix_edm_pair.cpp:52)");
    return EDM_HIP_ERR_ARG;
  }
  static const bool host_trace = getenv("EDM_HIP_TRACE") != nullptr;
  static double last_exit_us = 0;
  const double t_in = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (host_trace && b->bias) b->bias->ht_ref_us = t_in;
  // pre_add_hill first, as fix_edm_pair does (:174): a pending overflow flush is part of the bias the
  // forces see
  int rc = do_pre_add_hill(b, est_hill_count < 0 ? n_samples : est_hill_count);
  if (rc) return rc;
  if (b->b_outofbounds) {
    if (n > 0) EDM_HIP_TRY(hipMemset(d_force, 0, sizeof(double) * (size_t)n));
    return do_post_add_hill(b);
  }
  ht_mark(b->bias, 0);
  // forces (queued, not waited for), then the new hills behind them on the same stream: one host wait
  // (where the step's selection runs as a chained launch, the force kernel rides in that launch)
  b->pending = PendingForces();
  b->bias->wait_polled = false;
  if (n > 0) {
    b->pending.active = true;
    b->pending.n = n;
    b->pending.d_r = d_r;
    b->pending.d_force = d_force;
  }
  rc = process_new_hills(b, n_samples, d_sample_r, 1, d_runiform, -1);
  ht_mark(b->bias, 7);
  // (no hills this step, or they were skipped: the force kernel goes alone -- with tagged partial sums the host can
  //  look at instead of waiting for the stream, see edm_hip_gauss_pair_forces)
  unsigned long long tag = 0;
  if (b->pending.active && forces_poll_enabled()) tag = b->pending.tag = ++b->bias->force_seq;
  int rcf = pending_forces_flush(b->bias, &b->pending);
  if (rc) return rc;
  if (rcf) return rcf;
  double e = 0;
  if (tag && b->pending.tagged && poll_tagged_partials(b->bias, b->pending.nblk, tag, &e)) {
    b->bias->polled_forces++;
  } else {
    // (a polled hill batch has shown the stream's last kernel past its read-back: the force kernel, earlier on the
    //  stream, is complete and its partial sums are in host memory)
    if ((tag && b->pending.tagged) || !b->bias->wait_polled) EDM_HIP_TRY(hipStreamSynchronize(b->bias->stream));
    e = 0;
    for (int k = 0; k < b->pending.nblk; k++) e += b->bias->h_partials[(tag && b->pending.tagged) ? 2 * k : k];
  }
  if (energy) *energy = e;
  ht_mark(b->bias, 8);
  rc = do_post_add_hill(b);
  if (host_trace) {
    const double t_out = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (b->bias->ready_seq == 160) {
      fprintf(stderr, "[edm host] pair_step: %.2f us inside the call, %.2f us since the previous call returned\n", t_out - t_in,
              t_in - last_exit_us);
      const double *m = b->bias->ht_marks;
      fprintf(stderr, "[edm host] marks (us after entry): pre_add_hill done %.2f | first launch %.2f -> %.2f | second launch %.2f -> %.2f | "
              "poll %.2f -> %.2f | new hills processed %.2f | energy summed %.2f | exit %.2f\n", m[0], m[1], m[2], m[3], m[4], m[5], m[6],
              m[7], m[8], t_out - t_in);
    }
    last_exit_us = t_out;
  }
  return rc;
}

// fix edm_pair's hill step in the REFERENCE'S order (lammps/fix_edm_pair.cpp:173-247): the step's hills are applied as
// in edm_hip_bias_pair_step -- selection, limiter and grid update do not depend on the forces -- and the force of pair k
// is then interpolated on the bias as it stood when the reference's loop reached that pair: the grid after
// pre_add_hill plus the hills of the add_hill calls before it (OrderedForcesArgs, edm_kernels.h).
// the bias the first pair of a reference-order step sees: the node records behind pre_add_hill's overflow flush
static int ordered_snapshot(edm_hip_bias *b) {
  edm_hip_gauss *g = b->bias;
  if (!ordered_forces_supported(g->g)) {
    set_error("reference-order pair step: needs a 1-D bias whose stencil is not wider than a periodic grid");
    return EDM_HIP_ERR_ARG;
  }
  const size_t grid_doubles = (size_t)g->g.total * (size_t)g->g.rec;
  EDM_HIP_TRY(b->ord_rec0.reserve(grid_doubles));
  b->ord_snap_pending = true;
  b->last_batch.valid = false;
  b->ord_step_active = true;
  return EDM_HIP_OK;
}
// ... deferred: the step's selection launch carries the copy (SelectArgs::snap_*); whoever is about to touch the grid
// without having launched a selection makes it here
static void ordered_snapshot_ride(edm_hip_bias *b, SelectArgs *a) {
  if (!b->ord_snap_pending || b->bias->g.rec != 2) return;
  a->snap_src = b->bias->rec;
  a->snap_dst = b->ord_rec0.p;
  a->snap_n = (long long)b->bias->g.total;   // (1-D interpolating grid: records of two doubles)
  b->ord_snap_pending = false;
}
static int ordered_snapshot_now(edm_hip_bias *b) {
  if (!b->ord_snap_pending) return EDM_HIP_OK;
  b->ord_snap_pending = false;
  edm_hip_gauss *g = b->bias;
  EDM_HIP_TRY(hipMemcpyAsync(b->ord_rec0.p, g->rec, sizeof(double) * (size_t)g->g.total * (size_t)g->g.rec, hipMemcpyDeviceToDevice,
                             g->stream));
  return EDM_HIP_OK;
}
// ... and, once the step's hill batch has been applied (last_batch), the running records of its hills
static int ordered_records_enqueue(edm_hip_bias *b, OrderedForcesArgs *out) {
  edm_hip_gauss *g = b->bias;
  const bool sliced = b->last_batch.range_dev != nullptr || b->last_batch.local_cnt >= 0;
  const long long nh = b->last_batch.range_dev ? b->last_batch.local_cap
                       : (b->last_batch.local_cnt >= 0 ? b->last_batch.local_cnt : b->last_batch.nh);
  if (nh > ordered_max_hills()) {
    set_error("reference-order pair step: more than 16384 hills in one step; use edm_hip_bias_pair_step (all forces on the "
              "step-start bias) for all-samples deposition of a large system");
    return EDM_HIP_ERR_ARG;
  }
  long long cap = 256;   // (grown in powers of two: the buffers settle after the first steps)
  while (cap < nh) cap *= 2;
  if (ordered_record_doubles(g->g, cap) * sizeof(double) > ((size_t)4 << 30)) cap = nh;
  EDM_HIP_TRY(b->ord_records.reserve(ordered_record_doubles(g->g, cap)));
  EDM_HIP_TRY(b->ord_counts.reserve(ordered_count_shorts(g->g, cap)));
  EDM_HIP_TRY(b->ord_dirty.reserve_zeroed((size_t)(b->last_batch.nh > 4096 ? b->last_batch.nh : 4096)));
  OrderedForcesArgs a;
  memset(&a, 0, sizeof(a));
  a.nh = nh;
  a.nh_cap = cap;
  a.hill_off = sliced ? b->last_batch.local_off : 0;
  a.range_dev = b->last_batch.range_dev;
  a.res_dev = b->last_batch.res_dev;
  if (b->last_batch.wait_flag) {
    a.res_dev = nullptr;
    a.wait_flag = b->last_batch.wait_flag;
    a.wait_seq = b->last_batch.wait_seq;
    a.nh_dev = b->last_batch.d_nh;
    a.terms_ready = b->ord_ready.p;
    a.status = b->ord_status.p;
    a.status_host = b->d_ord_status;
    // (behind a collective the batch's launch starts when the slowest rank has arrived: 200 ms instead of 2)
    a.gate_ticks = b->comm ? 20000000ull : 0ull;
  }
  a.k = b->last_batch.k;
  a.heights = b->last_batch.heights;
  a.h_const = b->last_batch.h_const;
  a.tail_h1 = b->last_batch.tail_h1;
  a.tail_h2 = b->last_batch.tail_h2;
  a.hx = g->ws.hx.p;
  a.hc = g->ws.hc.p;
  a.ht = g->ws.ht.p;
  a.sel = b->last_batch.sel;
  a.rec0 = b->ord_rec0.p;
  a.records = b->ord_records.p;
  a.counts = b->ord_counts.p;
  a.dirty_hill = b->ord_dirty.p;
  if (b->last_batch.terms_emitted) {   // (the emitters of the batch's launch noted the first dirty hill under this number)
    a.terms = b->ord_terms.p;
    a.terms_rows = b->ord_terms_rows;
    a.dirty_seq = b->ord_seq;
  } else {
    a.dirty_seq = ++b->ord_seq;
  }
  // development aid (EDM_HIP_TRACE=ordered): stamps of the 100th record pass to stderr
  static const bool tracing = getenv("EDM_HIP_TRACE") && !strcmp(getenv("EDM_HIP_TRACE"), "ordered");
  const size_t trace_wgs = (size_t)((g->g.n[0] + 31) / 32);
  if (tracing && b->ord_seq == 100) {
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a.trace), trace_wgs * 64));
    EDM_HIP_TRY(hipMemset(a.trace, 0, trace_wgs * 64));
  }
  EDM_HIP_TRY(launch_ordered_records(g->g, g->tables(), a, b->ord_on_own_stream ? b->ord_stream : g->stream));
  if (a.trace) {
    EDM_HIP_TRY(hipStreamSynchronize(b->ord_on_own_stream ? b->ord_stream : g->stream));
    std::vector<unsigned long long> tr(trace_wgs * 8);
    EDM_HIP_TRY(hipMemcpy(tr.data(), a.trace, trace_wgs * 64, hipMemcpyDeviceToHost));
    (void)hipFree(a.trace);
    a.trace = nullptr;
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < trace_wgs; w++) if (tr[w * 8] && tr[w * 8] < t0) t0 = tr[w * 8];
    const char *names[7] = {"start", "node terms done", "chunk 0 listed", "chunk 0 terms", "chunk 0 run", "chunk 0 stored", "end"};
    for (int k = 0; k < 7; k++) {
      std::vector<double> v;
      for (size_t w = 0; w < trace_wgs; w++) if (tr[w * 8 + k]) v.push_back((double)(tr[w * 8 + k] - t0) * 0.01);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      fprintf(stderr, "[edm trace] records %-16s n=%4zu  min %6.2f  med %6.2f  max %6.2f us\n", names[k], v.size(), v.front(), v[v.size() / 2], v.back());
    }
    unsigned long long mx = 0, sum = 0;
    for (size_t w = 0; w < trace_wgs; w++) { sum += tr[w * 8 + 7]; if (tr[w * 8 + 7] > mx) mx = tr[w * 8 + 7]; }
    fprintf(stderr, "[edm trace] records: hills listed per tile in chunk 0: mean %.1f max %llu (of %lld hills)\n", (double)sum / trace_wgs, mx, nh);
  }
  *out = a;
  return EDM_HIP_OK;
}

// records of the batch's hills, then the force pass that reads them (b->last_batch, b->ord_early: the pairs)
static int ordered_forces_enqueue(edm_hip_bias *b) {
  edm_hip_gauss *g = b->bias;
  hipStream_t s = b->ord_on_own_stream ? b->ord_stream : g->stream;
  OrderedForcesArgs a;
  int rc = ordered_records_enqueue(b, &a);
  if (rc) return rc;
  if (b->ord_early.list) {
    PairListArgs pl = b->ord_early.pl;
    pl.partial_tag = b->ord_early.tag;
    EDM_HIP_TRY(launch_pairlist_forces_ordered(g->g, pl, a, g->d_partials, s, &b->ord_early.nblk));
    return EDM_HIP_OK;
  }
  a.n = b->ord_early.n;
  a.r = b->ord_early.d_r;
  a.first_sample = b->ord_early.d_first;
  a.force = b->ord_early.d_force;
  hipEvent_t e0, e1;
  profile_slot(g, &e0, &e1);
  // development aid (EDM_HIP_TRACE=k1o): stamps of the 100th force pass to stderr
  static const bool tracing = getenv("EDM_HIP_TRACE") && !strcmp(getenv("EDM_HIP_TRACE"), "k1o");
  const size_t trace_wgs = 65536;
  if (tracing && b->ord_seq == 100) {
    EDM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a.trace), trace_wgs * 64));
    EDM_HIP_TRY(hipMemset(a.trace, 0, trace_wgs * 64));
  }
  EDM_HIP_TRY(launch_pair_forces_ordered(g->g, a, g->d_partials, s, &b->ord_early.nblk, b->ord_early.tag, e0, e1));
  if (a.trace) {
    EDM_HIP_TRY(hipStreamSynchronize(s));
    std::vector<unsigned long long> tr(trace_wgs * 8);
    EDM_HIP_TRY(hipMemcpy(tr.data(), a.trace, trace_wgs * 64, hipMemcpyDeviceToHost));
    (void)hipFree(a.trace);
    a.trace = nullptr;
    unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < trace_wgs; w++) if (tr[w * 8] && tr[w * 8] < t0) t0 = tr[w * 8];
    const char *names[5] = {"start", "hills staged", "rows staged", "first trip done", "last trip done"};
    for (int k = 0; k < 5; k++) {
      std::vector<double> v;
      for (size_t w = 0; w < trace_wgs; w++) if (tr[w * 8 + k]) v.push_back((double)(tr[w * 8 + k] - t0) * 0.01);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      fprintf(stderr, "[edm trace] k1o %-16s n=%4zu  min %6.2f  p25 %6.2f  med %6.2f  p75 %6.2f  max %6.2f us\n", names[k], v.size(),
              v.front(), v[v.size() / 4], v[v.size() / 2], v[3 * v.size() / 4], v.back());
    }
  }
  return EDM_HIP_OK;
}

// The passes on the second stream left without doing their work: their gate wave was let in ahead of the batch it waited
// for and gave up (k_wait_word) -- something runs the process's kernels one at a time, a profiler collecting hardware
// counters does.  The batch is through by now (the caller has waited for it): both passes again, behind it on the
// object's stream, with a fresh tag; the second stream stays unused from here on.
static int ordered_step_redo_after_gate(edm_hip_bias *b, int *nblk) {
  edm_hip_gauss *g = b->bias;
  EDM_HIP_TRY(hipStreamSynchronize(b->ord_stream));
  EDM_HIP_TRY(hipStreamSynchronize(g->stream));
  b->ord_on_own_stream = false;
  b->ord_own_stream_off = true;
  b->ord_gate_giveups++;
  if (b->ord_early.tag) b->ord_early.tag = ++g->force_seq;
  // (a batch that accepted no hill leaves a record of zero hills: the passes then read the grid's step-start copy, which
  //  is the grid)
  if (!b->last_batch.valid) {
    set_error("reference-order step: the force pass has to be queued again but the batch's record is gone");
    return EDM_HIP_ERR_STATE;
  }
  int rc = ordered_forces_enqueue(b);
  if (rc) return rc;
  *nblk = b->ord_early.nblk;
  return EDM_HIP_OK;
}

static int pair_step_ordered_device(edm_hip_bias *b, long long n, const double *d_r, double *d_force,
                                    const int *d_first_sample, long long n_samples, const double *d_sample_r,
                                    const double *d_runiform, double *energy) {
  edm_hip_gauss *g = b->bias;
  hipStream_t s = g->stream;
  int rc = ordered_snapshot(b);
  if (rc) return rc;
  b->pending = PendingForces();
  g->wait_polled = false;
  const unsigned long long tag0 = forces_poll_enabled() ? ++g->force_seq : 0;
  b->ord_early = edm_hip_bias::OrderedEarly();
  b->ord_early.n = n;
  b->ord_early.d_r = d_r;
  b->ord_early.d_first = d_first_sample;
  b->ord_early.d_force = d_force;
  b->ord_early.tag = tag0;
  b->ord_early.armed = n > 0;
  rc = process_new_hills(b, n_samples, d_sample_r, 1, d_runiform, -1);
  ht_mark(g, 7);
  b->ord_early.armed = false;
  b->ord_step_active = false;
  b->ord_snap_pending = false;   // (no hill batch was applied: nobody needs the copy)
  if (rc) {
    if (b->ord_on_own_stream) (void)hipStreamSynchronize(b->ord_stream);
    return rc;
  }
  int nblk = 0;
  bool tagged = false;
  const unsigned long long tag = b->ord_early.tag;   // (a redone step has taken a fresh one)
  if (b->ord_early.done) {
    // (the batch's deferred log lines -- positions and per-hill bias from the read-back region -- are picked up NOW, while
    //  the record and force pass run: at the start of the next cycle, where they would be fetched otherwise, they sit in
    //  front of its first launch, ~2 us)
    rc = resolve_deferred_log(b);
    if (rc) return rc;
  }
  if (b->ord_early.done) {
    // (the force pass went out behind the hill batch, before the host had the limiter's result)
    if (b->ord_early.rc) return b->ord_early.rc;
    nblk = b->ord_early.nblk;
    tagged = tag != 0;
  } else if (n > 0 && b->last_batch.valid && b->last_batch.nh > 0) {
    rc = ordered_forces_enqueue(b);
    if (rc) return rc;
    nblk = b->ord_early.nblk;
    tagged = tag != 0;
  } else if (n > 0) {
    // no new hill this step (none accepted, or edm_bias.cpp:534-535 skipped them): every pair sees the same bias
    b->pending.active = true;
    b->pending.n = n;
    b->pending.d_r = d_r;
    b->pending.d_force = d_force;
    b->pending.tag = tag;
    rc = pending_forces_flush(g, &b->pending);
    if (rc) return rc;
    nblk = b->pending.nblk;
    tagged = tag && b->pending.tagged;
  }
  double e = 0;
  if (!(tagged && poll_tagged_partials(g, nblk, tag, &e))) {
    if (b->ord_on_own_stream) EDM_HIP_TRY(hipStreamSynchronize(b->ord_stream));
    EDM_HIP_TRY(hipStreamSynchronize(s));
    if (b->ord_on_own_stream && *reinterpret_cast<volatile int *>(b->h_ord_status) == 2) {
      rc = ordered_step_redo_after_gate(b, &nblk);
      if (rc) return rc;
      EDM_HIP_TRY(hipStreamSynchronize(s));
    }
    e = 0;
    for (int k = 0; k < nblk; k++) e += g->h_partials[tagged ? 2 * k : k];
  } else {
    g->polled_forces++;
  }
  if (energy) *energy = e;
  return EDM_HIP_OK;
}

int edm_hip_bias_pair_step_ordered(edm_hip_bias *b, long long n, const double *d_r, double *d_force,
                                   const int *d_first_sample, long long n_samples, const double *d_sample_r,
                                   const double *d_runiform, long long est_hill_count, double *energy) {
  if (energy) *energy = 0;
  if (!b->bias && !b->b_outofbounds) {
    set_error("pair_step before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  if (b->dim != 1) {
    set_error("pair_step: the pair-distance CV is 1-D (fix_edm_pair.cpp:52)");
    return EDM_HIP_ERR_ARG;
  }
  if (n < 0) n = 0;
  if (n_samples < 0) n_samples = 0;
  static const bool host_trace = getenv("EDM_HIP_TRACE") != nullptr;   // development aid: the call's host-side marks
  static double last_exit_us = 0;
  const double t_in = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (host_trace && b->bias) b->bias->ht_ref_us = t_in;
  int rc = do_pre_add_hill(b, est_hill_count < 0 ? n_samples : est_hill_count);
  if (rc) return rc;
  if (b->b_outofbounds) {
    if (n > 0) EDM_HIP_TRY(hipMemset(d_force, 0, sizeof(double) * (size_t)n));
    return do_post_add_hill(b);
  }
  ht_mark(b->bias, 0);
  rc = pair_step_ordered_device(b, n, d_r, d_force, d_first_sample, n_samples, d_sample_r, d_runiform, energy);
  if (rc) return rc;
  ht_mark(b->bias, 8);
  rc = do_post_add_hill(b);
  if (host_trace) {
    const double t_out = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (b->bias->ready_seq == 160) {
      const double *m = b->bias->ht_marks;
      fprintf(stderr, "[edm host] pair_step_ordered: %.2f us inside the call, %.2f us since the previous call returned\n"
              "[edm host] marks (us after entry): pre_add_hill done %.2f | selection launch %.2f -> %.2f | hill batch launch %.2f -> %.2f | "
              "force pass queued %.2f, poll %.2f -> %.2f | new hills processed %.2f | forces seen complete %.2f | exit %.2f\n",
              t_out - t_in, t_in - last_exit_us, m[0], m[1], m[2], m[3], m[4], m[9], m[5], m[6], m[7], m[8], t_out - t_in);
    }
    last_exit_us = t_out;
  }
  return rc;
}

// ... for a caller whose arrays live in HOST memory (the host-list fix edm_pair): copies queued on the object's stream
// around the kernels; the forces come down once the force pass -- the step's LAST kernel here -- has run
int edm_hip_bias_pair_step_ordered_host(edm_hip_bias *b, long long n, const double *h_r, double *h_force,
                                        const int *h_first_sample, long long n_samples, const double *h_sample_r,
                                        const double *h_runiform, long long est_hill_count, double *energy) {
  if (energy) *energy = 0;
  if (!b->bias && !b->b_outofbounds) {
    set_error("pair_step before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  if (b->dim != 1) {
    set_error("pair_step: the pair-distance CV is 1-D (fix_edm_pair.cpp:52)");
    return EDM_HIP_ERR_ARG;
  }
  if (n < 0) n = 0;
  if (n_samples < 0) n_samples = 0;
  int rc = do_pre_add_hill(b, est_hill_count < 0 ? n_samples : est_hill_count);
  if (rc) return rc;
  if (b->b_outofbounds) {
    if (n > 0) memset(h_force, 0, sizeof(double) * (size_t)n);
    return do_post_add_hill(b);
  }
  hipStream_t s = b->bias->stream;
  EDM_HIP_TRY(b->hs_r.reserve((size_t)(n > 0 ? n : 1)));
  EDM_HIP_TRY(b->hs_f.reserve((size_t)(n > 0 ? n : 1)));
  EDM_HIP_TRY(b->hs_x.reserve((size_t)(n_samples > 0 ? n_samples : 1)));
  EDM_HIP_TRY(b->hs_u.reserve((size_t)(n_samples > 0 ? n_samples : 1)));
  EDM_HIP_TRY(b->ord_first.reserve((size_t)(n > 0 ? n : 1)));
  // (whatever happens below, no copy may still be reading or writing the caller's arrays when the call returns)
  struct StreamGuard {
    hipStream_t s;
    ~StreamGuard() { (void)hipStreamSynchronize(s); }
  } guard{s};
  if (n > 0) {
    EDM_HIP_TRY(hipMemcpyAsync(b->hs_r.p, h_r, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    if (h_first_sample)
      EDM_HIP_TRY(hipMemcpyAsync(b->ord_first.p, h_first_sample, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
  }
  if (n_samples > 0) {
    EDM_HIP_TRY(hipMemcpyAsync(b->hs_x.p, h_sample_r, sizeof(double) * (size_t)n_samples, hipMemcpyHostToDevice, s));
    if (h_runiform)
      EDM_HIP_TRY(hipMemcpyAsync(b->hs_u.p, h_runiform, sizeof(double) * (size_t)n_samples, hipMemcpyHostToDevice, s));
  }
  if (!b->ord_up_event) EDM_HIP_TRY(hipEventCreateWithFlags(&b->ord_up_event, hipEventDisableTiming));
  EDM_HIP_TRY(hipEventRecord(b->ord_up_event, s));
  b->ord_wait_uploads = true;
  rc = pair_step_ordered_device(b, n, b->hs_r.p, b->hs_f.p, h_first_sample ? b->ord_first.p : nullptr, n_samples,
                                b->hs_x.p, h_runiform ? b->hs_u.p : nullptr, energy);
  b->ord_wait_uploads = false;
  if (rc) return rc;
  if (n > 0) EDM_HIP_TRY(hipMemcpyAsync(h_force, b->hs_f.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
  EDM_HIP_TRY(hipStreamSynchronize(s));
  return do_post_add_hill(b);
}

// The same step for a caller whose arrays live in HOST memory (the host-list fix edm_pair): staged through HBM with
// the copies queued around the kernels -- distances up, forces evaluated, then the forces travel down WHILE the hill
// samples and their uniforms travel up (opposite directions of the link, two streams), then the hill cycle.  With
// page-locked arrays (edm_hip_host_malloc) the copies are true DMA transfers; pageable arrays work, staged by the
// runtime.
int edm_hip_bias_pair_step_host(edm_hip_bias *b, long long n, const double *h_r, double *h_force, long long n_samples,
                                const double *h_sample_r, const double *h_runiform, long long est_hill_count,
                                double *energy) {
  if (energy) *energy = 0;
  if (!b->bias && !b->b_outofbounds) {
    set_error("pair_step before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  if (b->dim != 1) {
    set_error("pair_step: the pair-distance CV is 1-D (fix_edm_pair.cpp:52)");
    return EDM_HIP_ERR_ARG;
  }
  if (n < 0) n = 0;
  if (n_samples < 0) n_samples = 0;
  int rc = do_pre_add_hill(b, est_hill_count < 0 ? n_samples : est_hill_count);
  if (rc) return rc;
  if (b->b_outofbounds) {
    if (n > 0) memset(h_force, 0, sizeof(double) * (size_t)n);
    return do_post_add_hill(b);
  }
  hipStream_t s = b->bias->stream;
  if (!b->copy_stream) {
    EDM_HIP_TRY(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    EDM_HIP_TRY(hipEventCreateWithFlags(&b->copy_event, hipEventDisableTiming));
  }
  EDM_HIP_TRY(b->hs_r.reserve((size_t)(n > 0 ? n : 1)));
  EDM_HIP_TRY(b->hs_f.reserve((size_t)(n > 0 ? n : 1)));
  EDM_HIP_TRY(b->hs_x.reserve((size_t)(n_samples > 0 ? n_samples : 1)));
  EDM_HIP_TRY(b->hs_u.reserve((size_t)(n_samples > 0 ? n_samples : 1)));
  int nblk = 0;
  b->bias->wait_polled = false;
  // (whatever happens below -- a full overflow buffer, a communicator error -- no copy may still be reading or writing
  //  the caller's arrays when the call returns)
  struct CopyGuard {
    hipStream_t a, c;
    ~CopyGuard() {
      (void)hipStreamSynchronize(a);
      (void)hipStreamSynchronize(c);
    }
  } copy_guard{s, b->copy_stream};
  if (n > 0) {
    EDM_HIP_TRY(hipMemcpyAsync(b->hs_r.p, h_r, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    rc = pair_forces_enqueue(b->bias, n, b->hs_r.p, b->hs_f.p, &nblk);   // (behind the overflow flush, ahead of the new hills)
    if (rc) return rc;
    EDM_HIP_TRY(hipEventRecord(b->copy_event, s));
    EDM_HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->copy_event, 0));
    EDM_HIP_TRY(hipMemcpyAsync(h_force, b->hs_f.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, b->copy_stream));
  }
  const double *d_samples = b->hs_x.p;
  if (n_samples > 0) {
    if (h_sample_r == h_r && n_samples <= n)
      d_samples = b->hs_r.p;   // (the samples ARE the pair distances: already there)
    else
      EDM_HIP_TRY(hipMemcpyAsync(b->hs_x.p, h_sample_r, sizeof(double) * (size_t)n_samples, hipMemcpyHostToDevice, s));
    if (h_runiform)
      EDM_HIP_TRY(hipMemcpyAsync(b->hs_u.p, h_runiform, sizeof(double) * (size_t)n_samples, hipMemcpyHostToDevice, s));
  }
  b->pending = PendingForces();
  rc = process_new_hills(b, n_samples, d_samples, 1, h_runiform ? b->hs_u.p : nullptr, -1);
  if (rc) return rc;
  if (!b->bias->wait_polled) EDM_HIP_TRY(hipStreamSynchronize(s));
  if (n > 0) EDM_HIP_TRY(hipStreamSynchronize(b->copy_stream));
  const double e = pair_forces_finish(b->bias, nblk);
  if (energy) *energy = e;
  return do_post_add_hill(b);
}

// fix edm's post_force for a caller whose atom arrays live in HOST memory (lammps/fix_edm.cpp:134-162): positions up,
// the bias force comes back as a DELTA the host adds to atom->f -- the caller's force array is never uploaded (the
// reference's update_forces only ever subtracts dV/ds from it, edm_bias.cpp:287-293).  The position block is page-locked
// in place (hipHostRegister, remembered until the caller's pointer or size changes: LAMMPS keeps atom->x until it
// reallocates), so the upload is one DMA transfer; the delta lands in page-locked memory of the library's own.
static int host_block_register(edm_hip_bias *b, int slot, const void *p, size_t bytes) {
  if (b->reg_ptr[slot] == p && b->reg_bytes[slot] >= bytes) return EDM_HIP_OK;
  if (b->reg_ptr[slot]) {
    (void)hipHostUnregister(const_cast<void *>(b->reg_ptr[slot]));
    b->reg_ptr[slot] = nullptr;
    b->reg_bytes[slot] = 0;
  }
  if (b->reg_failed_ptr[slot] == p) return EDM_HIP_OK;   // (not registrable -- e.g. already page-locked by the caller: plain copies)
  if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess) {
    b->reg_ptr[slot] = p;
    b->reg_bytes[slot] = bytes;
  } else {
    (void)hipGetLastError();
    b->reg_failed_ptr[slot] = p;
  }
  return EDM_HIP_OK;
}

// the force delta's way down: pieces on the copy stream behind the launch that carries the force kernel, an event
// behind each -- the host adds piece k while piece k + 1 is on the link and the step's hills run on the main stream
struct DeltaCopy {
  edm_hip_bias *b;
  long long n;
  int dim, pieces;
  int rc;
  bool queued;
  long long per() const { return (((n + pieces - 1) / pieces) + 1) & ~1LL; }   // (even: pieces start 16-byte aligned)
};
static void delta_copy_queue(void *ctx) {
  DeltaCopy *d = static_cast<DeltaCopy *>(ctx);
  edm_hip_bias *b = d->b;
  if (d->queued || d->n <= 0) return;
  d->queued = true;
  auto ok = [&](hipError_t e) {
    if (e != hipSuccess && !d->rc) d->rc = (int)e;
    return e == hipSuccess;
  };
  if (!ok(hipEventRecord(b->copy_event, b->bias->stream))) return;
  if (!ok(hipStreamWaitEvent(b->copy_stream, b->copy_event, 0))) return;
  const long long per = d->per();
  for (int c = 0; c < d->pieces; c++) {
    const long long i0 = c * per, i1 = (i0 + per < d->n) ? i0 + per : d->n;
    if (i0 < i1)
      ok(launch_copy_to_host(b->hs_f.p + (size_t)i0 * d->dim, b->h_delta + (size_t)i0 * d->dim, (i1 - i0) * d->dim,
                             b->copy_stream));
    ok(hipEventRecord(b->delta_ev[c], b->copy_stream));
  }
}

int edm_hip_bias_step_host(edm_hip_bias *b, long long n, const double *h_x, int x_stride, double *h_f, int f_stride,
                           const int *h_mask, const double *h_runiform, int apply_mask, int hill_step,
                           long long est_hill_count, double *energy) {
  if (energy) *energy = 0;
  if (!b->bias && !b->b_outofbounds) {
    set_error("step before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  if (n < 0) n = 0;
  const int dim = (int)b->dim;
  if (n > 0 && (x_stride < dim || f_stride < dim)) {
    set_error("step_host: row strides shorter than the dimension");
    return EDM_HIP_ERR_ARG;
  }
  if (apply_mask >= 0 && n > 0 && !h_mask) {
    set_error("step_host: apply_mask >= 0 needs a mask");
    return EDM_HIP_ERR_ARG;
  }
  if (b->b_outofbounds) {
    if (!hill_step) return EDM_HIP_OK;
    int rc = do_pre_add_hill(b, est_hill_count < 0 ? n : est_hill_count);
    return rc ? rc : do_post_add_hill(b);
  }
  hipStream_t s = b->bias->stream;
  const double t_entry = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  const size_t xcount = n > 0 ? (size_t)(n - 1) * (size_t)x_stride + (size_t)dim : 0;
  EDM_HIP_TRY(b->hs_x.reserve(xcount > 0 ? xcount : 1));
  EDM_HIP_TRY(b->hs_f.reserve((size_t)(n > 0 ? n : 1) * dim));
  if (b->h_delta_cap < (size_t)n * dim) {
    if (b->h_delta) (void)hipHostFree(b->h_delta);
    b->h_delta = nullptr;
    b->h_delta_cap = (size_t)n * dim + (size_t)n * dim / 4 + 64;
    EDM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_delta), sizeof(double) * b->h_delta_cap, hipHostMallocDefault));
  }
  if (!b->copy_stream) {
    EDM_HIP_TRY(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    EDM_HIP_TRY(hipEventCreateWithFlags(&b->copy_event, hipEventDisableTiming));
  }
  // few, large pieces: each costs a launch and an event, and the add starts with the first
  constexpr int MAX_PIECES = 4;
  int PIECES = (int)(((size_t)n * dim * sizeof(double)) >> 21);
  PIECES = PIECES < 1 ? 1 : (PIECES > MAX_PIECES ? MAX_PIECES : PIECES);
  if (!b->delta_ev[0])
    for (int c = 0; c < MAX_PIECES; c++) EDM_HIP_TRY(hipEventCreateWithFlags(&b->delta_ev[c], hipEventDisableTiming));
  // (whatever happens below, no copy may still be reading or writing the caller's arrays when the call returns)
  struct StreamGuard {
    hipStream_t a, c;
    bool armed;
    ~StreamGuard() {
      if (!armed) return;
      (void)hipStreamSynchronize(a);
      (void)hipStreamSynchronize(c);
    }
  } guard{s, b->copy_stream, true};
  if (n > 0) {
    // (whole rows: a copy of the caller's own that spans the block must not straddle locked and unlocked memory)
    host_block_register(b, edm_hip_bias::REG_X, h_x, sizeof(double) * (size_t)n * (size_t)x_stride);
    EDM_HIP_TRY(hipMemcpyAsync(b->hs_x.p, h_x, sizeof(double) * xcount, hipMemcpyHostToDevice, s));
    EDM_HIP_TRY(hipMemsetAsync(b->hs_f.p, 0, sizeof(double) * (size_t)n * dim, s));
    if (apply_mask >= 0) {
      EDM_HIP_TRY(b->hs_mask.reserve((size_t)n));
      host_block_register(b, edm_hip_bias::REG_MASK, h_mask, sizeof(int) * (size_t)n);
      EDM_HIP_TRY(hipMemcpyAsync(b->hs_mask.p, h_mask, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
    }
    if (hill_step && h_runiform) {
      EDM_HIP_TRY(b->hs_u.reserve((size_t)n));
      host_block_register(b, edm_hip_bias::REG_U, h_runiform, sizeof(double) * (size_t)n);
      EDM_HIP_TRY(hipMemcpyAsync(b->hs_u.p, h_runiform, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    }
  }
  static const bool host_trace = getenv("EDM_HIP_TRACE") && strcmp(getenv("EDM_HIP_TRACE"), "step_host") == 0;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tr[4 + MAX_PIECES] = {0};
  if (host_trace) tr[1] = now_us();
  const int *saved_mask = b->d_mask;
  if (apply_mask >= 0) b->d_mask = b->hs_mask.p;
  DeltaCopy dc{b, n, dim, PIECES, 0, false};
  int rc;
  if (hill_step) {
    b->step_hook = delta_copy_queue;
    b->step_hook_ctx = &dc;
    rc = edm_hip_bias_step(b, n, b->hs_x.p, x_stride, b->hs_f.p, dim, h_runiform ? b->hs_u.p : nullptr, apply_mask,
                           est_hill_count, energy);
    b->step_hook = nullptr;
    b->step_hook_ctx = nullptr;
  } else {
    rc = edm_hip_bias_update_forces(b, n, b->hs_x.p, x_stride, b->hs_f.p, dim, apply_mask, energy);
  }
  b->d_mask = saved_mask;
  if (rc) return rc;
  if (host_trace) tr[2] = now_us();
  if (n > 0) {
    delta_copy_queue(&dc);   // (forces only, or a step that never launched them through the hook: now)
    if (dc.rc) {
      set_error(std::string("step_host: queueing the force delta's copies failed: ") + hipGetErrorString((hipError_t)dc.rc));
      return EDM_HIP_ERR_HIP;
    }
    DeltaAddJob job;
    job.h_f = h_f;
    job.dl = b->h_delta;
    job.n = n;
    job.per = dc.per();
    job.dim = dim;
    job.f_stride = f_stride;
    job.pieces = PIECES;
    job.ev = b->delta_ev;
    int dev = 0;
    (void)hipGetDevice(&dev);
    b->add_pool.resize(n * dim >= 65536 ? b->host_add_threads : 1, dev);
    const int rca = b->add_pool.run_job(job);
    if (rca) {
      set_error(std::string("step_host: waiting for the force delta failed: ") + hipGetErrorString((hipError_t)rca));
      return EDM_HIP_ERR_HIP;
    }
  }
  // (success: the step's completion record is behind the uploads on the main stream, every piece's event has been
  //  waited for on the copy stream -- nothing of the caller's is in flight, no stream wait needed: ~17 us each)
  guard.armed = false;
  if (host_trace) {   // development aid: the call's host marks, us since entry
    const double t_added = now_us();
    (void)hipStreamSynchronize(b->copy_stream);
    const double t_copy = now_us();
    (void)hipStreamSynchronize(s);
    fprintf(stderr, "[edm trace] step_host: uploads queued %.1f  step returned %.1f  delta added %.1f (%d pieces, %d threads)  copy stream idle %.1f  main stream idle %.1f\n",
            tr[1] - t_entry, tr[2] - t_entry, t_added - t_entry, PIECES, b->add_pool.threads(), t_copy - t_entry, now_us() - t_entry);
    for (int t = 0; t < b->add_pool.threads(); t++)
      fprintf(stderr, "[edm trace]   add thread %d: started %.1f  first piece seen %.1f  waited %.1f in all  done %.1f\n", t,
              g_add_trace[t][0] - t_entry, g_add_trace[t][1] - t_entry, g_add_trace[t][2], g_add_trace[t][3] - t_entry);
  }
  return EDM_HIP_OK;
}

int edm_hip_bias_pair_list_upload(edm_hip_bias *b, long long npairs, const int *h_pair_i, const int *h_pair_j,
                                  long long nall, const int *h_type) {
  if (npairs < 0 || nall < 0 || npairs > 1073741823LL) return EDM_HIP_ERR_ARG;   // (sample indices 2 e + slot are ints)
  for (long long p = 0; p < npairs; p++)
    if (h_pair_i[p] < 0 || h_pair_i[p] >= nall || h_pair_j[p] < 0 || h_pair_j[p] >= nall) {
      set_error("pair_list_upload: atom index outside [0, nall)");
      return EDM_HIP_ERR_ARG;
    }
  // entry indices grouped by atom (as i, as j), each group in list order: stable counting sort
  std::vector<long long> it_off((size_t)nall + 1, 0), jt_off((size_t)nall + 1, 0);
  for (long long p = 0; p < npairs; p++) {
    it_off[(size_t)h_pair_i[p] + 1]++;
    jt_off[(size_t)h_pair_j[p] + 1]++;
  }
  for (long long a = 0; a < nall; a++) {
    it_off[(size_t)a + 1] += it_off[(size_t)a];
    jt_off[(size_t)a + 1] += jt_off[(size_t)a];
  }
  std::vector<int> it_idx((size_t)(npairs > 0 ? npairs : 1)), jt_idx((size_t)(npairs > 0 ? npairs : 1));
  std::vector<int> it_ent((size_t)(npairs > 0 ? npairs : 1)), jt_ent((size_t)(npairs > 0 ? npairs : 1));
  {
    std::vector<long long> ci(it_off.begin(), it_off.end() - 1), cj(jt_off.begin(), jt_off.end() - 1);
    // (the OTHER atom of every entry, grouped by atom: what the force pass walks -- contiguous per atom; the entry's own
    //  index beside it, for the reference-order pass)
    for (long long p = 0; p < npairs; p++) {
      it_ent[(size_t)ci[(size_t)h_pair_i[p]]] = (int)p;
      it_idx[(size_t)ci[(size_t)h_pair_i[p]]++] = h_pair_j[p];
      jt_ent[(size_t)cj[(size_t)h_pair_j[p]]] = (int)p;
      jt_idx[(size_t)cj[(size_t)h_pair_j[p]]++] = h_pair_i[p];
    }
  }
  const size_t np1 = (size_t)(npairs > 0 ? npairs : 1), na1 = (size_t)(nall > 0 ? nall : 1);
  EDM_HIP_TRY(b->pl_i.reserve(np1));
  EDM_HIP_TRY(b->pl_j.reserve(np1));
  EDM_HIP_TRY(b->pl_it_idx.reserve(np1));
  EDM_HIP_TRY(b->pl_jt_idx.reserve(np1));
  EDM_HIP_TRY(b->pl_it_entry.reserve(np1));
  EDM_HIP_TRY(b->pl_jt_entry.reserve(np1));
  EDM_HIP_TRY(b->pl_type.reserve(na1));
  EDM_HIP_TRY(b->pl_it_off.reserve(na1 + 1));
  EDM_HIP_TRY(b->pl_jt_off.reserve(na1 + 1));
  if (npairs > 0) {
    EDM_HIP_TRY(hipMemcpy(b->pl_i.p, h_pair_i, sizeof(int) * (size_t)npairs, hipMemcpyHostToDevice));
    EDM_HIP_TRY(hipMemcpy(b->pl_j.p, h_pair_j, sizeof(int) * (size_t)npairs, hipMemcpyHostToDevice));
    EDM_HIP_TRY(hipMemcpy(b->pl_it_idx.p, it_idx.data(), sizeof(int) * (size_t)npairs, hipMemcpyHostToDevice));
    EDM_HIP_TRY(hipMemcpy(b->pl_jt_idx.p, jt_idx.data(), sizeof(int) * (size_t)npairs, hipMemcpyHostToDevice));
    EDM_HIP_TRY(hipMemcpy(b->pl_it_entry.p, it_ent.data(), sizeof(int) * (size_t)npairs, hipMemcpyHostToDevice));
    EDM_HIP_TRY(hipMemcpy(b->pl_jt_entry.p, jt_ent.data(), sizeof(int) * (size_t)npairs, hipMemcpyHostToDevice));
  }
  if (nall > 0) EDM_HIP_TRY(hipMemcpy(b->pl_type.p, h_type, sizeof(int) * (size_t)nall, hipMemcpyHostToDevice));
  EDM_HIP_TRY(hipMemcpy(b->pl_it_off.p, it_off.data(), sizeof(long long) * ((size_t)nall + 1), hipMemcpyHostToDevice));
  EDM_HIP_TRY(hipMemcpy(b->pl_jt_off.p, jt_off.data(), sizeof(long long) * ((size_t)nall + 1), hipMemcpyHostToDevice));
  b->pl_npairs = npairs;
  b->pl_nall = nall;
  b->pl_mask_valid = false;
  return EDM_HIP_OK;
}

int edm_hip_bias_pair_list_step(edm_hip_bias *b, int nlocal, int itype, int jtype, const double *d_x, double *d_fdelta,
                                int hill_step, long long est_hill_count, double *energy, long long *ncalls) {
  if (energy) *energy = 0;
  if (ncalls) *ncalls = 0;
  if (!b->bias && !b->b_outofbounds) {
    set_error("pair_list_step before subdivide");
    return EDM_HIP_ERR_STATE;
  }
  if (b->pl_npairs < 0) {
    set_error("pair_list_step before pair_list_upload");
    return EDM_HIP_ERR_STATE;
  }
  if (b->dim != 1) {
    set_error("pair_list_step: the pair-distance CV is 1-D (fix_edm_pair.cpp:52)");
    return EDM_HIP_ERR_ARG;
  }
  if (hill_step && !(b->hill_density < 0) && !b->device_rng) {
    set_error("pair_list_step: hill steps draw their uniforms on the device -- call set_device_rng first");
    return EDM_HIP_ERR_STATE;
  }
  const long long npairs = b->pl_npairs, nall = b->pl_nall;
  int rc = EDM_HIP_OK;
  if (hill_step) {
    rc = do_pre_add_hill(b, est_hill_count);   // overflow flush before any force (fix_edm_pair.cpp:174)
    if (rc) return rc;
  }
  if (b->b_outofbounds) {
    if (nall > 0) EDM_HIP_TRY(hipMemset(d_fdelta, 0, sizeof(double) * 3 * (size_t)nall));
    return hill_step ? do_post_add_hill(b) : EDM_HIP_OK;
  }
  hipStream_t s = b->bias->stream;
  PairListArgs a;
  memset(&a, 0, sizeof(a));
  a.npairs = npairs;
  a.pair_i = b->pl_i.p;
  a.pair_j = b->pl_j.p;
  a.type = b->pl_type.p;
  a.itype = itype;
  a.jtype = jtype;
  a.nlocal = nlocal;
  a.nall = (int)nall;
  a.x = d_x;
  a.it_off = b->pl_it_off.p;
  a.jt_off = b->pl_jt_off.p;
  a.it_partner = b->pl_it_idx.p;
  a.jt_partner = b->pl_jt_idx.p;
  a.it_entry = b->pl_it_entry.p;
  a.jt_entry = b->pl_jt_entry.p;
  a.fdelta = d_fdelta;
  if (hill_step && npairs > 0 && !(b->pl_mask_valid && b->pl_mask_nlocal == nlocal && b->pl_mask_itype == itype &&
                                   b->pl_mask_jtype == jtype)) {
    // live virtual samples of this list (fix_edm_pair.cpp:230-237): once per uploaded list
    EDM_HIP_TRY(b->vs_mask.reserve((size_t)2 * npairs));
    a.vs_mask = b->vs_mask.p;
    int mb = 0;
    EDM_HIP_TRY(launch_pairlist_mask(a, b->bias->d_partials, s, &mb));
    EDM_HIP_TRY(hipStreamSynchronize(s));
    double c = 0;
    for (int k = 0; k < mb; k++) c += b->bias->h_partials[k];
    b->pl_calls = (long long)c;
    b->pl_mask_valid = true;
    b->pl_mask_nlocal = nlocal;
    b->pl_mask_itype = itype;
    b->pl_mask_jtype = jtype;
  }
  // the force pass is queued like the pair forces of edm_hip_bias_pair_step: in the launch of the step's selection
  // where there is one, else on its own ahead of anything of the step that writes the grid
  b->bias->wait_polled = false;
  b->pending = PendingForces();
  // reference order (edm_hip_bias_set "reference_order"): the hills go first, the force pass reads each entry's bias as
  // it stood when the reference's loop reached the entry (see edm_hip_bias_pair_step_ordered)
  const bool ordered = hill_step && npairs > 0 && b->reference_order;
  if (ordered) {
    rc = ordered_snapshot(b);
    if (rc) return rc;
  } else {
    b->pending.active = true;
    b->pending.list = true;
    b->pending.pl = a;
  }
  if (hill_step && npairs > 0) {
    // add_hill(r, u) for the two virtual samples of every list entry, in list order; dead ones are masked out.
    // The CV of an accepted sample is recomputed from the positions when its hill is prepared; only the paths
    // that read sample positions as a plain array (synchronous multi-GPU exchange, target heights) get one.
    const double *sample_r = nullptr;
    if (b->comm || b->b_targeting) {
      EDM_HIP_TRY(b->vs_r.reserve((size_t)2 * npairs));
      a.vs_r = b->vs_r.p;
      EDM_HIP_TRY(launch_pairlist_samples(a, s));
      sample_r = b->vs_r.p;
    } else {
      b->pl_view_x = d_x;
    }
    const int *saved_mask = b->d_mask;
    b->d_mask = b->vs_mask.p;
    b->ord_early = edm_hip_bias::OrderedEarly();
    if (ordered) {   // (records and force pass go out from inside the hill batch's apply, see pair_step_ordered_device)
      b->ord_early.list = true;
      b->ord_early.pl = a;
      b->ord_early.tag = forces_poll_enabled() ? ++b->bias->force_seq : 0;
      b->ord_early.armed = true;
    }
    rc = process_new_hills(b, 2 * npairs, sample_r, 1, nullptr, 1);
    b->ord_early.armed = false;
    b->ord_step_active = false;
    b->ord_snap_pending = false;
    b->d_mask = saved_mask;
    b->pl_view_x = nullptr;
  }
  // (a step without hills: the force pass goes alone, its workgroups tag their partial energy sums and the host looks at
  //  the slots instead of waiting for the stream, see edm_hip_gauss_pair_forces)
  unsigned long long tag = 0;
  int nblk_ordered = 0;
  if (ordered) {
    if (rc) {
      if (b->ord_on_own_stream) (void)hipStreamSynchronize(b->ord_stream);
      return rc;
    }
    tag = b->ord_early.tag;   // (a step redone after an exceeded launch bound has taken a fresh one)
    a.partial_tag = tag;
    if (b->ord_early.done) {
      int rcl = resolve_deferred_log(b);   // (while the force pass runs, see pair_step_ordered_device)
      if (rcl) return rcl;
    }
    if (b->ord_early.done) {
      if (b->ord_early.rc) return b->ord_early.rc;
      nblk_ordered = b->ord_early.nblk;
    } else if (b->last_batch.valid && b->last_batch.nh > 0) {
      b->ord_early.list = true;
      b->ord_early.pl = a;
      rc = ordered_forces_enqueue(b);
      if (rc) return rc;
      nblk_ordered = b->ord_early.nblk;
    } else {   // (no new hill: every entry sees the same bias)
      EDM_HIP_TRY(launch_pairlist_forces(b->bias->g, b->bias->rec, a, b->bias->d_partials, s, &nblk_ordered));
    }
  } else if (b->pending.active && forces_poll_enabled()) {   // (no hills this step, or the step's new hills were skipped)
    tag = ++b->bias->force_seq;
    b->pending.pl.partial_tag = tag;
  }
  int rcf = pending_forces_flush(b->bias, &b->pending);   // (no hill launch carried it: nothing has touched the grid)
  if (rc) return rc;
  if (rcf) return rcf;
  int nblk = ordered ? nblk_ordered : b->pending.nblk;
  int nblk_redo = -1;
  double e = 0;
  if (tag && poll_tagged_partials(b->bias, nblk, tag, &e)) {
    b->bias->polled_forces++;
  } else {
    // (a polled hill batch has shown the stream past the force pass queued ahead of it)
    if (ordered && b->ord_on_own_stream) EDM_HIP_TRY(hipStreamSynchronize(b->ord_stream));
    if (tag || !b->bias->wait_polled) EDM_HIP_TRY(hipStreamSynchronize(s));
    if (ordered && b->ord_on_own_stream && *reinterpret_cast<volatile int *>(b->h_ord_status) == 2) {
      int nb2 = 0;
      rc = ordered_step_redo_after_gate(b, &nb2);
      if (rc) return rc;
      EDM_HIP_TRY(hipStreamSynchronize(s));
      nblk_redo = nb2;
    }
    if (nblk_redo >= 0) nblk = nblk_redo;
    e = 0;
    for (int k = 0; k < nblk; k++) e += b->bias->h_partials[tag ? 2 * k : k];
  }
  if (energy) *energy = e;
  if (ncalls) *ncalls = hill_step ? b->pl_calls : 0;
  return hill_step ? do_post_add_hill(b) : EDM_HIP_OK;
}

int edm_hip_bias_pre_add_hill(edm_hip_bias *b, long long est_hill_count) {
  b->staged_x.clear();
  b->staged_u.clear();
  b->in_cycle = true;
  return do_pre_add_hill(b, est_hill_count);
}

int edm_hip_bias_add_hill(edm_hip_bias *b, const double *position, double runiform) {
  if (b->temp_hill_prefactor < 0 && !b->b_outofbounds) {
    set_error("Must call pre_add_hill before add_hill");
    return EDM_HIP_ERR_STATE;
  }
  for (unsigned int d = 0; d < b->dim; d++) b->staged_x.push_back(position[d]);
  b->staged_u.push_back(runiform);
  return EDM_HIP_OK;
}

int edm_hip_bias_post_add_hill(edm_hip_bias *b) {
  int rc = EDM_HIP_OK;
  const long long n = (long long)b->staged_u.size();
  if ((n > 0 || b->comm) && !b->b_outofbounds) {
    hipStream_t s = b->bias->stream;
    EDM_HIP_TRY(b->stage_x.reserve(b->staged_x.size()));
    EDM_HIP_TRY(b->stage_u.reserve(b->staged_u.size()));
    EDM_HIP_TRY(hipMemcpyAsync(b->stage_x.p, b->staged_x.data(), sizeof(double) * b->staged_x.size(), hipMemcpyHostToDevice, s));
    EDM_HIP_TRY(hipMemcpyAsync(b->stage_u.p, b->staged_u.data(), sizeof(double) * b->staged_u.size(), hipMemcpyHostToDevice, s));
    EDM_HIP_TRY(hipStreamSynchronize(s));
    rc = process_new_hills(b, n, b->stage_x.p, (int)b->dim, b->stage_u.p, -1);
    if (rc) return rc;
  }
  b->staged_x.clear();
  b->staged_u.clear();
  b->in_cycle = false;
  return do_post_add_hill(b);
}

// edm_bias.cpp:224-262
// With a communicator the grids and the histogram are replicated bit-identically on every rank: like the
// reference's multi_write (rank 0 opens the file, grid.h:549) only rank 0 writes; the other ranks just let
// their queued updates finish, so a file is never truncated under another rank's writer.
static bool writes_files(const edm_hip_bias *b) {
  (void)resolve_deferred_log(const_cast<edm_hip_bias *>(b));
  const_cast<edm_hip_bias *>(b)->hills.drain();   // (a caller that writes its files expects the HILLS log on disk as well)
  if (b->nranks <= 1 || b->rank == 0) return true;
  if (b->bias) (void)hipStreamSynchronize(b->bias->stream);
  return false;
}
int edm_hip_bias_write_bias(const edm_hip_bias *b, const char *filename, int serial_format) {
  if (!b->bias) return EDM_HIP_ERR_STATE;
  if (!writes_files(b)) return EDM_HIP_OK;
  return serial_format ? edm_hip_gauss_write(b->bias, filename) : edm_hip_gauss_multi_write(b->bias, filename, 0);
}
int edm_hip_bias_write_lammps_table(const edm_hip_bias *b, const char *filename, int serial_format) {
  if (!b->bias) return EDM_HIP_ERR_STATE;
  if (!writes_files(b)) return EDM_HIP_OK;
  return serial_format ? edm_hip_gauss_write(b->bias, filename) : edm_hip_gauss_multi_write(b->bias, filename, 1);
}
int edm_hip_bias_write_histogram(const edm_hip_bias *b, int serial_format) {
  if (!b->hist) return EDM_HIP_ERR_STATE;
  if (!writes_files(b)) return EDM_HIP_OK;
  // (the histogram is updated by kernels on the bias stream, which may still be running behind a polled batch)
  if (b->bias) EDM_HIP_TRY(hipStreamSynchronize(b->bias->stream));
  if (serial_format) return edm_hip_grid_write(b->hist, b->hist_output.c_str());
  return edm_hip_grid_multi_write(b->hist, b->hist_output.c_str(), b->min.data(), b->max.data(), b->bper.data(), 0);
}
int edm_hip_bias_clear_histogram(edm_hip_bias *b) {
  if (!b->hist) return EDM_HIP_ERR_STATE;
  // (the histogram is updated by kernels on the bias stream, which may still be running behind a polled batch)
  if (b->bias) EDM_HIP_TRY(hipStreamSynchronize(b->bias->stream));
  return edm_hip_grid_clear(b->hist);
}
edm_hip_gauss *edm_hip_bias_gauss(edm_hip_bias *b) { return b->bias; }
edm_hip_grid *edm_hip_bias_histogram(edm_hip_bias *b) {
  // the handle has a stream of its own: hand it out only once the updates queued on the bias stream are done
  if (b->bias && b->hist) (void)hipStreamSynchronize(b->bias->stream);
  return b->hist;
}

int edm_hip_bias_get(const edm_hip_bias *b, const char *name, double *value) {
#define G(n, expr) if (strcmp(name, n) == 0) { *value = (double)(expr); return EDM_HIP_OK; }
  G("dim", b->dim)
  G("b_tempering", b->b_tempering)
  G("b_targeting", b->b_targeting)
  G("global_tempering", b->global_tempering)
  G("bias_factor", b->bias_factor)
  G("boltzmann_factor", b->boltzmann_factor)
  G("temperature", b->temperature)
  G("hill_prefactor", b->hill_prefactor)
  G("bias_per_step", b->bias_per_step)
  G("hill_density", b->hill_density)
  G("cum_bias", b->cum_bias)
  G("total_volume", b->total_volume)
  G("expected_target", b->expected_target)
  G("b_outofbounds", b->b_outofbounds)
  G("overflow_left", b->overflow_left)
  G("overflow_right", b->overflow_right)
  G("b_skip_hill_add", b->b_skip_hill_add)
  G("hills_added", b->hills_added)
  G("steps", b->steps)
  G("mpi_rank", b->mpi_rank)
  G("mpi_size", b->mpi_size)
  // telemetry of the polled completion: batches released by the polled word / by the stream wait
  G("polled_batches", b->bias ? b->bias->polled_batches : 0)
  G("poll_fallbacks", b->bias ? b->bias->poll_fallbacks : 0)
  G("header_releases", b->bias ? b->bias->header_releases : 0)
  G("polled_forces", b->bias ? b->bias->polled_forces : 0)
  G("lookup_prep_launches", b->bias ? b->bias->lookup_prep_launches : 0)
  G("bound_redos", b->bound_redos)
  G("reference_order", b->reference_order)
  G("ord_gate_giveups", b->ord_gate_giveups)
  G("host_add_threads", b->host_add_threads)
#undef G
  set_error(std::string("unknown EDMBias member ") + name);
  return EDM_HIP_ERR_ARG;
}

int edm_hip_bias_set(edm_hip_bias *b, const char *name, double value) {
#define S(n, lhs, type) if (strcmp(name, n) == 0) { lhs = (type)value; return EDM_HIP_OK; }
  S("b_tempering", b->b_tempering, int)
  S("global_tempering", b->global_tempering, double)
  S("bias_factor", b->bias_factor, double)
  S("hill_prefactor", b->hill_prefactor, double)
  S("bias_per_step", b->bias_per_step, double)
  S("hill_density", b->hill_density, double)
  S("cum_bias", b->cum_bias, double)
  S("total_volume", b->total_volume, double)
  S("debug_virtual_ranks", b->debug_virtual_ranks, int)
  S("debug_force_sync", b->debug_force_sync, int)
  S("reference_order", b->reference_order, int)
  if (strcmp(name, "host_add_threads") == 0) {
    if (value < 1 || value > 64) return EDM_HIP_ERR_ARG;
    b->host_add_threads = (int)value;
    return EDM_HIP_OK;
  }
  if (strcmp(name, "debug_tiles_first") == 0 && b->bias) { b->bias->debug_tiles_first = (int)value; return EDM_HIP_OK; }
#undef S
  set_error(std::string("unknown or read-only EDMBias member ") + name);
  return EDM_HIP_ERR_ARG;
}

int edm_hip_bias_get_array(const edm_hip_bias *b, const char *name, double *out) {
  const std::vector<double> *v = nullptr;
  if (strcmp(name, "bias_dx") == 0) v = &b->bias_dx;
  if (strcmp(name, "bias_sigma") == 0) v = &b->bias_sigma;
  if (strcmp(name, "min") == 0) v = &b->min;
  if (strcmp(name, "max") == 0) v = &b->max;
  if (!v) return EDM_HIP_ERR_ARG;
  for (size_t i = 0; i < v->size(); i++) out[i] = (*v)[i];
  return EDM_HIP_OK;
}

int edm_hip_bias_wait(edm_hip_bias *b) {
  if (b && b->bias) EDM_HIP_TRY(hipStreamSynchronize(b->bias->stream));
  if (b && b->copy_stream) EDM_HIP_TRY(hipStreamSynchronize(b->copy_stream));
  if (b && b->ord_stream) EDM_HIP_TRY(hipStreamSynchronize(b->ord_stream));
  return EDM_HIP_OK;
}

int edm_hip_bias_set_device_rng(edm_hip_bias *b, int enabled, unsigned long long seed) {
  b->device_rng = enabled != 0;
  b->rng_seed = seed;
  b->rng_cycle = 0;
  return EDM_HIP_OK;
}

int edm_hip_bias_set_hill_log(edm_hip_bias *b, int enabled) {
  if (!enabled) {   // (what was logged so far reaches the file now)
    (void)resolve_deferred_log(b);
    b->hills.submit(b->hill_events);
    b->hills.drain();
  }
  b->hill_log = enabled;
  return EDM_HIP_OK;
}

// ---- multi-GPU ------------------------------------------------------------------
int edm_hip_comm_unique_id(void *id_bytes, size_t cap) {
  if (cap < sizeof(ncclUniqueId)) return EDM_HIP_ERR_ARG;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) {
    set_error("ncclGetUniqueId failed");
    return EDM_HIP_ERR_COMM;
  }
  memcpy(id_bytes, &id, sizeof(id));
  return EDM_HIP_OK;
}

int edm_hip_bias_comm_init(edm_hip_bias *b, const void *id_bytes, int nranks, int rank) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return EDM_HIP_ERR_ARG;
  b->nranks = nranks;
  b->rank = rank;
  b->mpi_size = nranks;
  if (rank != b->mpi_rank) {
    // the HILLS log is per rank (<name>_<rank>, edm_bias.cpp:1104-1107)
    // (the handle was created before its rank was known and opened <name>_0 -- which IS rank 0's log: close our
    //  descriptor, never remove the file)
    (void)resolve_deferred_log(b);
    b->hills.submit(b->hill_events);
    b->hills.close();
    b->mpi_rank = rank;
    open_hills(b);
  }
  if (!id_bytes) return EDM_HIP_OK;  // rank bookkeeping only, no communicator
  delete b->comm;
  b->comm = nullptr;
  return make_rccl_transport(id_bytes, nranks, rank, &b->comm);
}

// the same exchange with the payloads staged through POSIX shared memory (edm_comm.h): ranks of one host
// without RCCL peer access, e.g. several ranks on ONE GPU (tests), or debugging
int edm_hip_bias_comm_init_shm(edm_hip_bias *b, const char *shm_name, int nranks, int rank) {
  int rc = edm_hip_bias_comm_init(b, nullptr, nranks, rank);
  if (rc) return rc;
  delete b->comm;
  b->comm = nullptr;
  return make_shm_transport(shm_name, nranks, rank, &b->comm);
}

int edm_hip_bias_comm_destroy(edm_hip_bias *b) {
  delete b->comm;
  b->comm = nullptr;
  return EDM_HIP_OK;
}

}  // extern "C"
