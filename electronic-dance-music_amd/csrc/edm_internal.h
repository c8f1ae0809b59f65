// edm_internal.h -- host-side objects behind the opaque handles of include/edm_hip.h.
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "../../include/edm_hip.h"
#include "edm_common.h"
#include "edm_kernels.h"

namespace edm {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);

#define EDM_HIP_TRY(expr)                                   \
  do {                                                      \
    hipError_t e__ = (expr);                                \
    if (e__ != hipSuccess) return edm::hip_fail(e__, #expr); \
  } while (0)

// grows a device allocation (contents are NOT preserved)
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      if (e != hipSuccess) return e;
      p = nullptr;
      cap = 0;
    }
    size_t want = n + n / 4 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  // as reserve(); a new allocation is zero-filled (contents are kept zero by its users)
  hipError_t reserve_zeroed(size_t n) {
    if (n <= cap) return hipSuccess;
    hipError_t e = reserve(n);
    if (e != hipSuccess) return e;
    e = hipMemset(p, 0, cap * sizeof(T));
    if (e != hipSuccess) return e;
    return hipDeviceSynchronize();  // (rare: only when the buffer grows)
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// workspace of the hill path, shared by add_values / the controller
struct HillWorkspace {
  DevBuf<double> hx, hx0, ht, added, partial, scratch, tail_h1, tail_h2, tail_a2, tail_cum, heights, slots, delta, gath;
  DevBuf<int> hc, tail_flags, tile_flags, tile_list;
  int tile_parity = 0;      // which of tile_list's two counters the next culled gather uses
  DevBuf<char> result;      // LimitResult
  DevBuf<char> rb;          // packed read-back region of small batches (one D2H instead of six)
  DevBuf<double> tagged;    // LimitArgs::tagged: {integral, batch number} per hill (zero-filled when allocated)
  unsigned long long tag_seq = 0;
  void release();
};

// derived DimmedGrid geometry from the constructor arguments (grid.h:190-213)
void make_geometry(Geom &g, int dim, const double *min, const double *max, const double *spacing,
                   const int *periodic, int has_deriv, int interpolate);
// geometry from PLUMED-file header fields (grid.h:799-806)
void finish_read_geometry(Geom &g);
void fill_public_geometry(const Geom &g, edm_hip_geometry *out);

// host text writers (byte-identical to grid.h:448-503 / :509-674)
int write_plumed(const Geom &g, const double *values, const double *derivs, const char *filename);

// parsed PLUMED grid file (grid.h:712-835)
struct GridFile {
  Geom g;
  std::vector<double> values, derivs;
};
int read_plumed(int dim, const char *filename, int b_interpolate, GridFile &out);

}  // namespace edm

struct edm_hip_grid {
  edm::Geom g;
  // device: g.total doubles, or -- when the grid stores derivatives (g.has_deriv) -- g.total node records of
  // g.rec doubles (V, dV/ds_0 .., pad), the layout of the gaussian grid's records
  double *values = nullptr;
  hipStream_t stream = nullptr;
  double *scratch = nullptr;  // lookup partial sums (record grids; allocated on demand)
};

struct edm_hip_gauss {
  edm::Geom g;
  double *rec = nullptr;                 // device node records
  double *tab[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};  // denom, dderiv per dim
  long long lookup_prep_launches = 0;    // fix edm steps whose force kernel shared a launch with the flush's preparation
  int *ball = nullptr;                   // 2-D / 3-D: Tables::ball (rebuilt when spacing, sigma or stencil half-widths change)
  int nball = 0;
  double ball_key[10] = {0};             // (dim, dx, sigma, msize the list was built for)
  double *node_tab = nullptr;            // 1-D grid with walls: Tables::node1d (rebuilt with the boundary / the geometry)
  hipStream_t stream = nullptr;
  double *scratch = nullptr;             // lookup partial sums
  double *d_scalars = nullptr;           // small device result slots
  double *h_scalars = nullptr;           // pinned host mirror
  double *h_partials = nullptr;          // host-mapped pinned block partial sums (energy)
  double *d_partials = nullptr;          // device view of h_partials
  char *h_stage = nullptr;               // pinned (host-mapped) staging for batched result read-back
  char *d_stage = nullptr;               // its device-side address
  size_t h_stage_bytes = 0;
  unsigned long long done_seq = 0;       // sequence number of the last polled read-back (see PostSpec::done_flag)
  // a batch released by its header line whose read-back region the host has not read yet (deferred_fetch): the
  // sequence number its completion word will carry and the region's size; saved to rb_saved before the region is reused
  unsigned long long rb_pending_seq = 0;
  size_t rb_pending_bytes = 0;
  std::vector<char> rb_saved;
  long long polled_batches = 0;          // hill batches whose completion was seen through the polled word ...
  long long poll_fallbacks = 0;          // ... and batches whose poll ran out (the stream wait took over)
  long long header_releases = 0;         // polled batches released by their header line alone (see LimitResult)
  unsigned long long force_seq = 0;      // tag of the last forces-only launch whose partial sums the host polled for
  long long polled_forces = 0;           // forces-only calls that returned on their workgroups' tagged sums (telemetry)
  bool wait_polled = false;              // the last apply_hills saw its results through the polled words: the
                                         // stream was NOT synchronised (its last kernel may still be retiring)
  bool shared_device = false;            // another rank of the job runs on this GPU (host-staged carrier): see LimitArgs::shared_device
  int debug_tiles_first = -1;            // tests: LimitArgs::tiles_first_mode
  int *d_dirty = nullptr;
  long long tiles_per_hill = 0;          // cached tiles_per_hill_bound() of the current geometry / boundary
  int *d_tickets = nullptr;              // three last-workgroup tickets (EDM_TICKET_INTS ints each), kept zero
  unsigned long long *d_ready = nullptr; // word the gather half of k_integrals_gather polls for ...
  unsigned long long ready_seq = 0;      // ... the sequence number of its launch
  double ht_ref_us = 0;                  // development aid (EDM_HIP_TRACE): host clock at the entry of the step being traced
  double ht_marks[12] = {0};             // ... and at marked places of that step (ht_mark), printed by the step's entry point
  // lookup replica of a 2-D / 3-D grid with a periodic boundary (see lookup_one / launch_build_faces): g.total
  // blocks of 128 bytes, 4x the node records -- memory spent so that a coordinate-CV sample reads 1 or 2 aligned
  // lines instead of 2.5 / 5.  faces_mode: -1 = automatic (grids beyond the L2s' reach), 0 = off, 1 = always.
  double *faces = nullptr;
  int faces_mode = -1;
  int faces_state = 0;                   // 0 = not built, 1 = current, 2 = stale (a path other than the in-place gather wrote the records)
  bool faces_unavailable = false;        // automatic mode: the allocation did not fit, stay on the node records
  long long faces_builds = 0;            // full rebuilds so far (telemetry / tests)
  // bench support: HIP events around the dominant lookup kernel
  // (a ring of event pairs: the launches of a timed loop are stamped without any host-side read in between;
  //  the elapsed times are summed up by profile_read, outside the loop)
  static const int PROF_RING = 1024;
  hipEvent_t *prof_ev = nullptr;         // 2 * PROF_RING events, created on first enable
  int prof_pending = 0;                  // stamped launches not yet summed (<= PROF_RING; later ones go untimed)
  int profiling = 0;                     // 0 = off, N > 0 = stamp every N-th launch of the lookup kernel
  long long prof_seen = 0;               // lookup launches since profiling was enabled
  double prof_ms = 0;
  long long prof_launches = 0;
  edm::HillWorkspace ws;
  edm::Tables tables() const;
};

// internal entry points shared between edm_gauss and edm_bias
namespace edm {
#define EDM_APPLY_BOUND_EXCEEDED 1000
// pair forces of a fused step (edm_hip_bias_pair_step) not launched yet: they ride in the launch of the step's
// selection when that is possible (launch_pair_forces_select), else they are queued on their own BEFORE anything
// of the step that writes the bias grid
struct PendingForces {
  bool active = false;
  long long n = 0;
  const double *d_r = nullptr;
  double *d_force = nullptr;
  int nblk = 0;              // K1 workgroups launched (partial energy sums to add up)
  unsigned long long tag = 0;   // != 0: when the forces are launched on their own, with tagged partial sums (launch_pair_forces)
  bool tagged = false;       // ... and the launch honoured it (the specialised 1-D kernels do)
  // ... or the force pass over a device-resident neighbour list (fix edm_pair gpu_list), queued the same way
  bool list = false;
  PairListArgs pl;
  // ... or K2, the coordinate-CV force kernel of a fix edm step (edm_hip_bias_step): it can share a launch with the
  // preparation of the overflow flush's hill list (launch_lookup_prep); `la.partial_tag` as for `tag` above
  bool lookup = false;
  LookupArgs la;
  // called once, right behind the launch that carries the forces (whichever launch that turns out to be): a caller that
  // wants the forces elsewhere while the rest of the step runs hangs its copies on it (edm_hip_bias_step_host)
  void (*on_launched)(void *) = nullptr;
  void *on_launched_ctx = nullptr;
  void launched() {
    if (on_launched) on_launched(on_launched_ctx);
    on_launched = nullptr;
  }
};
void ht_mark(edm_hip_gauss *g, int slot);   // development aid, edm_gauss.cpp
// EDM_HIP_TEST_FORCE tokens (tests only; edm_gauss.cpp)
bool test_force(const char *token);
long long test_force_value(const char *key);
// event pair for the next stamped launch of the handle's dominant lookup kernel (nullptrs when profiling is off)
void profile_slot(const edm_hip_gauss *g, hipEvent_t *e0, hipEvent_t *e1);

struct ApplySpec {
  long long nh = 0;
  const double *d_x = nullptr;     // sample positions
  int x_stride = 1;
  // ... or (d_x == NULL, 1-D): the virtual samples of a device-resident neighbour list, see HillList::pl_x
  const double *pl_x = nullptr;
  const int *pl_i = nullptr, *pl_j = nullptr;
  const long long *d_sel = nullptr;
  const double *d_h = nullptr;     // per-hill heights or NULL
  double h_const = 0;
  bool limited = false;            // run the limiter
  int flush_mode = 0;
  double limit = 0, cum_in = 0;
  // CV histogram to update on the device (output_hill, edm_bias.cpp:601-610); NULL = none
  const Geom *hist_g = nullptr;
  double *hist_values = nullptr;
  bool fetch_all = false;          // host wants position + bias_added of EVERY hill (HILLS log)
  // ... but can take them LATER (apply_hills_fetch_deferred): a batch every hill of which was added in full is then
  // released by its header line as if nothing had been asked for, and the log is written behind the step
  bool defer_fetch_ok = false;
  // a reference-order fix edm_pair step: where the hill batch's launch may store the hills' unit-height stencil terms
  // (LimitArgs::ord_terms; room for `nh` rows) -- only the launch that fuses integrals and gather does
  double *ord_terms = nullptr;
  unsigned *ord_dirty = nullptr;
  unsigned ord_seq = 0;
  // ... and a call made once every launch of the batch is queued, BEFORE the host waits for the limiter's result: what
  // depends on the batch only through device memory (the reference-order force pass) is queued here, behind the batch,
  // instead of after the host's turn-around
  // (ready_flag / ready_seq: the word the batch's limiter publishes for the workgroups that wait for it, LimitArgs; NULL
  //  when the batch did not go through the launch that has one.  d_nh: the batch's hill count on the device, or NULL)
  void (*before_wait)(void *ctx, const double *d_base_heights, const double *d_tail_h1, const double *d_tail_h2,
                      const LimitResult *d_res, bool terms_emitted, const unsigned long long *ready_flag,
                      unsigned long long ready_seq, const long long *d_nh) = nullptr;
  void *before_wait_ctx = nullptr;
  // a record pass that will run BESIDE the batch's launch (LimitArgs::ord_ready)
  unsigned *ord_ready = nullptr;
  bool fetch_heights = true;       // with d_h: copy the per-hill base heights back (a flush already has them)
  // optional: d_h is filled by the preparation kernel from this host-mapped array (nh doubles)
  const double *h_fetch_src = nullptr;
  // deferred count: the batch is queued with `nh` as a launch bound while the true count still sits in
  // device memory; apply_hills returns EDM_APPLY_BOUND_EXCEEDED (nothing applied) if the bound was too small
  const long long *d_nh = nullptr;
  double expected_nh = -1;  // expected batch size when it is a random variable (stochastic selection); < 0: nh
  // with a deferred count: selection chained in front of the hill preparation (one launch for both)
  const SelectArgs *sel_chain = nullptr;
  PendingForces *forces = nullptr;   // launched together with sel_chain where possible
  // multi-GPU packed exchange: the hill list is unpacked from the gathered packets (replaces preparation)
  const UnpackArgs *unpack_chain = nullptr;
  // sharded application of a dense batch on a replicated grid (multi-GPU): this rank gathers only its own
  // slice [shard_off, shard_off + shard_cnt) of the global hill list into a delta grid; the per-hill
  // integrals and the delta grid are summed over the ranks (ncclAllReduce) and every rank adds the same
  // total.  shard_virtual > 1 (tests): the slices of that many ranks are processed one after the other
  // in this process, with plain adds in place of the collectives.
  void *shard_comm = nullptr;  // edm::Transport *
  const long long *shard_counts = nullptr;   // hills per rank of the global list (nranks entries), with shard_comm
  long long shard_off = 0, shard_cnt = 0;
  int shard_virtual = 0;
  // heights that depend on the bias under construction (local tempering): strictly ordered kernel
  bool ordered = false;
  OrderedParams op;
};
struct ApplyOutcome {
  LimitResult res;
  double total_added = 0;          // unlimited mode: sum of added
  // ordered tail (res.n_tail entries): limiter flags, undo height, undo bias_added
  std::vector<int> flags;
  std::vector<double> h2, a2;
  // original positions (and bias_added when fetch_all) of hills [first, nh)
  long long first = 0;
  std::vector<double> pos, added;
  std::vector<double> heights;     // per-hill base heights of [first, nh) when spec.d_h was given
  // the batch was released by its header line alone (see LimitResult): every hill was added in full, nothing deferred,
  // nobody asked for positions / per-hill bias -- flags, h2, a2, pos, added above are EMPTY (they would read 1, 0, 0)
  bool plain_fast = false;
  bool terms_emitted = false;        // spec.ord_terms was filled by the batch's launch
  bool deferred_fetch = false;       // plain_fast with fetch_all: positions / per-hill bias wait in the read-back region
  long long deferred_bound = 0;      // ... laid out for this launch bound (see apply_hills_fetch_deferred)
  const double *d_added = nullptr;   // where the batch's per-hill bias_added lies on the device (valid until the next batch)
  // ... and the heights the batch was applied with (see HillHeights): per-hill base heights (NULL: the constant),
  // the limiter's tail arrays for hills >= res.k
  const double *d_base_heights = nullptr, *d_tail_h1 = nullptr, *d_tail_h2 = nullptr;
};
// prep -> integrals -> (limiter) -> ordered gather -> boundary duplication.
// Leaves per-hill `added` in g->ws.added and the tail arrays in g->ws.tail_*.
int apply_hills(edm_hip_gauss *g, const ApplySpec &spec, ApplyOutcome *out, bool want_total);
// original positions [nh][dim] and per-hill bias [nh] of the last batch released with deferred_fetch (waits for the
// batch's completion word if it has not arrived yet; apply_hills itself saves the region before it reuses it)
int apply_hills_fetch_deferred(edm_hip_gauss *g, long long nh_bound, long long nh, std::vector<double> &pos,
                               std::vector<double> &added);
// the lookup replica, built or rebuilt if it is wanted and not current; *faces = NULL when the grid does not use one
int faces_prepare(edm_hip_gauss *g, const double **faces);
inline void faces_touch(edm_hip_gauss *g) {   // the node records were written by a path that does not maintain the replica
  if (g->faces_state == 1) g->faces_state = 2;
}
int pair_forces_enqueue(const edm_hip_gauss *g, long long n, const double *d_r, double *d_force, int *nblk);
int update_forces_enqueue(const edm_hip_gauss *g, long long n, const double *d_x, int x_stride, double *d_f, int f_stride,
                          const int *d_mask, int apply_mask, int *nblk, unsigned long long tag = 0);
double pair_forces_finish(const edm_hip_gauss *g, int nblk);
// queues pending forces on their own (no-op when none are pending)
bool forces_poll_enabled();
bool poll_tagged_partials(const edm_hip_gauss *g, int nblk, unsigned long long tag, double *energy);
int pending_forces_flush(const edm_hip_gauss *g, PendingForces *pf);
// selection (+ preparation / packing) of a step, in one launch with the pending forces where possible
int select_prep_enqueue(const edm_hip_gauss *g, const SelectArgs &a, const HillList &h, PendingForces *pf);
}  // namespace edm
