// edm_kernels.h -- host-callable launchers of the gfx950 kernels (edm_kernels.hip).
// Internal to libedm_hip.so; the public boundary is include/edm_hip.h.
#pragma once

#include <hip/hip_runtime_api.h>
#include <string.h>

#include "edm_common.h"

namespace edm {

// McGovern-De Pablo tables in HBM: [dim][65536] doubles each (NULL rows for
// periodic boundary dimensions).
struct Tables {
  const double *denom[3];
  const double *dderiv[3];
  // 1-D grid with walls, optional: per node (t2, t4, 1 / bc_denom, inside ? 1 : 0) -- the node-only factors of a
  // stencil term's VALUE (gaussian_grid.h:311,:313,:318), computed once by launch_build_node_table with the very
  // code the kernels would run per stencil point (node_terms), so a walk that reads them yields the same bits
  const double *node1d;
  // 2-D / 3-D, optional: the stencil offsets that can lie inside a hill's support (dp2 < 8, gaussian_grid.h:299)
  // whatever the hill's position within its cell -- offsets o with sum_d (max(0, |o_d| - 1) dx_d / sigma_d)^2 <= 8,
  // |o_d| <= msize_d, packed (o_0 + 128) | (o_1 + 128) << 8 | (o_2 + 128) << 16 and ordered by that sum (inner
  // first): the per-hill integral walks this list instead of the (2 msize + 1)^DIM box (built by the host, edm_gauss.cpp)
  const int *ball;
  int nball;
};
// out: 4 doubles per node of the 1-D grid (t.node1d is not read)
hipError_t launch_build_node_table(const Geom &g, const Tables &t, double *out, hipStream_t s);

// ---- lookup path (K1/K2/K8) ------------------------------------------------
enum LookupMode {
  LOOKUP_FORCES = 0,   // f[i][d] -= dV/ds_d, masked, energy sum       (edm_bias.cpp:276-295)
  LOOKUP_VALUES = 1,   // energy[i] = V, deriv[i][d] = dV/ds_d          (gaussian_grid.h:118-138)
  LOOKUP_INDEX = 2     // flat[i] = start node or -1                     (grid.h:264-273)
};

struct LookupArgs {
  long long n;
  const double *x;
  int x_stride;
  double *f;          // FORCES: forces; VALUES: deriv [n][dim]
  int f_stride;
  double *energy;     // VALUES: per-sample energy (may be NULL)
  long long *flat;    // INDEX
  const int *mask;
  int apply_mask;
  // FORCES, optional: every workgroup stores {partial energy sum, partial_tag} as one 16-byte system-scope store at
  // scratch[2 * workgroup] (host-mapped memory the host polls, see launch_pair_forces); 0 = plain partial sums
  unsigned long long partial_tag;
};

// energy_out: device double receiving the (deterministic, fixed-order) sum of V;
// scratch: device doubles, at least lookup_scratch_doubles() of them.
size_t lookup_scratch_doubles();
// ev0/ev1 (may be NULL) are recorded on `s` directly around the main kernel, so a caller can
// time exactly that launch with hipEventElapsedTime.
// blocks_out (may be NULL) receives the number of per-block partial sums written to `scratch`;
// a caller that passes energy_out == NULL can sum them itself in index order.
// faces (may be NULL): the lookup replica of a 2-D / 3-D grid (launch_build_faces) -- interpolating lookups then
// read their corner records from it (1 or 2 aligned 128-byte lines per sample) with bit-identical results
hipError_t launch_lookup(const Geom &g, const double *rec, LookupMode mode, const LookupArgs &a,
                         double *scratch, double *energy_out, hipStream_t s,
                         hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int *blocks_out = nullptr,
                         const double *faces = nullptr);
// lookup replica of a 2-D / 3-D grid: g.total blocks of 16 doubles; block(i0, i1[, i2]) = the records of nodes
// (i0, i1), (i0+1, i1), (i0, i1+1), (i0+1, i1+1) [at i2], periodic wrap applied
hipError_t launch_build_faces(const Geom &g, const double *rec, double *faces, hipStream_t s);
// 1-D pair-distance form: force[i] = -dV/dr(r_i)
// fix edm_pair on a device-resident neighbour list (fix_edm_pair.cpp:177-238 over flattened pair records).
// ONE pass, 16 lanes per owned atom: the atom's bias force = sum over its entries as i of del * f_r minus the sum
// over its entries as j (newton off: ghost atoms receive nothing), each term computed on the spot from the two
// positions (pair distance, bias lookup) -- nothing per entry is written to memory, a pair is simply evaluated from
// both of its ends -- in a fixed order through the index lists built when the list was uploaded: no atomics,
// bit-reproducible forces.  Energy is summed on the i side (every entry has an owned i).
// The two virtual add_hill samples of an entry (fix_edm_pair.cpp:230-237: the second one live iff j is owned) need
// no per-step arrays either: which of them are live depends on the list, the types and nlocal only
// (launch_pairlist_mask, once per uploaded list), and the CV of an ACCEPTED sample is recomputed from the
// positions when its hill is prepared (HillList::pl_x).
struct PairListArgs {
  long long npairs;
  const int *pair_i, *pair_j;   // list entries in neighbour-list order (j already masked with NEIGHMASK)
  const int *type;              // [nall] atom types; itype/jtype select the pairs (fix_edm_pair.cpp:181-202)
  int itype, jtype;
  int nlocal, nall;
  const double *x;              // [nall][3]
  const long long *it_off, *jt_off;   // [nall + 1] CSR offsets of the entries with pair_i == a / pair_j == a
  const int *it_partner, *jt_partner; // [npairs] the OTHER atom of each of those entries (j of the entries with
                                      // pair_i == a, i of the entries with pair_j == a), each atom's run in list order
  const int *it_entry, *jt_entry;     // [npairs] the list entry behind each of those slots (read by the
                                      // reference-order pass only: launch_pairlist_forces_ordered)
  double *fdelta;               // out [nall][3]: bias force per atom (ghost atoms: zero)
  double *vs_r;                 // launch_pairlist_samples: out [2 * npairs] virtual-sample CVs
  int *vs_mask;                 // launch_pairlist_mask: out [2 * npairs] 1 = live sample
  unsigned long long partial_tag;   // launch_pairlist_forces, optional: tagged partial energy sums (see launch_pair_forces)
};
// partials[0 .. blocks): energy partial sums (launch_pairlist_forces) / live-sample counts (launch_pairlist_mask)
#define EDM_PAIRLIST_MAX_BLOCKS 1024
hipError_t launch_pairlist_forces(const Geom &g, const double *rec, const PairListArgs &a, double *partials,
                                  hipStream_t s, int *blocks_out);
// vs_mask[2 e + slot] for every entry, and the number of live samples (= add_hill calls of a hill step) as
// partial sums
hipError_t launch_pairlist_mask(const PairListArgs &a, double *partials, hipStream_t s, int *blocks_out);
// vs_r[2 e + slot] = r_e for every entry: only for the paths that read sample positions as a plain array (the
// synchronous multi-GPU exchange, target heights)
hipError_t launch_pairlist_samples(const PairListArgs &a, hipStream_t s);
// tag != 0 (honoured by the specialised 1-D kernels: *tagged_out = 1): every workgroup stores {partial sum, tag} as one
// 16-byte system-scope store at scratch[2 * workgroup] -- scratch then is host-mapped memory the host polls instead of
// waiting for the stream (energy_out must be NULL)
hipError_t launch_pair_forces(const Geom &g, const double *rec, long long n, const double *r,
                              double *force, double *scratch, double *energy_out, hipStream_t s,
                              hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int *blocks_out = nullptr,
                              unsigned long long tag = 0, int *tagged_out = nullptr);

// ---- fix edm_pair in the REFERENCE'S order (fix_edm_pair.cpp:177-238) ---------------------------------------------
// The reference walks the neighbour list once: update_force for pair k, then its one or two add_hill calls -- so
// pair k's force reads a bias that already holds the hills of pairs 0..k-1 of the same step.  Which samples become
// hills does not depend on the grid (edm_bias.cpp:543), and neither do the limiter's decisions (a hill's integrated
// bias is independent of grid contents), so the step's hill batch is applied exactly as in the batched step and the
// forces follow from three things the batch leaves on the device: a copy of the node records taken before the batch
// (rec0), the prepared hill list, and the heights the limiter settled on.
//   launch_ordered_records : per tile of 32 nodes, the tile's nodes' records after every hill that reaches the tile, in
//       hill order -- rec0 plus, hill after hill, h1 term and then h2 term: the reference's sequence of +=
//       (gaussian_grid.h:343-355) -- and, for every m, how many of the first m hills reached the tile.  A W1 step's
//       ~125 hills leave ~3 MB of records (a hill reaches ~36 of the C1D grid's 351 tiles).
//   launch_pair_forces_ordered : pair k counts the hills whose add_hill call precedes its update_force
//       (m = #{j : sample(j) < first_sample[k]}, a binary search over the ascending sample indices in LDS), takes for
//       each of its two corner nodes the record behind the last of those hills that reached the node's tile (one
//       16-bit count, one 16-byte record; rec0 where none did) and interpolates (grid.h:390-446, interp<1> :52-139).
//       The boundary duplication of gaussian_grid.h:571-630 (value of the first / last in-boundary node copied outward
//       after every hill with a non-zero correction) is applied on the fly: an outward copy node reads the value of
//       its source node at the same m once the first such hill (first_dirty, found by the record pass) lies before
//       the pair.
// 1-D grids only (fix_edm_pair.cpp:52), stencil not wider than a periodic grid, at most ordered_max_hills() hills.
struct LimitResult;
struct OrderedForcesArgs {
  long long nh;             // this rank's hills of the step's batch (true count; unused with range_dev)
  long long nh_cap;         // hills the record / count buffers were sized for (>= nh)
  // multi-GPU: the batch is the rank-major global list and this rank's hills are its slice [hill_off, hill_off + nh) --
  // the reference's ranks see their own hills while they walk their pairs, the other ranks' only from post_add_hill
  // on (edm_bias.cpp:565-583).  range_dev (device, {offset, count}) overrides hill_off / nh where the host does not
  // know them (the packed exchange); nh_cap then bounds the count.
  long long hill_off;
  const long long *range_dev;
  long long k;              // hills [0, k): base height; hills >= k: the limiter's tail arrays (see HillHeights)
  // the record pass BESIDE the hill batch's launch (another stream; see LimitArgs::ord_ready): it waits for the limiter's
  // word `wait_seq` at `wait_flag` (the split index and the error state come with it), counts the hills in *nh_dev (the
  // selection's count), waits for terms_ready[2 hill + part] == dirty_seq before it reads a hill's terms -- with
  // agent-scope loads -- and leaves the batch's error state in *status for the force pass behind it
  const unsigned long long *wait_flag;
  unsigned long long wait_seq;
  const long long *nh_dev;
  const unsigned *terms_ready;
  int *status;
  int *status_host;   // host-mapped: the gate wave ahead of the record pass writes 2 here (and to *status) when it gives up
  unsigned long long gate_ticks;   // ... after this many ticks of the 100 MHz wall clock (0: the default, 2 ms)
  const LimitResult *res_dev;   // when set (single rank, launches queued before the host has seen the limiter's result):
                                // nh and k are read from the limiter's device-side result, nh_cap bounds the count; a
                                // batch the limiter refused (error != 0: nothing applied, the step is redone) counts 0 hills
  const double *heights;    // per-hill base heights or NULL (h_const)
  double h_const;
  const double *tail_h1, *tail_h2;
  const double *hx;         // prepared hills: remapped position, centre node, (t1, t3)
  const int *hc;
  const double *ht;
  const long long *sel;     // sample index of this rank's hill j, ascending (NULL: hill j is sample j)
  const double *rec0;       // the node records before the batch
  const double *terms;      // optional: [nh_cap rows][2 msize + 1][2] unit-height terms of the batch's hills (LimitArgs::ord_terms;
                            // row = index in the batch's hill list); NULL: the record pass computes them
  long long terms_rows;     // rows `terms` has room for
  double *records;          // [tiles][nh_cap][32][2], see launch_ordered_records
  unsigned short *counts;   // [nh + 1][tiles]
  unsigned *dirty_hill;     // [rows of the batch's hill list] == dirty_seq where the hill met a non-zero boundary correction
                            // (ordered_dirty_note in edm_kernels.hip; zero-initialised once, never reset)
  unsigned dirty_seq;       // this step's number (> 0, grows from step to step)
  long long n;              // pairs
  const double *r;          // [n] pair distances
  const int *first_sample;  // [n] sample index of pair k's first add_hill call, or NULL: 2 k (the virtual samples of a
                            // device-resident neighbour list)
  double *force;            // [n] out: -dV/dr
  unsigned long long *trace;   // development aid (EDM_HIP_TRACE=ordered): 8 wall-clock stamps per workgroup of the record pass, or NULL
};
size_t ordered_record_doubles(const Geom &g, long long nh_cap);
size_t ordered_count_shorts(const Geom &g, long long nh_cap);
long long ordered_max_hills();
bool ordered_forces_supported(const Geom &g);
hipError_t launch_ordered_records(const Geom &g, const Tables &t, const OrderedForcesArgs &a, hipStream_t s);
// tagged partial energy sums like launch_pair_forces (tag != 0: scratch is host-mapped, polled by the host)
hipError_t launch_pair_forces_ordered(const Geom &g, const OrderedForcesArgs &a, double *scratch, hipStream_t s,
                                      int *blocks_out, unsigned long long tag, hipEvent_t ev0 = nullptr,
                                      hipEvent_t ev1 = nullptr);
// the same pass over a device-resident neighbour list (a.n / r / first_sample / force unused: list entry e is
// "pair" e, its first sample 2 e; pl.fdelta receives the per-atom sums, pl.partial_tag as in launch_pairlist_forces)
hipError_t launch_pairlist_forces_ordered(const Geom &g, const PairListArgs &pl, const OrderedForcesArgs &a, double *partials,
                                          hipStream_t s, int *blocks_out);

// ---- record layout conversion ---------------------------------------------------
hipError_t launch_pack(const Geom &g, double *rec, const double *values, const double *derivs, hipStream_t s);
hipError_t launch_unpack(const Geom &g, const double *rec, double *values, double *derivs, hipStream_t s);

// base[i * rec] = values[i] (derivative slots of a record grid kept)
hipError_t launch_set_values(const Geom &g, double *base, const double *values, hipStream_t s);

// DimmedGaussGrid::remap (gaussian_grid.h:504-541) batched: out rows of dim doubles
hipError_t launch_remap(const Geom &g, long long n, const double *x, int x_stride, double *out, hipStream_t s);

// ---- plain DimmedGrid lookups and Grid::add --------------------------------------
// get_value / get_value_deriv on a grid WITHOUT derivative records (grid.h:343-365): nearest-lower node value,
// 0 outside in_grid, derivative 0.  out_value / out_deriv [n][dim] may be NULL.
hipError_t launch_nearest_values(const Geom &g, const double *values, long long n, const double *x, int x_stride,
                                 double *out_value, double *out_deriv, hipStream_t s);
// Grid::add (grid.h:275-290) in two kernels around a lookup of `other`: node coordinates min + dx * index of
// nodes [first, first + count) (rows of dim doubles), then grid_[i] += scale * E + offset (and the derivative
// slots += scale * D where the grid has them) for the same nodes
hipError_t launch_node_coords(const Geom &g, long long first, long long count, double *out, hipStream_t s);
hipError_t launch_axpy_nodes(const Geom &g, double *base, long long first, long long count, const double *E,
                             const double *D, double scale, double offset, hipStream_t s);

// ---- plain-grid histogram add (K7) ------------------------------------------
hipError_t launch_hist_add(const Geom &g, double *values, long long n, const double *x, int x_stride,
                           const long long *sel, const double *w, double w_const, hipStream_t s);

// ---- hill path (K3/K4/K5/K6) -----------------------------------------------
// order-preserving selection of accepted samples (edm_bias.cpp:543 and the mask
// tests of :406): sel[j] = index of the j-th accepted sample, *count = how many.
size_t select_scratch_ints(long long n);
// count2 (may be NULL): a second place the count is written to (e.g. device memory for kernels that
// consume a deferred count while `count` is host-mapped for the CPU)
hipError_t launch_select(long long n, const double *runiform, double threshold, int use_threshold,
                         const int *mask, int apply_mask, long long *sel, long long *count,
                         int *scratch, hipStream_t s, long long *count2 = nullptr, unsigned long long rng = 0);

struct HillList {
  long long nh;
  const double *x;        // sample positions [.. ][x_stride]
  int x_stride;
  // alternative source of 1-D sample positions (x == NULL): sample s is the virtual add_hill sample of entry s >> 1
  // of a device-resident neighbour list, its CV the pair distance |pl_x[pl_i[e]] - pl_x[pl_j[e]]| (rows of 3)
  const int *pl_i = nullptr, *pl_j = nullptr;
  const double *pl_x = nullptr;
  const long long *sel;   // optional indirection into x (NULL = identity)
  // prepared per-hill records (device, capacity >= nh):
  double *hx;             // remapped position            [nh][dim]
  int *hc;                // centre node index (INT_MIN in hc[i*dim] = rejected) [nh][dim]
  double *ht;             // (t1, t3) of gaussian_grid.h:310,:312 per dim [nh][2*dim]
  double *hx0;            // original (un-remapped) position, compacted [nh][dim] (may be NULL)
  // deferred count: when set, the true number of hills is min(*nh_dev, nh) and `nh` is only the launch
  // bound -- lets a batch be queued before the selection count has travelled back to the host
  const long long *nh_dev;
};
hipError_t launch_hill_prep(const Geom &g, const HillList &h, hipStream_t s, const double *fetch_src = nullptr,
                            double *fetch_dst = nullptr);
// fix edm step on a 2-D / 3-D grid: K2 (forces, four lanes per atom) and the preparation of a hill list in one launch
bool lookup_prep_fusable(const Geom &g, const HillList &h);
// ... or, without a flush, K2 and the step's selection + preparation (k_select_prep's body) in one launch
struct SelectArgs;
hipError_t launch_lookup_select(const Geom &g, const double *rec, const LookupArgs &la, double *scratch, hipStream_t s,
                                hipEvent_t ev0, hipEvent_t ev1, int *blocks_out, const double *faces, const SelectArgs &a,
                                const HillList &h);
hipError_t launch_lookup_prep(const Geom &g, const double *rec, const LookupArgs &a, double *scratch, hipStream_t s,
                              hipEvent_t ev0, hipEvent_t ev1, int *blocks_out, const double *faces, const HillList &h,
                              const double *fetch_src, double *fetch_dst);

// --- chained launches for short hill steps (see last_block_done in edm_kernels.hip) ---
// a ticket = zero-initialised device ints: top counter + FAN sub-counters, one 128-byte line each
#define EDM_TICKET_FAN 16
#define EDM_TICKET_INTS (32 * (1 + EDM_TICKET_FAN))
// selection + preparation in one launch; h.nh is the launch bound of the step, h.sel == a.sel
struct SelectArgs {
  long long n;
  const double *ru;       // uniforms, or NULL: taken from the device stream `rng` (see device_uniform)
  unsigned long long rng;
  double thr;
  int use_thr;
  const int *mask;
  int apply_mask;
  int *counts;            // [blocks] scratch (select_scratch_ints covers it)
  int *stage;             // [select_stage_ints(n)] scratch
  long long *sel;         // out: ordered accepted sample indices (first min(count, h.nh))
  long long *count_host;  // out: count, host-mapped
  long long *count_dev;   // out: count, device memory
  int *ticket;            // zero-initialised device int, left at zero
  double *pack;           // multi-GPU: instead of sel/prep, write this rank's exchange packet
                          // [count, x_0 .. x_{bound-1}] (bound = h.nh, dim doubles per position)
  unsigned long long *trace;   // development aid (EDM_HIP_TRACE): 8 wall-clock stamps per workgroup of k_pair_forces_select, or NULL
  // reference-order pair step: the step-start copy of the 1-D grid's records rides in this launch (nothing has touched
  // the grid yet: the hills are applied by a later launch); snap_n 16-byte records, or snap_dst == NULL
  const double *snap_src;
  double *snap_dst;
  long long snap_n;
};
// receive side of the packed exchange (see k_unpack_prep)
#define EDM_MAX_RANKS 16
struct UnpackArgs {
  const double *recv;     // nranks packets of `packet` doubles
  int nranks;
  long long bound;        // per-rank capacity of a packet
  long long packet;       // 1 + bound * dim
  double *all;            // out: positions of the global list, stride dim
  long long *count_dev;   // out: global count (poisoned when a rank overflowed its packet)
  long long *count_host;
  int rank;               // this rank, and (optional) where its slice {offset, count} of the global list goes
  long long *local_range;
};
hipError_t launch_unpack_prep(const UnpackArgs &a, const Geom &g, const HillList &h, hipStream_t s);
struct RankHeights {
  int nranks;
  long long offset[EDM_MAX_RANKS + 1];
  double height[EDM_MAX_RANKS];
};
hipError_t launch_rank_heights(const RankHeights &rh, double *out, hipStream_t s);
size_t select_stage_ints(long long n);
hipError_t launch_select_prep(const SelectArgs &a, const Geom &g, const HillList &h, hipStream_t s);
// K1 and the selection (+ preparation) of a fix edm_pair hill step as ONE launch (short pair arrays only)
bool pair_forces_select_fusable(const Geom &g, long long n_pairs, long long n_samples);
hipError_t launch_pair_forces_select(const SelectArgs &a, const Geom &g, const HillList &h, const double *rec, long long n,
                                     const double *r, double *force, double *scratch, hipStream_t s, hipEvent_t ev0,
                                     hipEvent_t ev1, int *blocks_out);
// the force pass over a device-resident neighbour list and the selection (+ preparation) of the same step as ONE launch
hipError_t launch_pairlist_forces_select(const SelectArgs &a, const Geom &g, const HillList &h, const double *rec,
                                         const PairListArgs &pl, double *partials, hipStream_t s, int *blocks_out);
struct LimitArgs;
struct PostSpec;
struct LimitResult;
struct HillHeights {
  const double *h;        // per-hill base height or NULL
  double h_const;
  long long k;            // hills [0,k) use (base height, 0); hills >= k use the tail arrays
  const double *tail_h1;  // first add (0 = hill deferred, not applied)
  const double *tail_h2;  // second add (the "undo" hill) or 0
  const LimitResult *res_dev;  // when set, k (and the error flag) are read from the limiter's
                               // device-side result: no host round trip between K4 and K5
  long long tail_shift;   // the hill list is a slice of the list the limiter saw, starting at this global index
                          // (sharded multi-GPU application); 0 otherwise
};
// per-hill integrated bias for the BASE heights (the value add_value returns)
// `mark` (workgroup-per-hill launches of a 2-D / 3-D grid only): extra workgroups of the same launch list the tiles the
// hills touch for the culled gather that follows (k_mark_tiles' body: it needs the prepared hills, nothing of the
// integrals) -- one launch and its ~6-9 us of dependent atomics less per hill batch
long long mark_tiles_threads(const Geom &g, long long nh);   // threads k_mark_tiles' body needs for nh hills
struct MarkArgs {
  int *flags;        // GatherPlan::tile_flags
  int *list;         // GatherPlan::tile_list
  long long ntiles;
  int parity;
};
hipError_t launch_hill_integrals(const Geom &g, const Tables &t, const HillList &h, const double *heights,
                                 double h_const, double *added, hipStream_t s, const LimitArgs *chain = nullptr,
                                 const MarkArgs *mark = nullptr);

struct GatherPlan {
  int groups;             // hill groups (partial buffers) -- 1 = accumulate in place
  int adaptive;           // groups is an upper bound: the device uses min(groups, max(1, count / 128)), so a
                          // batch queued with a deferred count is split exactly as if the count had been known
  double *partial;        // [groups (+1)][total][rec] when groups > 1 or in fused mode
  int *tile_flags;        // [ntiles] scratch when culling, else NULL
  int *tile_list;         // [ntiles + 2]: list, then two counters used alternately (each k_mark_tiles launch
                          // zeroes the counter of the NEXT batch, so no memset precedes it); kept zero in between
  int tile_parity;        // which counter this batch uses
  long long tile_bound;   // launch bound for culled gathers
  int compact_waves;      // 2-D / 3-D: a wave owns a compact block of its tile (hill_gather_body), not 64 consecutive nodes
  int tiles_marked;       // the tile list of this batch was built by an earlier launch (launch_hill_integrals with `mark`)
  // fused mode (dense batches on small grids): the gather runs BEFORE the limiter with the base
  // heights, writes per-group deltas and, as a by-product, the per-(hill, tile) pieces of each
  // hill's integrated bias into `slots` [nh][slots_per_hill]; a correction pass then fixes up the
  // (rare) hills the limiter changed.
  double *slots;
  int slots_per_hill;
  // lookup replica kept current by an in-place gather (groups == 1, MODE 0) of a 2-D / 3-D grid, or NULL
  double *faces;
};
// slots per hill needed by the fused gather for this geometry
int gather_slots_per_hill(const Geom &g);
// fused gather: partial[0..groups) += base-height contributions; added[i] = integrated bias of hill i
hipError_t launch_hill_gather_fused(const Geom &g, const Tables &t, const HillList &h, const double *heights,
                                    double h_const, const GatherPlan &plan, double *added, int *dirty_flag,
                                    hipStream_t s);
// correction: partial[groups] = sum over the limiter's tail hills of (tail_h1 - base, tail_h2) terms
// (zero when the limiter changed nothing); then rec += partial[0] + ... + partial[groups] in order
hipError_t launch_hill_gather_correct_and_apply(const Geom &g, const Tables &t, double *rec, const HillList &h,
                                                const HillHeights &hh, const GatherPlan &plan, int with_correction,
                                                int *dirty_flag, hipStream_t s);
long long gather_tiles(const Geom &g);
// applies the hills to the record array in list order; dirty_flag (device int) is set
// when any boundary correction was non-zero (gaussian_grid.h:357-358)
hipError_t launch_hill_gather(const Geom &g, const Tables &t, double *rec, const HillList &h,
                              const HillHeights &hh, const GatherPlan &plan, int *dirty_flag, hipStream_t s,
                              const PostSpec *chain = nullptr);
// Short 1-D batches: the per-hill integrals + chained limiter (launch_hill_integrals with `chain`) and the in-place
// gather (launch_hill_gather with plan.groups == 1) as ONE launch whose two halves run side by side; the gather
// waits for the limiter's word (chain.ready_flag == chain.ready_seq, a zero-initialised device word and a sequence
// number that grows with every such launch) only before it applies heights.  hh.res_dev is the limiter's result.
bool integrals_gather_fusable(const Geom &g, long long nh_bound, const GatherPlan &plan);
hipError_t launch_integrals_gather(const Geom &g, const Tables &t, double *rec, const HillList &h, const double *heights,
                                   double h_const, double *added, const LimitArgs &chain, const HillHeights &hh,
                                   const GatherPlan &plan, int *dirty_flag, hipStream_t s, const PostSpec *post_chain);

// pieces of launch_hill_gather_correct_and_apply for the sharded multi-GPU application: the correction pass
// alone (writes partial buffer [plan.groups], zeroed first) and dst[i] += sum of `groups` partial buffers
hipError_t launch_hill_gather_correction(const Geom &g, const Tables &t, const HillList &h, const HillHeights &hh,
                                         const GatherPlan &plan, int *dirty_flag, hipStream_t s);
hipError_t launch_add_partials(const Geom &g, double *dst, const double *partial, int groups,
                               const LimitResult *res_dev, hipStream_t s);
// gaussian_grid.h:571-630, executed iff *dirty_flag != 0; clears the flag
hipError_t launch_duplicate_boundary(const Geom &g, double *rec, int *dirty_flag, hipStream_t s);

// fused bookkeeping after a limited batch: boundary duplication + CV histogram (+1 per hill, or per
// replayed hill of a flush; -1 per undo)
struct LimitResult;
hipError_t launch_post_batch(const Geom &g, double *rec, int *dirty_flag, const Geom &hist_geom, double *hist,
                             long long nh, const double *hx0, const LimitResult *res_dev, const int *flags,
                             int flush_mode, hipStream_t s);

// limiter (edm_bias.cpp:444-526 for new hills, :313-380 for the overflow flush)
#define EDM_TAIL_CAP 12288
#define EDM_CHUNK 4096
struct LimitResult {
  double cum_out;          // temp_hill_cum_ after the batch (flush: bias added by the flush)
  long long k;             // first hill handled by the ordered tail
  long long nh;            // number of hills actually in the batch (deferred count resolved here)
  int n_tail;              // nh - k
  int stop;                // flush mode: tail-relative index where the flush stopped, or n_tail
  int n_deferred;          // new-hill mode: hills (whole or remainder) to append to the overflow buffer
  int error;               // 1 = tail longer than EDM_TAIL_CAP, 2 = deferred count exceeded the launch bound
  // new hills only: every hill of the batch was added in full -- no undo, nothing deferred -- so flags, undo heights
  // and undo bias need not be read at all (they are 1, 0, 0)
  int all_plain;
  int pad0;
  double h2_stop;                   // flush mode: the undo height of the hill the flush stopped at (0 if it ran through) -- all a
                                    // host without a HILLS log needs of the tail arrays
  unsigned long long reserved;      // (the header of the packed read-back region is one 64-byte line)
};
// The limiter's wave also sends its result to a line of its own in host memory (LimitArgs::fast_line) as ONE 64-byte
// store (header_line_to_host in edm_kernels.hip):
// eight words [seq | cum_out | k, nh | n_tail, stop | n_deferred, error | all_plain | h2_stop | seq], the batch's number
// first and last.  The host may take that line the moment both show the number it waits for, without waiting for the
// acknowledgement of every other store into the region (~3 us on PCIe).
inline bool edm_header_line_decode(const unsigned long long line[8], unsigned long long want, LimitResult *out) {
  if (line[0] != want || line[7] != want) return false;
  memset(out, 0, sizeof(*out));
  memcpy(&out->cum_out, &line[1], 8);
  out->k = (long long)(line[2] & 0xFFFFFFFFull);        // (chained batches: at most 2048 hills)
  out->nh = (long long)(line[2] >> 32);
  out->n_tail = (int)(unsigned)(line[3] & 0xFFFFFFFFull);
  out->stop = (int)(unsigned)(line[3] >> 32);
  out->n_deferred = (int)(unsigned)(line[4] & 0xFFFFFFFFull);
  out->error = (int)(unsigned)(line[4] >> 32);
  out->all_plain = (int)(unsigned)(line[5] & 0xFFFFFFFFull);
  memcpy(&out->h2_stop, &line[6], 8);
  return true;
}
static_assert(sizeof(LimitResult) == 64, "the limiter's result is one 64-byte line");
// flags per tail hill: bit0 = applied (an 'h'/'b' hill was added), bit1 = undo hill
// added too ('u'/'v'), bit2 = deferred to the overflow buffer
struct LimitTail {
  double *h1, *h2, *added2, *cum_after;
  int *flags;
};
size_t limit_scratch_doubles(long long nh);
hipError_t launch_limit(long long nh, const double *added, const double *heights, double h_const,
                        double limit, double cum_in, int flush_mode, const LimitTail &tail,
                        LimitResult *result_dev, double *scratch, hipStream_t s,
                        const long long *nh_dev = nullptr);

// limiter chained onto the integrals kernel (h.nh <= 2048, no chunk skipping)
struct LimitArgs {
  int enabled;
  int *ticket;
  double limit, cum_in;
  int flush_mode;
  LimitTail tail;
  LimitResult *res;
  // optional: once the limiter has run, the packed read-back region of the batch (apply_hills' layout: header +
  // flags | h2 | added2 | added | positions, sized by the launch bound) is complete -- nothing the host reads is
  // written by the gather -- and the last workgroup copies the part the true hill count fills to rb_dst
  // (host-mapped) and then stores done_seq to *done_flag: the host polls that word and returns while the gather
  // still runs (every later call is ordered behind it on the stream)
  const char *rb_src;
  char *rb_dst;
  unsigned long long *done_flag;
  unsigned long long done_seq;
  // optional, with done_flag: 64 bytes of host-mapped memory that receive the limiter's result as one store the moment
  // the limiter's wave has it (see edm_header_line_decode)
  unsigned long long *fast_line;
  // k_integrals_gather: device word the gather workgroups of the same launch poll for ready_seq
  unsigned long long *ready_flag;
  unsigned long long ready_seq;
  int early_word;   // the last wave of the limiter's workgroup may publish EDM_READY_BELOW ahead of the limiter
  // optional (a workgroup per hill, launch wider than the hill count): every hill's workgroup stores {integral, tag_seq}
  // as ONE 16-byte agent-scope store into tagged[2 * hill], takes no ticket and is done; the first workgroup WITHOUT a
  // hill polls the slots and runs the limiter on what it has read -- one dependent round trip between the last integral
  // and the limiter instead of three (store acknowledged -> ticket -> reload).  NULL: the last-arrival ticket.
  double *tagged;
  unsigned long long tag_seq;
  // host-side hint: the hill count the batch is expected to have (<= the launch bound h.nh); 0: unknown
  long long expected_hills;
  // optional, k_integrals_gather (a reference-order fix edm_pair step): extra workgroups at the END of the launch store the
  // unit-height stencil terms (value, derivative) of every hill -- ord_parts workgroups per hill, ord_terms[hill][2 msize + 1][2],
  // zeros where the hill does not reach -- and note the hills with a non-zero boundary correction in ord_dirty
  // (ordered_dirty_note).  They wait for nobody; the record pass that follows (launch_ordered_records) reads the terms
  // instead of computing them tile by tile, where a dense tile's hills queue on one CU.
  double *ord_terms;
  unsigned *ord_dirty;      // OrderedForcesArgs::dirty_hill
  unsigned ord_seq;
  // ... and (ord_ready != NULL) for a record pass that runs BESIDE this launch, on another stream, instead of behind it:
  // terms and notes leave as agent-scope stores, each emitter workgroup writes ord_seq to ord_ready[2 hill + part] once
  // its stores are acknowledged, and the emitters are dispatched between the integrals and the gather tiles (which never
  // go first then), so that the record pass finds them done when the limiter's word arrives
  unsigned *ord_ready;
  // host-side, k_integrals_gather: another process runs kernels on this device (ranks sharing a GPU): the waiting
  // gather tiles are never dispatched ahead of the integrals; tiles_first_mode: -1 = the launcher's choice (memset
  // leaves 0 = integrals first, so callers set it), 0 / 1 = forced (tests)
  int shared_device;
  int tiles_first_mode;
  // development aid (EDM_HIP_TRACE=1): 8 wall-clock stamps (10 ns units) per workgroup of k_integrals_gather, or NULL
  unsigned long long *trace;
};
bool hill_integrals_can_chain_limit(long long nh);
// boundary duplication + histogram chained onto the gather (plan.groups == 1, hh.res_dev and h.hx0 set)
struct PostSpec {
  int *ticket;
  const Geom *hist_geom;
  double *hist;
  const int *flags;
  int flush_mode;
  const char *rb_src;   // optional: rb_bytes (multiple of 8) copied to rb_dst (host-mapped) at the very end
  char *rb_dst;
  long long rb_bytes;
};


// histogram side of output_hill for the ordered tail: -1 for every hill whose undo was added
// (flags bit1) and, when plus_for_applied, +1 for every hill that was applied (bit0), at its
// original position hx0[(k + j)]
hipError_t launch_hist_tail(const Geom &hist, double *values, const LimitResult *res_dev, const int *flags,
                            const double *hx0, int plus_for_applied, hipStream_t s);

// Ordered hill application for heights that depend on the bias under construction (local tempering,
// edm_bias.cpp:547-549): ONE workgroup applies the hills strictly one after another -- height from the
// current grid, integrated bias, limiter step, stencil update, boundary duplication -- with workgroup
// barriers in between, exactly the reference's sequence.  Fills the same tail arrays as launch_limit
// (k = 0, n_tail = nh <= EDM_TAIL_CAP), the per-hill heights and bias.
struct OrderedParams {
  double prefactor;        // temp_hill_prefactor_
  int use_target;          // b_targeting_
  Geom target;
  const double *target_values;
  double expected_target;
  int use_tempering;       // b_tempering_ && global_tempering_ < 0
  double temper_scale;     // (bias_factor_ - 1) * boltzmann_factor_
  double divisor;          // est_hill_count_ or hill_density_
  double clamp;            // BIAS_CLAMP * bias_per_step_
  double limit, cum_in;
};
hipError_t launch_hills_ordered(const Geom &g, const Tables &t, double *rec, const HillList &h, const OrderedParams &op,
                                const LimitTail &tail, double *heights_out, double *added_out,
                                LimitResult *result_dev, int *dirty_flag, hipStream_t s);

// hill heights with a target factor (edm_bias.cpp:537-558):
//   h_i = min( prefactor * exp(T(x_i) - expected) / divisor , clamp )
// T = Grid::get_value on the target grid read WITHOUT interpolation (nearest-lower node, 0 outside
// in_grid; grid.h:343-365), evaluated at the un-remapped sample position.
hipError_t launch_target_heights(const Geom &target, const double *target_values, long long n, const double *x,
                                 int x_stride, const long long *sel, double prefactor, double expected,
                                 double divisor, double clamp, double *out_h, hipStream_t s);

// out[i][0..dim) = x[(sel ? sel[i] : i) * x_stride + 0..dim): contiguous hill records for the exchange
hipError_t launch_gather_positions(long long n, const double *x, int x_stride, const long long *sel, int dim,
                                   double *out, hipStream_t s);

hipError_t launch_sum(long long n, const double *v, double *out, double *scratch, hipStream_t s);

// device array -> page-locked host memory by zero-copy stores (both 16-byte aligned), see edm_kernels.hip
hipError_t launch_copy_to_host(const double *d_src, double *h_dst_mapped, long long n, hipStream_t s);

// the completion protocol on a self-checking payload (edm_hip_debug_flag_order_stress)
hipError_t launch_flag_order_stress(long long *dst_mapped, long long words, unsigned long long seq, unsigned long long *flag_mapped,
                                    hipStream_t s);

}  // namespace edm
