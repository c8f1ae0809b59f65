// edm_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the EDM bias hot path.
//
// No MFMA anywhere: the path is gather (cubic-Hermite lookup), transcendental
// stencil evaluation and ordered accumulation.  All arithmetic is IEEE double in
// the reference's operation order (-ffp-contract=off), so node indices are
// bit-exact and values differ from the CPU reference only through exp() ulps and
// the documented reduction orders.
//
//   K1/K2  k_pair_forces_fast / k_lookup   bias-grid interpolation, force update, energy sum
//   sel    k_select_prep (k_sel_*)         ordered selection of accepted samples (+ hill preparation,
//                                          or the packet of the multi-GPU exchange; k_unpack_prep receives)
//   K3     k_hill_integrals                per-hill integrated bias (value add_value returns)
//   K4     limit_wave (k_limit)            ordered bias limiter (undo + overflow decisions)
//   K5     k_hill_gather (+ k_mark_tiles)  tile-owned, ORDER-PRESERVING gather of hills onto nodes
//   K6/K7  boundary duplication, CV histogram (chained onto K5, or k_post_batch)
//   K8     block/wave reductions           fixed-order energy and bias sums
// A short hill step is three launches: sel -> K3 (+K4, read-back) -> K5 (+K6, K7), each stage's small serial
// tail run by the last workgroup to finish (last_block_done); in a fix edm_pair step K1 rides in the first launch
// (k_pair_forces_select; the force pass over a device-resident neighbour list likewise, k_pairlist_forces_select).  The
// host is released by a word K3's last workgroup stores behind the read-back region in host-mapped memory: it polls
// that word, not the stream.  fix edm_pair's default order (pair k's force behind the hills of pairs 0..k-1) adds
// k_ordered_records and k_pair_forces_ordered behind the hill batch (OrderedForcesArgs, edm_kernels.h).
#include "edm_kernels.h"

#include <hip/hip_ext.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

// launch with the kernel's own begin/end timestamps in (ev0, ev1) when profiling is on: hipEventRecord
// brackets would add the dispatch latency (~3 us) to a 10 us kernel
#define EDM_LAUNCH_TIMED(kernel, grid, block, lds, s, ev0, ev1, ...)                                   \
  do {                                                                                                 \
    if (ev0)                                                                                           \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, s, ev0, ev1, 0, __VA_ARGS__);                    \
    else                                                                                               \
      hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__);                                    \
  } while (0)

namespace edm {

bool test_force(const char *token);        // EDM_HIP_TEST_FORCE tokens (tests only; edm_gauss.cpp)
long long test_force_value(const char *key);

static constexpr int BLOCK = 256;
static constexpr int HIST_LDS_BINS = 2048;  // histograms up to this many bins are accumulated per workgroup in LDS
static constexpr int MAX_BLOCKS = 2048;  // 256 CUs x 8 resident 256-thread blocks

// ---------------------------------------------------------------------------
// fixed-order reductions (deterministic: same launch shape -> same bits)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// A workgroup's partial energy sum as ONE 16-byte system-scope store {sum, tag} into host-mapped memory: a host that
// polls the slot sees the sum the moment it sees the launch's tag -- a forces-only call then needs no stream wait
// (20 -> ~13 us per call of 1 M pairs), only a look at every workgroup's slot.
__device__ __forceinline__ void store_partial_tagged(double *base, unsigned bid, double s, unsigned long long tag) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i d;
  d.x = __double2loint(s);
  d.y = __double2hiint(s);
  d.z = (int)(unsigned)(tag & 0xFFFFFFFFull);
  d.w = (int)(unsigned)(tag >> 32);
  double *p = base + 2 * (size_t)bid;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(d) : "memory");
}

// {value, tag} as ONE 16-byte store / load at agent scope (LimitArgs::tagged): the reader that finds the tag has the value
#define EDM_TAG_CAP 1024   // hills a polling limiter workgroup keeps in LDS
__device__ __forceinline__ void store_tagged_agent(double *base, long long idx, double v, unsigned long long tag) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i d;
  d.x = __double2loint(v);
  d.y = __double2hiint(v);
  d.z = (int)(unsigned)(tag & 0xFFFFFFFFull);
  d.w = (int)(unsigned)(tag >> 32);
  double *p = base + 2 * (size_t)idx;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ bool load_tagged_agent(const double *base, long long idx, unsigned long long tag, double *v) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i d;
  const double *p = base + 2 * (size_t)idx;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(p) : "memory");
  *v = __hiloint2double(d.y, d.x);
  return ((unsigned long long)(unsigned)d.z | ((unsigned long long)(unsigned)d.w << 32)) == tag;
}

// number of hills of a batch (resolves a deferred count)
__device__ __forceinline__ long long hill_count(const HillList &h) {
  long long n = h.nh;
  if (h.nh_dev) {
    const long long t = *h.nh_dev;
    n = t < n ? t : n;
  }
  return n;
}

// hill groups of an adaptive gather plan (a function of the true hill count only)
__device__ __host__ __forceinline__ int adaptive_groups(int cap, long long nh) {
  long long G = nh / 128;
  if (G > cap) G = cap;
  if (G < 1) G = 1;
  return (int)G;
}

// sum over the block, valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; w++) r += lds[w];
  }
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(BLOCK) k_sum_partials(const double *__restrict__ v, long long n, double *out) {
  __shared__ double lds[BLOCK / 64];
  double acc = 0;
  for (long long i = threadIdx.x; i < n; i += BLOCK) acc += v[i];
  double r = block_sum(acc, lds);
  if (threadIdx.x == 0) *out = r;
}

__global__ void __launch_bounds__(BLOCK) k_block_sums(const double *__restrict__ v, long long n, double *partial) {
  __shared__ double lds[BLOCK / 64];
  double acc = 0;
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) acc += v[i];
  double r = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// Grid::get_value on a grid read WITHOUT interpolation (grid.h:343-365): nearest-lower node, 0 outside
template <int DIM>
__device__ __forceinline__ double target_value(const Geom &g, const double *__restrict__ values, const double *xx) {
  if (!in_grid<DIM>(g, xx)) return 0;
  long long idx[DIM];
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    double w;
    idx[d] = node_index(g, d, xx[d], &w);
    if (idx[d] > g.n[d] - 1) idx[d] = g.n[d] - 1;
    if (idx[d] < 0) idx[d] = 0;
  }
  long long flat = idx[DIM - 1];
#pragma unroll
  for (int d = DIM - 1; d > 0; d--) flat = flat * g.n[d - 1] + idx[d - 1];
  return values[flat];
}

template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_target_heights(Geom g, const double *__restrict__ values, long long n,
                                                          const double *__restrict__ x, int x_stride,
                                                          const long long *__restrict__ sel, double prefactor,
                                                          double expected, double divisor, double clamp,
                                                          double *__restrict__ out_h) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const long long src = sel ? sel[i] : i;
    double xx[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) xx[d] = x[src * x_stride + d];
    const double t = target_value<DIM>(g, values, xx);
    double h = prefactor;
    h *= exp(t - expected);   // :546
    h /= divisor;             // :552-555
    h = fmin(h, clamp);       // :558
    out_h[i] = h;
  }
}
hipError_t launch_target_heights(const Geom &target, const double *target_values, long long n, const double *x,
                                 int x_stride, const long long *sel, double prefactor, double expected,
                                 double divisor, double clamp, double *out_h, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  long long b = (n + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  switch (target.dim) {
    case 1: hipLaunchKernelGGL(k_target_heights<1>, dim3((unsigned)b), dim3(BLOCK), 0, s, target, target_values, n, x, x_stride, sel, prefactor, expected, divisor, clamp, out_h); break;
    case 2: hipLaunchKernelGGL(k_target_heights<2>, dim3((unsigned)b), dim3(BLOCK), 0, s, target, target_values, n, x, x_stride, sel, prefactor, expected, divisor, clamp, out_h); break;
    default: hipLaunchKernelGGL(k_target_heights<3>, dim3((unsigned)b), dim3(BLOCK), 0, s, target, target_values, n, x, x_stride, sel, prefactor, expected, divisor, clamp, out_h); break;
  }
  return hipGetLastError();
}

// DimmedGrid::get_value / get_value_deriv on a grid that stores NO derivatives (grid.h:343-365): the value of
// the nearest-lower node, 0 outside in_grid; the derivative (which the reference would read through a NULL
// grid_deriv_) is reported as 0
template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_nearest_values(Geom g, const double *__restrict__ values, long long n,
                                                          const double *__restrict__ x, int x_stride,
                                                          double *__restrict__ out_value, double *__restrict__ out_deriv) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    double xx[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) xx[d] = x[i * x_stride + d];
    if (out_value) out_value[i] = target_value<DIM>(g, values, xx);
    if (out_deriv) {
#pragma unroll
      for (int d = 0; d < DIM; d++) out_deriv[i * DIM + d] = 0.0;
    }
  }
}
hipError_t launch_nearest_values(const Geom &g, const double *values, long long n, const double *x, int x_stride,
                                 double *out_value, double *out_deriv, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  long long b = (n + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_nearest_values<1>, dim3((unsigned)b), dim3(BLOCK), 0, s, g, values, n, x, x_stride, out_value, out_deriv); break;
    case 2: hipLaunchKernelGGL(k_nearest_values<2>, dim3((unsigned)b), dim3(BLOCK), 0, s, g, values, n, x, x_stride, out_value, out_deriv); break;
    default: hipLaunchKernelGGL(k_nearest_values<3>, dim3((unsigned)b), dim3(BLOCK), 0, s, g, values, n, x, x_stride, out_value, out_deriv); break;
  }
  return hipGetLastError();
}

// DimmedGaussGrid::remap (gaussian_grid.h:504-541) as the lookup and hill kernels apply it, on its own: out rows of
// dim doubles
template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_remap(Geom g, long long n, const double *__restrict__ x, int x_stride,
                                                 double *__restrict__ out) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    double xx[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) xx[d] = x[i * x_stride + d];
    remap<DIM>(g, xx);
#pragma unroll
    for (int d = 0; d < DIM; d++) out[i * DIM + d] = xx[d];
  }
}
hipError_t launch_remap(const Geom &g, long long n, const double *x, int x_stride, double *out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  long long b = (n + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_remap<1>, dim3((unsigned)b), dim3(BLOCK), 0, s, g, n, x, x_stride, out); break;
    case 2: hipLaunchKernelGGL(k_remap<2>, dim3((unsigned)b), dim3(BLOCK), 0, s, g, n, x, x_stride, out); break;
    default: hipLaunchKernelGGL(k_remap<3>, dim3((unsigned)b), dim3(BLOCK), 0, s, g, n, x, x_stride, out); break;
  }
  return hipGetLastError();
}

// Grid::add (grid.h:275-290), device side.  k_node_coords: the coordinates min + dx * index (:282-284) of nodes
// [first, first + count), rows of dim doubles; k_axpy_nodes: grid_[i] += scale * E + offset and
// grid_deriv_[i][j] += scale * D[j] (:285-287) for the same nodes (derivative slots only where the grid has them).
__global__ void __launch_bounds__(BLOCK) k_node_coords(Geom g, long long first, long long count, double *__restrict__ out) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < count; i += stride) {
    long long index = first + i;
    int d;
    for (d = 0; d < g.dim - 1; d++) {
      const long long k = index % g.n[d];
      index = (index - k) / g.n[d];
      out[i * g.dim + d] = g.min[d] + g.dx[d] * (double)(unsigned long long)k;
    }
    out[i * g.dim + d] = g.min[d] + g.dx[d] * (double)(unsigned long long)index;
  }
}
hipError_t launch_node_coords(const Geom &g, long long first, long long count, double *out, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  long long b = (count + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  hipLaunchKernelGGL(k_node_coords, dim3((unsigned)b), dim3(BLOCK), 0, s, g, first, count, out);
  return hipGetLastError();
}
__global__ void __launch_bounds__(BLOCK) k_axpy_nodes(Geom g, double *__restrict__ base, long long first, long long count,
                                                      const double *__restrict__ E, const double *__restrict__ D,
                                                      double scale, double offset) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < count; i += stride) {
    double *r = base + (first + i) * g.rec;
    r[0] += scale * E[i] + offset;
    if (g.has_deriv)
      for (int d = 0; d < g.dim; d++) r[1 + d] += scale * D[i * g.dim + d];
  }
}
hipError_t launch_axpy_nodes(const Geom &g, double *base, long long first, long long count, const double *E,
                             const double *D, double scale, double offset, hipStream_t s) {
  if (count <= 0) return hipSuccess;
  long long b = (count + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  hipLaunchKernelGGL(k_axpy_nodes, dim3((unsigned)b), dim3(BLOCK), 0, s, g, base, first, count, E, D, scale, offset);
  return hipGetLastError();
}
// base[i * rec] = values[i]: node values of a record grid replaced, derivative slots kept
__global__ void __launch_bounds__(BLOCK) k_set_values(Geom g, double *__restrict__ base, const double *__restrict__ values) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < g.total; i += stride) base[i * g.rec] = values[i];
}
hipError_t launch_set_values(const Geom &g, double *base, const double *values, hipStream_t s) {
  long long b = (g.total + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(k_set_values, dim3((unsigned)b), dim3(BLOCK), 0, s, g, base, values);
  return hipGetLastError();
}

__global__ void __launch_bounds__(BLOCK) k_gather_positions(long long n, const double *__restrict__ x, int x_stride,
                                                            const long long *__restrict__ sel, int dim,
                                                            double *__restrict__ out) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const long long src = sel ? sel[i] : i;
    for (int d = 0; d < dim; d++) out[i * dim + d] = x[src * x_stride + d];
  }
}
hipError_t launch_gather_positions(long long n, const double *x, int x_stride, const long long *sel, int dim,
                                   double *out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  long long b = (n + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  hipLaunchKernelGGL(k_gather_positions, dim3((unsigned)b), dim3(BLOCK), 0, s, n, x, x_stride, sel, dim, out);
  return hipGetLastError();
}

hipError_t launch_sum(long long n, const double *v, double *out, double *scratch, hipStream_t s) {
  int blocks = (int)((n + BLOCK - 1) / BLOCK);
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_block_sums, dim3(blocks), dim3(BLOCK), 0, s, v, n, scratch);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(BLOCK), 0, s, scratch, (long long)blocks, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K1/K2: lookup  (gaussian_grid.h:118-138 -> grid.h:390-446 -> interp grid.h:52-139)
// ---------------------------------------------------------------------------
template <int R>
struct Rec;
template <>
struct Rec<2> {
  double v[2];
  __device__ __forceinline__ void load(const double *__restrict__ rec, long long node) {
    const double2 t = reinterpret_cast<const double2 *>(rec)[node];
    v[0] = t.x;
    v[1] = t.y;
  }
};
template <>
struct Rec<4> {
  double v[4];
  __device__ __forceinline__ void load(const double *__restrict__ rec, long long node) {
    const double4 t = reinterpret_cast<const double4 *>(rec)[node];
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
};

// Returns the flat start node (or -1 where the reference returns 0) and the
// interpolated value / POSITIVE gradient.
// (SRC: where the node records come from -- src.load(Rec<R> &, node).  The grid's record array for every caller but
//  the reference-order pair forces, which read each node as it stood after a prefix of the step's hills.)
template <int DIM, class SRC>
__device__ __forceinline__ long long lookup_one_src(const Geom &g, const SRC &src, const double *xin, double &value,
                                                    double *der) {
  constexpr int R = (DIM == 1) ? 2 : 4;
  double xx[DIM];
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    xx[d] = xin[d];
    der[d] = 0;
  }
  value = 0;
  if (!in_bounds<DIM>(g, xx)) {
    remap<DIM>(g, xx);
    if (!in_bounds<DIM>(g, xx)) return -1;
  }
  if (!in_grid<DIM>(g, xx)) return -1;

  long long idx[DIM], stride[DIM];
  double where[DIM];
  stride[0] = 1;
#pragma unroll
  for (int d = 1; d < DIM; d++) stride[d] = stride[d - 1] * g.n[d - 1];
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    double w;
    idx[d] = node_index(g, d, xx[d], &w);
    // the reference reads one node past the array when the periodic wrap rounds
    // up to max exactly (undefined there); stay inside the allocation instead
    if (idx[d] > g.n[d] - 1) idx[d] = g.n[d] - 1;
    if (idx[d] < 0) idx[d] = 0;
    where[d] = w - g.min[d] - idx[d] * g.dx[d];
  }
  long long flat = idx[DIM - 1];
#pragma unroll
  for (int d = DIM - 1; d > 0; d--) flat = flat * g.n[d - 1] + idx[d - 1];

  if (!g.interp) {
    Rec<R> r;
    src.load(r, flat);
    value = r.v[0];
#pragma unroll
    for (int d = 0; d < DIM; d++) der[d] = r.v[1 + d];
    return flat;
  }
#pragma unroll
  for (int d = 0; d < DIM; d++)
    if (g.periodic[d] && idx[d] == g.n[d] - 1) stride[d] *= (1 - g.n[d]);

  // scaled coordinate and its powers per dimension do not depend on the corner's node
  double f = 0;
#pragma unroll
  for (int corner = 0; corner < (1 << DIM); corner++) {
    long long shift = 0;
#pragma unroll
    for (int d = 0; d < DIM; d++) shift += stride[d] * ((corner >> d) & 1);
    Rec<R> r;
    src.load(r, flat + shift);
    const double tf = r.v[0];
    double C[DIM], D[DIM];
    double ff = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      const int bit = (corner >> d) & 1;
      const int sgn = bit ? -1 : 1;
      const double X = fabs(where[d] / g.dx[d] - bit);
      const double X2 = X * X;
      const double X3 = X2 * X;
      double qq;
      if (fabs(tf) < 0.0000001)  // grid.h:113-116: derivative term dropped near zero
        qq = 0.0;
      else
        qq = -r.v[1 + d] / tf;
      C[d] = (1 - 3 * X2 + 2 * X3) - sgn * qq * (X - 2 * X2 + X3) * g.dx[d];
      D[d] = (-6 * X + 6 * X2) - sgn * qq * (1 - 4 * X + 3 * X2) * g.dx[d];
      D[d] *= sgn / g.dx[d];
      ff *= C[d];
    }
    f += tf * ff;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      double fd = D[d];
#pragma unroll
      for (int e = 0; e < DIM; e++)
        if (e != d) fd *= C[e];
      der[d] += tf * fd;
    }
  }
  value = f;
  return flat;
}
template <int R>
struct RecArray {
  const double *__restrict__ rec;
  __device__ __forceinline__ void load(Rec<R> &r, long long node) const { r.load(rec, node); }
};
template <int DIM>
__device__ __forceinline__ long long lookup_one(const Geom &g, const double *__restrict__ rec,
                                                const double *xin, double &value, double *der) {
  const RecArray<(DIM == 1) ? 2 : 4> src{rec};
  return lookup_one_src<DIM>(g, src, xin, value, der);
}

template <int DIM, int MODE>
__global__ void __launch_bounds__(BLOCK) k_lookup(Geom g, const double *__restrict__ rec, LookupArgs a,
                                                  double *__restrict__ block_energy) {
  __shared__ double lds[BLOCK / 64];
  double e_acc = 0;
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < a.n; i += stride) {
    if (MODE == LOOKUP_FORCES && !(a.apply_mask < 0 || (a.mask[i] & a.apply_mask))) continue;
    double xin[DIM], der[DIM], v;
#pragma unroll
    for (int d = 0; d < DIM; d++) xin[d] = a.x[i * a.x_stride + d];
    const long long flat = lookup_one<DIM>(g, rec, xin, v, der);
    if (MODE == LOOKUP_FORCES) {
      e_acc += v;
#pragma unroll
      for (int d = 0; d < DIM; d++) a.f[i * a.f_stride + d] -= der[d];
    } else if (MODE == LOOKUP_VALUES) {
      if (a.energy) a.energy[i] = v;
      if (a.f) {
#pragma unroll
        for (int d = 0; d < DIM; d++) a.f[i * DIM + d] = der[d];
      }
      e_acc += v;
    } else {
      a.flat[i] = flat;
    }
  }
  double r = block_sum(e_acc, lds);
  if (threadIdx.x == 0) {
    if (a.partial_tag) store_partial_tagged(block_energy, blockIdx.x, r, a.partial_tag); else block_energy[blockIdx.x] = r;
  }
}

size_t lookup_scratch_doubles() { return 2 * MAX_BLOCKS + 16; }   // (tagged partial sums take two doubles per workgroup)

// ---------------------------------------------------------------------------
// K2 on the lookup replica: FOUR LANES PER SAMPLE.
//
// The replica keeps one aligned 128-byte block per node with the records of nodes (i0, i1), (i0+1, i1), (i0, i1+1),
// (i0+1, i1+1) [at i2] (periodic wrap applied): the four corners of a 2-D cell, one face of a 3-D cell.  Lane q of
// a sample's quad loads slot q of the cell's block (3-D: and of the block above), so the quad reads whole lines and
// a wave-instruction touches 16 lines / pages instead of 64.  That, not the byte count, is what the coordinate-CV
// lookup was short of: measured on MI355X (tools/randline.hip) random 128-byte lines of a 16 GiB buffer read at
// 5.9 TB/s with four lanes per line and at 1.1-1.2 TB/s with one lane per line -- 64 address translations per
// instruction -- while the node-record layout made every sample touch 2.5 (2-D) / 5 (3-D) lines of which it used
// 96 / 256 bytes.  The arithmetic is the reference's, operation for operation (grid.h:390-446, interp<DIM> :52-139),
// spread over the quad: lane d derives node index and cell coordinate of dimension d, lane q evaluates corner q
// (3-D: corners q and q + 4), and every lane adds the corner terms up in the reference's corner order, so values
// and gradients are bit-identical to k_lookup's.  Lane d < DIM updates force component d, lane 3 keeps the energy.
// ---------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ double quad_bcast(double v) {   // the value lane K of this lane's quad holds
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);   // DPP quad_perm:[K,K,K,K]
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ int quad_bcast_i(int v) {
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
  return __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true);
}
// quad_perm:[1,0,3,2] / [2,3,0,1]: the value of the lane whose index within the quad differs in bit 0 / bit 1
template <int XOR>
__device__ __forceinline__ double quad_xor(double v) {
  constexpr int ctrl = (XOR == 1) ? (1 | (0 << 2) | (3 << 4) | (2 << 6)) : (2 | (3 << 2) | (0 << 4) | (1 << 6));
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {   // every lane of the quad: the quad's sum, (0+1) + (2+3)
  v += quad_xor<1>(v);
  v += quad_xor<2>(v);
  return v;
}
// one corner of interp<DIM> (grid.h:85-131): tf * prod C and tf * D_d * prod_{e != d} C_e.  The reference's
// qq = -d / f per dimension (grid.h:117) is formed from ONE reciprocal of the corner's value, and D's scaling by
// sgn / dx (grid.h:123) from the launch's 1 / dx: a double division is ~30 instructions, a 3-D sample had fifteen of them
// per lane and the kernel is bound by fp64 issue, not by HBM.  Values move by ~1e-16 relative (as in K1's fast path).
template <int DIM>
__device__ __forceinline__ void corner_terms(const Geom &g, const double *inv_dx, const double4 &r, const double *wod,
                                             int b0, int b1, int b2, double &tF, double *tD) {
  const double tf = r.x;
  const double dv[3] = {r.y, r.z, r.w};
  const double rf = (fabs(tf) < 0.0000001) ? 0.0 : 1.0 / tf;   // grid.h:113-116: derivative term dropped near zero
  double C[DIM], D[DIM];
  double ff = 1.0;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    const int bit = (d == 0) ? b0 : (d == 1) ? b1 : b2;
    const double sgn = bit ? -1.0 : 1.0;
    const double X = fabs(wod[d] - bit);
    const double X2 = X * X;
    const double X3 = X2 * X;
    const double sq = sgn * (-dv[d] * rf) * g.dx[d];
    C[d] = (1 - 3 * X2 + 2 * X3) - sq * (X - 2 * X2 + X3);
    D[d] = ((-6 * X + 6 * X2) - sq * (1 - 4 * X + 3 * X2)) * (sgn * inv_dx[d]);
    ff *= C[d];
  }
  tF = tf * ff;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    double fd = D[d];
#pragma unroll
    for (int e = 0; e < DIM; e++)
      if (e != d) fd *= C[e];
    tD[d] = tf * fd;
  }
}
// REPLICA false: the same four-lanes-per-sample kernel on the NODE RECORDS (grids without a replica: walls, or small
// enough to sit in L2): lane q loads corner q (and q + 4) at its node's address -- a quad covers the 64-byte corner
// pairs of two grid rows per instruction, 32 lines per wave-instruction instead of 64, a quarter of the load
// instructions per sample.
// Lane d < DIM owns dimension d: it loads coordinate d of the sample (one 8-byte load per lane instead of DIM), runs
// the bounds / remap / in-grid tests of that dimension (gaussian_grid.h:490-541, grid.h:865-874: all per dimension) and
// derives its node index and cell coordinate; DPP quad moves hand the pieces round.  The corner terms are added up
// over the quad as (0 + 1) + (2 + 3), in 3-D after each lane has added its two faces -- not the reference's corner
// order, a difference of association (~1e-16).
// (body for workgroup `bid` of `nb`: shared by the plain launch and by the launch that also prepares a hill list)
template <int DIM, int MODE, bool REPLICA>
__device__ __forceinline__ void lookup_quad_body(const Geom &g, const double *__restrict__ faces, const LookupArgs &a,
                                                 double *__restrict__ block_energy, unsigned bid, unsigned nb) {
  static_assert(DIM == 2 || DIM == 3, "the replica serves 2-D and 3-D grids");
  __shared__ double lds[BLOCK / 64];
  const int q = threadIdx.x & 3;
  const int b0 = q & 1, b1 = q >> 1;
  // this lane's dimension (lane 3, and lane 2 in 2-D, shadow another lane's: their results are not read)
  const int myd = (q < DIM) ? q : DIM - 1;
  const double mn = (myd == 0) ? g.min[0] : (myd == 1) ? g.min[1] : g.min[2];
  const double mx = (myd == 0) ? g.max[0] : (myd == 1) ? g.max[1] : g.max[2];
  const double dxd = (myd == 0) ? g.dx[0] : (myd == 1) ? g.dx[1] : g.dx[2];
  const double bmn = (myd == 0) ? g.bmin[0] : (myd == 1) ? g.bmin[1] : g.bmin[2];
  const double bmx = (myd == 0) ? g.bmax[0] : (myd == 1) ? g.bmax[1] : g.bmax[2];
  const int nd = (myd == 0) ? g.n[0] : (myd == 1) ? g.n[1] : g.n[2];
  const int perd = (myd == 0) ? g.periodic[0] : (myd == 1) ? g.periodic[1] : g.periodic[2];
  const int bperd = (myd == 0) ? g.bper[0] : (myd == 1) ? g.bper[1] : g.bper[2];
  const double inv_dxd = 1.0 / dxd;
  double inv_dx[DIM];
#pragma unroll
  for (int d = 0; d < DIM; d++) inv_dx[d] = 1.0 / g.dx[d];
  double e_acc = 0;
  const long long stride = (long long)nb * (BLOCK / 4);
  for (long long i = (long long)bid * (BLOCK / 4) + (threadIdx.x >> 2); i < a.n; i += stride) {
    if (MODE == LOOKUP_FORCES && !(a.apply_mask < 0 || (a.mask[i] & a.apply_mask))) continue;   // (uniform over the quad)
    double xi = a.x[i * a.x_stride + myd];
    // the force component this lane will update: requested now, needed last
    double f_old = 0;
    if (MODE == LOOKUP_FORCES && q < DIM) f_old = a.f[i * a.f_stride + q];
    // in_bounds (closed interval on the boundary), remap if ANY dimension is outside, in_bounds again, in_grid
    int out_d = (xi < bmn || xi > bmx) ? 1 : 0;
    int any_out = quad_bcast_i<0>(out_d) | quad_bcast_i<1>(out_d);
    if (DIM == 3) any_out |= quad_bcast_i<2>(out_d);
    if (any_out) {   // gaussian_grid.h:504-541, one dimension
      if (xi < mn || xi > mx) {
        if (perd) {
          xi -= (mx - mn) * ifloor((xi - mn) / (mx - mn));
        } else if (bperd) {
          const double period = bmx - bmn;
          const double s0 = round_half((mn - xi) / (bmx - bmn)) * period;
          const double s1 = round_half((mx - xi) / (bmx - bmn)) * period;
          if (fabs(mn - xi - s0) < fabs(mx - xi - s1)) xi += s0; else xi += s1;
        }
      }
      out_d = (xi < bmn || xi > bmx) ? 1 : 0;
    }
    if (!perd && (xi < mn || xi >= mx - dxd)) out_d = 1;   // grid.h:865-874
    int bad = quad_bcast_i<0>(out_d) | quad_bcast_i<1>(out_d);
    if (DIM == 3) bad |= quad_bcast_i<2>(out_d);
    double value = 0, my_der = 0;
    if (!bad) {   // (uniform over the quad)
      // node index, offset inside the cell and scaled coordinate of this lane's dimension (grid.h:264-273, :426-430, :99)
      if (perd) xi -= (mx - mn) * ifloor((xi - mn) / (mx - mn));
      long long id = (long long)floor((xi - mn) / dxd);
      if (id > nd - 1) id = nd - 1;   // (the reference reads past the array when the wrap rounds up to max: stay inside)
      if (id < 0) id = 0;
      const double wd = (xi - mn - id * dxd) * inv_dxd;
      const int my_idx = (int)id;
      int idx[DIM];
      double wod[DIM];
      idx[0] = quad_bcast_i<0>(my_idx);
      wod[0] = quad_bcast<0>(wd);
      idx[1] = quad_bcast_i<1>(my_idx);
      wod[1] = quad_bcast<1>(wd);
      if (DIM == 3) {
        idx[DIM - 1] = quad_bcast_i<2>(my_idx);
        wod[DIM - 1] = quad_bcast<2>(wd);
      }
      long long blk = idx[DIM - 1];
#pragma unroll
      for (int d = DIM - 1; d > 0; d--) blk = blk * g.n[d - 1] + idx[d - 1];
      const double4 *f4 = reinterpret_cast<const double4 *>(faces);
      long long at;   // this lane's record of the lower face
      if (REPLICA) {
        at = blk * 4 + q;
      } else {
        // node records: corner (b0, b1) sits b0 nodes along dimension 0 and b1 rows along dimension 1 from the
        // cell's start node, each step wrapping to node 0 across a periodic seam (grid.h:432-433)
        long long s0 = 1, s1 = g.n[0];
        if (g.periodic[0] && idx[0] == g.n[0] - 1) s0 *= (1 - g.n[0]);
        if (g.periodic[1] && idx[1] == g.n[1] - 1) s1 *= (1 - g.n[1]);
        at = blk + (b0 ? s0 : 0) + (b1 ? s1 : 0);
      }
      const double4 rA = f4[at];
      double tF, tD[DIM];
      if (DIM == 3) {
        // the face above: node i2 + 1, or node 0 across a periodic seam (grid.h:432-433)
        long long up = (long long)g.n[0] * g.n[1];
        if (g.periodic[DIM - 1] && idx[DIM - 1] == g.n[DIM - 1] - 1) up *= (1 - g.n[DIM - 1]);
        const double4 rB = f4[REPLICA ? (blk + up) * 4 + q : at + up];
        double tFB, tDB[DIM];
        corner_terms<DIM>(g, inv_dx, rA, wod, b0, b1, 0, tF, tD);
        corner_terms<DIM>(g, inv_dx, rB, wod, b0, b1, 1, tFB, tDB);
        tF += tFB;
#pragma unroll
        for (int d = 0; d < DIM; d++) tD[d] += tDB[d];
      } else {
        corner_terms<DIM>(g, inv_dx, rA, wod, b0, b1, 0, tF, tD);
      }
      value = quad_sum(tF);
      double der[DIM];
#pragma unroll
      for (int d = 0; d < DIM; d++) der[d] = quad_sum(tD[d]);
      my_der = (q == 0) ? der[0] : (q == 1) ? der[1] : der[DIM - 1];
    }
    if (MODE == LOOKUP_FORCES) {
      if (q < DIM) a.f[i * a.f_stride + q] = f_old - my_der;
      if (q == 3) e_acc += value;
    } else {
      if (q < DIM && a.f) a.f[i * DIM + q] = my_der;
      if (q == 3) {
        if (a.energy) a.energy[i] = value;
        e_acc += value;
      }
    }
  }
  double r = block_sum(e_acc, lds);
  if (threadIdx.x == 0) {
    if (a.partial_tag) store_partial_tagged(block_energy, bid, r, a.partial_tag); else block_energy[bid] = r;
  }
}
template <int DIM, int MODE, bool REPLICA = true>
__global__ void __launch_bounds__(BLOCK) k_lookup_quad(Geom g, const double *__restrict__ faces, LookupArgs a,
                                                       double *__restrict__ block_energy) {
  lookup_quad_body<DIM, MODE, REPLICA>(g, faces, a, block_energy, blockIdx.x, gridDim.x);
}

static int cu_count();
template <int DIM>
static hipError_t lookup_dim(const Geom &g, const double *rec, LookupMode mode, const LookupArgs &a,
                             double *scratch, double *energy_out, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                             int *blocks_out, const double *faces) {
  int blocks = (int)((a.n + BLOCK - 1) / BLOCK);
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  const bool use_quad = DIM > 1 && g.interp && g.rec == 4 && mode != LOOKUP_INDEX;
  if (use_quad) {   // four lanes per sample: on the lookup replica where the grid keeps one, else on the node records
    constexpr int QD = (DIM > 1) ? DIM : 2;
    long long qb = (a.n * 4 + BLOCK - 1) / BLOCK;
    if (qb > MAX_BLOCKS) qb = MAX_BLOCKS;
    {
      // as many workgroups as the device keeps resident of this kernel (the workgroups stride over the samples, so any
      // count is correct): a launch of 2048 on a device that keeps 1280 resident runs a full round and a 60 % one
      static int resident[2][2] = {{0, 0}, {0, 0}};
      int &res = resident[faces ? 1 : 0][mode == LOOKUP_FORCES ? 1 : 0];
      if (res == 0) {
        int per_cu = 0;
        hipError_t eo;
        if (faces)
          eo = mode == LOOKUP_FORCES ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lookup_quad<QD, LOOKUP_FORCES, true>, BLOCK, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lookup_quad<QD, LOOKUP_VALUES, true>, BLOCK, 0);
        else
          eo = mode == LOOKUP_FORCES ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lookup_quad<QD, LOOKUP_FORCES, false>, BLOCK, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lookup_quad<QD, LOOKUP_VALUES, false>, BLOCK, 0);
        if (eo != hipSuccess) {
          (void)hipGetLastError();
          per_cu = 0;
        }
        res = per_cu > 0 ? per_cu * cu_count() : -1;
      }
      // (one round -- as many workgroups as stay resident, striding -- measured 12.1 / 17.5 us against 12.7 / 18.4 us for
      //  2048 workgroups on the 2048^2 / 512^3 grids at 262 144 atoms; two or three rounds: no different from 2048)
      if (res > 0 && qb > (long long)res) qb = (long long)res;
    }
    blocks = (int)(qb < 1 ? 1 : qb);
    if (faces) {
      if (mode == LOOKUP_FORCES)
        EDM_LAUNCH_TIMED((k_lookup_quad<QD, LOOKUP_FORCES, true>), dim3(blocks), dim3(BLOCK), 0, s, ev0, ev1, g, faces, a, scratch);
      else
        EDM_LAUNCH_TIMED((k_lookup_quad<QD, LOOKUP_VALUES, true>), dim3(blocks), dim3(BLOCK), 0, s, ev0, ev1, g, faces, a, scratch);
    } else {
      if (mode == LOOKUP_FORCES)
        EDM_LAUNCH_TIMED((k_lookup_quad<QD, LOOKUP_FORCES, false>), dim3(blocks), dim3(BLOCK), 0, s, ev0, ev1, g, rec, a, scratch);
      else
        EDM_LAUNCH_TIMED((k_lookup_quad<QD, LOOKUP_VALUES, false>), dim3(blocks), dim3(BLOCK), 0, s, ev0, ev1, g, rec, a, scratch);
    }
  } else {
    switch (mode) {
      case LOOKUP_FORCES:
        EDM_LAUNCH_TIMED((k_lookup<DIM, LOOKUP_FORCES>), dim3(blocks), dim3(BLOCK), 0, s, ev0, ev1, g, rec, a, scratch);
        break;
      case LOOKUP_VALUES:
        EDM_LAUNCH_TIMED((k_lookup<DIM, LOOKUP_VALUES>), dim3(blocks), dim3(BLOCK), 0, s, ev0, ev1, g, rec, a, scratch);
        break;
      default:
        EDM_LAUNCH_TIMED((k_lookup<DIM, LOOKUP_INDEX>), dim3(blocks), dim3(BLOCK), 0, s, ev0, ev1, g, rec, a, scratch);
        break;
    }
  }
  if (blocks_out) *blocks_out = blocks;
  if (energy_out)
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(BLOCK), 0, s, scratch, (long long)blocks, energy_out);
  return hipGetLastError();
}

hipError_t launch_lookup(const Geom &g, const double *rec, LookupMode mode, const LookupArgs &a,
                         double *scratch, double *energy_out, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                         int *blocks_out, const double *faces) {
  switch (g.dim) {
    case 1: return lookup_dim<1>(g, rec, mode, a, scratch, energy_out, s, ev0, ev1, blocks_out, nullptr);
    case 2: return lookup_dim<2>(g, rec, mode, a, scratch, energy_out, s, ev0, ev1, blocks_out, faces);
    default: return lookup_dim<3>(g, rec, mode, a, scratch, energy_out, s, ev0, ev1, blocks_out, faces);
  }
}

// ---------------------------------------------------------------------------
// lookup replica ("faces"): see k_lookup_quad.  Built from the node records by k_build_faces (one thread per
// 32-byte slot: coalesced stores, gathered loads); kept current by the in-place tile-owned gather, whose
// workgroups store every node record they rewrite into the four blocks it appears in (face_store).
// ---------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_build_faces(Geom g, const double *__restrict__ rec, double *__restrict__ faces) {
  const long long stride = (long long)gridDim.x * BLOCK;
  const long long nslots = g.total * 4;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < nslots; i += stride) {
    const long long blk = i >> 2;
    const int slot = (int)(i & 3);
    long long rest = blk;
    const int b0 = (int)(rest % g.n[0]);
    rest /= g.n[0];
    const int b1 = (int)(rest % g.n[1]);
    const long long b2 = rest / g.n[1];   // (0 in 2-D)
    int q0 = b0 + (slot & 1), q1 = b1 + (slot >> 1);
    bool none = false;   // the +1 neighbour of the last node of a non-periodic dimension does not exist (never read)
    if (q0 >= g.n[0]) { if (g.periodic[0]) q0 -= g.n[0]; else none = true; }
    if (q1 >= g.n[1]) { if (g.periodic[1]) q1 -= g.n[1]; else none = true; }
    double4 v = make_double4(0, 0, 0, 0);
    if (!none) v = reinterpret_cast<const double4 *>(rec)[q0 + (long long)g.n[0] * (q1 + (long long)g.n[1] * b2)];
    reinterpret_cast<double4 *>(faces)[i] = v;
  }
}
hipError_t launch_build_faces(const Geom &g, const double *rec, double *faces, hipStream_t s) {
  if (g.dim < 2) return hipErrorInvalidValue;
  long long b = (g.total * 4 + BLOCK - 1) / BLOCK;
  if (b > 8 * MAX_BLOCKS) b = 8 * MAX_BLOCKS;
  if (b < 1) b = 1;
  if (g.dim == 2)
    hipLaunchKernelGGL(k_build_faces<2>, dim3((unsigned)b), dim3(BLOCK), 0, s, g, rec, faces);
  else
    hipLaunchKernelGGL(k_build_faces<3>, dim3((unsigned)b), dim3(BLOCK), 0, s, g, rec, faces);
  return hipGetLastError();
}
// the record of node p (V, dV/ds_0 .., pad) into the four blocks of the replica it belongs to
template <int DIM>
__device__ __forceinline__ void face_store(const Geom &g, double *__restrict__ faces, const int *p, const double *acc) {
  if (DIM < 2) return;
  const double4 v = make_double4(acc[0], acc[1], acc[DIM >= 2 ? 2 : 0], DIM == 3 ? acc[DIM == 3 ? 3 : 0] : 0.0);
#pragma unroll
  for (int slot = 0; slot < 4; slot++) {
    int b0 = p[0] - (slot & 1), b1 = p[DIM >= 2 ? 1 : 0] - (slot >> 1);
    if (b0 < 0) { if (!g.periodic[0]) continue; b0 += g.n[0]; }
    if (b1 < 0) { if (!g.periodic[1]) continue; b1 += g.n[1]; }
    const long long blk = b0 + (long long)g.n[0] * (b1 + (long long)g.n[1] * (DIM == 3 ? p[DIM == 3 ? 2 : 0] : 0));
    reinterpret_cast<double4 *>(faces)[blk * 4 + slot] = v;
  }
}

// 1-D pair-distance form (fix_edm_pair.cpp:215-217 batched): 16 B in / 16 B out
// per lane (two samples), grid-stride, energy reduced in-kernel.
__global__ void __launch_bounds__(BLOCK) k_pair_forces(Geom g, const double *__restrict__ rec, long long n,
                                                       const double *__restrict__ r, double *__restrict__ force,
                                                       double *__restrict__ block_energy) {
  __shared__ double lds[BLOCK / 64];
  double e_acc = 0;
  const long long npair = n >> 1;
  const long long stride = (long long)gridDim.x * BLOCK;
  const double2 *r2 = reinterpret_cast<const double2 *>(r);
  double2 *f2 = reinterpret_cast<double2 *>(force);
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < npair; i += stride) {
    const double2 rr = r2[i];
    double v0, v1, d0, d1;
    lookup_one<1>(g, rec, &rr.x, v0, &d0);
    lookup_one<1>(g, rec, &rr.y, v1, &d1);
    e_acc += v0;
    e_acc += v1;
    double2 out;
    out.x = 0.0 - d0;
    out.y = 0.0 - d1;
    f2[i] = out;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double v, d;
    lookup_one<1>(g, rec, &r[n - 1], v, &d);
    e_acc += v;
    force[n - 1] = 0.0 - d;
  }
  double s = block_sum(e_acc, lds);
  if (threadIdx.x == 0) block_energy[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------
// K1 fast path: 1-D, interpolating, non-periodic grid (the fix_edm_pair geometry).
//  * the node records of a window [w0, w0+W) are staged ONCE per workgroup into LDS
//    (up to 10224 nodes = 159.75 KiB of the CU's 160 KiB) by coalesced 16-B loads;
//    the two corners of a sample are then two ds_read_b128 instead of two L2 gathers.
//    Samples whose cell falls outside the window take the global-memory path.
//  * the cubic-Hermite blend is evaluated without divisions: f*C = f*A(X) + s*d*B(X)*dx
//    is algebraically what grid.h:113-123 computes with qq = -d/f (the |f| < 1e-7
//    special case is kept); X = where * (1/dx).  Values move by ~1e-16 relative.
//  * the node index stays BIT-EXACT: floor((x-min)*(1/dx)) is used only when it cannot
//    differ from the reference's floor((x-min)/dx); near an integer the division is done.
// ---------------------------------------------------------------------------
static constexpr int FAST_BLOCK = 1024;
static constexpr int LDS_WINDOW_MAX = 10224;  // nodes: 10224*16 B + 256 B of reduction scratch = 163840 B

typedef double v2d __attribute__((ext_vector_type(2)));  // native vector: usable with nontemporal builtins

__device__ __forceinline__ void hermite_1d(double fa, double da, double fb, double db, double X, double dx,
                                           double inv_dx, double &value, double &der) {
#pragma clang fp contract(fast)
  // corner a (bit 0, s=+1) sits at distance X, corner b (bit 1, s=-1) at distance Y = 1-X.  With
  // A(t) = 1-3t^2+2t^3, B(t) = t-2t^2+t^3 (grid.h:120-121) the two corners share everything:
  //   A(Y) = X^2(3-2X) = 1-A(X),  B(X) = X Y^2,  B(Y) = Y X^2,  A'(X) = A'(Y) = -6XY,
  //   B'(X) = Y(1-3X),  B'(Y) = X(3X-2)
  const double Y = 1.0 - X;
  const double X2 = X * X, Y2 = Y * Y;
  const double Ab = X2 * (3.0 - 2.0 * X);
  const double Aa = Y2 * (1.0 + 2.0 * X);
  const double Ba = X * Y2, Bb = Y * X2;
  const double Ap = -6.0 * X * Y;
  const double Bpa = Y * (1.0 - 3.0 * X), Bpb = X * (3.0 * X - 2.0);
  const double ga = (fabs(fa) < 0.0000001) ? 0.0 : da * dx;  // grid.h:113-116
  const double gb = (fabs(fb) < 0.0000001) ? 0.0 : db * dx;
  value = (fa * Aa + ga * Ba) + (fb * Ab - gb * Bb);
  der = ((fa - fb) * Ap + ga * Bpa + gb * Bpb) * inv_dx;
}

// The same cubic in monomial form, from the corner values and the SCALED corner slopes g = d * dx (zero where
// |f| < 1e-7, grid.h:113-116): p(X) = fa + X (ga + X (c2 + X c3)), c2 = 3 (fb - fa) - 2 ga - gb,
// c3 = 2 (fa - fb) + ga + gb.  14 fp64 operations instead of 25 -- K1 is bound by fp64 issue, not by HBM -- and
// the same polynomial to ~1e-16 of the largest coefficient (the reference's own form cancels as much).
__device__ __forceinline__ void hermite_1d_horner(double fa, double ga, double fb, double gb, double X, double inv_dx,
                                                  double &value, double &der) {
#pragma clang fp contract(fast)
  const double df = fb - fa;
  const double gs = ga + gb;
  const double c3 = gs - 2.0 * df;
  const double c2 = 3.0 * df - (gs + ga);
  value = fa + X * (ga + X * (c2 + X * c3));
  der = (ga + X * (2.0 * c2 + X * (3.0 * c3))) * inv_dx;
}
// scaled slope of a node record as the interpolation uses it
__device__ __forceinline__ double scaled_slope(double f, double d, double dx) {
  return (fabs(f) < 0.0000001) ? 0.0 : d * dx;
}

// The reference's own two-corner form (grid.h:110-123 with X = fabs(where/dx - x0)), used when the
// scaled coordinate falls outside [0,1] by rounding: there fabs() mirrors the two corners
// inconsistently and the shared-corner identities above do not describe what the reference computes.
__device__ __forceinline__ void hermite_1d_mirrored(double fa, double da, double fb, double db, double Xs, double dx,
                                                    double inv_dx, double &value, double &der) {
  const double X = fabs(Xs), Y = fabs(Xs - 1.0);
  const double X2 = X * X, X3 = X2 * X, Y2 = Y * Y, Y3 = Y2 * Y;
  const double Aa = 1 - 3 * X2 + 2 * X3, Ab = 1 - 3 * Y2 + 2 * Y3;
  const double Ba = X - 2 * X2 + X3, Bb = Y - 2 * Y2 + Y3;
  const double Apa = -6 * X + 6 * X2, Apb = -6 * Y + 6 * Y2;
  const double Bpa = 1 - 4 * X + 3 * X2, Bpb = 1 - 4 * Y + 3 * Y2;
  const double ga = (fabs(fa) < 0.0000001) ? 0.0 : da * dx;
  const double gb = (fabs(fb) < 0.0000001) ? 0.0 : db * dx;
  value = (fa * Aa + ga * Ba) + (fb * Ab - gb * Bb);
  der = ((fa * Apa + ga * Bpa) - (fb * Apb - gb * Bpb)) * inv_dx;
}

typedef __attribute__((address_space(3))) const v2d lds_v2d;

// Fast-path precondition (checked by the launcher): 1-D, interpolating, grid and boundary both
// non-periodic, so remap() is the identity and a sample outside [bmin,bmax] or outside the
// grid's in_grid range simply contributes (0, 0) (gaussian_grid.h:128-135, grid.h:398-409).
// "some lane of the wave": straight from the lane mask (the library's __any materialises the predicate in a VGPR)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

template <bool USE_LDS>
__device__ __forceinline__ void pair_one(const Geom &g, const double *__restrict__ rec, lds_v2d *win,
                                         int w0, int w1, double inv_dx, double x, double &v, double &d) {
  // Straight-line (predicated) evaluation so the compiler can interleave the independent lookups of
  // a lane; the two rare cases -- an index within rounding of an integer, a cell outside the LDS
  // window -- sit behind wave-uniform branches.
  // in_bounds (gaussian_grid.h:490-499) and in_grid (grid.h:865-874) of a non-periodic 1-D grid
  // (the four comparisons folded into two launch-uniform bounds: x >= max(bmin, min) and
  //  x < min(nextafter(bmax), max - dx) -- the closed upper end of the boundary becomes an open one)
  const double lo_ok = fmax(g.bmin[0], g.min[0]);
  const double hi_open = fmin(nextafter(g.bmax[0], 1.0e308), g.max[0] - g.dx[0]);
  const bool in_range = (x >= lo_ok) & (x < hi_open);
  const double q = (x - g.min[0]) * inv_dx;
  double fq = floor(q);
  // floor(q) can differ from the reference's floor((x-min)/dx) only within rounding of an integer: the guard
  // is |q - nearest integer| <= 1e-11 * max(1, q), tested here against its launch-uniform upper bound
  const double eps = 1e-11 * fmax(1.0, (double)g.n[0]);
  const double frac = q - fq;
  const bool near = in_range & ((frac <= eps) | (frac >= 1.0 - eps));
  if (wave_any(near)) {
    if (near) fq = floor((x - g.min[0]) / g.dx[0]);
  }
  int idx = (int)fq;
  idx = idx < 0 ? 0 : idx;
  idx = idx > g.n[0] - 2 ? g.n[0] - 2 : idx;
  // (for a sample in range fq IS the index -- the clamps only keep the addresses of masked lanes legal -- so the
  //  int -> double conversion of the reference's idx * dx is not needed)
  const double where = x - g.min[0] - fq * g.dx[0];
  const double X = where * inv_dx;
  // corner records as (f, g = scaled slope): the LDS window was staged in that form, global records are converted
  v2d a, b;
  if (USE_LDS) {
    const int li = idx - w0;
    const bool inw = (unsigned)li < (unsigned)(w1 - w0 - 1);   // 0 <= li and idx + 1 < w1, in one comparison
    const int lc = inw ? li : 0;
    a = win[lc];
    b = win[lc + 1];
    if (wave_any(in_range & !inw)) {
      if (!inw) {
        a = reinterpret_cast<const v2d *>(rec)[idx];
        b = reinterpret_cast<const v2d *>(rec)[idx + 1];
        a.y = scaled_slope(a.x, a.y, g.dx[0]);
        b.y = scaled_slope(b.x, b.y, g.dx[0]);
      }
    }
  } else {
    a = reinterpret_cast<const v2d *>(rec)[idx];
    b = reinterpret_cast<const v2d *>(rec)[idx + 1];
    a.y = scaled_slope(a.x, a.y, g.dx[0]);
    b.y = scaled_slope(b.x, b.y, g.dx[0]);
  }
  double vv, dd;
  hermite_1d_horner(a.x, a.y, b.x, b.y, X, inv_dx, vv, dd);
  const bool outside = in_range & ((X < 0.0) | (X > 1.0));  // `where` off by an ulp at a node
  if (wave_any(outside)) {
    if (outside) {   // (the reference's fabs() mirroring: its own form, from the original records)
      const v2d ra = reinterpret_cast<const v2d *>(rec)[idx], rb = reinterpret_cast<const v2d *>(rec)[idx + 1];
      hermite_1d_mirrored(ra.x, ra.y, rb.x, rb.y, X, g.dx[0], inv_dx, vv, dd);
    }
  }
  v = in_range ? vv : 0.0;
  d = in_range ? dd : 0.0;
}

// (body of K1 for workgroup `bid` of `nb`: shared by the plain launch and by the launch fused with the selection)
template <bool USE_LDS, int NT>
__device__ __forceinline__ void pair_forces_fast_body(const Geom &g, const double *__restrict__ rec, long long n,
                                                      const double *__restrict__ r, double *__restrict__ force,
                                                      double *__restrict__ block_energy, long long w0, int wn,
                                                      double inv_dx, double2 *lds_all, unsigned bid, unsigned nb,
                                                      unsigned long long tag = 0) {
  double *red = reinterpret_cast<double *>(lds_all);  // first 256 B: reduction scratch
  if (USE_LDS) {
    // the window is staged as (f, scaled slope): the |f| < 1e-7 test and the product with dx are paid once per
    // node here instead of once per sample
    const double2 *src = reinterpret_cast<const double2 *>(rec) + w0;
    for (int i = threadIdx.x; i < wn; i += NT) {
      double2 t = src[i];
      t.y = scaled_slope(t.x, t.y, g.dx[0]);
      lds_all[16 + i] = t;
    }
    __syncthreads();
  }
  lds_v2d *win = (lds_v2d *)(lds_all + 16);
  const int w1 = (int)w0 + wn;
  double e_acc = 0;
  const long long npair = n >> 1;
  const long long stride = (long long)nb * NT;
  const v2d *r2 = reinterpret_cast<const v2d *>(r);
  v2d *f2 = reinterpret_cast<v2d *>(force);
  // two independent 16-B loads per lane per iteration, requested ONE ITERATION AHEAD: the next pairs travel
  // while the current four lookups are evaluated (the loop is latency-bound at four waves per SIMD)
  const long long i0 = (long long)bid * NT + threadIdx.x;
  v2d ra = {0.0, 0.0}, rb = {0.0, 0.0};
  if (i0 < npair) ra = __builtin_nontemporal_load(&r2[i0]);
  if (i0 + stride < npair) rb = __builtin_nontemporal_load(&r2[i0 + stride]);
  for (long long i = i0; i < npair; i += 2 * stride) {
    const long long j = i + stride;
    const bool has_b = j < npair;
    const long long in = i + 2 * stride, jn = in + stride;
    v2d na = {0.0, 0.0}, nb = {0.0, 0.0};
    if (in < npair) na = __builtin_nontemporal_load(&r2[in]);
    if (jn < npair) nb = __builtin_nontemporal_load(&r2[jn]);
    if (has_b) {  // steady state: four independent lookups in flight per lane
      double v0, v1, v2, v3, d0, d1, d2, d3;
      pair_one<USE_LDS>(g, rec, win, (int)w0, w1, inv_dx, ra.x, v0, d0);
      pair_one<USE_LDS>(g, rec, win, (int)w0, w1, inv_dx, ra.y, v1, d1);
      pair_one<USE_LDS>(g, rec, win, (int)w0, w1, inv_dx, rb.x, v2, d2);
      pair_one<USE_LDS>(g, rec, win, (int)w0, w1, inv_dx, rb.y, v3, d3);
      e_acc += v0;
      e_acc += v1;
      e_acc += v2;
      e_acc += v3;
      v2d oa, ob;
      oa.x = 0.0 - d0;
      oa.y = 0.0 - d1;
      ob.x = 0.0 - d2;
      ob.y = 0.0 - d3;
      __builtin_nontemporal_store(oa, &f2[i]);
      __builtin_nontemporal_store(ob, &f2[j]);
    } else {
      double v0, v1, d0, d1;
      pair_one<USE_LDS>(g, rec, win, (int)w0, w1, inv_dx, ra.x, v0, d0);
      pair_one<USE_LDS>(g, rec, win, (int)w0, w1, inv_dx, ra.y, v1, d1);
      e_acc += v0;
      e_acc += v1;
      v2d oa;
      oa.x = 0.0 - d0;
      oa.y = 0.0 - d1;
      __builtin_nontemporal_store(oa, &f2[i]);
    }
    ra = na;
    rb = nb;
  }
  if ((n & 1) && bid == 0 && threadIdx.x == 0) {
    double v, d;
    pair_one<USE_LDS>(g, rec, win, (int)w0, w1, inv_dx, r[n - 1], v, d);
    e_acc += v;
    force[n - 1] = 0.0 - d;
  }
  double s = block_sum(e_acc, red);
  if (threadIdx.x == 0) {
    // (untagged: a system-scope store too -- written through to the host-mapped array the host adds the sums up from)
    if (tag) store_partial_tagged(block_energy, bid, s, tag);
    else __hip_atomic_store(&block_energy[bid], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <bool USE_LDS, int NT>
__global__ void __launch_bounds__(NT) k_pair_forces_fast(Geom g, const double *__restrict__ rec, long long n,
                                                                 const double *__restrict__ r,
                                                                 double *__restrict__ force,
                                                                 double *__restrict__ block_energy, long long w0,
                                                                 int wn, double inv_dx, unsigned long long tag) {
  extern __shared__ double2 lds_all[];
  pair_forces_fast_body<USE_LDS, NT>(g, rec, n, r, force, block_energy, w0, wn, inv_dx, lds_all, blockIdx.x, gridDim.x, tag);
}

// ---------------------------------------------------------------------------
// fix edm_pair over a device-resident neighbour list (see PairListArgs)
// ---------------------------------------------------------------------------
// distance of list entry (i, j) exactly as every user computes it (the force pass, the sample positions)
__device__ __forceinline__ double pairlist_distance(const double *__restrict__ x, int i, int j, double &delx, double &dely,
                                                    double &delz) {
  delx = x[3 * (long long)i] - x[3 * (long long)j];
  dely = x[3 * (long long)i + 1] - x[3 * (long long)j + 1];
  delz = x[3 * (long long)i + 2] - x[3 * (long long)j + 2];
  return sqrt(delx * delx + dely * dely + delz * delz);
}
// the type filter of fix_edm_pair.cpp:181-202: i of type ipair pairs with j of type jpair, and vice versa
__device__ __forceinline__ bool pairlist_types_match(const PairListArgs &a, int ti, int tj) {
  if (ti == a.itype) return tj == a.jtype;
  if (ti == a.jtype) return tj == a.itype;
  return false;
}
// One side of an atom's sum: its entries q = beg + sub, beg + sub + 16, ... with partner[q] the other atom; `as_i`:
// the atom is the entry's i (the term is added and its energy counted), else its j (subtracted).  Four
// entries per trip: their partner, type and position loads are requested together (an entry is a chain of
// three dependent loads -- partner -> position -> grid record -- and a lane walks a handful of entries), the
// terms are still added in list order.
// ORD (default: none): the reference-order pass -- ord->lookup(r, entry, v, d) reads the bias as it stood when the
// reference's loop reached list entry `entry` (entry_of[q]: the list entry behind slot q of the per-atom lists)
struct NoOrder {};
template <bool FAST, class ORD = NoOrder>
__device__ __forceinline__ void pairlist_side(const Geom &g, const double *__restrict__ rec, const PairListArgs &a,
                                              double inv_dx, int atom, int ta, const double *xa, bool as_i,
                                              const int *__restrict__ partner, long long beg, long long end, int sub,
                                              double &fx, double &fy, double &fz, double &e_acc,
                                              const ORD *ord = nullptr, const int *__restrict__ entry_of = nullptr) {
  constexpr int ILP = 4;
  for (long long q0 = beg + sub; q0 < end; q0 += 16 * ILP) {
    int other[ILP];
    bool ok[ILP];
    double xo[ILP][3];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
      const long long q = q0 + 16 * u;
      ok[u] = q < end;
      other[u] = ok[u] ? partner[q] : 0;   // (contiguous per atom: built when the list was uploaded)
    }
#pragma unroll
    for (int u = 0; u < ILP; u++) {
      const int to = ok[u] ? a.type[other[u]] : 0;
      if (ok[u]) {
        xo[u][0] = a.x[3 * (long long)other[u]];
        xo[u][1] = a.x[3 * (long long)other[u] + 1];
        xo[u][2] = a.x[3 * (long long)other[u] + 2];
      } else {
        xo[u][0] = xo[u][1] = xo[u][2] = 0;
      }
      ok[u] = ok[u] && (as_i ? pairlist_types_match(a, ta, to) : pairlist_types_match(a, to, ta));
    }
    double px[ILP], py[ILP], pz[ILP], pv[ILP];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
      px[u] = py[u] = pz[u] = pv[u] = 0;
      if (ok[u]) {
        // del = x_i - x_j of the ENTRY (fix_edm_pair.cpp:206-213), whichever end this atom is
        double delx = as_i ? xa[0] - xo[u][0] : xo[u][0] - xa[0];
        double dely = as_i ? xa[1] - xo[u][1] : xo[u][1] - xa[1];
        double delz = as_i ? xa[2] - xo[u][2] : xo[u][2] - xa[2];
        double r = sqrt(delx * delx + dely * dely + delz * delz);
        const double rinv = 1.0 / r;
        delx *= rinv;
        dely *= rinv;
        delz *= rinv;
        double v, d;
        if constexpr (!std::is_same<ORD, NoOrder>::value)
          ord->lookup(r, entry_of[q0 + 16 * u], v, d);
        else if (FAST)
          pair_one<false>(g, rec, nullptr, 0, 0, inv_dx, r, v, d);
        else
          lookup_one<1>(g, rec, &r, v, &d);
        const double fr = 0.0 - d;   // update_force on a zeroed accumulator (fix_edm_pair.cpp:215-217)
        px[u] = delx * fr;
        py[u] = dely * fr;
        pz[u] = delz * fr;
        pv[u] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < ILP; u++) {
      if (ok[u]) {
        if (as_i) {
          fx += px[u];
          fy += py[u];
          fz += pz[u];
          e_acc += pv[u];
        } else {
          fx -= px[u];
          fy -= py[u];
          fz -= pz[u];
        }
      }
    }
  }
}
// (body of the pass for workgroup `bid` of `nblk`: shared by the plain launch and by the launch that also carries the
//  step's selection)
template <bool FAST, class ORD = NoOrder>
__device__ __forceinline__ void pairlist_forces_body(const Geom &g, const double *__restrict__ rec, const PairListArgs &a,
                                                     double *__restrict__ partials, double inv_dx, unsigned bid,
                                                     unsigned nblk, const ORD *ord = nullptr) {
  __shared__ double lds[BLOCK / 64];
  const int sub = threadIdx.x & 15;
  double e_acc = 0;
  const long long astride = ((long long)nblk * BLOCK) >> 4;
  for (long long atom = ((long long)bid * BLOCK + threadIdx.x) >> 4; atom < a.nall; atom += astride) {
    double fx = 0, fy = 0, fz = 0;
    if (atom < a.nlocal) {   // (ghost atoms never appear as i and receive nothing as j: newton off)
      const int ta = a.type[atom];
      const double xa[3] = {a.x[3 * atom], a.x[3 * atom + 1], a.x[3 * atom + 2]};
      const long long ib = a.it_off[atom], ie = a.it_off[atom + 1], jb = a.jt_off[atom], je = a.jt_off[atom + 1];
      pairlist_side<FAST, ORD>(g, rec, a, inv_dx, (int)atom, ta, xa, true, a.it_partner, ib, ie, sub, fx, fy, fz, e_acc, ord, a.it_entry);
      pairlist_side<FAST, ORD>(g, rec, a, inv_dx, (int)atom, ta, xa, false, a.jt_partner, jb, je, sub, fx, fy, fz, e_acc, ord, a.jt_entry);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {   // fixed tree within the 16-lane group
      fx += __shfl_xor(fx, o, 64);
      fy += __shfl_xor(fy, o, 64);
      fz += __shfl_xor(fz, o, 64);
    }
    if (sub == 0) {
      a.fdelta[3 * atom] = fx;
      a.fdelta[3 * atom + 1] = fy;
      a.fdelta[3 * atom + 2] = fz;
    }
  }
  const double se = block_sum(e_acc, lds);
  if (threadIdx.x == 0) {
    if (a.partial_tag) store_partial_tagged(partials, bid, se, a.partial_tag); else partials[bid] = se;
  }
}
template <bool FAST>
__global__ void __launch_bounds__(BLOCK) k_pairlist_forces(Geom g, const double *__restrict__ rec, PairListArgs a,
                                                           double *__restrict__ partials, double inv_dx) {
  pairlist_forces_body<FAST>(g, rec, a, partials, inv_dx, blockIdx.x, gridDim.x);
}

// which virtual samples of the list are live (static between list uploads) + their number
__global__ void __launch_bounds__(BLOCK) k_pairlist_mask(PairListArgs a, double *__restrict__ partials) {
  __shared__ double lds[BLOCK / 64];
  double calls = 0;
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long p = (long long)blockIdx.x * BLOCK + threadIdx.x; p < a.npairs; p += stride) {
    const int i = a.pair_i[p], j = a.pair_j[p];
    const bool ok = pairlist_types_match(a, a.type[i], a.type[j]);
    const bool second = ok && (j < a.nlocal);
    a.vs_mask[2 * p] = ok ? 1 : 0;
    a.vs_mask[2 * p + 1] = second ? 1 : 0;
    calls += (ok ? 1.0 : 0.0) + (second ? 1.0 : 0.0);
  }
  const double sc = block_sum(calls, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = sc;
}
__global__ void __launch_bounds__(BLOCK) k_pairlist_samples(PairListArgs a) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long p = (long long)blockIdx.x * BLOCK + threadIdx.x; p < a.npairs; p += stride) {
    double dx, dy, dz;
    const double r = pairlist_distance(a.x, a.pair_i[p], a.pair_j[p], dx, dy, dz);
    a.vs_r[2 * p] = r;
    a.vs_r[2 * p + 1] = r;
  }
}

hipError_t launch_pairlist_forces(const Geom &g, const double *rec, const PairListArgs &a, double *partials,
                                  hipStream_t s, int *blocks_out) {
  if (g.dim != 1) return hipErrorInvalidValue;
  if (blocks_out) *blocks_out = 0;
  if (a.nall <= 0) return hipSuccess;
  const long long threads = (long long)a.nall * 16;
  long long nb = (threads + BLOCK - 1) / BLOCK;
  if (nb > MAX_BLOCKS) nb = MAX_BLOCKS;   // (one energy partial per workgroup; the atoms are strided over)
  const bool fast = (g.interp && !g.periodic[0] && !g.bper[0] && g.n[0] >= 2);
  const double inv_dx = 1.0 / g.dx[0];
  if (fast)
    hipLaunchKernelGGL(k_pairlist_forces<true>, dim3((unsigned)nb), dim3(BLOCK), 0, s, g, rec, a, partials, inv_dx);
  else
    hipLaunchKernelGGL(k_pairlist_forces<false>, dim3((unsigned)nb), dim3(BLOCK), 0, s, g, rec, a, partials, inv_dx);
  if (blocks_out) *blocks_out = (int)nb;
  return hipGetLastError();
}
hipError_t launch_pairlist_mask(const PairListArgs &a, double *partials, hipStream_t s, int *blocks_out) {
  long long nb = (a.npairs + BLOCK - 1) / BLOCK;
  if (nb > EDM_PAIRLIST_MAX_BLOCKS) nb = EDM_PAIRLIST_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_pairlist_mask, dim3((unsigned)nb), dim3(BLOCK), 0, s, a, partials);
  if (blocks_out) *blocks_out = (int)nb;
  return hipGetLastError();
}
hipError_t launch_pairlist_samples(const PairListArgs &a, hipStream_t s) {
  long long nb = (a.npairs + BLOCK - 1) / BLOCK;
  if (nb > EDM_PAIRLIST_MAX_BLOCKS) nb = EDM_PAIRLIST_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_pairlist_samples, dim3((unsigned)nb), dim3(BLOCK), 0, s, a);
  return hipGetLastError();
}

static int cu_count() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}
// geometry served by the specialised K1 (the pair-distance CV: 1-D, interpolating, nothing periodic)
static bool pair_fast_path(const Geom &g) {
  return g.dim == 1 && g.interp && !g.periodic[0] && !g.bper[0] && g.n[0] >= 2;
}
// LDS staging costs ~160 KB of L2 reads per workgroup: worth it only for long sample arrays
// (measured break-even ~1 M pairs: 2 M pairs 12.3 us staged vs 14.7 us from L2, 4 M pairs 20 vs 31 us)
static constexpr long long PAIR_LDS_THRESHOLD = 1500000;
// short arrays: two workgroups per CU, each lane looping over groups of four pairs (measured faster than one
// pass of eight workgroups per CU -- 52 vs 45 G evals/s at 1 M pairs -- fewer waves to launch and retire)
static int pair_short_blocks(long long n) {
  const long long work = (n >> 1) + 1;
  long long blocks = (work + BLOCK - 1) / BLOCK;
  if (blocks > 2 * cu_count()) blocks = 2 * cu_count();
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

hipError_t launch_pair_forces(const Geom &g, const double *rec, long long n, const double *r, double *force,
                              double *scratch, double *energy_out, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                              int *blocks_out, unsigned long long tag, int *tagged_out) {
  if (tagged_out) *tagged_out = 0;
  int blocks;
  const bool fast = pair_fast_path(g);
  if (fast) {
    static bool attr_set = false;
    const int n_cu = cu_count();
    const double inv_dx = 1.0 / g.dx[0];
    const bool use_lds = n >= PAIR_LDS_THRESHOLD;
    if (use_lds) {
      int wn = g.n[0] < LDS_WINDOW_MAX ? g.n[0] : LDS_WINDOW_MAX;
      long long w0 = (long long)g.n[0] - wn;  // top-aligned: pair distances populate the upper range
      const size_t lds_bytes = 256 + (size_t)wn * 16;
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pair_forces_fast<true, FAST_BLOCK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 256 + LDS_WINDOW_MAX * 16);
        if (e != hipSuccess) return e;
        attr_set = true;
      }
      blocks = n_cu;
      EDM_LAUNCH_TIMED((k_pair_forces_fast<true, FAST_BLOCK>), dim3(blocks), dim3(FAST_BLOCK), lds_bytes, s, ev0, ev1, g, rec, n,
                       r, force, scratch, w0, wn, inv_dx, tag);
      if (tagged_out && tag) *tagged_out = 1;
    } else {
      // short arrays: small workgroups spread over every CU (latency-bound regime)
      blocks = pair_short_blocks(n);
      EDM_LAUNCH_TIMED((k_pair_forces_fast<false, BLOCK>), dim3(blocks), dim3(BLOCK), 256, s, ev0, ev1, g, rec, n, r, force,
                       scratch, 0LL, 0, inv_dx, tag);
      if (tagged_out && tag) *tagged_out = 1;
    }
  } else {
    long long work = (n >> 1) + 1;
    blocks = (int)((work + BLOCK - 1) / BLOCK);
    if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    EDM_LAUNCH_TIMED(k_pair_forces, dim3(blocks), dim3(BLOCK), 0, s, ev0, ev1, g, rec, n, r, force, scratch);
  }
  if (blocks_out) *blocks_out = blocks;
  if (energy_out)
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(BLOCK), 0, s, scratch, (long long)blocks, energy_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// record layout conversion (host SoA values[]/derivs[][dim]  <->  device AoS)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_pack(Geom g, double *__restrict__ rec, const double *__restrict__ values,
                                                const double *__restrict__ derivs) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < g.total; i += stride) {
    rec[i * g.rec] = values[i];
    for (int d = 0; d < g.rec - 1; d++)
      rec[i * g.rec + 1 + d] = (d < g.dim && derivs) ? derivs[i * g.dim + d] : 0.0;
  }
}
__global__ void __launch_bounds__(BLOCK) k_unpack(Geom g, const double *__restrict__ rec, double *__restrict__ values,
                                                  double *__restrict__ derivs) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < g.total; i += stride) {
    if (values) values[i] = rec[i * g.rec];
    if (derivs)
      for (int d = 0; d < g.dim; d++) derivs[i * g.dim + d] = rec[i * g.rec + 1 + d];
  }
}
static int blocks_for(long long n) {
  long long b = (n + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}
hipError_t launch_pack(const Geom &g, double *rec, const double *values, const double *derivs, hipStream_t s) {
  hipLaunchKernelGGL(k_pack, dim3(blocks_for(g.total)), dim3(BLOCK), 0, s, g, rec, values, derivs);
  return hipGetLastError();
}
hipError_t launch_unpack(const Geom &g, const double *rec, double *values, double *derivs, hipStream_t s) {
  hipLaunchKernelGGL(k_unpack, dim3(blocks_for(g.total)), dim3(BLOCK), 0, s, g, rec, values, derivs);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K7: histogram  (DimmedGrid::add_value grid.h:370-385 on cv_hist_, edm_bias.cpp:601-610)
// Bin contents are integer-valued doubles, so the atomic order cannot change bits.
// ---------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_hist_add(Geom g, double *__restrict__ values, long long n,
                                                    const double *__restrict__ x, int x_stride,
                                                    const long long *__restrict__ sel, const double *__restrict__ w,
                                                    double w_const) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const double wi = w ? w[i] : w_const;
    if (wi == 0.0) continue;
    const long long src = sel ? sel[i] : i;
    double xx[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) xx[d] = x[src * x_stride + d];
    if (!in_grid<DIM>(g, xx)) continue;
    long long idx[DIM];
    bool ok = true;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      double wr;
      idx[d] = node_index(g, d, xx[d], &wr);
      if (idx[d] < 0 || idx[d] >= g.n[d]) ok = false;  // reference: out-of-array write
    }
    if (!ok) continue;
    long long flat = idx[DIM - 1];
#pragma unroll
    for (int d = DIM - 1; d > 0; d--) flat = flat * g.n[d - 1] + idx[d - 1];
    atomicAdd(&values[flat * g.rec], wi);   // (rec == 1 for a histogram; a grid with derivative records keeps V first)
  }
}
hipError_t launch_hist_add(const Geom &g, double *values, long long n, const double *x, int x_stride,
                           const long long *sel, const double *w, double w_const, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int b = blocks_for(n);
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_hist_add<1>, dim3(b), dim3(BLOCK), 0, s, g, values, n, x, x_stride, sel, w, w_const); break;
    case 2: hipLaunchKernelGGL(k_hist_add<2>, dim3(b), dim3(BLOCK), 0, s, g, values, n, x, x_stride, sel, w, w_const); break;
    default: hipLaunchKernelGGL(k_hist_add<3>, dim3(b), dim3(BLOCK), 0, s, g, values, n, x, x_stride, sel, w, w_const); break;
  }
  return hipGetLastError();
}

// forward declaration target: LimitResult is defined in edm_kernels.h
template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_hist_tail(Geom g, double *__restrict__ values,
                                                     const LimitResult *__restrict__ res,
                                                     const int *__restrict__ flags, const double *__restrict__ hx0,
                                                     int plus_for_applied) {
  if (res->error) return;
  const int ntail = res->n_tail;
  const long long k = res->k;
  for (int j = threadIdx.x; j < ntail; j += BLOCK) {
    double wgt = 0;
    if (plus_for_applied && (flags[j] & 1)) wgt += 1.0;
    if (flags[j] & 2) wgt -= 1.0;
    if (wgt == 0.0) continue;
    double xx[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) xx[d] = hx0[(k + j) * DIM + d];
    if (!in_grid<DIM>(g, xx)) continue;
    long long idx[DIM];
    bool ok = true;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      double wr;
      idx[d] = node_index(g, d, xx[d], &wr);
      if (idx[d] < 0 || idx[d] >= g.n[d]) ok = false;
    }
    if (!ok) continue;
    long long flat = idx[DIM - 1];
#pragma unroll
    for (int d = DIM - 1; d > 0; d--) flat = flat * g.n[d - 1] + idx[d - 1];
    atomicAdd(&values[flat], wgt);
  }
}
hipError_t launch_hist_tail(const Geom &g, double *values, const LimitResult *res_dev, const int *flags,
                            const double *hx0, int plus_for_applied, hipStream_t s) {
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_hist_tail<1>, dim3(1), dim3(BLOCK), 0, s, g, values, res_dev, flags, hx0, plus_for_applied); break;
    case 2: hipLaunchKernelGGL(k_hist_tail<2>, dim3(1), dim3(BLOCK), 0, s, g, values, res_dev, flags, hx0, plus_for_applied); break;
    default: hipLaunchKernelGGL(k_hist_tail<3>, dim3(1), dim3(BLOCK), 0, s, g, values, res_dev, flags, hx0, plus_for_applied); break;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// order-preserving selection of accepted samples (edm_bias.cpp:406, :543)
// ---------------------------------------------------------------------------
static constexpr int SEL_PER_THREAD = 8;
static constexpr int SEL_CHUNK = BLOCK * SEL_PER_THREAD;

// uniform number i (0-based) of the device stream `rng`: output i + 1 of SplitMix64 started at `rng`, top 53
// bits -> [0, 1)  (edm_amd.workloads.uniform is the host twin)
__device__ __forceinline__ double device_uniform(unsigned long long rng, long long i) {
  unsigned long long z = rng + (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
// ru == NULL with use_thr: the uniforms come from the device stream `rng` (no array to generate, upload or read)
__device__ __forceinline__ bool sel_flag(long long i, long long n, const double *__restrict__ ru, double thr,
                                         int use_thr, const int *__restrict__ mask, int apply_mask,
                                         unsigned long long rng) {
  if (i >= n) return false;
  if (!(apply_mask < 0 || (apply_mask & mask[i]))) return false;
  if (use_thr && !((ru ? ru[i] : device_uniform(rng, i)) < thr)) return false;
  return true;
}

__global__ void __launch_bounds__(BLOCK) k_sel_count(long long n, const double *__restrict__ ru, double thr,
                                                     int use_thr, const int *__restrict__ mask, int apply_mask,
                                                     int *__restrict__ counts, unsigned long long rng) {
  __shared__ int lds[BLOCK / 64];
  const long long base = (long long)blockIdx.x * SEL_CHUNK + (long long)threadIdx.x * SEL_PER_THREAD;
  int c = 0;
#pragma unroll
  for (int j = 0; j < SEL_PER_THREAD; j++) c += sel_flag(base + j, n, ru, thr, use_thr, mask, apply_mask, rng) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < BLOCK / 64; w++) t += lds[w];
    counts[blockIdx.x] = t;
  }
}

// exclusive scan of the per-block counts by one block (counts -> offsets, total): 256 counts per
// pass, wave prefix sums by shuffles + a 4-entry cross-wave fix-up, running carry between passes
__global__ void __launch_bounds__(BLOCK) k_sel_scan(int nblocks, const int *__restrict__ counts,
                                                    long long *__restrict__ offsets, long long *__restrict__ total,
                                                    long long *__restrict__ total2) {
  __shared__ long long wsum[BLOCK / 64];
  __shared__ long long carry_sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_sh = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += BLOCK) {
    const int i = base + threadIdx.x;
    const long long c = (i < nblocks) ? counts[i] : 0;
    long long inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    long long before = carry_sh;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (i < nblocks) offsets[i] = before + inc - c;
    __syncthreads();
    if (threadIdx.x == BLOCK - 1) carry_sh = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *total = carry_sh;
    if (total2) *total2 = carry_sh;
  }
}

__global__ void __launch_bounds__(BLOCK) k_sel_scatter(long long n, const double *__restrict__ ru, double thr,
                                                       int use_thr, const int *__restrict__ mask, int apply_mask,
                                                       const long long *__restrict__ offsets,
                                                       long long *__restrict__ sel, unsigned long long rng) {
  __shared__ int lds[BLOCK];
  const long long base = (long long)blockIdx.x * SEL_CHUNK + (long long)threadIdx.x * SEL_PER_THREAD;
  bool fl[SEL_PER_THREAD];
  int c = 0;
#pragma unroll
  for (int j = 0; j < SEL_PER_THREAD; j++) {
    fl[j] = sel_flag(base + j, n, ru, thr, use_thr, mask, apply_mask, rng);
    c += fl[j] ? 1 : 0;
  }
  lds[threadIdx.x] = c;
  __syncthreads();
  // exclusive scan over the 256 thread counts (Hillis-Steele in LDS)
  for (int o = 1; o < BLOCK; o <<= 1) {
    int v = (threadIdx.x >= o) ? lds[threadIdx.x - o] : 0;
    __syncthreads();
    lds[threadIdx.x] += v;
    __syncthreads();
  }
  long long pos = offsets[blockIdx.x] + (lds[threadIdx.x] - c);
#pragma unroll
  for (int j = 0; j < SEL_PER_THREAD; j++)
    if (fl[j]) sel[pos++] = base + j;
}

size_t select_scratch_ints(long long n) {
  const long long nb = (n + SEL_CHUNK - 1) / SEL_CHUNK;
  return (size_t)(nb + 2) * 3;  // counts (int) + offsets (long long) in int units
}

hipError_t launch_select(long long n, const double *ru, double thr, int use_thr, const int *mask, int apply_mask,
                         long long *sel, long long *count, int *scratch, hipStream_t s, long long *count2,
                         unsigned long long rng) {
  const int nb = (int)((n + SEL_CHUNK - 1) / SEL_CHUNK);
  if (nb == 0) {
    if (count2) {
      hipError_t e = hipMemsetAsync(count2, 0, sizeof(long long), s);
      if (e != hipSuccess) return e;
    }
    return hipMemsetAsync(count, 0, sizeof(long long), s);
  }
  int *counts = scratch;
  long long *offsets = reinterpret_cast<long long *>(scratch + ((nb + 2) & ~1));
  hipLaunchKernelGGL(k_sel_count, dim3(nb), dim3(BLOCK), 0, s, n, ru, thr, use_thr, mask, apply_mask, counts, rng);
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(BLOCK), 0, s, nb, counts, offsets, count, count2);
  hipLaunchKernelGGL(k_sel_scatter, dim3(nb), dim3(BLOCK), 0, s, n, ru, thr, use_thr, mask, apply_mask, offsets, sel, rng);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// hill preparation: remap, rejection, centre node, hill-only exponentials
// (gaussian_grid.h:206-224 and the temp1/temp3 terms of :310,:312)
// ---------------------------------------------------------------------------
template <int DIM>
__device__ __forceinline__ void hill_prep_vals(const Geom &g, const HillList &h, long long i, double *x);
// position of sample `src`: a row of h.x, or (1-D, h.pl_x) the pair distance of neighbour-list entry src >> 1
template <int DIM>
__device__ __forceinline__ void sample_position(const HillList &h, long long src, double *x) {
  if (DIM == 1 && h.pl_x) {
    const long long e = src >> 1;
    double dx, dy, dz;
    x[0] = pairlist_distance(h.pl_x, h.pl_i[e], h.pl_j[e], dx, dy, dz);
    return;
  }
#pragma unroll
  for (int d = 0; d < DIM; d++) x[d] = h.x[src * h.x_stride + d];
}
template <int DIM>
__device__ __forceinline__ void hill_prep_one(const Geom &g, const HillList &h, long long i, long long src) {
  double x[DIM];
  sample_position<DIM>(h, src, x);
  hill_prep_vals<DIM>(g, h, i, x);
}
// the prepared fields of one hill from its CV: x is remapped in place (gaussian_grid.h:206-224), c = centre node
// (c[0] = INT_MIN: rejected, outside a wall), ht = the hill-only exponentials (t1, t3) of :310,:312 per dimension
template <int DIM>
__device__ __forceinline__ void hill_prep_compute(const Geom &g, double *x, int *c, double *ht) {
  remap<DIM>(g, x);
  bool ok = true;
#pragma unroll
  for (int d = 0; d < DIM; d++)
    if (!g.bper[d] && (x[d] < g.bmin[d] || x[d] > g.bmax[d])) ok = false;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    c[d] = ifloor((x[d] - g.min[d]) / g.dx[d]);
    double t1 = 0, t3 = 0;
    if (!g.bper[d]) {
      const double sg = g.sigma[d];
      t1 = exp(-((x[d] - g.bmin[d]) * (x[d] - g.bmin[d])) / (sg * sg));
      t3 = exp(-((x[d] - g.bmax[d]) * (x[d] - g.bmax[d])) / (sg * sg));
    }
    ht[2 * d] = t1;
    ht[2 * d + 1] = t3;
  }
  if (!ok) c[0] = INT_MIN;
}
// (x holds the sample's CV on entry and is remapped in place)
template <int DIM>
__device__ __forceinline__ void hill_prep_vals(const Geom &g, const HillList &h, long long i, double *x) {
  if (h.hx0) {
#pragma unroll
    for (int d = 0; d < DIM; d++) h.hx0[i * DIM + d] = x[d];
  }
  int c[DIM];
  double ht[2 * DIM];
  hill_prep_compute<DIM>(g, x, c, ht);
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    h.hx[i * DIM + d] = x[d];
    h.hc[i * DIM + d] = c[d];
    h.ht[i * 2 * DIM + 2 * d] = ht[2 * d];
    h.ht[i * 2 * DIM + 2 * d + 1] = ht[2 * d + 1];
  }
}

// (body for workgroup `bid` of `nb`)
template <int DIM>
__device__ __forceinline__ void hill_prep_body(const Geom &g, const HillList &h, const double *__restrict__ fetch_src,
                                               double *__restrict__ fetch_dst, unsigned bid, unsigned nb) {
  const long long stride = (long long)nb * BLOCK;
  const long long nh = hill_count(h);
  for (long long i = (long long)bid * BLOCK + threadIdx.x; i < nh; i += stride) {
    if (fetch_dst) fetch_dst[i] = fetch_src[i];
    hill_prep_one<DIM>(g, h, i, h.sel ? h.sel[i] : i);
  }
}
template <int DIM>
// (fetch_src/fetch_dst, optional: the per-hill heights of an overflow flush sit in host-mapped memory next to the
//  positions; this kernel brings them over while it prepares the hills -- no separate upload)
__global__ void __launch_bounds__(BLOCK) k_hill_prep(Geom g, HillList h, const double *__restrict__ fetch_src,
                                                     double *__restrict__ fetch_dst) {
  hill_prep_body<DIM>(g, h, fetch_src, fetch_dst, blockIdx.x, gridDim.x);
}
// fix edm step on a 2-D / 3-D grid: the force kernel (K2 on four lanes per atom) and the preparation of the overflow
// flush's hill list in ONE launch -- workgroups [0, nb_prep) prepare (they read positions and heights from host-mapped
// memory: dispatched first, done long before the lookups), the rest evaluate forces.  Neither touches what the other
// reads or writes (the flush's gather, which writes the grid, is a later launch): one launch and its 5 us less per step.
template <int DIM, bool REPLICA>
__global__ void __launch_bounds__(BLOCK) k_lookup_quad_prep(Geom g, const double *__restrict__ faces, LookupArgs a,
                                                            double *__restrict__ block_energy, HillList h,
                                                            const double *__restrict__ fetch_src,
                                                            double *__restrict__ fetch_dst, unsigned nb_prep) {
  if (blockIdx.x < nb_prep)
    hill_prep_body<DIM>(g, h, fetch_src, fetch_dst, blockIdx.x, nb_prep);
  else
    lookup_quad_body<DIM, LOOKUP_FORCES, REPLICA>(g, faces, a, block_energy, blockIdx.x - nb_prep, gridDim.x - nb_prep);
}

// ---------------------------------------------------------------------------
// "Last workgroup done" chaining.  A short hill step is a chain of tiny dependent kernels, each
// costing a launch; instead every workgroup of a stage publishes its results (agent-scope relaxed atomic stores,
// see publish() below -- NOT a fence, which would write back the XCD's whole L2),
// takes a ticket, and the workgroup that draws the last ticket runs the next (small, serial) stage
// in the same launch.  The ticket is reset by that workgroup, so the counters stay zero between
// launches.  Returns true (block-uniform) in the last workgroup, with the other workgroups' writes
// visible.
// ---------------------------------------------------------------------------
// (A ticket is EDM_TICKET_INTS ints: a top counter and EDM_TICKET_FAN sub-counters, each on its own
// 128-byte line.  Hundreds of workgroups incrementing ONE address serialise at ~12 ns per atomic -- 6 us
// for the 512 workgroups of the selection kernel, measured -- so workgroups count on sub-counter
// (id mod FAN) and only the last arrival of each sub-counter touches the top one.)
__device__ __forceinline__ bool last_block_done(int *ticket, unsigned total_blocks, unsigned id, bool flat = false) {
  __shared__ int s_is_last;
  // every thread's published stores (publish()) have reached the coherence point before the barrier
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    int last = 0;
    if (total_blocks <= 2 * EDM_TICKET_FAN || flat) {
      const unsigned t = (unsigned)__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == total_blocks - 1) {
        last = 1;
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const unsigned sub = id % EDM_TICKET_FAN;
      const unsigned subtotal = total_blocks / EDM_TICKET_FAN + (sub < total_blocks % EDM_TICKET_FAN ? 1u : 0u);
      int *mine = ticket + 32 * (1 + sub);
      const unsigned t = (unsigned)__hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == subtotal - 1) {
        __hip_atomic_store(mine, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t2 = (unsigned)__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t2 == EDM_TICKET_FAN - 1) {
          last = 1;
          __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    s_is_last = last;
  }
  __syncthreads();
  return s_is_last != 0;
}
__device__ __forceinline__ bool last_block_done(int *ticket, unsigned total_blocks) {
  return last_block_done(ticket, total_blocks, blockIdx.x + gridDim.x * blockIdx.y);
}
// Data handed from the other workgroups to the last one travels through agent-scope (L2-coherent
// across the XCDs) relaxed atomics: an agent-scope FENCE would write back the whole L2 of the XCD
// (measured: ~35 us per stage on MI355X), these cost nothing extra.
template <typename T>
__device__ __forceinline__ void publish(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T acquire(const T *p) {
  return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The word the limiter's workgroup of k_integrals_gather publishes and the gather's workgroups poll:
// [sequence number of the launch : 40][state : 8][k, first hill of the ordered tail (< 2^16: chained batches hold at most
// 2048 hills) : 16].  It may be published twice:
//   state EDM_READY_BELOW -- (early, optional) the batch provably stays below the limit: the limiter will change no
//                            height, every hill keeps its base height.  Published by a wave that adds the per-hill
//                            integrals up while the limiter's wave is still walking them -- one round trip after
//                            the last integral instead of the limiter's three -- with a margin far above the sum's
//                            rounding (a near-tie waits for the limiter).
//   state EDM_READY_FINAL | limiter error (0 = none) -- the limiter's outputs are published, k is valid
// Every publication is an atomic MAX: sequence numbers grow with the launches and FINAL > BELOW, so the limiter's own
// word always stands whichever of the two arrives first, and the early wave needs no look at the word before it
// writes (a compare-and-swap over the word it had read first cost it a dependent round trip, 1.2 us on the path of
// every tile).
#define EDM_READY_BELOW 0x01
#define EDM_READY_FINAL 0x80
__device__ __forceinline__ unsigned long long ready_word(unsigned long long seq, int state, long long k) {
  return ((seq & 0xFFFFFFFFFFull) << 24) | ((unsigned long long)(state & 0xFF) << 16) | (unsigned long long)((unsigned)k & 0xFFFFu);
}
__device__ __forceinline__ unsigned long long ready_seq_of(unsigned long long w) { return w >> 24; }
__device__ __forceinline__ int ready_state_of(unsigned long long w) { return (int)((w >> 16) & 0xFF); }
__device__ __forceinline__ long long ready_k_of(unsigned long long w) { return (long long)(w & 0xFFFFull); }
__device__ __forceinline__ void ready_publish(unsigned long long *word, unsigned long long w) {
  (void)__hip_atomic_fetch_max(word, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one thread waits until a word of launch `seq` is there (final: the limiter's own) and returns it (a few hundred
// workgroups poll the same word: spaced ~0.2 us apart so that the polls do not crowd the round trips of the
// workgroup that will store it)
__device__ __forceinline__ unsigned long long wait_for_word(const unsigned long long *word, unsigned long long seq,
                                                            bool final = true) {
  const unsigned long long want = seq & 0xFFFFFFFFFFull;
  unsigned long long w = acquire(word);
  if (ready_seq_of(w) == want && (!final || (ready_state_of(w) & EDM_READY_FINAL))) return w;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    __builtin_amdgcn_s_sleep(7);   // ~450 cycles
    w = acquire(word);
    if (ready_seq_of(w) == want && (!final || (ready_state_of(w) & EDM_READY_FINAL))) return w;
    if (wall_clock64() - t0 > 1000000000ull) __builtin_trap();   // 10 s at 100 MHz: never, short of a lost launch
  }
}

// Selection + hill preparation in one launch (stochastic steps with a deferred count): every
// workgroup compacts its SEL_CHUNK samples IN ORDER into its own stretch of `stage`, the last one
// scans the per-workgroup counts, moves at most `h.nh` (the launch bound of the step) entries to the
// dense ordered list and prepares those hills.
// accepted sample `src` becomes hill i of this rank: prepared in place, or (multi-GPU) packed for the exchange
template <int DIM>
__device__ __forceinline__ void select_emit(const SelectArgs &a, const Geom &g, const HillList &h, long long i,
                                            long long src) {
  if (a.pack) {
    double x[DIM];
    sample_position<DIM>(h, src, x);
#pragma unroll
    for (int d = 0; d < DIM; d++) a.pack[1 + i * DIM + d] = x[d];
    if (a.sel) a.sel[i] = src;   // (the reference-order force pass needs this rank's accepted sample indices)
  } else {
    a.sel[i] = src;
    hill_prep_one<DIM>(g, h, i, src);
  }
}

template <int DIM>
__device__ __forceinline__ void select_prep_body(const SelectArgs &a, const Geom &g, const HillList &h, unsigned bid,
                                                 unsigned nblk) {
  __shared__ int s_w[SEL_PER_THREAD][BLOCK / 64];
  __shared__ long long s_carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (a.snap_dst) {   // (the step-start copy of the grid for the reference-order force pass: 179 KB on W1, the first workgroups' job)
    typedef double vec2 __attribute__((ext_vector_type(2)));
    const vec2 *__restrict__ src = reinterpret_cast<const vec2 *>(a.snap_src);
    vec2 *__restrict__ dst = reinterpret_cast<vec2 *>(a.snap_dst);
    for (long long k = (long long)bid * BLOCK + threadIdx.x; k < a.snap_n; k += (long long)nblk * BLOCK) dst[k] = src[k];
  }
  {
    // sample (row j, thread t) = chunk base + j * BLOCK + t: every row is one coalesced load per wave.  A
    // sample's place in the chunk's ordered list = accepted samples of the rows before it + accepted samples of
    // its own row in lower waves + lower lanes of its wave (ballot): eight ballots, no shuffle scan.
    const long long base = (long long)bid * SEL_CHUNK + threadIdx.x;
    bool fl[SEL_PER_THREAD];
    unsigned long long bal[SEL_PER_THREAD];
    bool any = false;
    // sel_flag() of the thread's eight samples with every load requested before the first one is looked at (as a
    // loop over sel_flag the loads sat behind each other's compare: eight dependent round trips instead of one)
    double uu[SEL_PER_THREAD];
    int mk[SEL_PER_THREAD];
    const bool has_mask = a.apply_mask >= 0;
#pragma unroll
    for (int j = 0; j < SEL_PER_THREAD; j++) {
      const long long i = base + (long long)j * BLOCK;
      mk[j] = has_mask ? a.mask[i < a.n ? i : a.n - 1] : 0;
    }
    if (a.ru) {
#pragma unroll
      for (int j = 0; j < SEL_PER_THREAD; j++) {
        const long long i = base + (long long)j * BLOCK;
        uu[j] = a.ru[i < a.n ? i : a.n - 1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < SEL_PER_THREAD; j++) uu[j] = device_uniform(a.rng, base + (long long)j * BLOCK);
    }
#pragma unroll
    for (int j = 0; j < SEL_PER_THREAD; j++) {
      const long long i = base + (long long)j * BLOCK;
      fl[j] = (i < a.n) && (!has_mask || (a.apply_mask & mk[j])) && (!a.use_thr || uu[j] < a.thr);
      any |= fl[j];
    }
#pragma unroll
    for (int j = 0; j < SEL_PER_THREAD; j++) {
      bal[j] = __ballot(fl[j]);
      if (lane == 0) s_w[j][wave] = __popcll(bal[j]);
    }
    __syncthreads();
    int *mine = a.stage + (long long)bid * SEL_CHUNK;
    if (any || threadIdx.x == 0) {
      int run = 0;   // accepted samples of the rows walked so far
#pragma unroll
      for (int j = 0; j < SEL_PER_THREAD; j++) {
        int before = 0, row = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) {
          if (w < wave) before += s_w[j][w];
          row += s_w[j][w];
        }
        if (fl[j])
          publish(&mine[run + before + __popcll(bal[j] & ((1ull << lane) - 1ull))], (int)(j * BLOCK + threadIdx.x));
        run += row;
      }
      if (threadIdx.x == 0) publish(&a.counts[bid], run);
    }
  }
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)bid * 8 + 1] = wall_clock64();
  if (!last_block_done(a.ticket, nblk, bid)) {
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)bid * 8 + 2] = wall_clock64();
    return;
  }
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)bid * 8 + 3] = wall_clock64();

  // scan of the per-workgroup counts: every thread takes PERC consecutive workgroups and requests their
  // counts together (one memory round trip per BLOCK * PERC workgroups; 512 workgroups = one pass), the
  // exclusive offsets go to LDS, and the accepted samples of the pass are then dealt out evenly -- thread t
  // takes output slots t, t + BLOCK, ... and finds each one's workgroup by bisection -- so a workgroup
  // that accepted many samples costs no more than one that accepted one
  const int nblocks = (int)nblk;
  const long long bound = h.nh;
  constexpr int PERC_MAX = 8;
  int PERC = (nblocks + BLOCK - 1) / BLOCK;
  if (PERC > PERC_MAX) PERC = PERC_MAX;
  __shared__ int s_off[BLOCK * PERC_MAX + 1];
  __shared__ int s_ws[BLOCK / 64];
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int bb = 0; bb < nblocks; bb += BLOCK * PERC) {
    const int blk0 = bb + threadIdx.x * PERC;
    int cj[PERC_MAX];
    int c = 0;
#pragma unroll
    for (int j = 0; j < PERC_MAX; j++) {
      cj[j] = (j < PERC && blk0 + j < nblocks) ? acquire(&a.counts[blk0 + j]) : 0;
      c += cj[j];
    }
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) s_ws[wave] = inc;
    __syncthreads();
    int off = inc - c;
    int pass_total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) {
      if (w < wave) off += s_ws[w];
      pass_total += s_ws[w];
    }
#pragma unroll
    for (int j = 0; j < PERC_MAX; j++) {
      if (j < PERC) s_off[threadIdx.x * PERC + j] = off;
      off += cj[j];
    }
    if (threadIdx.x == 0) s_off[BLOCK * PERC] = pass_total;
    __syncthreads();
    const long long carry = s_carry;
    const int nloc = BLOCK * PERC;
    for (int e = threadIdx.x; e < pass_total; e += BLOCK) {
      if (carry + e >= bound) break;
      // last k with s_off[k] <= e  (empty workgroups share their successor's offset and are skipped)
      int lo = 0, hi = nloc;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= e) lo = mid; else hi = mid;
      }
      const long long blk = bb + lo;
      const long long src = blk * SEL_CHUNK + acquire(&a.stage[blk * SEL_CHUNK + (e - s_off[lo])]);
      select_emit<DIM>(a, g, h, carry + e, src);
    }
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + pass_total;
    __syncthreads();
  }
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)bid * 8 + 4] = wall_clock64();
  if (threadIdx.x == 0) {
    const long long total = s_carry;
    *a.count_host = total;
    *a.count_dev = total;
    if (a.pack) a.pack[0] = (double)total;
  }
}

template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_select_prep(SelectArgs a, Geom g, HillList h) {
  select_prep_body<DIM>(a, g, h, blockIdx.x, gridDim.x);
}
// fix edm step on a 2-D / 3-D grid without an overflow flush: the force kernel (K2 on four lanes per atom) and the
// step's selection + hill preparation in ONE launch -- workgroups [0, nsel) select (their serial tail is the longest
// chain: dispatched first), the rest evaluate forces.  The selection reads positions, mask and uniforms, the lookups the
// grid; the gather that writes the grid is a later launch.
template <int DIM, bool REPLICA>
__global__ void __launch_bounds__(BLOCK) k_lookup_quad_select(Geom g, const double *__restrict__ faces, LookupArgs la,
                                                              double *__restrict__ block_energy, SelectArgs a, HillList h,
                                                              unsigned nsel) {
  if (blockIdx.x < nsel)
    select_prep_body<DIM>(a, g, h, blockIdx.x, nsel);
  else
    lookup_quad_body<DIM, LOOKUP_FORCES, REPLICA>(g, faces, la, block_energy, blockIdx.x - nsel, gridDim.x - nsel);
}
hipError_t launch_lookup_select(const Geom &g, const double *rec, const LookupArgs &la, double *scratch, hipStream_t s,
                                hipEvent_t ev0, hipEvent_t ev1, int *blocks_out, const double *faces, const SelectArgs &a,
                                const HillList &h) {
  if (!(g.dim > 1 && g.interp && g.rec == 4) || la.n <= 0 || a.n <= 0) return hipErrorInvalidValue;
  long long qb = (la.n * 4 + BLOCK - 1) / BLOCK;
  if (qb > MAX_BLOCKS) qb = MAX_BLOCKS;
  const unsigned nb = (unsigned)(qb < 1 ? 1 : qb), nsel = (unsigned)((a.n + SEL_CHUNK - 1) / SEL_CHUNK);
  const dim3 grid(nsel + nb);
  if (g.dim == 2) {
    if (faces)
      EDM_LAUNCH_TIMED((k_lookup_quad_select<2, true>), grid, dim3(BLOCK), 0, s, ev0, ev1, g, faces, la, scratch, a, h, nsel);
    else
      EDM_LAUNCH_TIMED((k_lookup_quad_select<2, false>), grid, dim3(BLOCK), 0, s, ev0, ev1, g, rec, la, scratch, a, h, nsel);
  } else {
    if (faces)
      EDM_LAUNCH_TIMED((k_lookup_quad_select<3, true>), grid, dim3(BLOCK), 0, s, ev0, ev1, g, faces, la, scratch, a, h, nsel);
    else
      EDM_LAUNCH_TIMED((k_lookup_quad_select<3, false>), grid, dim3(BLOCK), 0, s, ev0, ev1, g, rec, la, scratch, a, h, nsel);
  }
  if (blocks_out) *blocks_out = (int)nb;
  return hipGetLastError();
}

// One launch for the two independent streaming passes of a fix edm_pair hill step: workgroups [0, nsel) run the
// selection (+ hill preparation in the last of them), the rest evaluate the pair forces (K1, arrays read from
// L2).  Neither touches what the other reads or writes; the selection's serial tail hides behind the lookups.
// (Selection workgroups keep ticket ids 0 .. nsel-1 whichever role is dispatched first: PairForcesArgs::sel_first.)
struct PairForcesArgs {
  const double *rec;
  long long n;
  const double *r;
  double *force;
  double *block_energy;
  double inv_dx;
  unsigned nsel, nk1;
  int sel_first;
};
__global__ void __launch_bounds__(BLOCK) k_pair_forces_select(SelectArgs a, Geom g, HillList h, PairForcesArgs f) {
  extern __shared__ double2 lds_all[];
  // role of this workgroup: selection workgroup `sb` (sb < nsel) or K1 workgroup `sb - nsel`
  const unsigned sb = f.sel_first ? blockIdx.x : (blockIdx.x >= f.nk1 ? blockIdx.x - f.nk1 : f.nsel + blockIdx.x);
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)sb * 8] = wall_clock64();   // (trace slots: selection first, then K1)
  if (sb < f.nsel)
    select_prep_body<1>(a, g, h, sb, f.nsel);
  else
    pair_forces_fast_body<false, BLOCK>(g, f.rec, f.n, f.r, f.force, f.block_energy, 0LL, 0, f.inv_dx, lds_all,
                                        sb - f.nsel, f.nk1);
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)sb * 8 + 7] = wall_clock64();
}

// The same pairing for fix edm_pair on a device-resident neighbour list: workgroups [0, nsel) run the selection over
// the list's virtual add_hill samples (+ hill preparation in the last of them), the rest the force pass
// (k_pairlist_forces' body).  The selection reads the live-sample mask and the uniforms' stream, the force pass the
// positions and the grid; the 2.26 M-sample selection of the 32 k-atom melt (14 us on its own) hides behind the pass.
template <bool FAST>
__global__ void __launch_bounds__(BLOCK) k_pairlist_forces_select(SelectArgs a, Geom g, HillList h, PairListArgs pl,
                                                                  const double *__restrict__ rec,
                                                                  double *__restrict__ partials, double inv_dx,
                                                                  unsigned nsel) {
  if (blockIdx.x < nsel)
    select_prep_body<1>(a, g, h, blockIdx.x, nsel);
  else
    pairlist_forces_body<FAST>(g, rec, pl, partials, inv_dx, blockIdx.x - nsel, gridDim.x - nsel);
}
hipError_t launch_pairlist_forces_select(const SelectArgs &a, const Geom &g, const HillList &h, const double *rec,
                                         const PairListArgs &pl, double *partials, hipStream_t s, int *blocks_out) {
  if (g.dim != 1 || pl.nall <= 0 || a.n <= 0) return hipErrorInvalidValue;
  const long long threads = (long long)pl.nall * 16;
  long long nb = (threads + BLOCK - 1) / BLOCK;
  if (nb > MAX_BLOCKS) nb = MAX_BLOCKS;
  const unsigned nsel = (unsigned)((a.n + SEL_CHUNK - 1) / SEL_CHUNK);
  const bool fast = (g.interp && !g.periodic[0] && !g.bper[0] && g.n[0] >= 2);
  const double inv_dx = 1.0 / g.dx[0];
  if (fast)
    hipLaunchKernelGGL(k_pairlist_forces_select<true>, dim3(nsel + (unsigned)nb), dim3(BLOCK), 0, s, a, g, h, pl, rec, partials,
                       inv_dx, nsel);
  else
    hipLaunchKernelGGL(k_pairlist_forces_select<false>, dim3(nsel + (unsigned)nb), dim3(BLOCK), 0, s, a, g, h, pl, rec, partials,
                       inv_dx, nsel);
  if (blocks_out) *blocks_out = (int)nb;
  return hipGetLastError();
}

size_t select_stage_ints(long long n) { return (size_t)((n + SEL_CHUNK - 1) / SEL_CHUNK) * SEL_CHUNK; }

hipError_t launch_select_prep(const SelectArgs &a, const Geom &g, const HillList &h, hipStream_t s) {
  const int nb = (int)((a.n + SEL_CHUNK - 1) / SEL_CHUNK);
  if (nb <= 0) return hipErrorInvalidValue;
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_select_prep<1>, dim3(nb), dim3(BLOCK), 0, s, a, g, h); break;
    case 2: hipLaunchKernelGGL(k_select_prep<2>, dim3(nb), dim3(BLOCK), 0, s, a, g, h); break;
    default: hipLaunchKernelGGL(k_select_prep<3>, dim3(nb), dim3(BLOCK), 0, s, a, g, h); break;
  }
  return hipGetLastError();
}

bool pair_forces_select_fusable(const Geom &g, long long n_pairs, long long n_samples) {
  return pair_fast_path(g) && n_pairs > 0 && n_pairs < PAIR_LDS_THRESHOLD && n_samples > 0;
}
hipError_t launch_pair_forces_select(const SelectArgs &a, const Geom &g, const HillList &h, const double *rec, long long n,
                                     const double *r, double *force, double *scratch, hipStream_t s, hipEvent_t ev0,
                                     hipEvent_t ev1, int *blocks_out) {
  if (!pair_forces_select_fusable(g, n, a.n)) return hipErrorInvalidValue;
  PairForcesArgs f;
  f.rec = rec;
  f.n = n;
  f.r = r;
  f.force = force;
  f.block_energy = scratch;
  f.inv_dx = 1.0 / g.dx[0];
  f.nsel = (unsigned)((a.n + SEL_CHUNK - 1) / SEL_CHUNK);
  f.nk1 = (unsigned)pair_short_blocks(n);
  // (selection workgroups dispatched AHEAD of K1's: their serial tail -- ticket, scan, preparation -- is the longest chain
  // of the launch and ends 0.9 us earlier when it starts first: list prepared at 8.6 instead of 9.5 us, launch 12.0-12.3
  // instead of 12.5-13.3 us)
  f.sel_first = 1;
  EDM_LAUNCH_TIMED(k_pair_forces_select, dim3(f.nsel + f.nk1), dim3(BLOCK), 256, s, ev0, ev1, a, g, h, f);
  if (blocks_out) *blocks_out = (int)f.nk1;
  return hipGetLastError();
}

hipError_t launch_hill_prep(const Geom &g, const HillList &h, hipStream_t s, const double *fetch_src, double *fetch_dst) {
  if (h.nh <= 0) return hipSuccess;
  const int b = blocks_for(h.nh);
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_hill_prep<1>, dim3(b), dim3(BLOCK), 0, s, g, h, fetch_src, fetch_dst); break;
    case 2: hipLaunchKernelGGL(k_hill_prep<2>, dim3(b), dim3(BLOCK), 0, s, g, h, fetch_src, fetch_dst); break;
    default: hipLaunchKernelGGL(k_hill_prep<3>, dim3(b), dim3(BLOCK), 0, s, g, h, fetch_src, fetch_dst); break;
  }
  return hipGetLastError();
}

bool lookup_prep_fusable(const Geom &g, const HillList &h) {
  return g.dim > 1 && g.interp && g.rec == 4 && h.nh > 0;
}
hipError_t launch_lookup_prep(const Geom &g, const double *rec, const LookupArgs &a, double *scratch, hipStream_t s,
                              hipEvent_t ev0, hipEvent_t ev1, int *blocks_out, const double *faces, const HillList &h,
                              const double *fetch_src, double *fetch_dst) {
  if (!lookup_prep_fusable(g, h) || a.n <= 0) return hipErrorInvalidValue;
  long long qb = (a.n * 4 + BLOCK - 1) / BLOCK;
  if (qb > MAX_BLOCKS) qb = MAX_BLOCKS;
  const unsigned nb = (unsigned)(qb < 1 ? 1 : qb), nb_prep = (unsigned)blocks_for(h.nh);
  const dim3 grid(nb_prep + nb);
  if (g.dim == 2) {
    if (faces)
      EDM_LAUNCH_TIMED((k_lookup_quad_prep<2, true>), grid, dim3(BLOCK), 0, s, ev0, ev1, g, faces, a, scratch, h, fetch_src, fetch_dst, nb_prep);
    else
      EDM_LAUNCH_TIMED((k_lookup_quad_prep<2, false>), grid, dim3(BLOCK), 0, s, ev0, ev1, g, rec, a, scratch, h, fetch_src, fetch_dst, nb_prep);
  } else {
    if (faces)
      EDM_LAUNCH_TIMED((k_lookup_quad_prep<3, true>), grid, dim3(BLOCK), 0, s, ev0, ev1, g, faces, a, scratch, h, fetch_src, fetch_dst, nb_prep);
    else
      EDM_LAUNCH_TIMED((k_lookup_quad_prep<3, false>), grid, dim3(BLOCK), 0, s, ev0, ev1, g, rec, a, scratch, h, fetch_src, fetch_dst, nb_prep);
  }
  if (blocks_out) *blocks_out = (int)nb;
  return hipGetLastError();
}

// Multi-GPU exchange, receive side: `recv` holds one fixed-size packet per rank, [count, x_0 .. x_{bound-1}]
// (what k_select_prep packed and ncclAllGather concatenated).  One workgroup builds the rank-major global
// hill list: positions into `all` (stride DIM), the prepared hill fields, and the global count (device +
// host-mapped).  A rank that overflowed its packet poisons the count, which the limiter turns into
// error 2 on every rank alike.
template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_unpack_prep(UnpackArgs a, Geom g, HillList h) {
  __shared__ long long s_off[EDM_MAX_RANKS + 1];
  __shared__ int s_bad;
  if (threadIdx.x == 0) {
    long long run = 0;
    int bad = 0;
    for (int r = 0; r < a.nranks; r++) {
      const long long c = (long long)a.recv[(long long)r * a.packet];
      s_off[r] = run;
      if (c < 0 || c > a.bound) bad = 1;
      run += (c < 0) ? 0 : (c > a.bound ? a.bound : c);
    }
    s_off[a.nranks] = run;
    s_bad = bad;
    if (a.local_range) {   // this rank's slice of the rank-major list (the reference-order force pass)
      a.local_range[0] = s_off[a.rank];
      a.local_range[1] = s_off[a.rank + 1] - s_off[a.rank];
    }
    const long long total = bad ? (long long)0x3fffffffffffffffLL : run;
    *a.count_dev = total;
    *a.count_host = total;
  }
  __syncthreads();
  if (s_bad) return;
  const long long total = s_off[a.nranks];
  for (long long e = threadIdx.x; e < total; e += BLOCK) {
    int r = 0;
    while (r + 1 < a.nranks && e >= s_off[r + 1]) r++;
    const long long i = e - s_off[r];
    double x[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      x[d] = a.recv[(long long)r * a.packet + 1 + i * DIM + d];
      a.all[e * DIM + d] = x[d];
    }
    hill_prep_vals<DIM>(g, h, e, x);
  }
}
hipError_t launch_unpack_prep(const UnpackArgs &a, const Geom &g, const HillList &h, hipStream_t s) {
  if (a.nranks < 1 || a.nranks > EDM_MAX_RANKS) return hipErrorInvalidValue;
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_unpack_prep<1>, dim3(1), dim3(BLOCK), 0, s, a, g, h); break;
    case 2: hipLaunchKernelGGL(k_unpack_prep<2>, dim3(1), dim3(BLOCK), 0, s, a, g, h); break;
    default: hipLaunchKernelGGL(k_unpack_prep<3>, dim3(1), dim3(BLOCK), 0, s, a, g, h); break;
  }
  return hipGetLastError();
}

// per-hill heights of a rank-major hill list when the height depends on the rank of origin
// (hill_density unset: prefactor / est_hill_count of the sending rank, edm_bias.cpp:552-556)
__global__ void __launch_bounds__(BLOCK) k_rank_heights(RankHeights rh, double *__restrict__ out) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < rh.offset[rh.nranks]; i += stride) {
    int r = 0;
    while (r + 1 < rh.nranks && i >= rh.offset[r + 1]) r++;
    out[i] = rh.height[r];
  }
}
hipError_t launch_rank_heights(const RankHeights &rh, double *out, hipStream_t s) {
  const long long n = rh.offset[rh.nranks];
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_rank_heights, dim3(blocks_for(n)), dim3(BLOCK), 0, s, rh, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// node-side and pair-side pieces of one stencil term (gaussian_grid.h:284-355)
// ---------------------------------------------------------------------------
template <int DIM>
struct NodeTerms {
  double xx[DIM];                    // node coordinate                     (:270)
  double t2[DIM], t4[DIM];           // zero-force blend at the two walls   (:311,:313)
  double t6[DIM], t7[DIM];           // its derivatives                     (:322-323)
  double dden[DIM];                  // derivative-table entry at bc_index  (:335)
  // bc_denom after dimension d is a product of node-only factors (table entries :318 or
  // sqrt(pi)*sigma :340), so it and its reciprocals are computed ONCE per node:
  double dprod[DIM], inv_dprod[DIM], inv_dprod2[DIM];
  bool inside;                       // node lies within every non-periodic boundary (:273)
};

// per-launch constants of the stencil term
template <int DIM>
struct TermConst {
  double inv_sigma[DIM], period[DIM], inv_period[DIM];
};
template <int DIM>
__device__ __forceinline__ void term_const(const Geom &g, TermConst<DIM> &tc) {
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    tc.inv_sigma[d] = 1.0 / g.sigma[d];
    tc.period[d] = g.max[d] - g.min[d];
    tc.inv_period[d] = 1.0 / (g.max[d] - g.min[d]);
  }
}

// PERB: every boundary dimension is periodic (no walls, no McGovern-De Pablo terms) -- known at compile time so
// that the wall blends, table reads and their registers disappear; the arithmetic that remains is unchanged
template <int DIM, bool PERB = false>
__device__ __forceinline__ void node_terms(const Geom &g, const Tables &t, const int *p, NodeTerms<DIM> &nt) {
  nt.inside = true;
  double running = 1.0;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    const double xx = g.min[d] + g.dx[d] * (size_t)p[d];
    nt.xx[d] = xx;
    nt.t2[d] = nt.t4[d] = nt.t6[d] = nt.t7[d] = 0;
    nt.dden[d] = 0;
    double factor = sqrt(M_PI) * g.sigma[d];
    if (!PERB && !g.bper[d]) {
      factor = 1.0;
      if (xx < g.bmin[d] || xx > g.bmax[d]) {
        nt.inside = false;
      } else {
        const double sg = g.sigma[d];
        const size_t ti = (size_t)((EDM_BC_TABLE_SIZE - 1) * (xx - g.bmin[d]) / (g.bmax[d] - g.bmin[d]));
        nt.t2[d] = smooth_step((xx - g.bmin[d]) / (sg * EDM_BC_MAR));
        nt.t4[d] = smooth_step((g.bmax[d] - xx) / (sg * EDM_BC_MAR));
        nt.t6[d] = smooth_step_dt((xx - g.bmin[d]) / (sg * EDM_BC_MAR)) / (EDM_BC_MAR * sg);
        nt.t7[d] = -smooth_step_dt((g.bmax[d] - xx) / (sg * EDM_BC_MAR)) / (EDM_BC_MAR * sg);
        factor = t.denom[d][ti];
        nt.dden[d] = t.dderiv[d][ti];
      }
    }
    running *= factor;
    nt.dprod[d] = running;
    nt.inv_dprod[d] = 1.0 / running;
    nt.inv_dprod2[d] = 1.0 / (running * running);
  }
}

__global__ void __launch_bounds__(BLOCK) k_build_node_table(Geom g, Tables t, double *__restrict__ out) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= g.n[0]) return;
  const int p[1] = {i};
  NodeTerms<1> nt;
  node_terms<1, false>(g, t, p, nt);
  out[4 * (size_t)i + 0] = nt.t2[0];
  out[4 * (size_t)i + 1] = nt.t4[0];
  out[4 * (size_t)i + 2] = nt.inside ? nt.inv_dprod[0] : -nt.inv_dprod[0];   // (1 / bc_denom > 0: the sign says inside / outside)
  out[4 * (size_t)i + 3] = 0.0;
}
hipError_t launch_build_node_table(const Geom &g, const Tables &t, double *out, hipStream_t s) {
  if (g.dim != 1 || g.bper[0]) return hipErrorInvalidValue;
  Tables tt = t;
  tt.node1d = nullptr;
  hipLaunchKernelGGL(k_build_node_table, dim3((unsigned)((g.n[0] + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, g, tt, out);
  return hipGetLastError();
}

// One (node, hill) term: val multiplies the height for V, dval[d] for dV/ds_d.  Returns false when
// the node is outside the hill's support (dp2 >= 8).  Same formulas as gaussian_grid.h:284-355 with
// every division replaced by a multiplication with a node- or launch-constant reciprocal (values
// move by ~1e-16 relative; the accumulation ORDER over hills is untouched).
template <int DIM, bool PERB = false>
__device__ __forceinline__ bool pair_term(const Geom &g, const TermConst<DIM> &tc, const NodeTerms<DIM> &nt,
                                          const double *hx, const double *ht, double &val, double *dval,
                                          bool &corr_nonzero, bool interior = false) {
  double dp[DIM], raw[DIM];
  double dp2 = 0;
  bool edge = false;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    raw[d] = nt.xx[d] - hx[d];
    if (g.periodic[d]) {
      const double turns = raw[d] * tc.inv_period[d];
      // the nearest-image choice (:287-291) must match the reference's round(dp / period) exactly
      if (fabs(fabs(turns - floor(turns)) - 0.5) < 1e-9) edge = true;
      raw[d] -= round_half(turns) * tc.period[d];
    }
    dp[d] = raw[d] * tc.inv_sigma[d];
    dp2 += dp[d] * dp[d];
  }
  // ... and so must the support test dp2 < 8 (:299): within rounding of the edge (nodes that sit
  // exactly sqrt(8) sigma away are common on coarse grids) redo it with the reference's divisions
  if (edge || fabs(dp2 - EDM_GAUSS_SUPPORT) < 1e-9) {
    dp2 = 0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      double e = nt.xx[d] - hx[d];
      if (g.periodic[d]) e -= round_half(e / (g.max[d] - g.min[d])) * (g.max[d] - g.min[d]);
      e /= g.sigma[d];
      dp[d] = e;
      dp2 += e * e;
    }
  }
  if (!(dp2 < EDM_GAUSS_SUPPORT)) return false;
  double expo = exp(-dp2);
  double corr = 0;
  double force[DIM];
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    force[d] = 0;
    if (!PERB && !g.bper[d]) {
      if (interior) {
        // no wall blend reaches this node (t2 = t4 = t6 = t7 = 0 for the whole wave): the general expressions
        // below reduce to these, bit for bit (their extra terms are products with zero)
        const double t5 = -2 * dp[d] * tc.inv_sigma[d];
        double F = t5 * expo;
        F = F * nt.dprod[d] - nt.dden[d] * expo;
        F *= nt.inv_dprod2[d];
        corr = 0;
        force[d] = F;
      } else {
        const double t1 = ht[2 * d], t3 = ht[2 * d + 1];
        corr = (t1 - expo) * nt.t2[d] + (t3 - expo) * nt.t4[d];  // overwritten per dim (:316)
        const double t5 = -2 * dp[d] * tc.inv_sigma[d];
        double F = t5 * expo;
        F += (t1 - expo) * nt.t6[d] - t5 * expo * nt.t2[d] + (t3 - expo) * nt.t7[d] - t5 * expo * nt.t4[d];
        F = F * nt.dprod[d] - nt.dden[d] * (expo + corr);
        F *= nt.inv_dprod2[d];
        corr *= nt.inv_dprod[d];
        force[d] = F;
      }
    }
  }
  expo *= nt.inv_dprod[DIM - 1];
  val = expo + corr;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    if (PERB || g.bper[d])
      dval[d] = -(2 * dp[d] * tc.inv_sigma[d] * expo);
    else
      dval[d] = force[d];
  }
  corr_nonzero = (corr * corr > 0);
  return true;
}

// Packed read-back region of a short limited batch (apply_hills' layout, sized by the launch bound nb) -> host-mapped
// memory; only the part the batch's true hill count na fills.  System-scope stores: written through to host
// memory now (plain stores would sit in L2 until the end-of-kernel write-back); agent-scope loads: the region
// was filled by other workgroups and launches.
template <int DIM>
__device__ __forceinline__ void readback_copy(const char *rb_src, char *rb_dst, long long nb, long long na, int me,
                                              int nthr) {
  const long long off_flags = 64, off_h2 = off_flags + ((4 * nb + 7) & ~7LL), off_a2 = off_h2 + 8 * nb,
                  off_added = off_a2 + 8 * nb, off_pos = off_added + 8 * nb;
  const long long seg_off[5] = {0, off_h2, off_a2, off_added, off_pos};
  const long long seg_len[5] = {off_flags + ((4 * na + 7) & ~7LL), 8 * na, 8 * na, 8 * na, 8 * na * DIM};
  const long long *src = reinterpret_cast<const long long *>(rb_src);
  long long *dst = reinterpret_cast<long long *>(rb_dst);
  long long longest = 0;
#pragma unroll
  for (int sgm = 0; sgm < 5; sgm++) longest = seg_len[sgm] > longest ? seg_len[sgm] : longest;
  // one word of every segment per round: the five loads travel together, then the five stores (a load -> store
  // pair per segment in turn made the copy a chain of five memory round trips)
  for (long long w = me; w < longest / 8; w += nthr) {
    long long v[5];
#pragma unroll
    for (int sgm = 0; sgm < 5; sgm++) v[sgm] = (w < seg_len[sgm] / 8) ? acquire(&src[seg_off[sgm] / 8 + w]) : 0;
#pragma unroll
    for (int sgm = 0; sgm < 5; sgm++)
      if (w < seg_len[sgm] / 8)
        __hip_atomic_store(&dst[seg_off[sgm] / 8 + w], v[sgm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// the limiter's part of the same region: result header + tail flags, undo heights, undo bias
template <int DIM>
__device__ __forceinline__ void readback_copy_limiter(const char *rb_src, char *rb_dst, long long nb, long long na, int me,
                                                      int nthr) {
  const long long off_flags = 64, off_h2 = off_flags + ((4 * nb + 7) & ~7LL), off_a2 = off_h2 + 8 * nb;
  const long long seg_off[3] = {0, off_h2, off_a2};
  const long long seg_len[3] = {off_flags + ((4 * na + 7) & ~7LL), 8 * na, 8 * na};
  const long long *src = reinterpret_cast<const long long *>(rb_src);
  long long *dst = reinterpret_cast<long long *>(rb_dst);
  const long long longest = seg_len[0] > seg_len[1] ? seg_len[0] : seg_len[1];
  for (long long w = me; w < longest / 8; w += nthr) {
    long long v[3];
#pragma unroll
    for (int sgm = 0; sgm < 3; sgm++) v[sgm] = (w < seg_len[sgm] / 8) ? acquire(&src[seg_off[sgm] / 8 + w]) : 0;
#pragma unroll
    for (int sgm = 0; sgm < 3; sgm++)
      if (w < seg_len[sgm] / 8)
        __hip_atomic_store(&dst[seg_off[sgm] / 8 + w], v[sgm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// the limiter-independent part of the same region: per-hill bias and original positions of hills [0, na)
template <int DIM>
__device__ __forceinline__ void readback_copy_hills(const char *rb_src, char *rb_dst, long long nb, long long na, int me,
                                                    int nthr, const double *s_added = nullptr) {
  const long long off_flags = 64, off_h2 = off_flags + ((4 * nb + 7) & ~7LL), off_a2 = off_h2 + 8 * nb,
                  off_added = off_a2 + 8 * nb, off_pos = off_added + 8 * nb;
  const long long *src = reinterpret_cast<const long long *>(rb_src);
  long long *dst = reinterpret_cast<long long *>(rb_dst);
  for (long long w = me; w < na * DIM; w += nthr) {
    const bool has_a = w < na;
    const long long va = has_a ? (s_added ? __double_as_longlong(s_added[w]) : acquire(&src[off_added / 8 + w])) : 0;
    const long long vp = acquire(&src[off_pos / 8 + w]);
    if (has_a) __hip_atomic_store(&dst[off_added / 8 + w], va, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst[off_pos / 8 + w], vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------
// K3: per-hill integrated bias -- one wave per hill walks the reference's stencil
// (gaussian_grid.h:227-281) and reduces h*(expo+corr)*vol in a fixed order.
// ---------------------------------------------------------------------------
// TPH threads cooperate on one hill: 64 (a wave per hill, long lists) or 256 (a workgroup per hill:
// four times shorter critical path for the few-hundred-hill batches of a stochastic hill step)
template <bool COHERENT>
__device__ __forceinline__ void limit_wave(long long nh_bound, const double *added, const double *heights,
                                           double h_const, double limit, double cum_in, int flush_mode,
                                           const LimitTail &tail, LimitResult *res, long long nchunks,
                                           const double *chunk_sum, const double *chunk_max,
                                           const long long *nh_dev, long long mirror = 0, long long *k_out = nullptr,
                                           int *err_out = nullptr, long long nh_known = -1,
                                           LimitResult *out_local = nullptr, const double *s_added = nullptr);

// The stencil walk of one hill by TPH cooperating threads (lt = this thread's index among them): the thread's share of
// height * (expo + corr) * vol over the reference's stencil (gaussian_grid.h:227-281), summed in stencil order.  The
// caller adds the shares up (wave_sum, then the waves in order): k_hill_integrals and the selection workgroups of
// every caller of this walk yields the same bits.
template <int DIM, int TPH, bool PERB>
__device__ __forceinline__ double hill_stencil_partial(const Geom &g, const Tables &t, const TermConst<DIM> &tc,
                                                       const int *c_r, const double *hx_r, const double *ht_r,
                                                       double height_r, bool live, int lt) {
  double acc = 0;
  const int c0 = live ? c_r[0] : INT_MIN;
  if (c0 != INT_MIN) {
    int c[DIM];
    double hx[DIM], ht[2 * DIM];
    double vol = 1;
    long long total = 1;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      c[d] = c_r[d];
      hx[d] = hx_r[d];
      ht[2 * d] = ht_r[2 * d];
      ht[2 * d + 1] = ht_r[2 * d + 1];
      vol *= g.dx[d];
      total *= (2 * g.msize[d] + 1);
    }
    const double height = height_r;
    // ILP stencil points per trip (independent decode + exp chains overlap; a workgroup-per-hill launch
    // has one wave per SIMD, so nothing else hides their latency); sums stay in stencil order.
    // 32-bit decode: the stencil has < 2^31 points whenever it fits a grid at all.
    constexpr int ILP = (DIM == 1) ? 8 : 4;  // (1-D: the whole 1131-point stencil in one trip of a 256-thread workgroup)
    const unsigned utotal = (unsigned)total;
    // the early distance test below works on un-wrapped offsets: valid while no stencil offset is
    // further than half a period from the hill (else the reference's minimum image folds it back, :287-291)
    bool ball_ok = true;
#pragma unroll
    for (int d = 0; d < DIM; d++)
      if (g.periodic[d] && 2 * (g.msize[d] + 1) > g.n[d]) ball_ok = false;
    if (DIM == 1 && !PERB && t.node1d) {
      // 1-D grid with walls: the node-only factors of every stencil point come from the table built by
      // launch_build_node_table (the same node_terms code ran there: same bits).  The walk is a serial instruction
      // stream per thread and the length of that stream is what a hill step waits for: the per-point divisions, wall
      // blends and the dependent read of the boundary table are gone, and the table rows of a trip's ILP points are
      // requested TOGETHER, before the first one is used (three phases, no load inside a branch).
      const double2 *tab2 = reinterpret_cast<const double2 *>(t.node1d);
      const int m0 = g.msize[0], n0 = g.n[0];
      const bool per0 = g.periodic[0] != 0;
      for (unsigned s0 = (unsigned)lt; s0 < utotal; s0 += ILP * TPH) {
        double term[ILP];
        int pw[ILP];
        bool ok[ILP];
        double2 qa[ILP];
        double qz[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) {
          const unsigned s = s0 + (unsigned)(u * TPH);
          bool v = s < utotal;
          int idx = (int)s - m0 + c[0];
          const double e = ((g.min[0] + idx * g.dx[0]) - hx[0]) * tc.inv_sigma[0];
          const double dp2_est = e * e;
          if (idx >= n0) {
            if (per0) idx %= n0; else v = false;
          }
          if (idx < 0) {
            if (per0) idx += n0; else v = false;
            if (idx < 0) v = false;
          }
          if (ball_ok && dp2_est > 8.0 * (1.0 + 1e-6)) v = false;
          ok[u] = v;
          pw[u] = v ? idx : 0;
        }
#pragma unroll
        for (int u = 0; u < ILP; u++) {
          qa[u] = tab2[2 * (size_t)pw[u]];
          qz[u] = t.node1d[4 * (size_t)pw[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < ILP; u++) {
          term[u] = 0;
          if (!ok[u] || !(qz[u] > 0.0)) continue;
          NodeTerms<DIM> nt;
          nt.xx[0] = g.min[0] + g.dx[0] * (size_t)pw[u];
          nt.t2[0] = qa[u].x;
          nt.t4[0] = qa[u].y;
          nt.inv_dprod[0] = qz[u];
          nt.t6[0] = nt.t7[0] = nt.dden[0] = nt.dprod[0] = nt.inv_dprod2[0] = 0;   // (derivative only: not used here)
          nt.inside = true;
          double val, dval[DIM];
          bool nz;
          if (!pair_term<DIM, PERB>(g, tc, nt, hx, ht, val, dval, nz)) continue;
          term[u] = height * val * vol;
        }
#pragma unroll
        for (int u = 0; u < ILP; u++) acc += term[u];
      }
      return acc;
    }
    if constexpr (DIM > 1) {
      if (t.ball && ball_ok) {
        // 2-D / 3-D (a workgroup per hill, or a wave per hill on long lists): only stencil points that can lie inside the hill's support (dp2 < 8, :299) are
        // dealt out -- the host's list of such offsets (Tables::ball: one coalesced load per point, no decode, nothing
        // shared between the threads; pair_term makes the exact test).  The box the reference walks holds
        // (2 msize + 1)^DIM points of which the support's ball covers 20 % in 2-D and 6 % in 3-D; dealt out point by
        // point, every wave met a few support points in most of its trips and walked all 24 trips of W4's 23^3 box as
        // a serial instruction stream: 16 us per hill however few hills a step had, 7 us this way (a point is ~350
        // dependent fp64 instructions and the hill's waves share one CU).  Same terms; a thread's terms are added in
        // list order, the threads' sums as before.
        const double thr = 8.0 * (1.0 + 1e-6);
        for (int j = lt; j < t.nball; j += TPH) {
          const int pk = t.ball[j];
          int p[DIM];
          bool skip = false;
          double dp2_est = 0;
#pragma unroll
          for (int d = 0; d < DIM; d++) {
            int idx = ((pk >> (8 * d)) & 255) - 128 + c[d];
            const double e = ((g.min[d] + idx * g.dx[d]) - hx[d]) * tc.inv_sigma[d];
            dp2_est += e * e;
            // (ball_ok: the stencil is narrower than the grid, so a periodic index is less than one period out)
            if (idx >= g.n[d]) {
              if (g.periodic[d]) idx -= g.n[d]; else skip = true;
              if (idx >= g.n[d]) idx %= g.n[d];
            }
            if (idx < 0) {
              if (g.periodic[d]) idx += g.n[d]; else skip = true;
              if (idx < 0) skip = true;
            }
            p[d] = idx;
          }
          if (skip || dp2_est > thr) continue;
          NodeTerms<DIM> nt;
          node_terms<DIM, PERB>(g, t, p, nt);
          if (!nt.inside) continue;
          double val, dval[DIM];
          bool nz;
          if (!pair_term<DIM, PERB>(g, tc, nt, hx, ht, val, dval, nz)) continue;
          acc += height * val * vol;
        }
        return acc;
      }
    }
    // The stencil offset of point s = lt + k * TPH is kept as a running mixed-radix number (digit d in base
    // 2 * msize[d] + 1) and advanced by the digits of TPH per point: the two 32-bit divisions per point that decoding s
    // afresh costs (~50 of a 3-D point's ~170 instructions) are paid once per thread.  Same points, same order.
    unsigned dig[DIM], inc[DIM], wd[DIM];
    {
      unsigned r0 = (unsigned)lt, r1 = (unsigned)TPH;
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        wd[d] = (unsigned)(2 * g.msize[d] + 1);
        if (d < DIM - 1) {
          dig[d] = r0 % wd[d];
          r0 /= wd[d];
          inc[d] = r1 % wd[d];
          r1 /= wd[d];
        } else {
          dig[d] = r0;   // (the last digit is not reduced: s < utotal bounds it)
          inc[d] = r1;
        }
      }
    }
    for (unsigned s0 = (unsigned)lt; s0 < utotal; s0 += ILP * TPH) {
      double term[ILP];
#pragma unroll
      for (int u = 0; u < ILP; u++) {
        term[u] = 0;
        const unsigned s = s0 + (unsigned)(u * TPH);
        unsigned offd[DIM];
        {
          // this point's digits, then the state moves on to the next point (whether or not this one is used)
          unsigned carry = 0;
#pragma unroll
          for (int d = 0; d < DIM; d++) {
            offd[d] = dig[d];
            unsigned nd = dig[d] + inc[d] + carry;
            carry = 0;
            if (d < DIM - 1 && nd >= wd[d]) {
              nd -= wd[d];
              carry = 1;
            }
            dig[d] = nd;
          }
        }
        if (s >= utotal) continue;
        int p[DIM];
        bool skip = false;
        double dp2_est = 0;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          const int off = (int)offd[d];
          int idx = off - g.msize[d] + c[d];
          {
            // distance to this (un-wrapped) stencil point; most of the stencil box lies outside dp2 < 8
            const double e = ((g.min[d] + idx * g.dx[d]) - hx[d]) * tc.inv_sigma[d];
            dp2_est += e * e;
          }
          if (idx >= g.n[d]) {
            if (g.periodic[d]) idx %= g.n[d]; else skip = true;
          }
          if (idx < 0) {
            if (g.periodic[d]) idx += g.n[d]; else skip = true;
            if (idx < 0) skip = true;  // reference: undefined (stencil wider than two grid lengths)
          }
          p[d] = idx;
        }
        if (skip) continue;
        if (ball_ok && dp2_est > 8.0 * (1.0 + 1e-6)) continue;  // conservative: the exact test is in pair_term (:294)
        NodeTerms<DIM> nt;
        node_terms<DIM, PERB>(g, t, p, nt);
        if (!nt.inside) continue;
        double val, dval[DIM];
        bool nz;
        if (!pair_term<DIM, PERB>(g, tc, nt, hx, ht, val, dval, nz)) continue;
        term[u] = height * val * vol;
      }
#pragma unroll
      for (int u = 0; u < ILP; u++) acc += term[u];
    }
  }
  return acc;
}

// The limiter's result to a 64-byte line of its own in host-mapped memory (LimitArgs::fast_line) as ONE write: lanes 0..7
// of the calling wave (which all hold the same result) store 8 bytes each with a single instruction --
//   [seq | cum_out | k, nh | n_tail, stop | n_deferred, error | all_plain | h2_stop | seq]
// -- the batch's sequence number in the FIRST and the LAST word, so that a reader who finds both has the line whole even
// if the write travelled as two halves (edm_header_line_decode in edm_kernels.h).
__device__ __forceinline__ void header_line_to_host(unsigned long long *host_line, const LimitResult &r, unsigned long long seq) {
  const int lane = threadIdx.x & 63;
  unsigned long long piece = seq;   // lanes 0 and 7
  if (lane == 1) piece = (unsigned long long)__double_as_longlong(r.cum_out);
  if (lane == 2) piece = (unsigned long long)(unsigned)r.k | ((unsigned long long)(unsigned)r.nh << 32);
  if (lane == 3) piece = (unsigned long long)(unsigned)r.n_tail | ((unsigned long long)(unsigned)r.stop << 32);
  if (lane == 4) piece = (unsigned long long)(unsigned)r.n_deferred | ((unsigned long long)(unsigned)r.error << 32);
  if (lane == 5) piece = (unsigned long long)(unsigned)r.all_plain;
  if (lane == 6) piece = (unsigned long long)__double_as_longlong(r.h2_stop);
  if (lane < 8) __hip_atomic_store(host_line + lane, piece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The serial stage behind the per-hill integrals, run by ONE workgroup once all of them are published: ordered
// limiter (wave 0), read-back region to host-mapped memory, release of the host.  n_true_known >= 0: the caller
// knows the batch's true hill count (else it is read from h.nh_dev).
template <int DIM, int NT>
__device__ __forceinline__ void limiter_stage(const HillList &h, const double *__restrict__ heights, double h_const,
                                              double *__restrict__ added, const LimitArgs &la, unsigned bid,
                                              long long n_true_known, const double *s_added = nullptr) {
  // wave 0 walks the limiter; with a read-back region (la.rb_dst) it stores its outputs to the device region and
  // to its host-mapped copy alike, while the other waves copy what does not depend on the limiter -- per-hill
  // bias and positions, by the true hill count -- so nothing is left to read back once the limiter is done.
  // k_integrals_gather (la.ready_flag): the gather workgroups of the same launch wait for the limiter, so its
  // outputs go to the device region only (agent scope), the word the gather polls follows as soon as wave 0's
  // stores are acknowledged -- no host round trip in front of it -- and the copy to the host comes after.
  const bool concurrent = la.ready_flag != nullptr;
  const long long mirror = (la.rb_dst && !concurrent) ? (long long)(la.rb_dst - la.rb_src) : 0;
  const long long nb = h.nh;   // the layout is sized by the launch bound
  const long long n_true = n_true_known >= 0 ? n_true_known : (h.nh_dev ? *h.nh_dev : nb);   // (already on its way for hill_count(): no new round trip)
  long long na = n_true;
  if (na > nb) na = 0;         // (bound exceeded: the limiter reports it, nothing is read)
  if (threadIdx.x < 64) {
    long long k_first = 0;
    int err = 0;
    LimitResult rl;
    limit_wave<true>(h.nh, added, heights, h_const, la.limit, la.cum_in, la.flush_mode, la.tail, la.res, 0, nullptr,
               nullptr, h.nh_dev, mirror, &k_first, &err, h.nh_dev ? n_true : -1, &rl, s_added);
    if (la.fast_line) header_line_to_host(la.fast_line, rl, la.done_seq);
    if (concurrent) {
      // the word carries what every gather workgroup needs first -- the error code and k, the first hill of the
      // ordered tail -- so that seeing it is all the waiting workgroups have to do when there is no tail
      __builtin_amdgcn_s_waitcnt(0);
      if (threadIdx.x == 0) ready_publish(la.ready_flag, ready_word(la.ready_seq, EDM_READY_FINAL | err, k_first));
      if (la.trace && threadIdx.x == 0) la.trace[(size_t)bid * 8 + 3] = wall_clock64();
    }
  } else if (concurrent && la.early_word && threadIdx.x >= NT - 64) {
    // the last wave, beside the limiter's wave and the two that copy: does the batch stay below the limit
    // whatever the order of the adds?
    if (na <= 64 * 64 && la.cum_in >= 0) {
      double part = 0;
      for (long long i = threadIdx.x - (NT - 64); i < na; i += 64) part += fabs(s_added ? s_added[i] : acquire(&added[i]));
      part = wave_sum(part);
      if (threadIdx.x == NT - 64 && n_true <= nb && (la.cum_in + part) * (1.0 + 1e-9) < la.limit)
        ready_publish(la.ready_flag, ready_word(la.ready_seq, EDM_READY_BELOW, na));   // (cannot displace the limiter's own word)
    }
  } else if (la.rb_dst) {
    readback_copy_hills<DIM>(la.rb_src, la.rb_dst, nb, na, (int)threadIdx.x - 64, (concurrent && la.early_word) ? NT - 128 : NT - 64,
                             s_added);
  }
  if (la.rb_dst) {
    if (concurrent) {
      __syncthreads();
      readback_copy_limiter<DIM>(la.rb_src, la.rb_dst, nb, na, (int)threadIdx.x, NT);
    }
    __builtin_amdgcn_s_waitcnt(0);   // every wave's stores into the host-mapped region have been acknowledged
    __syncthreads();
    // The flag is a RELAXED system-scope store on purpose.  A release at system scope would first write back the
    // XCD's whole L2 (~35 us measured, the cost this design exists to avoid).  Ordering rests on the hardware
    // instead: every store into the region above was itself a system-scope (write-through, uncached) store to
    // host memory; s_waitcnt(0) + the barrier mean each wave has its acknowledgements; PCIe posted writes of one
    // requester are not reordered, so the flag cannot overtake the data on the way to host memory.  The host
    // reads the flag (volatile) and then the data behind an acquire fence.  Put to the test by
    // k_flag_order_stress / edm_hip_debug_flag_order_stress (this protocol on a region whose every word is the
    // launch's number: tests/test_gpu_edge_cases.py runs 20 000 launches and requires zero words older than their
    // flag); the polled word is also checked against byte-identical results of the stream-wait path (EDM_HIP_POLL=0)
    // in test_polled_completion_equals_stream_wait, and a poll that does not see its word within 2 ms falls back to
    // hipStreamSynchronize.
    if (threadIdx.x == 0 && la.done_flag)
      __hip_atomic_store(la.done_flag, la.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// `la` (TPH == BLOCK only): the ordered limiter is chained onto the last workgroup to finish
// (bid: this workgroup's index among the integrals workgroups -- blockIdx.x, except inside k_integrals_gather)
// returns true (workgroup-uniform) in the workgroup that ran the chained limiter, once its results are out
template <int DIM, int TPH, bool PERB>
__device__ __forceinline__ bool hill_integrals_body(const Geom &g, const Tables &t, const HillList &h,
                                                    const double *__restrict__ heights, double h_const,
                                                    double *__restrict__ added, const LimitArgs &la, unsigned bid,
                                                    unsigned nwg) {
  constexpr int NT = (TPH > BLOCK) ? TPH : BLOCK;  // workgroup size: 512 / 1024 threads per hill for the 2-D / 3-D stencil
  __shared__ double s_red[NT / 64];
  const int lane = threadIdx.x & 63;
  const int lt = threadIdx.x % TPH;
  const long long hill = (long long)bid * (NT / TPH) + (threadIdx.x / TPH);
  // The hill's fields are requested BEFORE the hill count is known (the arrays hold h.nh entries, the launch bound:
  // an entry beyond the true count is stale and masked below): one memory round trip instead of three dependent
  // ones (count -> centre node -> the other fields).
  int c_r[DIM];
  double hx_r[DIM], ht_r[2 * DIM];
  double height_r = h_const;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    c_r[d] = INT_MIN;
    hx_r[d] = 0;
    ht_r[2 * d] = ht_r[2 * d + 1] = 0;
  }
  if (hill < h.nh) {
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      c_r[d] = h.hc[hill * DIM + d];
      hx_r[d] = h.hx[hill * DIM + d];
      ht_r[2 * d] = h.ht[hill * 2 * DIM + 2 * d];
      ht_r[2 * d + 1] = h.ht[hill * 2 * DIM + 2 * d + 1];
    }
    if (heights) height_r = heights[hill];
  }
  const long long nh_eff = hill_count(h);
  if (TPH == 64 && hill >= nh_eff) return false;  // (a whole workgroup shares one hill when TPH == BLOCK)
  const bool live = hill < nh_eff;
  // chained limiter (a workgroup per hill, launched against a bound on the hill count): only the workgroups that
  // own a hill take a ticket -- workgroup 0 alone when there is none -- so the last arrival is one of a few hundred
  // and a single counter (one atomic round trip) does
  const unsigned ticket_blocks = (unsigned)(nh_eff > 0 ? nh_eff : 1);
  // ... or none at all (LimitArgs::tagged): the launch is wider than the hill count, so its first workgroup WITHOUT a
  // hill can be the limiter's -- the hills' workgroups store {integral, batch number} as one 16-byte store and are
  // done; that workgroup polls the slots and runs the limiter on what it has read.  Between the last integral and the
  // limiter lies one round trip (store -> poll) instead of three (store acknowledged -> ticket -> reload).
  const bool tagged = TPH != 64 && la.enabled && la.tagged != nullptr && nh_eff >= 1 && nh_eff < (long long)nwg &&
                      nh_eff <= EDM_TAG_CAP;
  if constexpr (TPH != 64) {
    if (tagged && bid == (unsigned)nh_eff) {
      __shared__ double s_val[EDM_TAG_CAP];
      const int n = (int)nh_eff;
      constexpr int PER = (EDM_TAG_CAP + NT - 1) / NT;
      bool have[PER];
#pragma unroll
      for (int q = 0; q < PER; q++) have[q] = !((int)threadIdx.x + q * NT < n);
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < PER; q++) {
          if (have[q]) continue;
          const int i = (int)threadIdx.x + q * NT;
          double v;
          if (load_tagged_agent(la.tagged, i, la.tag_seq, &v)) {
            s_val[i] = v;
            publish(&added[i], v);   // (the device array later launches and the read-back read)
            have[q] = true;
          } else {
            ok = false;
          }
        }
        if (__syncthreads_and(ok ? 1 : 0)) break;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 1000000000ull) __builtin_trap();   // 10 s at 100 MHz: never, short of a lost workgroup
      }
      if (la.trace && threadIdx.x == 0) la.trace[(size_t)bid * 8 + 2] = wall_clock64();
      limiter_stage<DIM, NT>(h, heights, h_const, added, la, bid, -1, s_val);
      if (la.trace && threadIdx.x == 0) la.trace[(size_t)bid * 8 + 4] = wall_clock64();
      __syncthreads();
      return true;
    }
  }
  if (TPH != 64 && la.enabled && (tagged ? bid > (unsigned)nh_eff : bid >= ticket_blocks)) return false;
  TermConst<DIM> tc;
  term_const<DIM>(g, tc);
  unsigned long long *wtrace = (TPH != 64 && la.enabled && la.trace) ? la.trace + (size_t)bid * 8 : nullptr;
  if (wtrace && threadIdx.x == 0) wtrace[5] = wall_clock64();
  const double acc_part = hill_stencil_partial<DIM, TPH, PERB>(g, t, tc, c_r, hx_r, ht_r, height_r, live, lt);
  double acc = acc_part;
  acc = wave_sum(acc);
  if (TPH == 64) {
    if (lane == 0) added[hill] = acc;
  } else {
    if (lane == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && live) {
      double r = 0;
      for (int w = 0; w < NT / 64; w++) r += s_red[w];
      if (tagged) store_tagged_agent(la.tagged, hill, r, la.tag_seq);
      else if (la.enabled) publish(&added[hill], r);
      else added[hill] = r;
    }
    if (tagged) {
      if (la.trace && threadIdx.x == 0) la.trace[(size_t)bid * 8 + 1] = wall_clock64();
      return false;
    }
    if (la.enabled) {
      if (la.trace && threadIdx.x == 0) la.trace[(size_t)bid * 8 + 1] = wall_clock64();
      if (!last_block_done(la.ticket, ticket_blocks, bid, ticket_blocks <= 512)) return false;
      if (la.trace && threadIdx.x == 0) la.trace[(size_t)bid * 8 + 2] = wall_clock64();
      limiter_stage<DIM, NT>(h, heights, h_const, added, la, bid, -1);
      if (la.trace && threadIdx.x == 0) la.trace[(size_t)bid * 8 + 4] = wall_clock64();
      __syncthreads();
      return true;
    }
  }
  return false;
}
template <int DIM>
__device__ __forceinline__ void mark_tiles_body(const Geom &g, const HillList &h, int *__restrict__ flags,
                                                int *__restrict__ list, long long ntiles, int parity, long long id);
// (workgroups [0, nb_int): the integrals; the rest, if any: the tile list of the culled gather that follows -- MarkArgs)
template <int DIM, int TPH, bool PERB>
__global__ void __launch_bounds__((TPH > BLOCK) ? TPH : BLOCK) k_hill_integrals(Geom g, Tables t, HillList h,
                                                                                const double *__restrict__ heights,
                                                                                double h_const,
                                                                                double *__restrict__ added,
                                                                                LimitArgs la, MarkArgs mk, unsigned nb_int) {
  if (blockIdx.x >= nb_int) {
    constexpr int NT = (TPH > BLOCK) ? TPH : BLOCK;
    mark_tiles_body<DIM>(g, h, mk.flags, mk.list, mk.ntiles, mk.parity, (long long)(blockIdx.x - nb_int) * NT + threadIdx.x);
    return;
  }
  if (la.trace && threadIdx.x == 0) la.trace[(size_t)blockIdx.x * 8] = wall_clock64();
  (void)hill_integrals_body<DIM, TPH, PERB>(g, t, h, heights, h_const, added, la, blockIdx.x, nb_int);
  if (la.trace && threadIdx.x == 0) la.trace[(size_t)blockIdx.x * 8 + 7] = wall_clock64();
}

bool hill_integrals_can_chain_limit(long long nh) { return nh > 0 && nh <= 2048; }

hipError_t launch_hill_integrals(const Geom &g, const Tables &t, const HillList &h, const double *heights,
                                 double h_const, double *added, hipStream_t s, const LimitArgs *chain,
                                 const MarkArgs *mark) {
  if (h.nh <= 0) return hipSuccess;
  MarkArgs mk;
  memset(&mk, 0, sizeof(mk));
  if (mark) {
    if (h.nh > 2048 || g.dim < 2 || !mark->flags || !mark->list) return hipErrorInvalidValue;
    mk = *mark;
  }
  LimitArgs la;
  memset(&la, 0, sizeof(la));
  if (chain) {
    if (!hill_integrals_can_chain_limit(h.nh)) return hipErrorInvalidValue;
    la = *chain;
    la.enabled = 1;
  }
  bool perb = true;
  for (int d = 0; d < g.dim; d++)
    if (!g.bper[d]) perb = false;
#define EDM_INTEGRALS(D, TPHV, NTV, NB)                                                                            \
  do {                                                                                                             \
    if (perb)                                                                                                      \
      hipLaunchKernelGGL((k_hill_integrals<D, TPHV, true>), dim3((NB) + nmark(NTV)), dim3(NTV), 0, s, g, t, h,    \
                         heights, h_const, added, la, mk, (unsigned)(NB));                                        \
    else                                                                                                           \
      hipLaunchKernelGGL((k_hill_integrals<D, TPHV, false>), dim3((NB) + nmark(NTV)), dim3(NTV), 0, s, g, t, h,   \
                         heights, h_const, added, la, mk, (unsigned)(NB));                                        \
  } while (0)
  // marking workgroups behind the integrals' (which wait for nobody and are what the step waits for)
  const long long mark_threads = mark ? mark_tiles_threads(g, h.nh) : 0;
  auto nmark = [mark_threads](int nt) { return (unsigned)((mark_threads + nt - 1) / nt); };
  if (h.nh <= 2048) {
    const unsigned nb = (unsigned)h.nh;
    switch (g.dim) {
      case 1: EDM_INTEGRALS(1, BLOCK, BLOCK, nb); break;
      // (2-D / 3-D: the support's points are dealt out one or two per thread -- the walk is a serial instruction stream
      //  of ~300 dependent fp64 instructions per point, and a hill step waits for it)
      case 2: EDM_INTEGRALS(2, 2 * BLOCK, 2 * BLOCK, nb); break;
      default: EDM_INTEGRALS(3, 4 * BLOCK, 4 * BLOCK, nb); break;
    }
  } else {
    const unsigned nb = (unsigned)((h.nh + (BLOCK / 64) - 1) / (BLOCK / 64));
    switch (g.dim) {
      case 1: EDM_INTEGRALS(1, 64, BLOCK, nb); break;
      case 2: EDM_INTEGRALS(2, 64, BLOCK, nb); break;
      default: EDM_INTEGRALS(3, 64, BLOCK, nb); break;
    }
  }
#undef EDM_INTEGRALS
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K5: tile-owned gather.  Each 256-thread workgroup owns one tile of nodes (one
// node per thread) and walks the hill list IN ORDER, so every node accumulates
// its contributions in exactly the order the reference's sequential add_value
// calls would -- no atomics, bit-reproducible, identical on every GPU of a
// replicated run.  Node-only terms (table reads, wall blends) are computed once
// per node and amortised over all hills.
// ---------------------------------------------------------------------------
template <int DIM> struct Tile;
template <> struct Tile<1> { static constexpr int T[3] = {256, 1, 1}; };
template <> struct Tile<2> { static constexpr int T[3] = {16, 16, 1}; };
template <> struct Tile<3> { static constexpr int T[3] = {8, 8, 4}; };

__host__ __device__ inline long long floordiv(long long a, long long b) {
  long long q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
  return q;
}
__host__ __device__ inline long long ceildiv(long long a, long long b) { return -floordiv(-a, b); }

// number of stencil offsets of a hill centred at node c that land on node range
// [p0, p1] of dimension d under the reference's wrap rules (:251-266): index
// c+o with o in [-m, m]; >= n wraps by %, < 0 wraps by a single +n.
__host__ __device__ inline int images(const Geom &g, int d, int c, int p0, int p1) {
  const long long n = g.n[d], m = g.msize[d];
  long long lo = (long long)c - m, hi = (long long)c + m;
  if (!g.periodic[d]) return (lo <= p1 && hi >= p0) ? 1 : 0;
  if (2 * m + 1 <= n && c >= 0 && c < n && p0 >= 0 && p1 < n) {
    // the usual case (stencil narrower than the grid): only the images k = -1, 0, +1 can land on
    // [p0, p1], so the count needs no 64-bit divisions (this sits in the gather's inner loop)
    int cnt = 0;
#pragma unroll
    for (int k = -1; k <= 1; k++)
      if (lo <= p1 + k * n && hi >= p0 + k * n) cnt++;
    return cnt;
  }
  if (lo < -n) lo = -n;
  const long long kmin = ceildiv(lo - p1, n), kmax = floordiv(hi - p0, n);
  return kmax >= kmin ? (int)(kmax - kmin + 1) : 0;
}

long long gather_tiles(const Geom &g) {
  long long t = 1;
  for (int d = 0; d < g.dim; d++) {
    const int T = (g.dim == 1) ? Tile<1>::T[d] : (g.dim == 2) ? Tile<2>::T[d] : Tile<3>::T[d];
    t *= (g.n[d] + T - 1) / T;
  }
  return t;
}

// device-side twin of gather_tiles()
__device__ __forceinline__ long long gather_tiles_dev(const Geom &g) {
  long long t = 1;
  for (int d = 0; d < g.dim; d++) {
    const int T = (g.dim == 1) ? Tile<1>::T[d] : (g.dim == 2) ? Tile<2>::T[d] : Tile<3>::T[d];
    t *= (g.n[d] + T - 1) / T;
  }
  return t;
}

// slot of tile (by its per-dimension tile coordinates) among the tiles a hill centred at c overlaps
template <int DIM>
__device__ __forceinline__ int hill_tile_slot(const Geom &g, const int *c, const int *tile_coord) {
  int slot = 0, mul = 1;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    const int T = Tile<DIM>::T[d];
    const int nt = (g.n[d] + T - 1) / T;
    const int m = g.msize[d];
    int S = (2 * m) / T + 3;
    if (S > nt) S = nt;
    int sd;
    if (g.periodic[d]) {
      const int first = (int)((((long long)c[d] - m) % g.n[d] + g.n[d]) % g.n[d]);
      sd = (tile_coord[d] - first / T + nt) % nt;
    } else {
      const int first = (c[d] - m > 0) ? c[d] - m : 0;
      sd = tile_coord[d] - first / T;
    }
    if (sd < 0) sd = 0;
    if (sd > S - 1) sd = S - 1;
    slot += sd * mul;
    mul *= S;
  }
  return slot;
}

// MODE 0: heights from (base | limiter tail), in place (groups == 1) or per-group partials
// MODE 1: fused -- base heights only, always into partials, per-(hill, tile) integral pieces into slots
// MODE 2: correction -- only the limiter's tail hills, heights (tail_h1 - base, tail_h2), into partial[groups]
struct DupPlan {
  unsigned long long lo[3], hi[3];
};
__device__ __forceinline__ void duplicate_boundary_block(const Geom &g, double *__restrict__ rec, const DupPlan &dp);
__device__ __forceinline__ void duplicate_boundary_wave(const Geom &g, double *__restrict__ rec, const DupPlan &dp, int lane);
static DupPlan make_dup_plan(const Geom &g);
template <int DIM>
__device__ __forceinline__ void hist_batch(const Geom &hg, double *hist, long long nh, const double *hx0,
                                           const LimitResult *res, const int *flags, int flush_mode, long long first,
                                           long long stride);
// bookkeeping chained onto the last gather workgroup (see last_block_done)
struct PostArgs {
  int enabled;
  int skip_hist;   // the histogram is updated elsewhere (k_integrals_gather: by the limiter's workgroup)
  int *ticket;
  DupPlan dp;
  Geom hg;
  double *hist;
  const int *flags;
  int flush_mode;
  const char *rb_src;
  char *rb_dst;
  long long rb_bytes;
  // tile_ticket_duplicate: tiles [0, tk_lo_end) and [tk_hi_begin, ntile) take the ticket (both zero: every tile)
  unsigned tk_lo_end, tk_hi_begin;
};
static void dup_ticket_tiles_1d(const Geom &g, PostArgs &post, unsigned ntile, unsigned tile_nodes);
template <bool PERB>
__device__ __forceinline__ void tile_ticket_duplicate(const Geom &g, double *__restrict__ rec, const PostArgs &post,
                                                      unsigned ntile, unsigned tile, const int *s_dirty, int *s_last);

// boundary duplication (K6) and the histogram updates (K7) by the last gather workgroup to finish
// (a few hundred workgroups that finish spread over microseconds: one counter, one atomic round trip -- the
//  two-level ticket costs the last arrival two)
template <int DIM, bool PERB>
__device__ __forceinline__ void gather_post(const Geom &g, double *__restrict__ rec, const HillList &h, const HillHeights &hh,
                                            int *__restrict__ dirty_flag, const PostArgs &post, unsigned nblocks, unsigned id) {
  if (!last_block_done(post.ticket, nblocks, id, nblocks <= 512)) return;
  // the three chores are independent: the waves of the workgroup split them (wave 0 the boundary copies --
  // at most 4^DIM = 64, one per lane -- the lower half of the rest the histogram, the upper half the read-back)
  constexpr int NW = BLOCK / 64;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  if (wv == 0) {
    if (!PERB && acquire(dirty_flag) != 0) {   // (no walls, no boundary corrections, nothing to duplicate)
      duplicate_boundary_wave(g, rec, post.dp, ln);
      if (ln == 0) *dirty_flag = 0;
    }
  } else if (wv <= (NW - 1) / 2) {
    const int nh_w = (NW - 1) / 2;   // waves 1 .. nh_w
    if (!post.skip_hist && !hh.res_dev->error)
      hist_batch<DIM>(post.hg, post.hist, h.nh, h.hx0, hh.res_dev, post.flags, post.flush_mode, (wv - 1) * 64 + ln,
                      nh_w * 64);
  } else if (post.rb_dst) {
    // read-back region (written by the earlier launches of the step) -> host-mapped memory; only the part
    // the batch's true hill count fills: header + flags | h2 | added2 | added | positions (apply_hills' layout)
    const int first = (NW - 1) / 2 + 1, nthr = (NW - first) * 64, me = (wv - first) * 64 + ln;
    const long long nb = h.nh;                                  // the layout is sized by the launch bound
    long long na = hh.res_dev->nh < nb ? hh.res_dev->nh : nb;   // ... the hills are fewer
    if (hh.res_dev->error) na = 0;
    readback_copy<DIM>(post.rb_src, post.rb_dst, nb, na, me, nthr);
  }
}

// PARTS (1 or 8; 8 on the 1-D grid only): the tile is BLOCK / PARTS nodes wide and thread (part, node)
// accumulates every PARTS-th batch of the tile's hill list for its node; the parts are combined in LDS in a
// fixed order.  A node's serial chain is its number of overlapping hills, and a tile's work lands on ONE CU:
// on the 1-D grid hills pile up where the pair density is high (40 per 256-node tile at r ~ 2.7 against 10 at
// r ~ 1.3; timestamps: the dense tiles finished at 11.5 us, the sparse ones at 4), so 32-node tiles with eight
// hill-parts each spread the same work over eight times as many CUs (four parts: dense tiles at 8.2 us, sparse
// at 3.5; eight: W1 step 39.5 -> 37.9 us together with the flat ticket; sixteen: no further gain).
template <int DIM, int PARTS>
__device__ __forceinline__ constexpr int tile_extent(int d) {
  return (DIM == 1 && d == 0) ? BLOCK / PARTS : Tile<DIM>::T[d];
}
// DEFER (k_integrals_gather: 1-D, in place, the limiter running in OTHER workgroups of the same launch): the
// first chunk of the hill list is staged without heights, its stencil terms -- the exp-heavy part, which does not
// depend on the limiter -- are computed and parked in LDS, and only then does the workgroup wait for the limiter's
// word (ready_flag == ready_seq), fetch k and the tail heights and accumulate.  Hills the limiter deferred (height 0)
// are skipped at that point instead of at staging.
// (s_dirty, optional: boundary corrections are noted in that LDS word of the workgroup instead of the device flag.)
template <int DIM, int MODE, int PARTS, bool PERB, bool DEFER = false>
__device__ __forceinline__ void hill_gather_body(const Geom &g, const Tables &t, double *__restrict__ rec,
                                                 const HillList &h, const HillHeights &hh, const GatherPlan &plan,
                                                 int use_list, int *__restrict__ dirty_flag, int coherent,
                                                 long long tile, const unsigned long long *ready_flag = nullptr,
                                                 unsigned long long ready_seq = 0, unsigned long long *trace = nullptr,
                                                 int *s_dirty = nullptr) {
  static_assert(!DEFER || (DIM == 1 && MODE == 0), "deferred heights: the 1-D in-place gather only");
  constexpr int R = (DIM == 1) ? 2 : 4;
  constexpr int NODES = BLOCK / PARTS;
  const int tnode = threadIdx.x % NODES;   // this thread's node within the tile
  const int part = threadIdx.x / NODES;    // ... and its share of the hill batches
  if (use_list && threadIdx.x == 0) plan.tile_flags[tile] = 0;  // (k_mark_tiles relies on an all-zero flag array)
  // tile origin and this thread's node
  int t0[DIM], p[DIM], tcoord[DIM];
  {
    long long rest = tile;
    int lrest = tnode;
    // 2-D / 3-D: a WAVE owns a compact block of the tile -- 8 x 8 of the 16 x 16 nodes, 4 x 4 x 4 of the 8 x 8 x 4 --
    // instead of 64 consecutive nodes (a 16 x 4 strip, an 8 x 8 x 1 slab): a wave runs a hill's terms whenever ONE of
    // its nodes lies inside the hill's support, and a compact block meets fewer supports than a flat one (by the
    // blocks' Minkowski sums with the support ball: 84 instead of 108 wave-visits per 3-D hill of W4).  Which thread
    // owns which node changes nothing in a node's sum.
    int loc[3] = {0, 0, 0};
    if (DIM == 2 && PARTS == 1 && plan.compact_waves) {
      const int w = tnode >> 6, l = tnode & 63;
      loc[0] = (l & 7) + 8 * (w & 1);
      loc[1] = (l >> 3) + 8 * (w >> 1);
    } else if (DIM == 3 && PARTS == 1 && plan.compact_waves) {
      const int w = tnode >> 6, l = tnode & 63;
      loc[0] = (l & 3) + 4 * (w & 1);
      loc[1] = ((l >> 2) & 3) + 4 * (w >> 1);
      loc[2] = l >> 4;
    }
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      const int T = tile_extent<DIM, PARTS>(d);
      const int nt_d = (g.n[d] + T - 1) / T;
      tcoord[d] = (int)(rest % nt_d);
      t0[d] = tcoord[d] * T;
      rest /= nt_d;
      p[d] = t0[d] + ((DIM > 1 && PARTS == 1 && plan.compact_waves) ? loc[d] : (lrest % T));
      lrest /= T;
    }
  }
  static_assert(DIM == 1 || (BLOCK == 256 && Tile<2>::T[0] == 16 && Tile<2>::T[1] == 16 && Tile<3>::T[0] == 8 &&
                             Tile<3>::T[1] == 8 && Tile<3>::T[2] == 4), "the waves' blocks assume these tiles");
  bool active = true;
#pragma unroll
  for (int d = 0; d < DIM; d++)
    if (p[d] >= g.n[d]) active = false;
  long long flat = 0;
  if (active) {
    flat = p[DIM - 1];
#pragma unroll
    for (int d = DIM - 1; d > 0; d--) flat = flat * g.n[d - 1] + p[d - 1];
  }
  const bool in_grid0 = active;   // (before the boundary test below)
  NodeTerms<DIM> nt;
  bool node_interior = true;
  if (active) {
    node_terms<DIM, PERB>(g, t, p, nt);
    if (!nt.inside) active = false;
#pragma unroll
    for (int d = 0; d < DIM; d++)
      if (!PERB && !g.bper[d] && (nt.t2[d] != 0 || nt.t4[d] != 0 || nt.t6[d] != 0 || nt.t7[d] != 0)) node_interior = false;
  }
  // wave-uniform: away from the walls (95 % of the C1D nodes) the McGovern-De Pablo blends vanish
  const bool interior = !PERB && __all(node_interior || !active);
  // (limiter result and hill count in one round trip: the kernel is a chain of dependent loads)
  long long k_first_tail = hh.k;
  long long nh_eff = h.nh;
  if (DEFER) {
    nh_eff = hill_count(h);   // (the selection's count; the limiter's result is read after the wait below)
    k_first_tail = nh_eff;
  } else if (hh.res_dev) {
    const int err = hh.res_dev->error;
    const long long kk = hh.res_dev->k, nn = hh.res_dev->nh;
    if (err) return;  // limiter overflow / bound exceeded: the host handles it, nothing is applied
    k_first_tail = kk - hh.tail_shift;  // (may be negative: the ordered tail began before this slice)
    if (nn - hh.tail_shift < nh_eff) nh_eff = (nn - hh.tail_shift > 0) ? nn - hh.tail_shift : 0;
  } else {
    nh_eff = hill_count(h);
    if (k_first_tail > nh_eff) k_first_tail = nh_eff;
  }
  const int G = (MODE == 0 && plan.adaptive) ? adaptive_groups(plan.groups, nh_eff) : plan.groups;
  if (MODE == 0 && plan.adaptive && (int)blockIdx.y >= G) return;
  const int grp = (MODE == 2) ? G : blockIdx.y;   // the correction owns the extra partial buffer
  const long long per = (nh_eff + G - 1) / G;
  long long hbeg = per * blockIdx.y;
  long long hend = (hbeg + per < nh_eff) ? hbeg + per : nh_eff;
  if (MODE == 2) {
    hbeg = (k_first_tail > 0) ? k_first_tail : 0;
    hend = nh_eff;
  }
  const bool in_place = (MODE == 0) && (G == 1);
  double vol = 1;
#pragma unroll
  for (int d = 0; d < DIM; d++) vol *= g.dx[d];

  double acc[1 + DIM];
  if (in_place && active && part == 0) {
    // in-place: start from the stored record so the adds follow the reference's
    // sequence V0 + h0*t0 + h1*t1 + ... exactly
#pragma unroll
    for (int j = 0; j <= DIM; j++) acc[j] = rec[flat * R + j];
  } else {
#pragma unroll
    for (int j = 0; j <= DIM; j++) acc[j] = 0;
  }
  bool any_corr = false;
  bool touched = false;  // some hill reached this node (an untouched node of an in-place gather is not rewritten)

  // The hill list is walked in order, a chunk of BLOCK hills at a time: every thread tests ONE hill
  // of the chunk against this workgroup's tile (coalesced loads instead of a dependent scalar-load
  // chain), the overlapping hills are compacted IN ORDER into LDS (ballot prefix), and all threads
  // then walk that short list together -- ILP entries at a time so independent exp chains overlap
  // -- accumulating strictly in list order: the sums are those of the sequential reference.
  __shared__ int s_c[BLOCK][DIM];
  __shared__ double s_x[BLOCK][DIM];
  __shared__ double s_t[PERB ? 1 : BLOCK][2 * DIM];  // (hill-side wall exponentials: none without walls)
  __shared__ double s_h1[BLOCK], s_h2[BLOCK];
  __shared__ int s_wcnt[BLOCK / 64];
  using id_t = long long;
  __shared__ id_t s_id[(MODE == 1 || DEFER) ? BLOCK : 1];
  // DEFER: parked stencil terms of the first chunk, NTS per thread (value, derivative, multiplicity | nz << 30)
  constexpr int DEFER_ILP = 4;
  // (a tile meets ~13 of the ~125 hills of a W1 step, 40 where the pairs are dense: 4 per part cover 32.  Eight cost
  //  20 KB more LDS and 30 registers -- two workgroups per CU instead of three -- and bought the W1 step nothing)
  constexpr int NTS = 4;
  using tm_t = int;
  constexpr int TM_NZ = 30;
  __shared__ double s_tv[DEFER ? NTS : 1][DEFER ? BLOCK : 1], s_td[DEFER ? NTS : 1][DEFER ? BLOCK : 1];
  __shared__ tm_t s_tm[DEFER ? NTS : 1][DEFER ? BLOCK : 1];
  bool waited = !DEFER;   // (block-uniform) the limiter's result is known
  __shared__ double s_wpart[(MODE == 1) ? BLOCK / 64 : 1][(MODE == 1) ? BLOCK : 1];
  __shared__ double s_pacc[(PARTS > 1) ? PARTS - 1 : 1][(PARTS > 1) ? NODES : 1][1 + DIM];
  __shared__ int s_ptouch[(PARTS > 1) ? PARTS - 1 : 1][(PARTS > 1) ? NODES : 1];
  TermConst<DIM> tc;
  term_const<DIM>(g, tc);
  // launch-uniform: no dimension whose stencil can reach a node through more than one periodic image
  bool single_image = true;
#pragma unroll
  for (int d = 0; d < DIM; d++)
    if (g.periodic[d] && 2 * g.msize[d] + 1 > g.n[d]) single_image = false;
  constexpr int ILP = (DIM == 1) ? 4 : (DIM == 2) ? 2 : 1;  // (2-D/3-D tiles meet one to three hills of a sparse batch; fewer live registers -- 3-D: 128, four waves per SIMD -- more workgroups per CU)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long base = hbeg; base < hend; base += BLOCK) {
    const long long cur = base + threadIdx.x;   // (every thread stages one hill of the chunk)
    const bool defer_chunk = DEFER && !waited;
    bool take = false;
    int c[DIM];
    double h1 = 0, h2 = 0;
    double hx_r[DIM], ht_r[2 * DIM];
    if (cur < hend) {
      // 1-D: all of this hill's fields are requested together (one memory round trip; a tile overlaps a
      // good part of the hills).  2-D/3-D: a tile meets a few hills out of hundreds, so only the centre
      // node is fetched for the test and the rest follows for the hills that pass.
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        c[d] = h.hc[cur * DIM + d];
        if (DIM == 1) {
          hx_r[d] = h.hx[cur * DIM + d];
          if (!PERB) {
            ht_r[2 * d] = h.ht[cur * 2 * DIM + 2 * d];
            ht_r[2 * d + 1] = h.ht[cur * 2 * DIM + 2 * d + 1];
          }
        }
      }
      const double hb = hh.h ? hh.h[cur] : hh.h_const;
      const bool in_tail = !defer_chunk && ((MODE == 2) || (MODE == 0 && cur >= k_first_tail));
      double th1 = 0, th2 = 0;
      if (in_tail) {
        th1 = DEFER ? acquire(&hh.tail_h1[cur - k_first_tail]) : hh.tail_h1[cur - k_first_tail];
        th2 = DEFER ? acquire(&hh.tail_h2[cur - k_first_tail]) : hh.tail_h2[cur - k_first_tail];
      }
      if (c[0] != INT_MIN) {
        take = true;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          const int T = tile_extent<DIM, PARTS>(d);
          int t1 = t0[d] + T - 1;
          if (t1 > g.n[d] - 1) t1 = g.n[d] - 1;
          if (images(g, d, c[d], t0[d], t1) == 0) take = false;
        }
        if (take && DIM > 1) {
#pragma unroll
          for (int d = 0; d < DIM; d++) {
            hx_r[d] = h.hx[cur * DIM + d];
            if (!PERB) {
              ht_r[2 * d] = h.ht[cur * 2 * DIM + 2 * d];
              ht_r[2 * d + 1] = h.ht[cur * 2 * DIM + 2 * d + 1];
            }
          }
        }
        if (take && !defer_chunk) {
          if (MODE == 1) {
            h1 = hb;
            h2 = 0;
          } else if (MODE == 2) {
            h1 = th1 - hb;  // what the fused pass added too much
            h2 = th2;
          } else if (!in_tail) {
            h1 = hb;
            h2 = 0;
          } else {
            h1 = th1;
            h2 = th2;
          }
          if (h1 == 0 && h2 == 0) take = false;  // nothing to add (deferred hill / unchanged hill)
        }
      }
    }
    const unsigned long long bal = __ballot(take);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int pos = __popcll(bal & ((1ull << lane) - 1ull));
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) {
      if (w < wave) pos += s_wcnt[w];
      cnt += s_wcnt[w];
    }
    if (take) {
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        s_c[pos][d] = c[d];
        s_x[pos][d] = hx_r[d];
        if (!PERB) {
          s_t[pos][2 * d] = ht_r[2 * d];
          s_t[pos][2 * d + 1] = ht_r[2 * d + 1];
        }
      }
      s_h1[pos] = h1;
      s_h2[pos] = h2;
      if (MODE == 1 || DEFER) s_id[pos] = (id_t)cur;
    }
    __syncthreads();
    if (DEFER && defer_chunk) {
      // 1. the stencil terms of this thread's first NTS staged hills, computed while the limiter still runs
      static_assert(!DEFER || ILP == DEFER_ILP, "parked-term layout");
      if (active) {
        int it = 0;
        for (int q0 = part * ILP; q0 < cnt && it < NTS / ILP; q0 += PARTS * ILP, it++) {
#pragma unroll
          for (int q = 0; q < ILP; q++) {
            double v = 0, dv[DIM];
#pragma unroll
            for (int d = 0; d < DIM; d++) dv[d] = 0;
            int m = 0;
            bool nz = false;
            if (q0 + q < cnt) {
              m = 1;
#pragma unroll
              for (int d = 0; d < DIM; d++) m *= images(g, d, s_c[q0 + q][d], p[d], p[d]);
              if (!(m > 0 && pair_term<DIM, PERB>(g, tc, nt, s_x[q0 + q], s_t[PERB ? 0 : q0 + q], v, dv, nz, interior))) m = 0;
            }
            s_tv[it * ILP + q][threadIdx.x] = v;
            s_td[it * ILP + q][threadIdx.x] = dv[0];
            s_tm[it * ILP + q][threadIdx.x] = (tm_t)(m | (nz ? (1 << TM_NZ) : 0));
          }
        }
      }
      // 2. the limiter's word (its workgroups were dispatched ahead of this one and wait for nobody)
      if (trace && threadIdx.x == 0) trace[1] = wall_clock64();
      __shared__ unsigned long long s_word;
      if (threadIdx.x == 0) s_word = wait_for_word(ready_flag, ready_seq, false);
      __syncthreads();
      if (trace && threadIdx.x == 0) trace[2] = wall_clock64();
      waited = true;
      const unsigned long long word = s_word;
      const int state = ready_state_of(word);
      if (state == EDM_READY_BELOW) {
        k_first_tail = nh_eff;   // no hill is touched by the limiter: base heights throughout
      } else {
        if (state & ~EDM_READY_FINAL) return;  // limiter overflow / bound exceeded: the host handles it, nothing is applied
        k_first_tail = ready_k_of(word);
      }
      // 3. heights of the staged hills
      if ((int)threadIdx.x < cnt) {
        const long long id = s_id[threadIdx.x];
        double a1 = hh.h ? hh.h[id] : hh.h_const, a2 = 0;
        if (id >= k_first_tail) {
          a1 = acquire(&hh.tail_h1[id - k_first_tail]);
          a2 = acquire(&hh.tail_h2[id - k_first_tail]);
        }
        if (id >= nh_eff) a1 = a2 = 0;
        s_h1[threadIdx.x] = a1;
        s_h2[threadIdx.x] = a2;
      }
      __syncthreads();
    }
    if (active || MODE == 1) {
      int it_acc = 0;
      for (int q0 = part * ILP; q0 < cnt; q0 += PARTS * ILP, it_acc++) {
        double val[ILP], dval[ILP][DIM];
        int mult[ILP];
        bool nzq[ILP];
        const bool parked = DEFER && defer_chunk && it_acc < NTS / ILP;
#pragma unroll
        for (int q = 0; q < ILP; q++) {
          mult[q] = 0;
          nzq[q] = false;
          if (parked) {
            const int mm = s_tm[DEFER ? it_acc * ILP + q : 0][DEFER ? threadIdx.x : 0];
            val[q] = s_tv[DEFER ? it_acc * ILP + q : 0][DEFER ? threadIdx.x : 0];
            dval[q][0] = s_td[DEFER ? it_acc * ILP + q : 0][DEFER ? threadIdx.x : 0];
            mult[q] = mm & ((1 << TM_NZ) - 1);
            nzq[q] = (mm >> TM_NZ) != 0;
          } else if (active && q0 + q < cnt) {
            int m = 1;
#pragma unroll
            for (int d = 0; d < DIM; d++) m *= images(g, d, s_c[q0 + q][d], p[d], p[d]);
            bool nz = false;
            if (m > 0 && pair_term<DIM, PERB>(g, tc, nt, s_x[q0 + q], s_t[PERB ? 0 : q0 + q], val[q], dval[q], nz, interior)) {
              mult[q] = m;
              nzq[q] = nz;
            }
          }
          if (DEFER && mult[q] > 0 && s_h1[q0 + q] == 0 && s_h2[q0 + q] == 0) mult[q] = 0;  // a deferred / unchanged hill adds nothing
          if (mult[q] > 0) {
            any_corr |= nzq[q];
            touched = true;
          }
        }
#pragma unroll
        for (int q = 0; q < ILP; q++) {
          if (mult[q] > 0) {
            const double a1 = s_h1[q0 + q], a2 = s_h2[q0 + q];
            // (a stencil that wraps a small periodic grid more than once hits a node `mult` times: repeated adds,
            //  as the reference's loop makes them; everywhere else mult is 1 and the loops below are one add)
            const int reps = single_image ? 1 : mult[q];
            for (int rep = 0; rep < reps; rep++) {
              acc[0] += a1 * val[q];
#pragma unroll
              for (int d = 0; d < DIM; d++) acc[1 + d] += a1 * dval[q][d];
            }
            if (a2 != 0) {
              for (int rep = 0; rep < reps; rep++) {
                acc[0] += a2 * val[q];
#pragma unroll
                for (int d = 0; d < DIM; d++) acc[1 + d] += a2 * dval[q][d];
              }
            }
          }
        }
        if (MODE == 1) {
          // this tile's piece of each hill's integrated bias (gaussian_grid.h:349), fixed-order sums
          double piece[ILP];
#pragma unroll
          for (int q = 0; q < ILP; q++) {
            piece[q] = 0;
            if (mult[q] > 0) {
              const double term = s_h1[q0 + q] * val[q] * vol;
              const int reps = single_image ? 1 : mult[q];
              for (int rep = 0; rep < reps; rep++) piece[q] += term;
            }
          }
          if (ILP == 4) {
            // four wave sums for the price of ~1.2: after the first two butterfly steps each 16-lane group
            // carries ONE of the four hills (hill = 2 * (lane >= 32) + ((lane & 16) != 0)), the remaining four
            // steps run on a single value -- 7 shuffles instead of 24, the same fixed order for every launch
            const bool hi = (lane & 32) != 0, q16 = (lane & 16) != 0;
            double k0 = hi ? piece[2] : piece[0], s0 = hi ? piece[0] : piece[2];
            double k1 = hi ? piece[3] : piece[1], s1 = hi ? piece[1] : piece[3];
            k0 += __shfl_xor(s0, 32, 64);
            k1 += __shfl_xor(s1, 32, 64);
            double c = q16 ? k1 : k0;
            const double sd = q16 ? k0 : k1;
            c += __shfl_xor(sd, 16, 64);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            const int mine = (hi ? 2 : 0) + (q16 ? 1 : 0);
            if ((lane & 15) == 0 && q0 + mine < cnt) s_wpart[wave][q0 + mine] = c;
          } else {
#pragma unroll
            for (int q = 0; q < ILP; q++) {
              const double tot = wave_sum(piece[q]);
              if (lane == 0 && q0 + q < cnt) s_wpart[wave][q0 + q] = tot;
            }
          }
        }
      }
    }
    if (MODE == 1) {
      __syncthreads();
      if ((int)threadIdx.x < cnt) {
        double tot = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) tot += s_wpart[w][threadIdx.x];
        plan.slots[s_id[threadIdx.x] * plan.slots_per_hill + hill_tile_slot<DIM>(g, s_c[threadIdx.x], tcoord)] = tot;
      }
    }
    __syncthreads();
  }
  if (PARTS > 1) {
    // parts 1.. hand their sums to part 0, which adds them in part order
    if (part > 0) {
#pragma unroll
      for (int j = 0; j <= DIM; j++) s_pacc[part - 1][tnode][j] = acc[j];
      s_ptouch[part - 1][tnode] = touched ? 1 : 0;
    }
    __syncthreads();
    if (part > 0) {
      if (any_corr && active) {
        if (s_dirty) *s_dirty = 1; else if (coherent) publish(dirty_flag, 1); else *dirty_flag = 1;
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < PARTS - 1; q++) {
#pragma unroll
      for (int j = 0; j <= DIM; j++) acc[j] += s_pacc[q][tnode][j];
      touched |= (s_ptouch[q][tnode] != 0);
    }
  }
  if (active && (touched || !in_place)) {
    double *dst = in_place ? rec + flat * R : plan.partial + ((long long)grp * g.total + flat) * R;
    if (coherent) {
      // the chained boundary duplication (another workgroup) reads node values and the flag
      publish(&dst[0], acc[0]);
#pragma unroll
      for (int j = 1; j <= DIM; j++) dst[j] = acc[j];
      if (any_corr) {
        if (s_dirty) *s_dirty = 1; else publish(dirty_flag, 1);
      }
    } else {
#pragma unroll
      for (int j = 0; j <= DIM; j++) dst[j] = acc[j];
      if (any_corr) *dirty_flag = 1;
    }
    if (DIM > 1 && in_place && plan.faces) face_store<DIM>(g, plan.faces, p, acc);   // keep the lookup replica current
  } else if (!in_place) {
    // inactive nodes of a partial buffer must read as zero in the reduction
    bool in_grid_node = true;
#pragma unroll
    for (int d = 0; d < DIM; d++)
      if (p[d] >= g.n[d]) in_grid_node = false;
    if (in_grid_node) {
      long long fl = p[DIM - 1];
#pragma unroll
      for (int d = DIM - 1; d > 0; d--) fl = fl * g.n[d - 1] + p[d - 1];
      double *dst = plan.partial + ((long long)grp * g.total + fl) * R;
#pragma unroll
      for (int j = 0; j <= DIM; j++) dst[j] = 0;
    }
  }
}

template <int DIM, int MODE, int PARTS, bool PERB>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((MODE == 1 && DIM == 1) ? 4 : 1, 8))) k_hill_gather(Geom g, Tables t, double *__restrict__ rec, HillList h,
                                                               HillHeights hh, GatherPlan plan, int use_list,
                                                               int *__restrict__ dirty_flag, PostArgs post) {
  if (use_list) {
    // culled launch: the workgroups share the listed tiles (one each when the launch bound holds, which it
    // does by construction -- the stride loop keeps the result right even if a bound were ever too small)
    const long long count = plan.tile_list[gather_tiles_dev(g) + plan.tile_parity];
    for (long long i = blockIdx.x; i < count; i += gridDim.x) {
      hill_gather_body<DIM, MODE, PARTS, PERB>(g, t, rec, h, hh, plan, 1, dirty_flag, (MODE == 0) ? post.enabled : 0,
                                         plan.tile_list[i]);
      __syncthreads();  // (the body's LDS staging is reused by the next tile)
    }
  } else {
    hill_gather_body<DIM, MODE, PARTS, PERB>(g, t, rec, h, hh, plan, 0, dirty_flag, (MODE == 0) ? post.enabled : 0, blockIdx.x);
  }
  if (MODE == 0 && post.enabled)
    gather_post<DIM, PERB>(g, rec, h, hh, dirty_flag, post, gridDim.x * gridDim.y, blockIdx.x + gridDim.x * blockIdx.y);
}

// ---------------------------------------------------------------------------
// One launch for the two halves of a short 1-D hill step that do not depend on each other: workgroups
// [0, nb_int) compute the per-hill integrals and -- the last of them -- run the ordered limiter and the read-back
// (k_hill_integrals' body); the rest are the tile-owned gather, which computes its stencil terms while the limiter
// still runs and waits for the limiter's word only before it applies heights (hill_gather_body<.., DEFER>).  The
// integrals workgroups have the lower ids: they are dispatched first and wait for nobody, so the waiting gather
// workgroups can never keep them off the machine.  Arithmetic per node and per hill is that of the two separate
// launches; what changes is that the gather's ~10 us of exp-bound work no longer starts after the limiter's ~10 us
// chain of dependent round trips but beside it (W1 step: 37.9 -> see DESIGN.md section 5).
// "hill j of this step's batch met a non-zero boundary correction" (gaussian_grid.h:357-358: the boundary duplication
// follows such a hill): one word per hill holding the STEP'S NUMBER where it did -- no memset between steps, stale
// numbers do not match.  Per hill, not one minimum: with several ranks a rank's pairs see its own slice of the list only.
__device__ __forceinline__ void ordered_dirty_note(unsigned *dirty_hill, unsigned seq, long long hill) {
  publish(&dirty_hill[hill], seq);   // (every writer stores the same value; agent scope: the reader may be a launch on another stream)
}
// unit-height stencil terms of hill `hill` (LimitArgs::ord_terms): this workgroup's share `part` of `parts` of the
// 2 msize + 1 stencil offsets, one (value, derivative) pair per offset -- zeros where the node lies outside the grid, a
// wall or the hill's support.  The same node_terms / pair_term the gather and the record pass evaluate.
static constexpr int ORD_EMIT_PARTS = 2;
template <bool PERB>
__device__ __forceinline__ void ordered_emit_terms(const Geom &g, const Tables &t, const HillList &h, const LimitArgs &la,
                                                   long long hill, int part) {
  if (hill >= hill_count(h)) return;
  const int c = h.hc[hill];
  const double hx[1] = {h.hx[hill]};
  const double ht[2] = {PERB ? 0.0 : h.ht[2 * hill], PERB ? 0.0 : h.ht[2 * hill + 1]};
  const int m = g.msize[0], W = 2 * m + 1;
  double2 *row = reinterpret_cast<double2 *>(la.ord_terms) + hill * (long long)W;
  TermConst<1> tc;
  term_const<1>(g, tc);
  __shared__ int s_nz;
  if (threadIdx.x == 0) s_nz = 0;
  __syncthreads();
  bool any_nz = false;
  const int per = (W + ORD_EMIT_PARTS - 1) / ORD_EMIT_PARTS;
  const int o_end = ((part + 1) * per < W) ? (part + 1) * per : W;
  for (int o = part * per + threadIdx.x; o < o_end; o += BLOCK) {
    double2 out;
    out.x = out.y = 0.0;
    if (c != INT_MIN) {
      int idx = c + o - m;   // the reference's wrap rules (gaussian_grid.h:251-266): >= n wraps by %, < 0 by a single + n
      bool ok = true;
      if (idx >= g.n[0]) {
        if (g.periodic[0]) idx %= g.n[0]; else ok = false;
      }
      if (idx < 0) {
        if (g.periodic[0]) idx += g.n[0]; else ok = false;
        if (idx < 0) ok = false;
      }
      if (ok) {
        const int p[1] = {idx};
        NodeTerms<1> nt;
        node_terms<1, PERB>(g, t, p, nt);
        double val, dval[1];
        bool nz = false;
        if (nt.inside && pair_term<1, PERB>(g, tc, nt, hx, ht, val, dval, nz, false)) {
          out.x = val;
          out.y = dval[0];
          any_nz |= nz;
        }
      }
    }
    if (la.ord_ready) {   // (read by a launch that runs beside this one: past this XCD's L2)
      publish(&reinterpret_cast<double *>(row)[2 * o], out.x);
      publish(&reinterpret_cast<double *>(row)[2 * o + 1], out.y);
    } else {
      row[o] = out;
    }
  }
  if (any_nz) s_nz = 1;
  if (la.ord_ready) __builtin_amdgcn_s_waitcnt(0);   // this wave's term stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_nz) ordered_dirty_note(la.ord_dirty, la.ord_seq, hill);
    if (la.ord_ready) {
      __builtin_amdgcn_s_waitcnt(0);
      publish(&la.ord_ready[ORD_EMIT_PARTS * hill + part], la.ord_seq);
    }
  }
}

template <bool PERB>
__global__ void __launch_bounds__(BLOCK) k_integrals_gather(Geom g, Tables t, double *__restrict__ rec, HillList h,
                                                            const double *__restrict__ heights, double h_const,
                                                            double *__restrict__ added, LimitArgs la, HillHeights hh,
                                                            GatherPlan plan, int *__restrict__ dirty_flag, PostArgs post,
                                                            unsigned nb_int, unsigned tiles_first, unsigned nb_emit) {
  // (the term emitters of a reference-order step: the last nb_emit workgroups of the launch, dispatched behind everybody
  //  else, waiting for nobody)
  // (... or, with a record pass waiting for them beside this launch, LimitArgs::ord_ready: between the integrals and
  //  the tiles -- the block index is rotated so that the rest of the kernel sees the usual order)
  const unsigned bidx = (la.ord_ready && nb_emit)
                            ? (blockIdx.x < nb_int ? blockIdx.x
                                                   : (blockIdx.x < nb_int + nb_emit ? gridDim.x - nb_emit + (blockIdx.x - nb_int)
                                                                                    : blockIdx.x - nb_emit))
                            : blockIdx.x;
  if (bidx >= gridDim.x - nb_emit) {
    const unsigned e = bidx - (gridDim.x - nb_emit);
    ordered_emit_terms<PERB>(g, t, h, la, (long long)(e / ORD_EMIT_PARTS), (int)(e % ORD_EMIT_PARTS));
    return;
  }
  // Dispatch order.  The tiles wait for the limiter's word, so they must never keep the integrals' workgroups off the
  // machine: with the integrals first that holds by construction, but the launch is sized by a BOUND on the hill count
  // (628 integrals workgroups for ~125 hills) and the tiles -- whose terms are the longest stretch of the launch --
  // then start 2-4 us late, behind hundreds of workgroups that exit at once.  When the tiles cannot fill the machine
  // (host-checked against two resident workgroups per CU) they go first.
  // (`wg`: this workgroup's index in the order integrals | tiles, which the stamps and the rest of the code use)
  const unsigned ntile_all = gridDim.x - nb_emit - nb_int;
  const unsigned wg = tiles_first ? (bidx < ntile_all ? nb_int + bidx : bidx - ntile_all) : bidx;
#define EDM_STAMP(k) do { if (la.trace && threadIdx.x == 0) la.trace[(size_t)wg * 8 + (k)] = wall_clock64(); } while (0)
  EDM_STAMP(0);
  if (wg < nb_int) {
    const bool ran_limiter = hill_integrals_body<1, BLOCK, PERB>(g, t, h, heights, h_const, added, la, wg, nb_int);
    // the CV histogram needs the limiter's flags and the hills' positions, nothing of the gather: the limiter's
    // workgroup updates it while the gather applies heights (edm_bias.cpp:601-610)
    if (ran_limiter && post.enabled && !la.res->error)
      hist_batch<1>(post.hg, post.hist, h.nh, h.hx0, la.res, post.flags, post.flush_mode, threadIdx.x, BLOCK);
    EDM_STAMP(7);
    return;
  }
  const unsigned tile = wg - nb_int, ntile = ntile_all;
  // (chained bookkeeping that is only the boundary duplication: boundary corrections are noted in LDS and travel in
  //  the ticket, see tile_ticket_duplicate)
  __shared__ int s_dirty, s_last;
  const bool ticket_dirty = post.enabled && post.skip_hist && !post.rb_dst && ntile < 0xFFFFu;
  if (threadIdx.x == 0) s_dirty = 0;
  hill_gather_body<1, 0, 8, PERB, true>(g, t, rec, h, hh, plan, 0, dirty_flag, post.enabled, tile, la.ready_flag, la.ready_seq,
                                        la.trace ? la.trace + (size_t)wg * 8 : nullptr, ticket_dirty ? &s_dirty : nullptr);
  EDM_STAMP(6);
  if (!post.enabled) return;
  if (ticket_dirty) {
    tile_ticket_duplicate<PERB>(g, rec, post, ntile, tile, &s_dirty, &s_last);
    EDM_STAMP(7);
    return;
  }
  // (the bookkeeping reads the limiter's result: a workgroup whose tile met no hill has not waited for it yet)
  __syncthreads();
  if (threadIdx.x == 0) (void)wait_for_word(la.ready_flag, la.ready_seq);
  __syncthreads();
  gather_post<1, PERB>(g, rec, h, hh, dirty_flag, post, ntile, tile);
  EDM_STAMP(7);
#undef EDM_STAMP
}

// rec[p] += partial[0][p] + partial[1][p] + ... in group (= hill list) order
__global__ void __launch_bounds__(BLOCK) k_reduce_partials(Geom g, double *__restrict__ rec,
                                                           const double *__restrict__ partial, int groups,
                                                           const LimitResult *__restrict__ res, int adaptive,
                                                           long long nh, const long long *__restrict__ nh_dev) {
  if (res && res->error) return;  // limiter overflow: the batch is not applied
  if (adaptive) {
    if (nh_dev && *nh_dev < nh) nh = *nh_dev;
    groups = adaptive_groups(groups, nh);
    if (groups == 1) return;      // the single group accumulated in place
  }
  const long long n = g.total * g.rec;
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    double v = rec[i];
    for (int q = 0; q < groups; q++) v += partial[(long long)q * n + i];
    rec[i] = v;
  }
}

// tile culling for large grids with few hills: list the tiles touched by any hill.  One thread per
// (hill, stencil-corner offset); a tile is appended by whoever flips its flag first (wave-aggregated
// append), and the gather workgroup that later owns the tile clears the flag again, so `flags` is all
// zero between batches and nothing has to be memset or compacted.
// (id: this thread's index among the marking threads -- whole waves, see the wave-aggregated append)
template <int DIM>
__device__ __forceinline__ void mark_tiles_body(const Geom &g, const HillList &h, int *__restrict__ flags,
                                                int *__restrict__ list, long long ntiles, int parity, long long id) {
  int *count = list + ntiles + parity;
  if (id == 0) list[ntiles + (1 - parity)] = 0;  // the next batch's counter
  int ntile[DIM], steps[DIM];
  long long combos = 1;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    const int T = Tile<DIM>::T[d];
    ntile[d] = (g.n[d] + T - 1) / T;
    // sample nodes -m, -m+T, ... and the end point +m; a periodic dimension adds the four nodes next to the
    // wrap seams (the samples are T apart in un-wrapped coordinates, so when n is not a multiple of T the
    // partial last tile before a seam would fall between two of them; not needed when T divides n)
    steps[d] = (2 * g.msize[d]) / T + 2 + ((g.periodic[d] && g.n[d] % T != 0) ? 4 : 0);
    combos *= steps[d];
  }
  const long long i = id / combos;
  bool emit = false;
  long long tflat = 0;
  if (i < hill_count(h) && h.hc[i * DIM] != INT_MIN) {
    long long rest = id - i * combos, tstride = 1;
    bool skip = false;
    double dp2_min = 0;  // lower bound of dp2 (gaussian_grid.h:292) over the nodes of this tile
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      const int T = Tile<DIM>::T[d];
      const int sidx = (int)(rest % steps[d]);
      rest /= steps[d];
      const int nreg = (2 * g.msize[d]) / T + 2;
      int unwrapped;
      if (sidx < nreg) {
        int off = -g.msize[d] + sidx * T;
        if (off > g.msize[d]) off = g.msize[d];
        unwrapped = h.hc[i * DIM + d] + off;
      } else {
        const int e = sidx - nreg;  // seam nodes -1 | 0 and n-1 | n
        unwrapped = (e == 0) ? -1 : (e == 1) ? 0 : (e == 2) ? g.n[d] - 1 : g.n[d];
        if (unwrapped < h.hc[i * DIM + d] - g.msize[d] || unwrapped > h.hc[i * DIM + d] + g.msize[d]) skip = true;
      }
      int idx = unwrapped;
      if (idx >= g.n[d]) {
        // (non-periodic: a sample point past the end still stands for the last tile, whose leading nodes
        //  the stencil reaches -- the samples are T apart and the centre node lies inside the grid)
        if (g.periodic[d]) idx %= g.n[d]; else idx = unwrapped = g.n[d] - 1;
      }
      if (idx < 0) {
        if (g.periodic[d]) idx += g.n[d]; else idx = unwrapped = 0;
        if (idx < 0) skip = true;
      }
      if (!skip) {
        tflat += (long long)(idx / T) * tstride;
        // the tile's node range, shifted back to the periodic image this offset belongs to
        const int shift = unwrapped - idx;
        const int n0 = (idx / T) * T + shift;
        int n1 = (idx / T) * T + T - 1;
        if (n1 > g.n[d] - 1) n1 = g.n[d] - 1;
        n1 += shift;
        const double xh = h.hx[i * DIM + d];
        const double lo = g.min[d] + n0 * g.dx[d], hi = g.min[d] + n1 * g.dx[d];
        double gap = 0;
        if (xh < lo) gap = lo - xh;
        if (xh > hi) gap = xh - hi;
        gap /= g.sigma[d];
        dp2_min += gap * gap;
      }
      tstride *= ntile[d];
    }
    // tiles of the stencil's bounding box that lie wholly outside the dp2 < 8 ball receive nothing
    // (:294); the margin keeps the cut conservative against rounding in the per-node test
    if (!skip && dp2_min < 8.0 * (1.0 + 1e-6)) emit = (atomicExch(&flags[tflat], 1) == 0);
  }
  const unsigned long long bal = __ballot(emit);
  if (bal) {
    const int lane = threadIdx.x & 63;
    const int leader = (int)__builtin_ctzll(bal);
    int base = 0;
    if (lane == leader) base = atomicAdd(count, (int)__popcll(bal));
    base = __shfl(base, leader, 64);
    if (emit) list[base + (int)__popcll(bal & ((1ull << lane) - 1ull))] = (int)tflat;
  }
}
template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_mark_tiles(Geom g, HillList h, int *__restrict__ flags,
                                                      int *__restrict__ list, long long ntiles, int parity) {
  mark_tiles_body<DIM>(g, h, flags, list, ntiles, parity, (long long)blockIdx.x * BLOCK + threadIdx.x);
}
long long mark_tiles_threads(const Geom &g, long long nh) {
  long long combos = 1;
  for (int d = 0; d < g.dim; d++) {
    const int T = (g.dim == 1) ? Tile<1>::T[d] : (g.dim == 2) ? Tile<2>::T[d] : Tile<3>::T[d];
    combos *= (2 * g.msize[d]) / T + 2 + ((g.periodic[d] && g.n[d] % T != 0) ? 4 : 0);
  }
  return nh * combos;
}

template <int DIM>
static hipError_t gather_dim(const Geom &g, const Tables &t, double *rec, const HillList &h, const HillHeights &hh,
                             const GatherPlan &plan, int *dirty_flag, hipStream_t s, const PostSpec *chain) {
  PostArgs post;
  memset(&post, 0, sizeof(post));
  if (chain) {
    if (plan.groups != 1 || !hh.res_dev || !h.hx0) return hipErrorInvalidValue;
    post.enabled = 1;
    post.ticket = chain->ticket;
    post.dp = make_dup_plan(g);
    post.hg = *chain->hist_geom;
    post.hist = chain->hist;
    post.flags = chain->flags;
    post.flush_mode = chain->flush_mode;
    post.rb_src = chain->rb_src;
    post.rb_dst = chain->rb_dst;
    post.rb_bytes = chain->rb_bytes;
  }
  const long long ntiles = gather_tiles(g);
  int use_list = 0;
  long long launch_tiles = ntiles;
  if (plan.tile_flags && plan.tile_list && plan.groups == 1) {
    const long long mt = mark_tiles_threads(g, h.nh);
    if (!plan.tiles_marked)
      hipLaunchKernelGGL(k_mark_tiles<DIM>, dim3((unsigned)((mt + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, g, h, plan.tile_flags,
                         plan.tile_list, ntiles, plan.tile_parity);
    use_list = 1;
    launch_tiles = plan.tile_bound < ntiles ? plan.tile_bound : ntiles;
    // ... and never more workgroups than a few per CU: the workgroups stride over the list, and DISPATCHING a
    // workgroup that finds nothing to do is not free -- the bound is several times the list's length (it is taken on
    // the launch bound of the hill count), 18 000 workgroups for a list of 9 000 tiles on W4, and the launch's
    // duration followed the launched count, not the list (PMC: busy cycles 1.0 M at 410 workgroups, 5.4 M at 18 000)
    static const long long cap_env = test_force_value("gather_wgs");   // (tests: a launch narrower than its tile list)
    // (four per CU = what is resident at once at the kernel's 120-128 registers: every workgroup starts at once and
    //  strides; measured on W4: 512 -> 0.127, 768 -> 0.115, 1024 -> 0.110, 1536 -> 0.115, 2048 -> 0.113, unbounded
    //  0.124 ms per step; the 2-D gather, whose bound is closer to its list, does not care)
    const long long cap = cap_env > 0 ? cap_env : (long long)4 * cu_count();
    if (launch_tiles > cap) launch_tiles = cap;
  }
  bool perb = true;
  for (int d = 0; d < DIM; d++)
    if (!g.bper[d]) perb = false;
  const dim3 grid((unsigned)launch_tiles, (unsigned)plan.groups);
  if (DIM == 1 && !use_list) {
    // 1-D: 32-node tiles, eight hill-parts per node (see hill_gather_body)
    constexpr int P1 = (DIM == 1) ? 8 : 1;
    const dim3 grid4((unsigned)((g.n[0] + BLOCK / 8 - 1) / (BLOCK / 8)), (unsigned)plan.groups);
    if (perb)
      hipLaunchKernelGGL((k_hill_gather<DIM, 0, P1, true>), grid4, dim3(BLOCK), 0, s, g, t, rec, h, hh, plan,
                         use_list, dirty_flag, post);
    else
      hipLaunchKernelGGL((k_hill_gather<DIM, 0, P1, false>), grid4, dim3(BLOCK), 0, s, g, t, rec, h, hh, plan,
                         use_list, dirty_flag, post);
  } else {
    if (perb)
      hipLaunchKernelGGL((k_hill_gather<DIM, 0, 1, true>), grid, dim3(BLOCK), 0, s, g, t, rec, h, hh, plan, use_list, dirty_flag,
                         post);
    else
      hipLaunchKernelGGL((k_hill_gather<DIM, 0, 1, false>), grid, dim3(BLOCK), 0, s, g, t, rec, h, hh, plan, use_list, dirty_flag,
                         post);
  }
  if (plan.groups > 1)
    hipLaunchKernelGGL(k_reduce_partials, dim3(blocks_for(g.total * g.rec)), dim3(BLOCK), 0, s, g, rec, plan.partial,
                       plan.groups, hh.res_dev, plan.adaptive, h.nh, h.nh_dev);
  return hipGetLastError();
}

hipError_t launch_hill_gather(const Geom &g, const Tables &t, double *rec, const HillList &h, const HillHeights &hh,
                              const GatherPlan &plan, int *dirty_flag, hipStream_t s, const PostSpec *chain) {
  if (h.nh <= 0) return chain ? hipErrorInvalidValue : hipSuccess;
  switch (g.dim) {
    case 1: return gather_dim<1>(g, t, rec, h, hh, plan, dirty_flag, s, chain);
    case 2: return gather_dim<2>(g, t, rec, h, hh, plan, dirty_flag, s, chain);
    default: return gather_dim<3>(g, t, rec, h, hh, plan, dirty_flag, s, chain);
  }
}

bool integrals_gather_fusable(const Geom &g, long long nh_bound, const GatherPlan &plan) {
  return g.dim == 1 && hill_integrals_can_chain_limit(nh_bound) && plan.groups == 1 && !plan.tile_list;
}
hipError_t launch_integrals_gather(const Geom &g, const Tables &t, double *rec, const HillList &h, const double *heights,
                                   double h_const, double *added, const LimitArgs &chain, const HillHeights &hh,
                                   const GatherPlan &plan, int *dirty_flag, hipStream_t s, const PostSpec *post_chain) {
  if (!integrals_gather_fusable(g, h.nh, plan) || !chain.ready_flag || !hh.res_dev) return hipErrorInvalidValue;
  LimitArgs la = chain;
  la.enabled = 1;
  PostArgs post;
  memset(&post, 0, sizeof(post));
  if (post_chain) {
    if (!h.hx0) return hipErrorInvalidValue;
    post.enabled = 1;
    post.ticket = post_chain->ticket;
    post.dp = make_dup_plan(g);
    post.hg = *post_chain->hist_geom;
    post.hist = post_chain->hist;
    post.flags = post_chain->flags;
    post.flush_mode = post_chain->flush_mode;
    post.rb_src = post_chain->rb_src;
    post.rb_dst = post_chain->rb_dst;
    post.rb_bytes = post_chain->rb_bytes;
    post.skip_hist = 1;
  }
  const unsigned nb_int = (unsigned)h.nh;
  const unsigned nb_tiles = (unsigned)((g.n[0] + BLOCK / 8 - 1) / (BLOCK / 8));
  // tiles first iff the waiting tiles can never fill the machine: the kernel keeps three workgroups per CU resident
  // (139 registers: three waves per SIMD; 47 KB of LDS), so with 64 slots to spare the integrals' workgroups -- which
  // wait for nobody -- always find room to run through
  // ... and the hills' workgroups must find room beside them: a launch whose tiles and expected hills together exceed
  // the residency dispatches the integrals first (the neighbour-list melt at two workgroups per CU: 351 tiles + ~250
  // hills against 512 slots -- tiles first, 86 hills waited for a slot until the first ones had finished: integrals
  // done at 11 us instead of 6; a launch bound without a hint counts as its own expectation)
  const long long live = (chain.expected_hills > 0 && chain.expected_hills < h.nh) ? chain.expected_hills : h.nh;
  // how many workgroups of THIS kernel a CU keeps resident: asked of the runtime for the instantiation that is launched
  // (registers and LDS as this compiler allotted them -- 139 registers and 47 KB gave three when this was written),
  // never more than the three the reasoning above was measured with.  The waiting tiles go first only on a device this
  // process has to itself: a second rank on the same GPU (the host-staged carrier of the tests) brings waiting tiles of
  // its own, and the slots the integrals count on may be theirs.
  static int per_cu_cached[2] = {0, 0};
  int &per_cu = per_cu_cached[g.bper[0] ? 1 : 0];
  if (per_cu == 0) {
    int nblk = 0;
    hipError_t eo = g.bper[0] ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_integrals_gather<true>, BLOCK, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_integrals_gather<false>, BLOCK, 0);
    if (eo != hipSuccess) {
      (void)hipGetLastError();
      nblk = 1;
    }
    per_cu = nblk < 1 ? 1 : (nblk > 3 ? 3 : nblk);
  }
  const long long slots = (long long)per_cu * cu_count();
  unsigned tiles_first = ((long long)nb_tiles + 64 <= slots && (long long)nb_tiles + live + live / 4 <= slots) ? 1u : 0u;
  if (chain.shared_device) tiles_first = 0;             // integrals first: deadlock-free by construction
  if (la.ord_ready) tiles_first = 0;                    // (a record pass on another stream shares the machine: likewise)
  if (chain.tiles_first_mode >= 0) tiles_first = chain.tiles_first_mode ? 1u : 0u;   // (tests)
  if (post.enabled) dup_ticket_tiles_1d(g, post, nb_tiles, BLOCK / 8);
  // (term emitters of a reference-order step, sized by the expected hill count like the rest of the launch)
  const unsigned nb_emit = la.ord_terms ? (unsigned)h.nh * ORD_EMIT_PARTS : 0u;
  if (!g.bper[0])
    hipLaunchKernelGGL((k_integrals_gather<false>), dim3(nb_int + nb_tiles + nb_emit), dim3(BLOCK), 0, s, g, t, rec, h, heights,
                       h_const, added, la, hh, plan, dirty_flag, post, nb_int, tiles_first, nb_emit);
  else
    hipLaunchKernelGGL((k_integrals_gather<true>), dim3(nb_int + nb_tiles + nb_emit), dim3(BLOCK), 0, s, g, t, rec, h, heights,
                       h_const, added, la, hh, plan, dirty_flag, post, nb_int, tiles_first, nb_emit);
  return hipGetLastError();
}

// Tail of a 1-D gather tile whose launch chains the boundary duplication (gaussian_grid.h:571-630, due iff some tile of
// the batch met a non-zero boundary correction).  The ticket carries that bit: every tile adds 1, and 0x10000 if it
// saw one (*s_dirty, LDS), in ONE atomic round trip -- the last arrival learns both that it is the last and whether
// anybody was dirty (a separate flag word cost the last tile a second dependent round trip).
// Only tiles that can matter take it (dup_ticket_tiles_1d, host): those holding a node a wall blend reaches -- no other
// node can meet a boundary correction -- or a node the duplication reads or writes.  Where the walls sit on the grid's
// first and last node (the usual fix edm_pair set-up, W1 included) there is nothing to duplicate and nobody takes it:
// the launch ends with the last tile's stores instead of two dependent round trips later.
template <bool PERB>
__device__ __forceinline__ void tile_ticket_duplicate(const Geom &g, double *__restrict__ rec, const PostArgs &post,
                                                      unsigned ntile, unsigned tile, const int *s_dirty, int *s_last) {
  const bool everyone = post.tk_lo_end == 0 && post.tk_hi_begin == 0;
  if (!everyone) {
    if (PERB || (tile >= post.tk_lo_end && tile < post.tk_hi_begin)) return;   // (uniform over the workgroup)
    ntile = post.tk_lo_end + (ntile - post.tk_hi_begin);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned mine = *s_dirty ? 0x10001u : 1u;
    const unsigned t0 = (unsigned)__hip_atomic_fetch_add(post.ticket, (int)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int last = 0;
    if ((t0 & 0xFFFFu) == ntile - 1) {
      __hip_atomic_store(post.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = ((t0 + mine) >> 16) ? 2 : 1;
    }
    *s_last = last;
  }
  __syncthreads();
  if (!PERB && *s_last == 2 && threadIdx.x < 64) duplicate_boundary_wave(g, rec, post.dp, threadIdx.x);
}

int gather_slots_per_hill(const Geom &g) {
  static const int T1[3] = {256, 1, 1}, T2[3] = {16, 16, 1}, T3[3] = {8, 8, 4};
  const int *T = g.dim == 1 ? T1 : g.dim == 2 ? T2 : T3;
  int total = 1;
  for (int d = 0; d < g.dim; d++) {
    const int nt = (g.n[d] + T[d] - 1) / T[d];
    int S = (2 * g.msize[d]) / T[d] + 3;
    if (S > nt) S = nt;
    total *= S;
  }
  return total;
}

// added[i] = slots[i][0] + slots[i][1] + ... (tile order: fixed)
__global__ void __launch_bounds__(BLOCK) k_sum_slots(long long nh, int S, const double *__restrict__ slots,
                                                     double *__restrict__ added) {
  const long long stride = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < nh; i += stride) {
    double a = 0;
    for (int q = 0; q < S; q++) a += slots[i * S + q];
    added[i] = a;
  }
}

hipError_t launch_hill_gather_fused(const Geom &g, const Tables &t, const HillList &h, const double *heights,
                                    double h_const, const GatherPlan &plan, double *added, int *dirty_flag,
                                    hipStream_t s) {
  if (h.nh <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(plan.slots, 0, sizeof(double) * (size_t)h.nh * plan.slots_per_hill, s);
  if (e != hipSuccess) return e;
  HillHeights hh;
  hh.h = heights;
  hh.h_const = h_const;
  hh.k = h.nh;
  hh.tail_h1 = nullptr;
  hh.tail_h2 = nullptr;
  hh.res_dev = nullptr;
  hh.tail_shift = 0;
  const long long ntiles = gather_tiles(g);
  const dim3 grid((unsigned)ntiles, (unsigned)plan.groups);
  PostArgs nopost;
  memset(&nopost, 0, sizeof(nopost));
  switch (g.dim) {
    case 1: hipLaunchKernelGGL((k_hill_gather<1, 1, 1, false>), grid, dim3(BLOCK), 0, s, g, t, (double *)nullptr, h, hh, plan, 0, dirty_flag, nopost); break;
    case 2: hipLaunchKernelGGL((k_hill_gather<2, 1, 1, false>), grid, dim3(BLOCK), 0, s, g, t, (double *)nullptr, h, hh, plan, 0, dirty_flag, nopost); break;
    default: hipLaunchKernelGGL((k_hill_gather<3, 1, 1, false>), grid, dim3(BLOCK), 0, s, g, t, (double *)nullptr, h, hh, plan, 0, dirty_flag, nopost); break;
  }
  hipLaunchKernelGGL(k_sum_slots, dim3(blocks_for(h.nh)), dim3(BLOCK), 0, s, h.nh, plan.slots_per_hill, plan.slots, added);
  return hipGetLastError();
}

hipError_t launch_hill_gather_correct_and_apply(const Geom &g, const Tables &t, double *rec, const HillList &h,
                                                const HillHeights &hh, const GatherPlan &plan, int with_correction,
                                                int *dirty_flag, hipStream_t s) {
  if (h.nh <= 0) return hipSuccess;
  int groups = plan.groups;
  if (with_correction) {
    const size_t one = sizeof(double) * (size_t)g.total * g.rec;
    hipError_t e = hipMemsetAsync(plan.partial + (size_t)plan.groups * g.total * g.rec, 0, one, s);
    if (e != hipSuccess) return e;
    const long long ntiles = gather_tiles(g);
    const dim3 grid((unsigned)ntiles, 1);
    PostArgs nopost;
    memset(&nopost, 0, sizeof(nopost));
    switch (g.dim) {
      case 1: hipLaunchKernelGGL((k_hill_gather<1, 2, 1, false>), grid, dim3(BLOCK), 0, s, g, t, rec, h, hh, plan, 0, dirty_flag, nopost); break;
      case 2: hipLaunchKernelGGL((k_hill_gather<2, 2, 1, false>), grid, dim3(BLOCK), 0, s, g, t, rec, h, hh, plan, 0, dirty_flag, nopost); break;
      default: hipLaunchKernelGGL((k_hill_gather<3, 2, 1, false>), grid, dim3(BLOCK), 0, s, g, t, rec, h, hh, plan, 0, dirty_flag, nopost); break;
    }
    groups += 1;
  }
  hipLaunchKernelGGL(k_reduce_partials, dim3(blocks_for(g.total * g.rec)), dim3(BLOCK), 0, s, g, rec, plan.partial, groups,
                     with_correction ? hh.res_dev : (const LimitResult *)nullptr, 0, h.nh, (const long long *)nullptr);
  return hipGetLastError();
}

hipError_t launch_hill_gather_correction(const Geom &g, const Tables &t, const HillList &h, const HillHeights &hh,
                                         const GatherPlan &plan, int *dirty_flag, hipStream_t s) {
  const size_t one = sizeof(double) * (size_t)g.total * g.rec;
  hipError_t e = hipMemsetAsync(plan.partial + (size_t)plan.groups * g.total * g.rec, 0, one, s);
  if (e != hipSuccess) return e;
  if (h.nh <= 0) return hipSuccess;
  const long long ntiles = gather_tiles(g);
  const dim3 grid((unsigned)ntiles, 1);
  PostArgs nopost;
  memset(&nopost, 0, sizeof(nopost));
  double *none = nullptr;
  switch (g.dim) {
    case 1: hipLaunchKernelGGL((k_hill_gather<1, 2, 1, false>), grid, dim3(BLOCK), 0, s, g, t, none, h, hh, plan, 0, dirty_flag, nopost); break;
    case 2: hipLaunchKernelGGL((k_hill_gather<2, 2, 1, false>), grid, dim3(BLOCK), 0, s, g, t, none, h, hh, plan, 0, dirty_flag, nopost); break;
    default: hipLaunchKernelGGL((k_hill_gather<3, 2, 1, false>), grid, dim3(BLOCK), 0, s, g, t, none, h, hh, plan, 0, dirty_flag, nopost); break;
  }
  return hipGetLastError();
}
hipError_t launch_add_partials(const Geom &g, double *dst, const double *partial, int groups,
                               const LimitResult *res_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_partials, dim3(blocks_for(g.total * g.rec)), dim3(BLOCK), 0, s, g, dst, partial, groups,
                     res_dev, 0, 0LL, (const long long *)nullptr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Ordered (serial-dependence) hill application: one workgroup, hills strictly in sequence.
// ---------------------------------------------------------------------------
static constexpr int ORD_BLOCK = 1024;

template <int DIM>
__global__ void __launch_bounds__(ORD_BLOCK) k_hills_ordered(Geom g, Tables t, double *__restrict__ rec, HillList h,
                                                             OrderedParams op, LimitTail tail,
                                                             double *__restrict__ heights_out,
                                                             double *__restrict__ added_out,
                                                             LimitResult *__restrict__ res, DupPlan dp,
                                                             int *__restrict__ dirty_flag) {
  constexpr int R = (DIM == 1) ? 2 : 4;
  __shared__ double red[ORD_BLOCK / 64];
  __shared__ double s_height, s_h1, s_h2;
  __shared__ int s_dirty;
  TermConst<DIM> tc;
  term_const<DIM>(g, tc);
  double vol = 1;
#pragma unroll
  for (int d = 0; d < DIM; d++) vol *= g.dx[d];
  double cum = op.cum_in;  // meaningful in thread 0
  int n_def = 0;
  for (long long i = 0; i < h.nh; i++) {
    // ---- 1. height from the CURRENT grid (edm_bias.cpp:537-558) ----
    if (threadIdx.x == 0) {
      double x0[DIM];
#pragma unroll
      for (int d = 0; d < DIM; d++) x0[d] = h.hx0[i * DIM + d];
      double hgt = op.prefactor;
      if (op.use_target) hgt *= exp(target_value<DIM>(op.target, op.target_values, x0) - op.expected_target);
      if (op.use_tempering) {
        double v, der[DIM];
        lookup_one<DIM>(g, rec, x0, v, der);  // bias_->get_value(position)
        hgt *= exp(-v / op.temper_scale);
      }
      hgt /= op.divisor;
      hgt = fmin(hgt, op.clamp);
      s_height = hgt;
      s_dirty = 0;
    }
    __syncthreads();
    const double height = s_height;
    // the hill's distinct nodes: per dimension a run of `cnt` nodes starting at `lo` (wrapped)
    const bool valid = (h.hc[i * DIM] != INT_MIN);
    int c[DIM], lo[DIM], cntd[DIM];
    double hx[DIM], ht[2 * DIM];
    long long total = valid ? 1 : 0;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      c[d] = h.hc[i * DIM + d];
      hx[d] = h.hx[i * DIM + d];
      ht[2 * d] = h.ht[i * 2 * DIM + 2 * d];
      ht[2 * d + 1] = h.ht[i * 2 * DIM + 2 * d + 1];
      if (valid) {
        const int m = g.msize[d];
        if (g.periodic[d]) {
          cntd[d] = (2 * m + 1 < g.n[d]) ? 2 * m + 1 : g.n[d];
          lo[d] = (int)((((long long)c[d] - m) % g.n[d] + g.n[d]) % g.n[d]);
        } else {
          const int a = (c[d] - m > 0) ? c[d] - m : 0;
          const int b = (c[d] + m < g.n[d] - 1) ? c[d] + m : g.n[d] - 1;
          lo[d] = a;
          cntd[d] = (b >= a) ? b - a + 1 : 0;
        }
        total *= cntd[d];
      } else {
        lo[d] = 0;
        cntd[d] = 0;
      }
    }
    // ---- 2. integrated bias of the hill (what add_value returns) ----
    double part = 0;
    for (long long sidx = threadIdx.x; sidx < total; sidx += ORD_BLOCK) {
      int p[DIM];
      long long rest = sidx;
      int mult = 1;
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        const int o = (int)(rest % cntd[d]);
        rest /= cntd[d];
        int idx = lo[d] + o;
        if (g.periodic[d] && idx >= g.n[d]) idx -= g.n[d];
        p[d] = idx;
        mult *= images(g, d, c[d], idx, idx);
      }
      if (mult == 0) continue;
      NodeTerms<DIM> nt;
      node_terms<DIM>(g, t, p, nt);
      if (!nt.inside) continue;
      double val, dval[DIM];
      bool nz;
      if (!pair_term<DIM>(g, tc, nt, hx, ht, val, dval, nz)) continue;
      for (int rep = 0; rep < mult; rep++) part += height * val * vol;
    }
    const double added = block_sum(part, red);
    // ---- 3. limiter step (edm_bias.cpp:465-495), thread 0 ----
    if (threadIdx.x == 0) {
      double h1 = 0, h2 = 0, a2 = 0;
      int fl = 0;
      if (cum < op.limit) {
        h1 = height;
        fl = 1;
        cum += added;
        if (cum > op.limit) {
          h2 = fmax(op.limit - cum, -height);
          a2 = (height != 0.0) ? h2 * (added / height) : 0.0;
          cum += a2;
          fl |= 2 | 4;
          n_def++;
        }
      } else {
        fl = 4;
        n_def++;
      }
      s_h1 = h1;
      s_h2 = h2;
      tail.h1[i] = h1;
      tail.h2[i] = h2;
      tail.added2[i] = a2;
      tail.cum_after[i] = cum;
      tail.flags[i] = fl;
      heights_out[i] = height;
      added_out[i] = (fl & 1) ? added : 0.0;
    }
    __syncthreads();
    const double h1 = s_h1, h2 = s_h2;
    // ---- 4. stencil update (plain stores: every node of the run is owned by one thread) ----
    bool any_corr = false;
    if (h1 != 0 || h2 != 0) {
      for (long long sidx = threadIdx.x; sidx < total; sidx += ORD_BLOCK) {
        int p[DIM];
        long long rest = sidx;
        int mult = 1;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          const int o = (int)(rest % cntd[d]);
          rest /= cntd[d];
          int idx = lo[d] + o;
          if (g.periodic[d] && idx >= g.n[d]) idx -= g.n[d];
          p[d] = idx;
          mult *= images(g, d, c[d], idx, idx);
        }
        if (mult == 0) continue;
        NodeTerms<DIM> nt;
        node_terms<DIM>(g, t, p, nt);
        if (!nt.inside) continue;
        double val, dval[DIM];
        bool nz;
        if (!pair_term<DIM>(g, tc, nt, hx, ht, val, dval, nz)) continue;
        any_corr |= nz;
        long long flat = p[DIM - 1];
#pragma unroll
        for (int d = DIM - 1; d > 0; d--) flat = flat * g.n[d - 1] + p[d - 1];
        double acc[1 + DIM];
#pragma unroll
        for (int j = 0; j <= DIM; j++) acc[j] = rec[flat * R + j];
        for (int rep = 0; rep < mult; rep++) {
          acc[0] += h1 * val;
#pragma unroll
          for (int d = 0; d < DIM; d++) acc[1 + d] += h1 * dval[d];
        }
        if (h2 != 0)
          for (int rep = 0; rep < mult; rep++) {
            acc[0] += h2 * val;
#pragma unroll
            for (int d = 0; d < DIM; d++) acc[1 + d] += h2 * dval[d];
          }
#pragma unroll
        for (int j = 0; j <= DIM; j++) rec[flat * R + j] = acc[j];
      }
    }
    if (any_corr) s_dirty = 1;
    __threadfence_block();
    __syncthreads();
    // ---- 5. boundary duplication (gaussian_grid.h:365-368) ----
    if (s_dirty) duplicate_boundary_block(g, rec, dp);
    __threadfence_block();
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    res->cum_out = cum;
    res->k = 0;
    res->nh = h.nh;
    res->n_tail = (int)h.nh;
    res->stop = (int)h.nh;
    res->n_deferred = n_def;
    res->error = 0;
    *dirty_flag = 0;
  }
}

// ---------------------------------------------------------------------------
// K6: duplicate_boundary (gaussian_grid.h:571-630): copies the VALUE of the first /
// last in-boundary node to its outward neighbour for the 4^dim index combinations.
// ---------------------------------------------------------------------------
// copies for the 4^dim index combinations; must be called by >= 64 threads of one workgroup
// (c = this thread's combination index; BLOCK_SYNC: the callers are a whole workgroup, else one wave, whose
//  lanes run the loads and the stores in lockstep)
template <bool BLOCK_SYNC>
__device__ __forceinline__ void duplicate_boundary_lanes(const Geom &g, double *__restrict__ rec, const DupPlan &dp, int c) {
  int combos = 1;
  for (int d = 0; d < g.dim; d++) combos *= 4;
  bool do_copy = false;
  long long outer_flat = 0, inner_flat = 0;
  if (c < combos) {
    unsigned long long outer[3], inner[3];
    int rest = c;
    bool skip = false;
    for (int d = 0; d < g.dim; d++) {
      const int which = rest % 4;
      rest /= 4;
      switch (which) {
        case 0:
          skip |= (g.bper[d] != 0);
          skip |= (dp.lo[d] == 0);
          outer[d] = dp.lo[d] - 1;
          inner[d] = dp.lo[d];
          break;
        case 1:
          outer[d] = dp.lo[d];
          inner[d] = dp.lo[d];
          break;
        case 2:
          outer[d] = dp.hi[d];
          inner[d] = dp.hi[d];
          break;
        default:
          skip |= (g.bper[d] != 0);
          skip |= (dp.hi[d] == (unsigned long long)(g.n[d] - 1));
          outer[d] = dp.hi[d] + 1;
          inner[d] = dp.hi[d];
          break;
      }
    }
    if (!skip) {
      bool oob = false;
      for (int d = 0; d < g.dim; d++)
        if (outer[d] >= (unsigned long long)g.n[d] || inner[d] >= (unsigned long long)g.n[d]) oob = true;
      if (!oob) {  // the reference would write out of bounds here
        outer_flat = (long long)outer[g.dim - 1];
        inner_flat = (long long)inner[g.dim - 1];
        for (int d = g.dim - 1; d > 0; d--) {
          outer_flat = outer_flat * g.n[d - 1] + (long long)outer[d - 1];
          inner_flat = inner_flat * g.n[d - 1] + (long long)inner[d - 1];
        }
        do_copy = true;
      }
    }
  }
  // inner nodes lie inside the boundary, non-trivial outer nodes outside it, so the
  // copies are independent of each other: read all, then write all
  double v = 0;
  if (do_copy) v = acquire(&rec[inner_flat * g.rec]);  // (may have been published by another workgroup of this launch)
  if (BLOCK_SYNC) __syncthreads();
  if (do_copy) rec[outer_flat * g.rec] = v;
}
__device__ __forceinline__ void duplicate_boundary_block(const Geom &g, double *__restrict__ rec, const DupPlan &dp) {
  duplicate_boundary_lanes<true>(g, rec, dp, threadIdx.x);
}
__device__ __forceinline__ void duplicate_boundary_wave(const Geom &g, double *__restrict__ rec, const DupPlan &dp, int lane) {
  duplicate_boundary_lanes<false>(g, rec, dp, lane);
}

__global__ void k_duplicate_boundary(Geom g, double *__restrict__ rec, DupPlan dp, int *__restrict__ dirty_flag) {
  if (*dirty_flag == 0) return;
  duplicate_boundary_block(g, rec, dp);
  __syncthreads();
  if (threadIdx.x == 0) *dirty_flag = 0;
}

static DupPlan make_dup_plan(const Geom &g) {
  // grid.h:264-273 applied to the boundary corners, then the two while loops of
  // gaussian_grid.h:582-588 (host arithmetic, identical to the reference)
  DupPlan dp;
  for (int d = 0; d < 3; d++) dp.lo[d] = dp.hi[d] = 0;
  for (int d = 0; d < g.dim; d++) {
    double w;
    unsigned long long lo = (unsigned long long)node_index(g, d, g.bmin[d], &w);
    unsigned long long hi = (unsigned long long)node_index(g, d, g.bmax[d], &w);
    long long guard = 0;
    while ((double)lo * g.dx[d] + g.min[d] < g.bmin[d] && guard++ < (1LL << 31)) lo += 1;
    guard = 0;
    while (((double)hi * g.dx[d] + g.min[d] > g.bmax[d] || hi == (unsigned long long)g.n[d]) && guard++ < (1LL << 31)) hi -= 1;
    dp.lo[d] = lo;
    dp.hi[d] = hi;
  }
  return dp;
}

// Which 1-D gather tiles (tile_nodes nodes each) take the duplication ticket: see tile_ticket_duplicate.  A node can
// meet a non-zero boundary correction only where a wall blend is non-zero (pair_term: corr = (..) t2 + (..) t4, and
// node_terms: t2 != 0 only for bmin <= x < bmin + EDM_BC_MAR sigma, t4 != 0 only for bmax - EDM_BC_MAR sigma < x <= bmax);
// the duplication copies node lo -> lo - 1 and hi -> hi + 1 where those exist (duplicate_boundary_lanes).  Ranges are
// widened by two nodes against rounding in the node positions.  EDM_HIP_TEST_FORCE=dup_ticket_all: every tile (tests).
static void dup_ticket_tiles_1d(const Geom &g, PostArgs &post, unsigned ntile, unsigned tile_nodes) {
  post.tk_lo_end = post.tk_hi_begin = 0;   // every tile
  static const bool all_env = test_force("dup_ticket_all");
  if (all_env || g.dim != 1 || ntile == 0 || ntile >= 0xFFFFu) return;
  if (g.bper[0]) {   // no walls: no corrections, nothing to duplicate
    post.tk_hi_begin = ntile;
    return;
  }
  const DupPlan &dp = post.dp;
  const long long n = g.n[0];
  const bool lo_copy = dp.lo[0] > 0 && dp.lo[0] < (unsigned long long)n;
  const bool hi_copy = dp.hi[0] + 1 < (unsigned long long)n;
  if (!lo_copy && !hi_copy) {   // the walls sit on the grid's end nodes: nobody needs to know who was dirty
    post.tk_hi_begin = ntile;
    return;
  }
  const double reach = EDM_BC_MAR * g.sigma[0];
  long long a1 = (long long)ceil((g.bmin[0] + reach - g.min[0]) / g.dx[0]) + 2;    // last node of the lower set
  long long b0 = (long long)floor((g.bmax[0] - reach - g.min[0]) / g.dx[0]) - 2;   // first node of the upper set
  if (lo_copy && (long long)dp.lo[0] + 2 > a1) a1 = (long long)dp.lo[0] + 2;
  if ((long long)dp.hi[0] - 2 < b0) b0 = (long long)dp.hi[0] - 2;
  if (a1 < 0) a1 = 0;
  if (b0 < 0) b0 = 0;
  if (a1 > n - 1) a1 = n - 1;
  if (b0 > n - 1) b0 = n - 1;
  const unsigned lo_end = (unsigned)(a1 / tile_nodes) + 1, hi_begin = (unsigned)(b0 / tile_nodes);
  if (lo_end >= hi_begin || lo_end >= ntile) return;   // the two sets meet: every tile
  post.tk_lo_end = lo_end;
  post.tk_hi_begin = hi_begin;
}

hipError_t launch_hills_ordered(const Geom &g, const Tables &t, double *rec, const HillList &h, const OrderedParams &op,
                                const LimitTail &tail, double *heights_out, double *added_out,
                                LimitResult *result_dev, int *dirty_flag, hipStream_t s) {
  if (h.nh <= 0) return hipSuccess;
  const DupPlan dp = make_dup_plan(g);
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_hills_ordered<1>, dim3(1), dim3(ORD_BLOCK), 0, s, g, t, rec, h, op, tail, heights_out, added_out, result_dev, dp, dirty_flag); break;
    case 2: hipLaunchKernelGGL(k_hills_ordered<2>, dim3(1), dim3(ORD_BLOCK), 0, s, g, t, rec, h, op, tail, heights_out, added_out, result_dev, dp, dirty_flag); break;
    default: hipLaunchKernelGGL(k_hills_ordered<3>, dim3(1), dim3(ORD_BLOCK), 0, s, g, t, rec, h, op, tail, heights_out, added_out, result_dev, dp, dirty_flag); break;
  }
  return hipGetLastError();
}

// One launch for the bookkeeping that follows a limited hill batch: boundary duplication (K6, block 0)
// and both histogram updates (K7, remaining blocks): +1 per hill (new hills) or per replayed hill
// (flush), -1 per undo.
template <int DIM>
__global__ void __launch_bounds__(BLOCK) k_post_batch(Geom g, double *__restrict__ rec, DupPlan dp,
                                                      int *__restrict__ dirty_flag, Geom hg,
                                                      double *__restrict__ hist, long long nh,
                                                      const double *__restrict__ hx0,
                                                      const LimitResult *__restrict__ res,
                                                      const int *__restrict__ flags, int flush_mode) {
  if (blockIdx.x == 0) {
    if (*dirty_flag != 0) {
      duplicate_boundary_block(g, rec, dp);
      __syncthreads();
      if (threadIdx.x == 0) *dirty_flag = 0;
    }
    return;
  }
  if (res->error) return;
  // Small histograms (the 1-D CV has ~113 bins) are privatised: a million hills hammering a hundred global
  // addresses took 1.1 ms of a 8.8 ms all-samples step; per-workgroup LDS counts (integer-valued, so the order
  // of the adds is immaterial) cost one global atomic per touched bin and workgroup.
  __shared__ double s_hist[HIST_LDS_BINS];
  const bool priv = hg.total <= HIST_LDS_BINS;
  if (priv) {
    for (int k = threadIdx.x; k < (int)hg.total; k += BLOCK) s_hist[k] = 0.0;
    __syncthreads();
  }
  hist_batch<DIM>(hg, priv ? s_hist : hist, nh, hx0, res, flags, flush_mode,
                  (long long)(blockIdx.x - 1) * BLOCK + threadIdx.x, (long long)(gridDim.x - 1) * BLOCK);
  if (priv) {
    __syncthreads();
    for (int k = threadIdx.x; k < (int)hg.total; k += BLOCK) {
      const double c = s_hist[k];
      if (c != 0.0) atomicAdd(&hist[k], c);
    }
  }
}

template <int DIM>
__device__ __forceinline__ void hist_batch(const Geom &hg, double *hist, long long nh, const double *hx0,
                                           const LimitResult *res, const int *flags, int flush_mode, long long first,
                                           long long stride) {
  const long long k = res->k;
  if (res->nh < nh) nh = res->nh;  // deferred count
  for (long long i = first; i < nh; i += stride) {
    double wgt = flush_mode ? 0.0 : 1.0;
    if (i >= k) {
      const int fl = flags[i - k];
      if (flush_mode && (fl & 1)) wgt += 1.0;
      if (fl & 2) wgt -= 1.0;
    }
    if (wgt == 0.0) continue;
    double xx[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) xx[d] = hx0[i * DIM + d];
    if (!in_grid<DIM>(hg, xx)) continue;
    long long idx[DIM];
    bool ok = true;
#pragma unroll
    for (int d = 0; d < DIM; d++) {
      double wr;
      idx[d] = node_index(hg, d, xx[d], &wr);
      if (idx[d] < 0 || idx[d] >= hg.n[d]) ok = false;
    }
    if (!ok) continue;
    long long flat = idx[DIM - 1];
#pragma unroll
    for (int d = DIM - 1; d > 0; d--) flat = flat * hg.n[d - 1] + idx[d - 1];
    atomicAdd(&hist[flat], wgt);
  }
}

hipError_t launch_post_batch(const Geom &g, double *rec, int *dirty_flag, const Geom &hg, double *hist, long long nh,
                             const double *hx0, const LimitResult *res_dev, const int *flags, int flush_mode,
                             hipStream_t s) {
  const DupPlan dp = make_dup_plan(g);
  long long hb = (nh + BLOCK - 1) / BLOCK;
  if (hb > 1024) hb = 1024;
  if (hb < 1) hb = 1;
  const unsigned blocks = (unsigned)hb + 1;
  switch (g.dim) {
    case 1: hipLaunchKernelGGL(k_post_batch<1>, dim3(blocks), dim3(BLOCK), 0, s, g, rec, dp, dirty_flag, hg, hist, nh, hx0, res_dev, flags, flush_mode); break;
    case 2: hipLaunchKernelGGL(k_post_batch<2>, dim3(blocks), dim3(BLOCK), 0, s, g, rec, dp, dirty_flag, hg, hist, nh, hx0, res_dev, flags, flush_mode); break;
    default: hipLaunchKernelGGL(k_post_batch<3>, dim3(blocks), dim3(BLOCK), 0, s, g, rec, dp, dirty_flag, hg, hist, nh, hx0, res_dev, flags, flush_mode); break;
  }
  return hipGetLastError();
}

hipError_t launch_duplicate_boundary(const Geom &g, double *rec, int *dirty_flag, hipStream_t s) {
  const DupPlan dp = make_dup_plan(g);
  hipLaunchKernelGGL(k_duplicate_boundary, dim3(1), dim3(64), 0, s, g, rec, dp, dirty_flag);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K4: ordered bias limiter
//   new hills : edm_bias.cpp:465-495  (add, maybe undo, maybe defer)
//   flush     : edm_bias.cpp:323-356  (replay until the limit, undo the crossing hill)
// Hills [0,k) provably precede the first crossing and keep their full height; the
// ordered tail [k, nh) is walked by one thread exactly like the reference's loop.
// ---------------------------------------------------------------------------
// per chunk of EDM_CHUNK hills: sum and the maximum inclusive prefix
__global__ void __launch_bounds__(BLOCK) k_chunk_stats(long long nh, const double *__restrict__ added,
                                                       double *__restrict__ chunk_sum, double *__restrict__ chunk_max) {
  __shared__ double lds[BLOCK];
  constexpr int PER = EDM_CHUNK / BLOCK;
  const long long base = (long long)blockIdx.x * EDM_CHUNK + (long long)threadIdx.x * PER;
  double loc[PER];
  double s = 0, mx = -1e308;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    loc[j] = (base + j < nh) ? added[base + j] : 0.0;
    s += loc[j];
    if (base + j < nh && s > mx) mx = s;
  }
  lds[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    // exclusive prefix over thread sums (sequential: 256 adds)
    double run = 0;
    for (int j = 0; j < BLOCK; j++) {
      const double v = lds[j];
      lds[j] = run;
      run += v;
    }
    chunk_sum[blockIdx.x] = run;
  }
  __syncthreads();
  double my = (mx > -1e307) ? mx + lds[threadIdx.x] : -1e308;
  __syncthreads();
  // block max
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double other = __shfl_down(my, o, 64);
    if (other > my) my = other;
  }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = my;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = lds[0];
    for (int w = 1; w < BLOCK / 64; w++)
      if (lds[w] > m) m = lds[w];
    chunk_max[blockIdx.x] = m;
  }
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// (run by ONE wave: lanes 0..63 of a workgroup; COHERENT: `added` was published by other workgroups
// of the same launch)
template <bool COHERENT>
__device__ __forceinline__ void limit_wave(long long nh_bound, const double *added, const double *heights,
                                           double h_const, double limit, double cum_in, int flush_mode,
                                           const LimitTail &tail, LimitResult *res, long long nchunks,
                                           const double *chunk_sum, const double *chunk_max,
                                           const long long *nh_dev, long long mirror, long long *k_out, int *err_out,
                                           long long nh_known, LimitResult *out_local, const double *s_added) {
  // (out_local: the result as every lane of the wave holds it -- the walk is uniform -- for a caller that writes the
  //  header line to the host in one instruction)
  // (k_out / err_out: the first tail hill and the error code, for a caller that hands them on in registers;
  //  nh_known >= 0: the caller has already read *nh_dev)
  // `mirror` != 0: the result and the tail's flags / h2 / added2 live in the packed read-back region, whose copy in
  // host-mapped memory sits `mirror` bytes away -- they are stored to both as they are produced (see LimitArgs)
  // (COHERENT: the outputs are read by other workgroups of the SAME launch -- the gather of k_integrals_gather --
  //  so the device copies travel at agent scope like everything else that is handed between workgroups)
  auto put_f64 = [mirror](double *p, double v) {
    if (COHERENT) publish(p, v); else *p = v;
    if (mirror)
      __hip_atomic_store(reinterpret_cast<double *>(reinterpret_cast<char *>(p) + mirror), v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  };
  auto put_i32 = [mirror](int *p, int v) {
    if (COHERENT) publish(p, v); else *p = v;
    if (mirror)
      __hip_atomic_store(reinterpret_cast<int *>(reinterpret_cast<char *>(p) + mirror), v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  };
  auto put_i64 = [mirror](long long *p, long long v) {
    if (COHERENT) publish(p, v); else *p = v;
    if (mirror)
      __hip_atomic_store(reinterpret_cast<long long *>(reinterpret_cast<char *>(p) + mirror), v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  };
  int plain_l = flush_mode ? 0 : 1;   // new hills: so far every hill was added in full
  double h2_stop_l = 0;               // flush: undo height of the hill the flush stopped at
  auto put_result = [&](double cum_out, long long k_, long long nh_, int n_tail, int stop_, int n_def_, int error) {
    put_f64(&res->cum_out, cum_out);
    put_i64(&res->k, k_);
    put_i64(&res->nh, nh_);
    put_i32(&res->n_tail, n_tail);
    put_i32(&res->stop, stop_);
    put_i32(&res->n_deferred, n_def_);
    put_i32(&res->error, error);
    put_i32(&res->all_plain, (error == 0 && n_def_ == 0) ? plain_l : 0);
  };
  auto keep_local = [&](double cum_out, long long k_, long long nh_, int n_tail, int stop_, int n_def_, int error) {
    if (!out_local) return;
    out_local->cum_out = cum_out;
    out_local->k = k_;
    out_local->nh = nh_;
    out_local->n_tail = n_tail;
    out_local->stop = stop_;
    out_local->n_deferred = n_def_;
    out_local->error = error;
    out_local->all_plain = (error == 0 && n_def_ == 0) ? plain_l : 0;
    out_local->h2_stop = h2_stop_l;
  };
  long long nh = nh_bound;
  if (k_out) *k_out = 0;
  if (err_out) *err_out = 0;
  if (nh_dev) {
    nh = nh_known >= 0 ? nh_known : *nh_dev;
    if (nh > nh_bound) {  // the batch was queued with too small a bound: nothing is applied
      if (threadIdx.x == 0) put_result(cum_in, 0, nh, 0, 0, 0, 2);
      keep_local(cum_in, 0, nh, 0, 0, 0, 2);
      if (err_out) *err_out = 2;
      return;
    }
  }
  // One wave.  Every lane runs the same (uniform) walk; the hills of a 64-wide slab sit one per
  // lane in registers and are broadcast by v_readlane, so an iteration is a handful of dependent
  // fp64 instructions instead of an LDS round trip; lane j keeps the outcome of hill j and the
  // slab is stored coalesced.
  const int lane = threadIdx.x;
  long long k = 0;
  double cum = cum_in;  // new-hill mode: temp_hill_cum_; flush mode: bias added by this flush
  // skip whole chunks that cannot reach the limit (not in flush mode: the flush list is short)
  if (!flush_mode && nchunks > 0) {
    long long c = 0;
    bool done = false;
    for (long long cb = 0; cb < nchunks && !done; cb += 64) {
      const double cs_l = (cb + lane < nchunks) ? chunk_sum[cb + lane] : 0.0;
      const double cm_l = (cb + lane < nchunks) ? chunk_max[cb + lane] : 0.0;
      const int lim = (nchunks - cb < 64) ? (int)(nchunks - cb) : 64;
      for (int j = 0; j < lim; j++) {
        const double cs = readlane_f64(cs_l, j), cm = readlane_f64(cm_l, j);
        if (!(cum < limit) || cum + cm >= limit) {
          done = true;
          break;
        }
        cum += cs;
        c++;
      }
    }
    k = c * EDM_CHUNK;
    if (k > nh) k = nh;
  }
  const long long ntail = nh - k;
  if (ntail > EDM_TAIL_CAP) {
    if (lane == 0) put_result(cum, k, nh, 0, 0, 0, 1);
    keep_local(cum, k, nh, 0, 0, 0, 1);
    if (err_out) *err_out = 1;
    return;
  }
  // Ordered walk, one 64-hill slab at a time.  Between two crossings of the limit nothing depends
  // on more than the running sum, so each segment is one wave prefix-sum plus a ballot that finds
  // the next hill at which edm_bias.cpp:465 / :474 (or :334 of the flush) change regime; the loop
  // below runs (crossings + 1) times per slab instead of once per hill.
  int n_def = 0;
  int stop = (int)ntail;
  bool stopped = false;
  // (the next slab's hills are requested before the current slab is walked: one memory round trip per slab hidden)
  auto load_a = [&](long long i) {
    if (!(i < nh)) return 0.0;
    if (s_added) return s_added[i];   // (LDS: the polling limiter workgroup has the integrals already)
    return COHERENT ? acquire(&added[i]) : added[i];
  };
  auto load_h = [&](long long i) { return (i < nh) ? (heights ? heights[i] : h_const) : 0.0; };
  double a_next = load_a(k + lane), h_next = load_h(k + lane);
  for (long long base = 0; base < ntail; base += 64) {
    const double a_l = a_next, h_l = h_next;
    if (base + 64 < ntail) {
      a_next = load_a(k + base + 64 + lane);
      h_next = load_h(k + base + 64 + lane);
    }
    // add_value(pos, h) is linear in h: the undo hill's bias is h2 * (added / height)
    const double q_l = (h_l != 0.0) ? a_l / h_l : 0.0;
    double o_h1 = 0, o_h2 = 0, o_a2 = 0, o_cum = 0;
    int o_fl = 0;
    const int lim = (ntail - base < 64) ? (int)(ntail - base) : 64;
    int start = 0;
    while (start < lim) {
      if (flush_mode ? stopped : !(cum < limit)) {
        // flush already stopped: nothing more is replayed; new hills: every remaining hill is deferred
        if (lane >= start && lane < lim) {
          o_fl = flush_mode ? 0 : 4;
          o_cum = cum;
        }
        if (!flush_mode) n_def += lim - start;
        start = lim;
        break;
      }
      // inclusive prefix of the bias over lanes [start, lim)
      double ps = (lane >= start && lane < lim) ? a_l : 0.0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(ps, o, 64);
        if (lane >= o) ps += up;
      }
      const double P = cum + ps;
      // first hill of the segment after which the regime changes
      const bool hit = (lane >= start && lane < lim) && (flush_mode ? (P > limit) : !(P < limit));
      const unsigned long long bal = __ballot(hit);
      const int j1 = bal ? (int)__builtin_ctzll(bal) : lim;
      if (lane >= start && lane < j1 && lane < lim) {  // added in full
        o_h1 = h_l;
        o_fl = 1;
        o_cum = P;
      }
      if (j1 >= lim) {
        cum = readlane_f64(P, lim - 1);
        start = lim;
        break;
      }
      // hill j1: added, and undone if it overshoots (:474-490 / :334-355)
      const double Pj = readlane_f64(P, j1), hj = readlane_f64(h_l, j1), qj = readlane_f64(q_l, j1);
      double h2 = 0, a2 = 0;
      int fl = 1;
      double after = Pj;
      if (Pj > limit) {
        h2 = fmax(limit - Pj, -hj);
        a2 = h2 * qj;
        after = Pj + a2;
        fl = flush_mode ? 3 : 7;
        if (!flush_mode) n_def++;
        if (flush_mode) {
          stop = (int)(base + j1);
          stopped = true;
          h2_stop_l = h2;
        }
      }
      if (lane == j1) {
        o_h1 = hj;
        o_h2 = h2;
        o_a2 = a2;
        o_cum = after;
        o_fl = fl;
      }
      cum = after;
      start = j1 + 1;
    }
    if (__ballot(lane < lim && o_fl != 1) != 0ull) plain_l = 0;
    if (lane < lim) {
      const long long ti = base + lane;
      if (COHERENT) publish(&tail.h1[ti], o_h1); else tail.h1[ti] = o_h1;
      put_f64(&tail.h2[ti], o_h2);
      put_f64(&tail.added2[ti], o_a2);
      tail.cum_after[ti] = o_cum;
      put_i32(&tail.flags[ti], o_fl);
    }
  }
  if (lane == 0) put_result(cum, k, nh, (int)ntail, stop, n_def, 0);
  keep_local(cum, k, nh, (int)ntail, stop, n_def, 0);
  if (k_out) *k_out = k;
}

__global__ void __launch_bounds__(64) k_limit(long long nh_bound, const double *__restrict__ added,
                                              const double *__restrict__ heights, double h_const, double limit,
                                              double cum_in, int flush_mode, LimitTail tail,
                                              LimitResult *__restrict__ res, long long nchunks,
                                              const double *__restrict__ chunk_sum,
                                              const double *__restrict__ chunk_max,
                                              const long long *__restrict__ nh_dev) {
  limit_wave<false>(nh_bound, added, heights, h_const, limit, cum_in, flush_mode, tail, res, nchunks, chunk_sum,
                    chunk_max, nh_dev);
}

size_t limit_scratch_doubles(long long nh) { return (size_t)(2 * ((nh + EDM_CHUNK - 1) / EDM_CHUNK) + 8); }

hipError_t launch_limit(long long nh, const double *added, const double *heights, double h_const, double limit,
                        double cum_in, int flush_mode, const LimitTail &tail, LimitResult *result_dev,
                        double *scratch, hipStream_t s, const long long *nh_dev) {
  long long nchunks = 0;
  double *csum = scratch, *cmax = scratch;
  if (!flush_mode && nh > EDM_CHUNK) {
    nchunks = (nh + EDM_CHUNK - 1) / EDM_CHUNK;
    csum = scratch;
    cmax = scratch + nchunks;
    hipLaunchKernelGGL(k_chunk_stats, dim3((unsigned)nchunks), dim3(BLOCK), 0, s, nh, added, csum, cmax);
  }
  hipLaunchKernelGGL(k_limit, dim3(1), dim3(64), 0, s, nh, added, heights, h_const, limit, cum_in, flush_mode, tail,
                     result_dev, nchunks, csum, cmax, nh_dev);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------
// fix edm_pair in the reference's order (see OrderedForcesArgs in edm_kernels.h)
// ---------------------------------------------------------------------------
static constexpr int ORD_NODES = 32, ORD_PARTS = BLOCK / ORD_NODES, ORD_CHUNK = 128;   // (16-node tiles, twice the workgroups: no faster)
// (chunks of 64 hills -- one wave lists -- made a W1 step's 125 hills two passes of list / fetch / run / store, each pass
//  three dependent round trips: the densest tiles finished at 10 us; two waves list 128 hills in one pass)
static constexpr long long ORD_MAX_HILLS = 16384;   // (sample indices in LDS: 64 KB; list counts fit 16 bits)
// this rank's hills of the batch and the limiter's split index, wherever they are known (see OrderedForcesArgs)
__device__ __forceinline__ void ordered_batch_counts(const OrderedForcesArgs &a, long long &off, long long &nloc, long long &k,
                                                     bool force_pass = true) {
  off = a.range_dev ? a.range_dev[0] : a.hill_off;
  nloc = a.range_dev ? a.range_dev[1] : a.nh;
  k = a.k;
  if (a.res_dev) {
    if (!a.range_dev) nloc = a.res_dev->nh;
    if (a.res_dev->error) nloc = 0;
    k = a.res_dev->k;
  }
  if (a.wait_flag) {   // (the record pass ran beside the hill batch: the selection's count; the split index came with the
                       //  limiter's word, the record pass took it from there and left the error state for the force pass)
    if (a.range_dev) {   // (multi-GPU: this rank's slice of the global list, as k_unpack_prep left it)
      off = acquire(&a.range_dev[0]);
      nloc = acquire(&a.range_dev[1]);
    } else {
      nloc = acquire(a.nh_dev);
    }
    if (force_pass && *a.status) nloc = 0;
  }
  if (nloc > a.nh_cap) nloc = a.nh_cap;
  if (nloc < 0) nloc = 0;
}
long long ordered_max_hills() { return ORD_MAX_HILLS; }
long long ordered_tiles(const Geom &g) { return (g.n[0] + ORD_NODES - 1) / ORD_NODES; }
size_t ordered_record_doubles(const Geom &g, long long nh_cap) {
  return (size_t)ordered_tiles(g) * (size_t)(nh_cap > 0 ? nh_cap : 1) * ORD_NODES * 2;
}
size_t ordered_count_shorts(const Geom &g, long long nh_cap) {
  return (size_t)ordered_tiles(g) * (size_t)((nh_cap > 0 ? nh_cap : 0) + 1);
}
bool ordered_forces_supported(const Geom &g) {
  return g.dim == 1 && g.rec == 2 && g.n[0] >= 2 && !(g.periodic[0] && 2 * g.msize[0] + 1 > g.n[0]);
}

// signed stencil offset of node n in the stencil of a hill centred at node c (|o| <= msize when the hill covers
// the node), through the one periodic image a stencil narrower than the grid can reach a node by
__device__ __forceinline__ int ordered_offset(const Geom &g, int n, int c) {
  int o = n - c;
  if (g.periodic[0]) {
    if (o > g.msize[0]) o -= g.n[0];
    else if (o < -g.msize[0]) o += g.n[0];
  }
  return o;
}

// The running records of a step's hills, tile by tile.  A workgroup owns a tile of ORD_NODES nodes and walks the hill list
// ORD_CHUNK hills at a time.  Waves 0 and 1 test the chunk's hills against the tile and list the ones that reach it IN ORDER
// (ballot prefix; ~15 % of a W1 step's hills reach a given tile); the workgroup's ORD_PARTS parts compute the
// unit-height stencil terms of the listed hills side by side (value and derivative of every (listed hill, node) into
// LDS); part 0 then runs the heights over them in hill order -- rec += h1 term, then += h2 term where the limiter
// added an undo hill: the reference's sequence of += (gaussian_grid.h:343-355, edm_bias.cpp:474-490) -- and every
// listed hill's running record of the tile's nodes goes to records[tile][list position][node]; counts[m][tile] = how
// many of the first m hills the tile listed.  A node's record after the first m hills is then
// records[tile][counts[m][tile] - 1][node] -- or the node's record before the batch when that count is zero.
template <bool PERB>
__global__ void __launch_bounds__(BLOCK) k_ordered_records(Geom g, Tables t, OrderedForcesArgs a) {
  static_assert(ORD_CHUNK == 128 && BLOCK >= 128, "two waves test and list a chunk");
  const int tnode = threadIdx.x % ORD_NODES, part = threadIdx.x / ORD_NODES;
  const int tile = blockIdx.x, ntiles = gridDim.x;
  const int t0 = tile * ORD_NODES;
  const int n = t0 + tnode;
  const bool in_grid = n < g.n[0];
  const int t1 = (t0 + ORD_NODES - 1 < g.n[0] - 1) ? t0 + ORD_NODES - 1 : g.n[0] - 1;
  const int p[1] = {in_grid ? n : 0};
  unsigned long long *tr = a.trace ? a.trace + (size_t)blockIdx.x * 8 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = wall_clock64();
  NodeTerms<1> nt;
  nt.inside = true;
  if (!a.terms) node_terms<1, PERB>(g, t, p, nt);   // (with the terms stored by the batch's launch the node side is not needed)
  const bool active = in_grid && nt.inside;   // (hills skip nodes outside a wall, gaussian_grid.h:273)
  TermConst<1> tc;
  term_const<1>(g, tc);
  if (tr && threadIdx.x == 0) tr[1] = wall_clock64();
  double acc0 = 0, acc1 = 0;
  if (in_grid && part == 0 && !a.wait_flag) {   // (beside the hill batch: the grid's step-start copy is read behind the wait below)
    const double2 r0 = reinterpret_cast<const double2 *>(a.rec0)[n];
    acc0 = r0.x;
    acc1 = r0.y;
  }
  double2 *R = reinterpret_cast<double2 *>(a.records) + (long long)tile * a.nh_cap * ORD_NODES;
  __shared__ int s_c[ORD_CHUNK], s_cnt, s_row[ORD_CHUNK], s_wcnt[2];
  __shared__ double s_x[ORD_CHUNK], s_t[ORD_CHUNK][2], s_a1[ORD_CHUNK], s_a2[ORD_CHUNK];
  __shared__ double s_v[ORD_CHUNK][ORD_NODES], s_d[ORD_CHUNK][ORD_NODES];
  int listed = 0;       // hills of the earlier chunks the tile listed
  if (threadIdx.x == 0) a.counts[tile] = 0;   // (row m = 0)
  // this rank's hills: the whole batch, or (multi-GPU) its slice [off, off + nloc) of the rank-major global list --
  // the reference's ranks see their OWN hills of the step while they walk their pairs and replay the other ranks'
  // only in post_add_hill (edm_bias.cpp:565-583)
  long long off, nloc, k_split;
  if (a.wait_flag) {
    // this launch runs BESIDE the hill batch's (another stream, no event between them): everything above needed nothing
    // of the step.  The limiter's word is waited for first -- it comes from the batch's launch, which its stream started
    // behind the selection's: the prepared hills, their count and the grid's copy are complete, and read from here on
    // with agent-scope loads (no line of this XCD's L2 from an earlier step); then the limiter's tail heights and the
    // emitters' terms
    // (the gate wave ahead of this launch gave up -- kernels of different streams are being run one at a time: leave, the
    //  host queues this pass again behind the batch)
    if (acquire(a.status) == 2) return;
    __shared__ unsigned long long s_word;
    if (threadIdx.x == 0) s_word = wait_for_word(a.wait_flag, a.wait_seq, false);
    __syncthreads();
    ordered_batch_counts(a, off, nloc, k_split, false);
    const int state = ready_state_of(s_word);
    const bool failed = state != EDM_READY_BELOW && (state & ~EDM_READY_FINAL) != 0;   // limiter overflow / launch bound exceeded
    if (tile == 0 && threadIdx.x == 0) publish(a.status, failed ? 1 : 0);
    if (failed) return;   // (nothing was applied and the host redoes the step: the force pass counts zero hills)
    k_split = (state == EDM_READY_BELOW) ? off + nloc : (long long)ready_k_of(s_word);   // (indices of the batch's list: no tail)
    if (in_grid && part == 0) {   // (the selection's launch wrote this step's copy)
      acc0 = acquire(&a.rec0[2 * (long long)n]);
      acc1 = acquire(&a.rec0[2 * (long long)n + 1]);
    }
  } else {
    ordered_batch_counts(a, off, nloc, k_split, false);
  }
  for (long long base = 0; base < nloc; base += ORD_CHUNK) {
    const int cnt = (nloc - base < ORD_CHUNK) ? (int)(nloc - base) : ORD_CHUNK;
    // waves 0 and 1: one hill of the chunk per lane
    bool take = false;
    int c = INT_MIN, pos = 0;
    double hx = 0, ht0 = 0, ht1 = 0, a1 = 0, a2 = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < ORD_CHUNK) {
      if ((int)threadIdx.x < cnt) {
        const long long cur = off + base + threadIdx.x;   // (index in the batch's hill list)
        if (a.wait_flag) {
          c = acquire(&a.hc[cur]);
          hx = acquire(&a.hx[cur]);
          if (!PERB) {
            ht0 = acquire(&a.ht[2 * cur]);
            ht1 = acquire(&a.ht[2 * cur + 1]);
          }
          a1 = a.heights ? acquire(&a.heights[cur]) : a.h_const;
        } else {
          c = a.hc[cur];
          hx = a.hx[cur];
          if (!PERB) {
            ht0 = a.ht[2 * cur];
            ht1 = a.ht[2 * cur + 1];
          }
          a1 = a.heights ? a.heights[cur] : a.h_const;
        }
        if (cur >= k_split) {
          a1 = a.wait_flag ? acquire(&a.tail_h1[cur - k_split]) : a.tail_h1[cur - k_split];
          a2 = a.wait_flag ? acquire(&a.tail_h2[cur - k_split]) : a.tail_h2[cur - k_split];
        }
        // (c == INT_MIN: a hill rejected at preparation -- outside a wall, gaussian_grid.h:214-216; both heights zero:
        //  a hill the limiter deferred whole)
        take = c != INT_MIN && !(a1 == 0 && a2 == 0) && images(g, 0, c, t0, t1) != 0;
        if (take && a.terms_ready) {   // (the hill's two emitter workgroups, in the launch beside this one)
          const unsigned long long t_spin = wall_clock64();
          while (acquire(&a.terms_ready[ORD_EMIT_PARTS * cur]) != a.dirty_seq ||
                 acquire(&a.terms_ready[ORD_EMIT_PARTS * cur + 1]) != a.dirty_seq) {
            __builtin_amdgcn_s_sleep(7);
            if (wall_clock64() - t_spin > 1000000000ull) __builtin_trap();   // 10 s at 100 MHz: never, short of a lost launch
          }
        }
      }
      const unsigned long long bal = __ballot(take);
      pos = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wcnt[wave] = __popcll(bal);
    }
    __syncthreads();
    if (threadIdx.x < ORD_CHUNK) {
      if (wave == 1) pos += s_wcnt[0];   // (listed hills among the chunk's hills before this one)
      if ((int)threadIdx.x < cnt)        // row m = base + thread + 1: the first m hills
        a.counts[(base + threadIdx.x + 1) * ntiles + tile] = (unsigned short)(listed + pos + (take ? 1 : 0));
      if (take) {
        s_c[pos] = c;
        s_row[pos] = (int)(off + base + threadIdx.x);
        s_x[pos] = hx;
        s_t[pos][0] = ht0;
        s_t[pos][1] = ht1;
        s_a1[pos] = a1;
        s_a2[pos] = a2;
      }
      if (threadIdx.x == 0) s_cnt = s_wcnt[0] + s_wcnt[1];
    }
    __syncthreads();
    if (tr && threadIdx.x == 0 && base == 0) tr[2] = wall_clock64();
    const int nl = s_cnt;
    if (tr && threadIdx.x == 0 && base == 0) tr[7] = (unsigned long long)nl;
    // (one listed hill per thread and trip: four of them unrolled side by side were tried -- the terms' branches keep
    //  the chains from interleaving, and with ~6 listed hills per chunk the parts beyond the first two sat idle: slower)
    if (a.terms) {
      // the terms were stored by the emitters of the hill batch's launch (LimitArgs::ord_terms): one 16-byte load per
      // (listed hill, node) -- all of a thread's loads requested before the first is used
      for (int e = part; e < nl; e += ORD_PARTS) {
        double2 tv;
        tv.x = tv.y = 0.0;
        const int c = s_c[e];
        if (in_grid && images(g, 0, c, n, n) != 0) {
          const long long at = (long long)s_row[e] * (2 * g.msize[0] + 1) + (ordered_offset(g, n, c) + g.msize[0]);
          if (a.wait_flag) {   // (stored beside this launch: agent-scope loads, past this XCD's L2)
            tv.x = acquire(&a.terms[2 * at]);
            tv.y = acquire(&a.terms[2 * at + 1]);
          } else {
            tv = reinterpret_cast<const double2 *>(a.terms)[at];
          }
        }
        s_v[e][tnode] = tv.x;
        s_d[e][tnode] = tv.y;
      }
    } else {
      for (int e = part; e < nl; e += ORD_PARTS) {
        double val = 0, dval[1] = {0};
        if (active && images(g, 0, s_c[e], n, n) != 0) {
          bool nz = false;
          double v1, d1[1];
          if (pair_term<1, PERB>(g, tc, nt, &s_x[e], s_t[e], v1, d1, nz, false)) {
            val = v1;
            dval[0] = d1[0];
            if (nz) ordered_dirty_note(a.dirty_hill, a.dirty_seq, s_row[e]);
          }
        }
        s_v[e][tnode] = val;
        s_d[e][tnode] = dval[0];
      }
    }
    __syncthreads();
    if (tr && threadIdx.x == 0 && base == 0) tr[3] = wall_clock64();
    if (part == 0) {
      // (four listed hills per trip: their LDS reads are requested together and the products h * term computed off the
      //  chain; the adds -- rec += h1 term, then += h2 term where the limiter added an undo hill (a branch the whole
      //  wave takes or not), hill after hill -- stay one dependent chain.  No test for zero terms: a node the hill's
      //  stencil does not reach adds h * 0, which leaves the record's value as it is -- a per-lane branch around it
      //  cost more than the adds: ~100 ns per listed hill, 3.6 us for the densest tile's 37)
      for (int e0 = 0; e0 < nl; e0 += 4) {
        double pv[4], pd[4], qv[4], qd[4];
        bool undo[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int e = (e0 + q < nl) ? e0 + q : e0;
          const double v = s_v[e][tnode], d = s_d[e][tnode], h1 = s_a1[e], h2 = s_a2[e];
          pv[q] = h1 * v;
          pd[q] = h1 * d;
          qv[q] = h2 * v;
          qd[q] = h2 * d;
          undo[q] = h2 != 0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (e0 + q < nl) {
            acc0 += pv[q];
            acc1 += pd[q];
            if (undo[q]) {
              acc0 += qv[q];
              acc1 += qd[q];
            }
            s_v[e0 + q][tnode] = acc0;
            s_d[e0 + q][tnode] = acc1;
          }
        }
      }
    }
    __syncthreads();
    if (tr && threadIdx.x == 0 && base == 0) tr[4] = wall_clock64();
    for (int e = part; e < nl; e += ORD_PARTS) {
      double2 out;
      out.x = s_v[e][tnode];
      out.y = s_d[e][tnode];
      R[(long long)(listed + e) * ORD_NODES + tnode] = out;
    }
    listed += nl;
    __syncthreads();
    if (tr && threadIdx.x == 0 && base == 0) tr[5] = wall_clock64();
  }
  if (tr && threadIdx.x == 0) tr[6] = wall_clock64();
}

// One wave that waits for the limiter's word of a batch and then for every emitter's flag: queued ahead of a record pass
// that runs beside the batch's launch (OrderedForcesArgs::wait_flag), it holds that pass's workgroups back until
// nothing they need is outstanding.  They would wait themselves -- but with 70 KB of LDS each on two thirds of the CUs,
// out of which the batch's own integrals and emitters would have to stay (measured: the W1 step 67 us with the record
// pass dispatched at once, 60 behind this wave); and a workgroup that holds LDS while it waits for workgroups that
// still need a CU is how two tenants of one GPU can lock each other up: this wave holds nothing.
// (Letting the record pass out as soon as the emitters are done, to list and fetch terms before the word is there, was
//  tried: the emitters finish about when the word arrives -- 10 us into the launch -- and until the parked gather tiles
//  have run there is no LDS for a record workgroup anyway: no faster.)
// It does not wait for ever: where kernels of different streams are run one at a time (a profiler collecting hardware
// counters serialises every queue of the process), this wave may be let in ahead of the batch it waits for, and the
// batch then never starts.  After ORD_GATE_TICKS it writes 2 to *status and its host-mapped twin and leaves; the record and force
// pass behind it see that and leave at once, and the host, once the batch is through, queues both again behind it
// (edm_bias.cpp, ordered_step_finish) and stops using the second stream.
static constexpr unsigned long long ORD_GATE_TICKS = 200000ull;   // 2 ms of the 100 MHz wall clock: 100x a healthy wait
// (with a communicator the batch sits behind a collective, which waits for the slowest rank: OrderedForcesArgs::gate_ticks)
__global__ void __launch_bounds__(64) k_wait_word(const unsigned long long *word, unsigned long long seq,
                                                  const long long *nh_dev, long long nh_cap, const unsigned *terms_ready,
                                                  unsigned ready_seq, int *status, int *status_host, unsigned long long limit) {
  __shared__ unsigned long long s_w;
  __shared__ int s_gave_up;
  if (threadIdx.x == 0) {
    const unsigned long long want = seq & 0xFFFFFFFFFFull;
    const unsigned long long t0 = wall_clock64();
    s_gave_up = 0;
    for (;;) {
      const unsigned long long w = acquire(word);
      if (ready_seq_of(w) == want) {
        s_w = w;
        break;
      }
      __builtin_amdgcn_s_sleep(7);
      if (wall_clock64() - t0 > limit) {
        s_gave_up = 1;
        break;
      }
    }
  }
  __syncthreads();
  if (s_gave_up) {
    if (threadIdx.x == 0) {
      publish(status, 2);
      __hip_atomic_store(status_host, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const int state = ready_state_of(s_w);
  if (state != EDM_READY_BELOW && (state & ~EDM_READY_FINAL) != 0) return;   // (the batch was refused: no terms are needed)
  // (the word comes from the batch's launch, which started behind the selection's: the count is this step's)
  long long n = acquire(nh_dev);
  if (n > nh_cap) n = nh_cap;
  const unsigned long long t0 = wall_clock64();
  bool late = false;
  for (long long i = threadIdx.x; i < n * ORD_EMIT_PARTS && !late; i += 64)
    while (acquire(&terms_ready[i]) != ready_seq) {
      __builtin_amdgcn_s_sleep(7);
      if (wall_clock64() - t0 > limit) {
        late = true;
        break;
      }
    }
  if (late) {
    publish(status, 2);
    __hip_atomic_store(status_host, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
hipError_t launch_ordered_records(const Geom &g, const Tables &t, const OrderedForcesArgs &a, hipStream_t s) {
  if (!ordered_forces_supported(g) || (!a.range_dev && !a.res_dev && a.nh > a.nh_cap) || a.nh_cap > ORD_MAX_HILLS) return hipErrorInvalidValue;
  const unsigned nb = (unsigned)ordered_tiles(g);
  if (a.wait_flag) {
    // (tests: EDM_HIP_TEST_FORCE=ord_gate_giveup makes the gate wait for a word that never comes)
    static const bool never = test_force("ord_gate_giveup");
    hipLaunchKernelGGL(k_wait_word, dim3(1), dim3(64), 0, s, a.wait_flag, a.wait_seq + (never ? 777777ull : 0ull), a.nh_dev, a.nh_cap,
                       a.terms_ready, a.dirty_seq, a.status, a.status_host, a.gate_ticks ? a.gate_ticks : ORD_GATE_TICKS);
  }
  if (g.bper[0])
    hipLaunchKernelGGL(k_ordered_records<true>, dim3(nb), dim3(BLOCK), 0, s, g, t, a);
  else
    hipLaunchKernelGGL(k_ordered_records<false>, dim3(nb), dim3(BLOCK), 0, s, g, t, a);
  return hipGetLastError();
}

struct OrderedCommon {
  int H;                        // hills
  int ntiles;
  const int *samples;           // LDS: sample index of hill j, ascending
  int first_dirty;
  int lo_t, lo_s, hi_t, hi_s;   // outward copy nodes of the boundary duplication and their sources (-1: none)
  bool fast;                    // the specialised 1-D lookup applies (pair_fast_path)
  double inv_dx;
};
__device__ __forceinline__ void ordered_common_init(const Geom &g, const OrderedForcesArgs &a, const DupPlan &dp, int *s_samples,
                                                    OrderedCommon &oc) {
  long long off, nloc, k_split;
  ordered_batch_counts(a, off, nloc, k_split);
  oc.H = (int)nloc;
  for (int i = threadIdx.x; i < oc.H; i += blockDim.x) s_samples[i] = a.sel ? (int)a.sel[i] : i;
  __syncthreads();
  oc.samples = s_samples;
  oc.ntiles = (g.n[0] + ORD_NODES - 1) / ORD_NODES;
  {
    // the first of THIS RANK'S hills with a non-zero boundary correction (list position within its slice)
    __shared__ int s_fd;
    if (threadIdx.x == 0) s_fd = INT_MAX;
    __syncthreads();
    for (int i = threadIdx.x; i < oc.H; i += blockDim.x)
      if (a.dirty_hill[off + i] == a.dirty_seq) atomicMin(&s_fd, i);
    __syncthreads();
    oc.first_dirty = s_fd;
  }
  oc.lo_t = oc.lo_s = oc.hi_t = oc.hi_s = -1;
  if (!g.bper[0]) {   // duplicate_boundary_lanes' cases 0 and 3 in one dimension
    if (dp.lo[0] > 0 && dp.lo[0] < (unsigned long long)g.n[0]) {
      oc.lo_t = (int)dp.lo[0] - 1;
      oc.lo_s = (int)dp.lo[0];
    }
    if (dp.hi[0] + 1 < (unsigned long long)g.n[0]) {
      oc.hi_t = (int)dp.hi[0] + 1;
      oc.hi_s = (int)dp.hi[0];
    }
  }
  oc.fast = g.interp && !g.periodic[0] && !g.bper[0];
  oc.inv_dx = 1.0 / g.dx[0];
}
// the node records as they stood after the first m hills of the batch, with the boundary duplication of
// gaussian_grid.h:571-630 applied: after every hill with a non-zero correction the value (not the derivative) of the
// first / last in-boundary node is copied to its outward neighbour -- so from the first such hill on an outward copy
// node reads its source's value
struct OrderedSource {
  const OrderedForcesArgs &a;
  const OrderedCommon &oc;
  int m;
  // optional: rows [row0, row0 + nrows) of the counts staged in LDS by the workgroup (its pairs are contiguous, so
  // their hill counts m span a handful of rows): the count then costs no global round trip
  const unsigned short *rows = nullptr;
  int row0 = 0, nrows = 0;
  __device__ __forceinline__ double2 raw(int node) const {
    const int tile = node / ORD_NODES;
    const int u = (m >= row0 && m < row0 + nrows) ? rows[(m - row0) * oc.ntiles + tile]
                                                  : a.counts[(long long)m * oc.ntiles + tile];
    if (u == 0) return reinterpret_cast<const double2 *>(a.rec0)[node];
    return reinterpret_cast<const double2 *>(a.records)[((long long)tile * a.nh_cap + (u - 1)) * ORD_NODES + (node % ORD_NODES)];
  }
  __device__ __forceinline__ double2 get(int node) const {
    double2 own = raw(node);
    if (m > oc.first_dirty && (node == oc.lo_t || node == oc.hi_t)) own.x = raw(node == oc.lo_t ? oc.lo_s : oc.hi_s).x;
    return own;
  }
  __device__ __forceinline__ void load(Rec<2> &r, long long node) const {
    const double2 v = get((int)node);
    r.v[0] = v.x;
    r.v[1] = v.y;
  }
};
// pair_one<false> (the specialised 1-D lookup: same index rule, same blend) on records that come from a source
template <class SRC>
__device__ __forceinline__ void pair_one_src(const Geom &g, const SRC &src, double inv_dx, double x, double &v, double &d) {
  const double lo_ok = fmax(g.bmin[0], g.min[0]);
  const double hi_open = fmin(nextafter(g.bmax[0], 1.0e308), g.max[0] - g.dx[0]);
  const bool in_range = (x >= lo_ok) & (x < hi_open);
  const double q = (x - g.min[0]) * inv_dx;
  double fq = floor(q);
  const double eps = 1e-11 * fmax(1.0, (double)g.n[0]);
  const double frac = q - fq;
  const bool near = in_range & ((frac <= eps) | (frac >= 1.0 - eps));
  if (near) fq = floor((x - g.min[0]) / g.dx[0]);
  int idx = (int)fq;
  idx = idx < 0 ? 0 : idx;
  idx = idx > g.n[0] - 2 ? g.n[0] - 2 : idx;
  const double where = x - g.min[0] - fq * g.dx[0];
  const double X = where * inv_dx;
  const double2 ra = src.get(idx), rb = src.get(idx + 1);
  double vv, dd;
  if ((X < 0.0) | (X > 1.0))   // `where` off by an ulp at a node: the reference's fabs() mirroring
    hermite_1d_mirrored(ra.x, ra.y, rb.x, rb.y, X, g.dx[0], inv_dx, vv, dd);
  else
    hermite_1d_horner(ra.x, scaled_slope(ra.x, ra.y, g.dx[0]), rb.x, scaled_slope(rb.x, rb.y, g.dx[0]), X, inv_dx, vv, dd);
  v = in_range ? vv : 0.0;
  d = in_range ? dd : 0.0;
}
// energy and dV/dr at r as the reference's loop saw them at the sample index `fs` of the pair's first add_hill call
__device__ __forceinline__ int ordered_hills_before(const OrderedCommon &oc, long long fs, int lo = 0, int hi = -1) {
  if (hi < 0) hi = oc.H;   // number of hills whose sample index is below fs, known to lie in [lo, hi]
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((long long)oc.samples[mid] < fs) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ void ordered_lookup_m(const Geom &g, const OrderedForcesArgs &a, const OrderedCommon &oc, double x,
                                                 int m, double &v, double &d, const unsigned short *rows = nullptr,
                                                 int row0 = 0, int nrows = 0) {
  const OrderedSource src{a, oc, m, rows, row0, nrows};
  if (oc.fast)
    pair_one_src(g, src, oc.inv_dx, x, v, d);
  else
    lookup_one_src<1>(g, src, &x, v, &d);
}
__device__ __forceinline__ void ordered_lookup(const Geom &g, const OrderedForcesArgs &a, const OrderedCommon &oc, double x,
                                               long long fs, double &v, double &d) {
  ordered_lookup_m(g, a, oc, x, ordered_hills_before(oc, fs), v, d);
}

// One pair of the force pass, lean form: the specialised 1-D lookup (pair_one_src's index rule and blend, word for word)
// for a workgroup whose rows of the counts are all in LDS.  No flat loads -- a load that might be LDS or global waits for both counters, and with
// them for the next pair's prefetched distance -- no divergent branches but the two rare ones, 32-bit index
// arithmetic.  ~230 VALU instructions per pair in the general form, which the pass is bound by (16 waves per CU busy
// through all four trips, in-kernel stamps), against ~63 in K1.
struct OrderedLean {
  const double2 *rec0;
  long long records_minus_rec0;   // (records - rec0, in 16-byte records: both come from hipMalloc, 16-byte aligned at least)
  const unsigned short *rows;   // LDS
  int row0, ntiles, nh_cap;
  int lo_t, hi_t, first_dirty;   // outward copy nodes of the boundary duplication (-1: none), first hill that duplicates
  double lo_ok, hi_open, eps, inv_dx;
};
__device__ __forceinline__ double2 ordered_lean_record(const OrderedLean &L, int m, int node) {
  const int tile = node >> 5;   // ORD_NODES == 32
  const int u = L.rows[(m - L.row0) * L.ntiles + tile];
  // (one address, selected arithmetically: a pointer chosen by `u ? ... : ...` became a divergent branch per record)
  const long long in_records = L.records_minus_rec0 + (long long)((tile * L.nh_cap + (u - 1)) * ORD_NODES + (node & (ORD_NODES - 1)));
  const long long idx = u ? in_records : (long long)node;
  return L.rec0[idx];
}
// in two halves, so that a thread can have two pairs' records in flight: the loads ...
struct OrderedLeanLoaded {
  double2 ra, rb;
  double X;
  bool in_range, ok;   // ok == false: the pair touches an outward copy node after a hill that duplicates -- one grid cell
                       // per wall; the caller takes the general form for it
};
__device__ __forceinline__ OrderedLeanLoaded ordered_lean_load(const Geom &g, const OrderedLean &L, int m, double x) {
  OrderedLeanLoaded o;
  o.in_range = (x >= L.lo_ok) & (x < L.hi_open);
  const double q = (x - g.min[0]) * L.inv_dx;
  double fq = floor(q);
  const double frac = q - fq;
  const bool near = o.in_range & ((frac <= L.eps) | (frac >= 1.0 - L.eps));
  if (near) fq = floor((x - g.min[0]) / g.dx[0]);
  int idx = (int)fq;
  idx = idx < 0 ? 0 : idx;
  idx = idx > g.n[0] - 2 ? g.n[0] - 2 : idx;
  const double where = x - g.min[0] - fq * g.dx[0];
  o.X = where * L.inv_dx;
  o.ok = !((m > L.first_dirty) & ((idx == L.lo_t) | (idx + 1 == L.lo_t) | (idx == L.hi_t) | (idx + 1 == L.hi_t)));
  o.ra = ordered_lean_record(L, m, idx);
  o.rb = ordered_lean_record(L, m, idx + 1);
  return o;
}
// ... and the blend
__device__ __forceinline__ void ordered_lean_blend(const Geom &g, const OrderedLean &L, const OrderedLeanLoaded &o, double &v,
                                                   double &d) {
  double vv, dd;
  if ((o.X < 0.0) | (o.X > 1.0))   // `where` off by an ulp at a node: the reference's fabs() mirroring
    hermite_1d_mirrored(o.ra.x, o.ra.y, o.rb.x, o.rb.y, o.X, g.dx[0], L.inv_dx, vv, dd);
  else
    hermite_1d_horner(o.ra.x, scaled_slope(o.ra.x, o.ra.y, g.dx[0]), o.rb.x, scaled_slope(o.rb.x, o.rb.y, g.dx[0]), o.X, L.inv_dx,
                      vv, dd);
  v = o.in_range ? vv : 0.0;
  d = o.in_range ? dd : 0.0;
}

__global__ void __launch_bounds__(BLOCK) k_pair_forces_ordered(Geom g, OrderedForcesArgs a, DupPlan dp,
                                                               double *__restrict__ block_energy, unsigned long long tag,
                                                               long long per_block) {
  extern __shared__ int s_samples[];
  __shared__ double red[BLOCK / 64];
  __shared__ int s_fd;
  constexpr int ROWS_LDS = 4096;   // 16-bit counts staged per workgroup: 8 KB, eleven rows of the C1D grid's 351 tiles
  __shared__ unsigned short s_rows[ROWS_LDS];
  // a workgroup owns a contiguous run of pairs: their hill counts m span a handful of consecutive rows of the counts
  // (a W1 step has ~8 000 pairs between two hills), staged in LDS so that a pair's chain is distance -> record, two
  // round trips, not distance -> count -> record
  // XCD-aware: workgroup i runs on XCD i mod 8, each with an L2 of its own.  Handing XCD x the x-th EIGHTH of the pairs
  // (workgroups x, x + 8, x + 16, ... take consecutive runs of it) keeps each L2 to the records behind an eighth of
  // the step's hills: with runs dealt out round robin every XCD pulled the whole table through the fabric -- counted
  // traffic 38 MB per launch against 20 MB of distances, sample indices and forces.
  unsigned run = blockIdx.x;
  if ((gridDim.x & 7u) == 0) run = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long long beg = (long long)run * per_block;
  const long long end = (beg + per_block < a.n) ? beg + per_block : a.n;
  // (the record pass ahead of this launch left without its records -- its gate gave up, see k_wait_word: leave too, without
  //  a partial sum: the host's poll runs out, and it queues both passes again)
  if (a.wait_flag && *a.status == 2) return;
  unsigned long long *tr = a.trace ? a.trace + (size_t)blockIdx.x * 8 : nullptr;   // (development aid: EDM_HIP_TRACE=k1o)
  if (tr && threadIdx.x == 0) tr[0] = wall_clock64();
  // ---- prologue: everything it reads from memory leaves at once (the hills' sample indices and correction flags, the
  //      run's first and last sample index, the first pair), two barriers ----
  long long i = beg + threadIdx.x;
  double x_next = 0.0;
  int fs_next = 0;   // (kept as loaded: a conversion here would wait for the load)
  if (i < end) {
    x_next = a.r[i];
    if (a.first_sample) fs_next = a.first_sample[i];
  }
  long long fs0 = 0, fs1 = -1;
  if (beg < end) {
    fs0 = a.first_sample ? (long long)a.first_sample[beg] : 2 * beg;
    fs1 = a.first_sample ? (long long)a.first_sample[end - 1] : 2 * (end - 1);
  }
  OrderedCommon oc;
  {
    long long off, nloc, k_split;
    ordered_batch_counts(a, off, nloc, k_split);
    oc.H = (int)nloc;
    int my_first = INT_MAX;   // the first of THIS RANK'S hills with a non-zero boundary correction (position in its slice)
    for (int k = threadIdx.x; k < oc.H; k += blockDim.x) {
      s_samples[k] = a.sel ? (int)a.sel[k] : k;
      if (a.dirty_hill[off + k] == a.dirty_seq && k < my_first) my_first = k;
    }
    if (threadIdx.x == 0) s_fd = INT_MAX;
    __syncthreads();
    if (my_first != INT_MAX) atomicMin(&s_fd, my_first);
    oc.samples = s_samples;
    oc.ntiles = (g.n[0] + ORD_NODES - 1) / ORD_NODES;
    oc.lo_t = oc.lo_s = oc.hi_t = oc.hi_s = -1;
    if (!g.bper[0]) {   // duplicate_boundary_lanes' cases 0 and 3 in one dimension
      if (dp.lo[0] > 0 && dp.lo[0] < (unsigned long long)g.n[0]) {
        oc.lo_t = (int)dp.lo[0] - 1;
        oc.lo_s = (int)dp.lo[0];
      }
      if (dp.hi[0] + 1 < (unsigned long long)g.n[0]) {
        oc.hi_t = (int)dp.hi[0] + 1;
        oc.hi_s = (int)dp.hi[0];
      }
    }
    oc.fast = g.interp && !g.periodic[0] && !g.bper[0];
    oc.inv_dx = 1.0 / g.dx[0];
  }
  if (tr && threadIdx.x == 0) tr[1] = wall_clock64();
  double e_acc = 0;
  int row0 = 0, row1 = 0, nrows = 0;
  if (beg < end) {
    row0 = ordered_hills_before(oc, fs0);
    row1 = ordered_hills_before(oc, fs1, row0);
    nrows = row1 - row0 + 1;
    if (nrows * oc.ntiles > ROWS_LDS) nrows = ROWS_LDS / oc.ntiles;
    for (int e = threadIdx.x; e < nrows * oc.ntiles; e += BLOCK) s_rows[e] = a.counts[(long long)row0 * oc.ntiles + e];
  }
  __syncthreads();
  oc.first_dirty = s_fd;
  if (tr && threadIdx.x == 0) tr[2] = wall_clock64();
  const bool lean = oc.fast && nrows == row1 - row0 + 1 &&
                    (long long)oc.ntiles * a.nh_cap * ORD_NODES < (1ll << 31);   // (32-bit record offsets)
  OrderedLean L;
  L.rec0 = reinterpret_cast<const double2 *>(a.rec0);
  L.records_minus_rec0 = reinterpret_cast<const double2 *>(a.records) - reinterpret_cast<const double2 *>(a.rec0);
  L.rows = s_rows;
  L.row0 = row0;
  L.ntiles = oc.ntiles;
  L.nh_cap = (int)a.nh_cap;
  L.lo_t = oc.lo_t;
  L.hi_t = oc.hi_t;
  L.first_dirty = oc.first_dirty;
  L.lo_ok = fmax(g.bmin[0], g.min[0]);
  L.hi_open = fmin(nextafter(g.bmax[0], 1.0e308), g.max[0] - g.dx[0]);
  L.eps = 1e-11 * fmax(1.0, (double)g.n[0]);
  L.inv_dx = oc.inv_dx;
  if (lean) {
    // two pairs per trip: both pairs' records are requested before either is blended (a record comes from the L2 of
    // the XCD that wrote it or from memory: ~1.2 us under load, the trip's length); the next two pairs' distances and
    // sample indices travel meanwhile (the loads do not move above the force stores by themselves)
    // (the prefetches are unconditional loads at clamped indices -- behind `if (j < end)` the compiler could not count
    //  the loads in flight and waited for ALL of them, the just-issued ones included, at the top of every trip; without
    //  sample indices the distances stand in as something valid to read, the value is not used)
    const bool has_fs = a.first_sample != nullptr;
    const int *fsp = has_fs ? a.first_sample : reinterpret_cast<const int *>(a.r);
    const long long last = end - 1;
    double x_next1 = a.r[(i + BLOCK < end) ? i + BLOCK : last];
    int fs_next1 = fsp[(i + BLOCK < end) ? i + BLOCK : last];
    for (; i < end; i += 2 * BLOCK) {
      const long long i1 = i + BLOCK;
      const bool two = i1 < end;
      const double x0 = x_next, x1 = x_next1;
      const long long f0 = has_fs ? (long long)fs_next : 2 * i;
      const long long f1 = has_fs ? (long long)fs_next1 : 2 * i1;
      {
        const long long j0 = (i + 2 * BLOCK < end) ? i + 2 * BLOCK : last, j1 = (i + 3 * BLOCK < end) ? i + 3 * BLOCK : last;
        x_next = a.r[j0];
        fs_next = fsp[j0];
        x_next1 = a.r[j1];
        fs_next1 = fsp[j1];
      }
      // (sample indices ascend with the pair index in the fix's list, so the count lies between the run's ends: a step
      //  or none of the search instead of log2(hills); any other order: the whole list, and the general form -- the
      //  pair's row of the counts is not staged)
      const bool in0 = (f0 >= fs0) & (f0 <= fs1), in1 = two & (f1 >= fs0) & (f1 <= fs1);
      const int m0 = ordered_hills_before(oc, f0, in0 ? row0 : 0, in0 ? row1 : oc.H);
      const int m1 = two ? ordered_hills_before(oc, f1, in1 ? row0 : 0, in1 ? row1 : oc.H) : row0;
      const OrderedLeanLoaded o0 = ordered_lean_load(g, L, in0 ? m0 : row0, x0);
      const OrderedLeanLoaded o1 = ordered_lean_load(g, L, in1 ? m1 : row0, x1);
      double v0, d0, v1 = 0.0, d1 = 0.0;
      ordered_lean_blend(g, L, o0, v0, d0);
      ordered_lean_blend(g, L, o1, v1, d1);
      if (!in0 || !o0.ok) ordered_lookup_m(g, a, oc, x0, m0, v0, d0, s_rows, row0, nrows);
      if (two && (!in1 || !o1.ok)) ordered_lookup_m(g, a, oc, x1, m1, v1, d1, s_rows, row0, nrows);
      e_acc += v0;
      a.force[i] = 0.0 - d0;
      if (two) {
        e_acc += v1;
        a.force[i1] = 0.0 - d1;
      }
      if (tr && threadIdx.x == 0 && i < beg + BLOCK) tr[3] = wall_clock64();   // (first trip done)
    }
  } else {
    for (; i < end; i += BLOCK) {
      const double x = x_next;
      const long long fs = a.first_sample ? (long long)fs_next : 2 * i;
      const long long j = i + BLOCK;
      if (j < end) {
        x_next = a.r[j];
        if (a.first_sample) fs_next = a.first_sample[j];
      }
      const bool inside = (fs >= fs0) & (fs <= fs1);
      const int m = ordered_hills_before(oc, fs, inside ? row0 : 0, inside ? row1 : oc.H);
      double v, d;
      ordered_lookup_m(g, a, oc, x, m, v, d, s_rows, row0, nrows);
      e_acc += v;
      a.force[i] = 0.0 - d;
    }
  }
  if (tr && threadIdx.x == 0) tr[4] = wall_clock64();
  const double sum = block_sum(e_acc, red);
  if (threadIdx.x == 0) {
    if (tag) store_partial_tagged(block_energy, blockIdx.x, sum, tag);
    else block_energy[blockIdx.x] = sum;
  }
}

// The force pass for LONG pair arrays (>= PAIR_LDS_THRESHOLD pairs), K1's own layout: one workgroup of 1024 threads per CU,
// a window of the grid in LDS -- here the window holds the node records as they stood behind the first m hills, m = the
// hill count of the pairs being looked up.  A workgroup owns a contiguous run of pairs; sample indices ascend with the
// pair index in the fix's list, so the run's hill counts are a handful of consecutive values m = row0 .. row1 and the
// pairs of one value are contiguous: per value the window is staged once (from the running records, through the
// counts: ~125 KB of L2 reads) and every thread walks on through its own pairs -- pair t, t + 1024, ... -- until it
// meets one of a later value, where it waits for that pass.  Pairs the scheme does not cover -- a sample index outside the
// run's ends (an array that does not ascend), a corner outside the window, the reference's mirrored blend -- take the
// general form with the records read from memory, like the short-array kernel's rare cases.  With two random 128-byte
// lines per pair out of a 2.5 MB table the short-array kernel is bound by L1 line fills at this size: 402 us for the
// 38.8 M pairs of W2, 0.19 of the HBM roofline.
static constexpr int ORD_WIN_HILLS = 2048;                                              // sample indices staged in LDS
static constexpr int ORD_WIN_NODES = (163840 - 256 - ORD_WIN_HILLS * 4 - 64) / 16;      // 9708 nodes
// one pair of the window form: from the window if its hill count is the staged one and its cell lies inside, else the
// general form
struct OrderedWin {
  const double2 *win;   // LDS
  int w0, wn, m;
  double lo_ok, hi_open, eps;
};
__device__ __forceinline__ bool ordered_win_pair(const Geom &g, const OrderedCommon &oc, const OrderedWin &W, double x, bool staged,
                                                 double &v, double &d) {
  // straight-line (predicated) like K1's pair_one, so that a thread's two lookups interleave.  Returns whether the pair
  // has to take the general form instead: a hill count that is not the staged one, a cell outside the window, `where`
  // off by an ulp at a node (the reference's mirrored blend)
  const bool in_range = (x >= W.lo_ok) & (x < W.hi_open);
  const double q = (x - g.min[0]) * oc.inv_dx;
  double fq = floor(q);
  const double frac = q - fq;
  const bool near = in_range & ((frac <= W.eps) | (frac >= 1.0 - W.eps));
  if (wave_any(near)) {
    if (near) fq = floor((x - g.min[0]) / g.dx[0]);
  }
  int idx = (int)fq;
  idx = idx < 0 ? 0 : idx;
  idx = idx > g.n[0] - 2 ? g.n[0] - 2 : idx;
  const double where = x - g.min[0] - fq * g.dx[0];
  const double X = where * oc.inv_dx;
  const int li = idx - W.w0;
  const bool inw = (unsigned)li < (unsigned)(W.wn - 1);   // 0 <= li and idx + 1 inside the window, in one comparison
  const int lc = inw ? li : 0;
  const double2 ra = W.win[lc], rb = W.win[lc + 1];
  double vv, dd;
  hermite_1d_horner(ra.x, ra.y, rb.x, rb.y, X, oc.inv_dx, vv, dd);
  v = in_range ? vv : 0.0;
  d = in_range ? dd : 0.0;
  return !staged | (in_range & (!inw | (X < 0.0) | (X > 1.0)));
}
__global__ void __launch_bounds__(FAST_BLOCK) k_pair_forces_ordered_win(Geom g, OrderedForcesArgs a, DupPlan dp,
                                                                        double *__restrict__ block_energy,
                                                                        unsigned long long tag, long long per_block, int w0,
                                                                        int wn) {
  extern __shared__ double2 lds_win[];
  if (a.wait_flag && *a.status == 2) return;   // (as k_pair_forces_ordered)
  double *red = reinterpret_cast<double *>(lds_win);   // first 256 B: reduction scratch
  double2 *win = lds_win + 16;
  int *s_samples = reinterpret_cast<int *>(lds_win + 16 + wn);
  __shared__ int s_fd;
  // (per_block is even: a thread takes two neighbouring pairs per trip, 16 bytes of distances in, 16 bytes of forces out)
  const long long beg = (long long)blockIdx.x * per_block;
  const long long end = (beg + per_block < a.n) ? beg + per_block : a.n;
  const long long end2 = beg < end ? beg + ((end - beg) & ~1LL) : beg;   // the run's whole twos end here; an odd last pair goes alone
  const bool has_fs = a.first_sample != nullptr;
  typedef double vd2 __attribute__((ext_vector_type(2)));
  typedef int vi2 __attribute__((ext_vector_type(2)));
  const vd2 *r2 = reinterpret_cast<const vd2 *>(a.r);
  const vi2 *f2 = reinterpret_cast<const vi2 *>(has_fs ? a.first_sample : reinterpret_cast<const int *>(a.r));
  vd2 *o2 = reinterpret_cast<vd2 *>(a.force);
  long long i = beg + 2 * (long long)threadIdx.x;     // first pair of this thread's next two
  const long long last2 = end2 > beg ? end2 - 2 : beg;
  constexpr long long S = 2 * FAST_BLOCK;   // a thread's next two pairs lie S pairs on
  vd2 x_next = {0.0, 0.0}, xb_next = {0.0, 0.0};
  vi2 fs_next = {0, 0}, fb_next = {0, 0};
  if (beg < end2) {
    const long long j = (i < end2 ? i : last2) >> 1, jb = (i + S < end2 ? i + S : last2) >> 1;
    x_next = __builtin_nontemporal_load(&r2[j]);
    fs_next = __builtin_nontemporal_load(&f2[j]);
    xb_next = __builtin_nontemporal_load(&r2[jb]);
    fb_next = __builtin_nontemporal_load(&f2[jb]);
  }
  long long fs0 = 0, fs1 = -1;
  if (beg < end) {
    fs0 = has_fs ? (long long)a.first_sample[beg] : 2 * beg;
    fs1 = has_fs ? (long long)a.first_sample[end - 1] : 2 * (end - 1);
  }
  OrderedCommon oc;
  {
    long long off, nloc, k_split;
    ordered_batch_counts(a, off, nloc, k_split);
    oc.H = (int)nloc;
    int my_first = INT_MAX;
    for (int k = threadIdx.x; k < oc.H; k += FAST_BLOCK) {
      s_samples[k] = a.sel ? (int)a.sel[k] : k;
      if (a.dirty_hill[off + k] == a.dirty_seq && k < my_first) my_first = k;
    }
    if (threadIdx.x == 0) s_fd = INT_MAX;
    __syncthreads();
    if (my_first != INT_MAX) atomicMin(&s_fd, my_first);
    oc.samples = s_samples;
    oc.ntiles = (g.n[0] + ORD_NODES - 1) / ORD_NODES;
    oc.lo_t = oc.lo_s = oc.hi_t = oc.hi_s = -1;
    if (!g.bper[0]) {   // duplicate_boundary_lanes' cases 0 and 3 in one dimension
      if (dp.lo[0] > 0 && dp.lo[0] < (unsigned long long)g.n[0]) {
        oc.lo_t = (int)dp.lo[0] - 1;
        oc.lo_s = (int)dp.lo[0];
      }
      if (dp.hi[0] + 1 < (unsigned long long)g.n[0]) {
        oc.hi_t = (int)dp.hi[0] + 1;
        oc.hi_s = (int)dp.hi[0];
      }
    }
    oc.fast = true;   // (the launcher took this kernel for an interpolating, non-periodic 1-D grid)
    oc.inv_dx = 1.0 / g.dx[0];
    __syncthreads();
    oc.first_dirty = s_fd;
  }
  int row0 = 0, row1 = -1;
  if (beg < end) {
    row0 = ordered_hills_before(oc, fs0);
    row1 = ordered_hills_before(oc, fs1, row0);
  }
  OrderedWin W;
  W.win = win;
  W.w0 = w0;
  W.wn = wn;
  W.lo_ok = fmax(g.bmin[0], g.min[0]);
  W.hi_open = fmin(nextafter(g.bmax[0], 1.0e308), g.max[0] - g.dx[0]);
  W.eps = 1e-11 * fmax(1.0, (double)g.n[0]);
  double e_acc = 0;
  for (int m = row0; m <= row1; m++) {
    // the window behind the first m hills: (value, scaled slope) per node, the boundary duplication applied
    {
      // (a node is count -> record, two dependent loads: all of a thread's counts are requested before its first
      //  record, all records before the first is used -- one node after the other this took ~19 us per window)
      constexpr int PER = 5;   // (nodes per thread and batch: ten at once cost 40 registers the loop below needs)
      const unsigned short *row = a.counts + (long long)m * oc.ntiles;
      const double2 *rec0 = reinterpret_cast<const double2 *>(a.rec0);
      const double2 *recs = reinterpret_cast<const double2 *>(a.records);
      for (int k0 = 0; k0 < wn; k0 += PER * FAST_BLOCK) {
        // (behind the run's first hill count only the tiles the next hill reached change -- a tenth of them: the others'
        //  counts are what they were, and so are their nodes in the window)
        int u[PER];
        bool changed[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
          const int k = k0 + threadIdx.x + q * FAST_BLOCK;
          const int tile = (w0 + (k < wn ? k : 0)) >> 5;
          u[q] = k < wn ? (int)row[tile] : 0;
          changed[q] = k < wn && (m == row0 || u[q] != (int)row[tile - oc.ntiles]);
        }
        double2 r[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
          const int k = k0 + threadIdx.x + q * FAST_BLOCK;
          const int node = w0 + (k < wn ? k : 0);
          const int tile = node >> 5;
          r[q].x = r[q].y = 0.0;
          if (changed[q])
            r[q] = u[q] ? recs[((long long)tile * a.nh_cap + (u[q] - 1)) * ORD_NODES + (node & (ORD_NODES - 1))] : rec0[node];
        }
#pragma unroll
        for (int q = 0; q < PER; q++) {
          const int k = k0 + threadIdx.x + q * FAST_BLOCK;
          if (changed[q]) {
            r[q].y = scaled_slope(r[q].x, r[q].y, g.dx[0]);
            win[k] = r[q];
          }
        }
      }
      __syncthreads();
      // (outward copy nodes of the boundary duplication: their value is their source's from the first hill with a
      //  non-zero correction on -- two nodes, through the general source)
      if (threadIdx.x < 2) {
        const int node = threadIdx.x == 0 ? oc.lo_t : oc.hi_t;
        if (node >= w0 && node < w0 + wn) {
          const OrderedSource src{a, oc, m};
          double2 t = src.get(node);
          t.y = scaled_slope(t.x, t.y, g.dx[0]);
          win[node - w0] = t;
        }
      }
    }
    __syncthreads();
    W.m = m;
    const long long lo_s = m > 0 ? (long long)oc.samples[m - 1] : LLONG_MIN, hi_s = m < oc.H ? (long long)oc.samples[m] : LLONG_MAX;
    // two twos per trip -- pairs (i, i + 1) and (i + S, i + S + 1), four lookups in flight like K1 -- with the next two
    // twos' distances and sample indices requested a trip ahead
    while (i < end2) {
      const bool has_b = i + S < end2;
      const vd2 xa = x_next, xb = xb_next;
      const vi2 ia = fs_next, ib = fb_next;
      const long long fa0 = has_fs ? (long long)ia.x : 2 * i, fa1 = has_fs ? (long long)ia.y : 2 * (i + 1);
      const long long fb0 = has_fs ? (long long)ib.x : 2 * (i + S), fb1 = has_fs ? (long long)ib.y : 2 * (i + S + 1);
      // a pair's hill count is the staged m iff its sample index lies in (sample index of hill m - 1, that of hill m]:
      // two comparisons, no search; beyond the upper end it belongs to a later pass, below the lower one (an array
      // that does not ascend) it takes the general form, which searches the whole list
      if (fa0 > hi_s) break;   // (a later pass's pairs: their window is not staged yet)
      const bool st_a0 = (fa0 > lo_s) & (fa0 <= hi_s), st_a1 = (fa1 > lo_s) & (fa1 <= hi_s);
      const bool st_b0 = (fb0 > lo_s) & (fb0 <= hi_s), st_b1 = (fb1 > lo_s) & (fb1 <= hi_s);
      const bool b_now = has_b && !(fb0 > hi_s);   // (the second two may have to wait for a later pass)
      vd2 nxa, nxb;
      vi2 nfa, nfb;
      {
        const long long ja = ((i + 2 * S < end2) ? i + 2 * S : last2) >> 1, jb = ((i + 3 * S < end2) ? i + 3 * S : last2) >> 1;
        nxa = __builtin_nontemporal_load(&r2[ja]);   // (streamed once, like K1's)
        nfa = __builtin_nontemporal_load(&f2[ja]);
        nxb = __builtin_nontemporal_load(&r2[jb]);
        nfb = __builtin_nontemporal_load(&f2[jb]);
      }
      double va0, da0, va1, da1, vb0 = 0.0, db0 = 0.0, vb1 = 0.0, db1 = 0.0;
      const bool ga0 = ordered_win_pair(g, oc, W, xa.x, st_a0, va0, da0);
      // (the second of a two may already be behind the next hill -- one such two per hill and run: the general form)
      const bool ga1 = ordered_win_pair(g, oc, W, xa.y, st_a1, va1, da1);
      bool gb0 = false, gb1 = false;
      if (b_now) {
        gb0 = ordered_win_pair(g, oc, W, xb.x, st_b0, vb0, db0);
        gb1 = ordered_win_pair(g, oc, W, xb.y, st_b1, vb1, db1);
      }
      if (wave_any(ga0 | ga1 | gb0 | gb1)) {
        // (the general form searches the hills' sample indices itself: the pair's count may be any)
        if (ga0) ordered_lookup(g, a, oc, xa.x, fa0, va0, da0);
        if (ga1) ordered_lookup(g, a, oc, xa.y, fa1, va1, da1);
        if (gb0) ordered_lookup(g, a, oc, xb.x, fb0, vb0, db0);
        if (gb1) ordered_lookup(g, a, oc, xb.y, fb1, vb1, db1);
      }
      e_acc += va0;
      e_acc += va1;
      vd2 out;
      out.x = 0.0 - da0;
      out.y = 0.0 - da1;
      __builtin_nontemporal_store(out, &o2[i >> 1]);
      if (b_now) {
        e_acc += vb0;
        e_acc += vb1;
        out.x = 0.0 - db0;
        out.y = 0.0 - db1;
        __builtin_nontemporal_store(out, &o2[(i + S) >> 1]);
        i += 2 * S;
        x_next = nxa;
        fs_next = nfa;
        xb_next = nxb;
        fb_next = nfb;
      } else {
        i += S;           // (the second two becomes the next trip's first: now, or in the pass it waits for)
        x_next = xb;
        fs_next = ib;
        xb_next = nxa;
        fb_next = nfa;
        if (has_b) break;
      }
    }
    __syncthreads();   // (everybody is through with this window before it is overwritten)
  }
  // (twos no pass took -- none, unless the array does not ascend -- and the run's odd last pair)
  for (; i < end2; i += S)
    for (int q = 0; q < 2; q++) {
      const long long p = i + q;
      const long long fs = has_fs ? (long long)a.first_sample[p] : 2 * p;
      double v, d;
      ordered_lookup_m(g, a, oc, a.r[p], ordered_hills_before(oc, fs), v, d);
      e_acc += v;
      a.force[p] = 0.0 - d;
    }
  if (beg < end && end2 < end && threadIdx.x == 0) {
    const long long p = end - 1;
    const long long fs = has_fs ? (long long)a.first_sample[p] : 2 * p;
    double v, d;
    ordered_lookup_m(g, a, oc, a.r[p], ordered_hills_before(oc, fs), v, d);
    e_acc += v;
    a.force[p] = 0.0 - d;
  }
  const double sum = block_sum(e_acc, red);
  if (threadIdx.x == 0) {
    if (tag) store_partial_tagged(block_energy, blockIdx.x, sum, tag);
    else block_energy[blockIdx.x] = sum;
  }
}

hipError_t launch_pair_forces_ordered(const Geom &g, const OrderedForcesArgs &a, double *scratch, hipStream_t s,
                                      int *blocks_out, unsigned long long tag, hipEvent_t ev0, hipEvent_t ev1) {
  if (!ordered_forces_supported(g) || a.nh_cap > ORD_MAX_HILLS) return hipErrorInvalidValue;
  // long arrays: the LDS-window form (see k_pair_forces_ordered_win), where a workgroup's run of pairs spans few hills
  static const bool win_env = !test_force("no_k1o_window");   // (tests: the short-array kernel on long arrays too)
  // (hills per workgroup's run: a pass and ~125 KB of window each -- 2048 hills over 256 runs are nine passes of a few
  //  microseconds beside a run's 100+ us of lookups)
  // (at W1's 1 M pairs the window form is slower: the step 71 against 59 us -- K1's own break-even is ~1.5 M pairs too)
  if (win_env && a.n >= PAIR_LDS_THRESHOLD && pair_fast_path(g) && a.nh_cap <= ORD_WIN_HILLS) {
    const int blocks = cu_count();
    const long long per_block = (((a.n + blocks - 1) / blocks) + 1) & ~1LL;   // (even: two pairs per thread and trip)
    const int wn = g.n[0] < ORD_WIN_NODES ? g.n[0] : ORD_WIN_NODES;
    const int w0 = g.n[0] - wn;   // top-aligned: pair distances populate the upper range
    const size_t lds = 256 + (size_t)wn * 16 + (size_t)ORD_WIN_HILLS * 4;
    static bool attr_win = false;
    if (!attr_win) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pair_forces_ordered_win),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 256 + ORD_WIN_NODES * 16 + ORD_WIN_HILLS * 4);
      if (e != hipSuccess) return e;
      attr_win = true;
    }
    const DupPlan dpw = make_dup_plan(g);
    EDM_LAUNCH_TIMED(k_pair_forces_ordered_win, dim3((unsigned)blocks), dim3(FAST_BLOCK), lds, s, ev0, ev1, g, a, dpw, scratch, tag,
                     per_block, w0, wn);
    if (blocks_out) *blocks_out = blocks;
    return hipGetLastError();
  }
  // four pairs per thread: 17.8 us per 1 M pairs; two or one (more workgroups, each paying the prologue that stages the
  // hills' sample indices and its rows of the counts) 20.7 us; fewer, fatter workgroups (2 / 1 per CU) 22 / 33 us
  // (after the lean form: 2 / 4 / 8 / 16 pairs per thread = 20.1 / 15.8 / 15.3 / 18.8 us per 1 M pairs)
  long long blocks = (a.n + 4 * BLOCK - 1) / (4 * BLOCK);
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  const long long per_block = (a.n + blocks - 1) / blocks;
  const DupPlan dp = make_dup_plan(g);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pair_forces_ordered),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int) * ORD_MAX_HILLS));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const size_t lds = sizeof(int) * (size_t)(((a.range_dev || a.res_dev) ? a.nh_cap : a.nh) > 0 ? ((a.range_dev || a.res_dev) ? a.nh_cap : a.nh) : 1);
  EDM_LAUNCH_TIMED(k_pair_forces_ordered, dim3((unsigned)blocks), dim3(BLOCK), lds, s, ev0, ev1, g, a, dp, scratch, tag, per_block);
  if (blocks_out) *blocks_out = (int)blocks;
  return hipGetLastError();
}

// ... and over a device-resident neighbour list: list entry e's two virtual add_hill samples are 2 e and 2 e + 1
// (fix_edm_pair.cpp:230-237), so the entry's update_force sees the hills whose sample index is below 2 e
struct OrderedListLookup {
  const Geom &g;
  const OrderedForcesArgs &a;
  const OrderedCommon &oc;
  // the lean form of k_pair_forces_ordered with the counts read from memory (a list's entries come in atom order: their
  // hill counts are all over the table, no rows to stage): one 16-bit count per corner tile -- one when both corners
  // share a tile, 31 times out of 32 --, 32-bit record offsets, the address selected arithmetically
  bool lean;
  const double2 *rec0;
  long long records_minus_rec0;
  int ntiles, nh_cap;
  double lo_ok, hi_open, eps;
  __device__ __forceinline__ double2 record(int u, int tile, int node) const {
    const long long in_records = records_minus_rec0 + (long long)((tile * nh_cap + (u - 1)) * ORD_NODES + (node & (ORD_NODES - 1)));
    return rec0[u ? in_records : (long long)node];
  }
  __device__ __forceinline__ void lookup(double x, int entry, double &v, double &d) const {
    const int m = ordered_hills_before(oc, 2 * (long long)entry);
    if (lean) {
      const bool in_range = (x >= lo_ok) & (x < hi_open);
      const double q = (x - g.min[0]) * oc.inv_dx;
      double fq = floor(q);
      const double frac = q - fq;
      const bool near = in_range & ((frac <= eps) | (frac >= 1.0 - eps));
      if (near) fq = floor((x - g.min[0]) / g.dx[0]);
      int idx = (int)fq;
      idx = idx < 0 ? 0 : idx;
      idx = idx > g.n[0] - 2 ? g.n[0] - 2 : idx;
      const bool special = (m > oc.first_dirty) & ((idx == oc.lo_t) | (idx + 1 == oc.lo_t) | (idx == oc.hi_t) | (idx + 1 == oc.hi_t));
      if (!special) {
        const double where = x - g.min[0] - fq * g.dx[0];
        const double X = where * oc.inv_dx;
        const int t0 = idx >> 5, t1 = (idx + 1) >> 5;
        const unsigned short *row = a.counts + (long long)m * ntiles;
        const int u0 = row[t0];
        const int u1 = (t1 == t0) ? u0 : (int)row[t1];
        const double2 ra = record(u0, t0, idx), rb = record(u1, t1, idx + 1);
        double vv, dd;
        if ((X < 0.0) | (X > 1.0))   // `where` off by an ulp at a node: the reference's fabs() mirroring
          hermite_1d_mirrored(ra.x, ra.y, rb.x, rb.y, X, g.dx[0], oc.inv_dx, vv, dd);
        else
          hermite_1d_horner(ra.x, scaled_slope(ra.x, ra.y, g.dx[0]), rb.x, scaled_slope(rb.x, rb.y, g.dx[0]), X, oc.inv_dx, vv, dd);
        v = in_range ? vv : 0.0;
        d = in_range ? dd : 0.0;
        return;
      }
    }
    ordered_lookup_m(g, a, oc, x, m, v, d);
  }
};
__global__ void __launch_bounds__(BLOCK) k_pairlist_forces_ordered(Geom g, PairListArgs pl, OrderedForcesArgs a, DupPlan dp,
                                                                   double *__restrict__ partials) {
  extern __shared__ int s_samples[];
  if (a.wait_flag && *a.status == 2) return;   // (as k_pair_forces_ordered)
  OrderedCommon oc;
  ordered_common_init(g, a, dp, s_samples, oc);
  OrderedListLookup ord{g, a, oc};
  ord.ntiles = oc.ntiles;
  ord.nh_cap = (int)a.nh_cap;
  ord.lean = oc.fast && (long long)oc.ntiles * a.nh_cap * ORD_NODES < (1ll << 31);   // (32-bit record offsets)
  ord.rec0 = reinterpret_cast<const double2 *>(a.rec0);
  ord.records_minus_rec0 = reinterpret_cast<const double2 *>(a.records) - reinterpret_cast<const double2 *>(a.rec0);
  ord.lo_ok = fmax(g.bmin[0], g.min[0]);
  ord.hi_open = fmin(nextafter(g.bmax[0], 1.0e308), g.max[0] - g.dx[0]);
  ord.eps = 1e-11 * fmax(1.0, (double)g.n[0]);
  pairlist_forces_body<false, OrderedListLookup>(g, a.rec0, pl, partials, 0.0, blockIdx.x, gridDim.x, &ord);
}
hipError_t launch_pairlist_forces_ordered(const Geom &g, const PairListArgs &pl, const OrderedForcesArgs &a, double *partials,
                                          hipStream_t s, int *blocks_out) {
  if (!ordered_forces_supported(g) || !pl.it_entry || !pl.jt_entry || a.nh_cap > ORD_MAX_HILLS) return hipErrorInvalidValue;
  if (blocks_out) *blocks_out = 0;
  if (pl.nall <= 0) return hipSuccess;
  const long long threads = (long long)pl.nall * 16;
  long long nb = (threads + BLOCK - 1) / BLOCK;
  if (nb > MAX_BLOCKS) nb = MAX_BLOCKS;
  const DupPlan dp = make_dup_plan(g);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pairlist_forces_ordered),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int) * ORD_MAX_HILLS));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const size_t lds = sizeof(int) * (size_t)(((a.range_dev || a.res_dev) ? a.nh_cap : a.nh) > 0 ? ((a.range_dev || a.res_dev) ? a.nh_cap : a.nh) : 1);
  hipLaunchKernelGGL(k_pairlist_forces_ordered, dim3((unsigned)nb), dim3(BLOCK), lds, s, g, pl, a, dp, partials);
  if (blocks_out) *blocks_out = (int)nb;
  return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------------
// Device array -> page-locked host memory through the shader (zero-copy stores over PCIe), for results that go down
// BESIDE a running step (edm_hip_bias_step_host): which SDMA engine a stream's copies get is the runtime's choice, and
// the engines differ (measured on one box, same 2 MB device-to-host copy: 42 GB/s on one stream, 11 GB/s on the stream
// of the process's third object -- tools/nd_steps.py ND_PRE=2); a kernel's stores take the same path every time.
// 16 bytes per lane, a wave writes 1 KB contiguous; few workgroups -- the link, not the CUs, is the limit.
__global__ void __launch_bounds__(256) k_copy_to_host(const double *__restrict__ src, double *__restrict__ dst, long long n) {
  const long long pairs = n >> 1;
  typedef double vec2 __attribute__((ext_vector_type(2)));
  const vec2 *__restrict__ s2 = reinterpret_cast<const vec2 *>(src);
  vec2 *__restrict__ d2 = reinterpret_cast<vec2 *>(dst);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride)
    __builtin_nontemporal_store(s2[i], &d2[i]);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
}
hipError_t launch_copy_to_host(const double *d_src, double *h_dst_mapped, long long n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(h_dst_mapped)) & 15) return hipErrorInvalidValue;
  long long nb = ((n >> 1) + 255) / 256;
  if (nb > 8) nb = 8;   // (8 ... 256 workgroups measured alike; the fewest leave the step's kernels alone)
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_copy_to_host, dim3((unsigned)nb), dim3(256), 0, s, d_src, h_dst_mapped, n);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------------
// The completion protocol of limit_and_readback (relaxed system-scope stores into the host-mapped region, s_waitcnt(0),
// barrier, relaxed system-scope flag) on a payload that makes an overtaking flag VISIBLE: every word of the region is
// the launch's sequence number, so a host that sees the flag and then finds an older word has caught the flag ahead of
// the data (edm_hip_debug_flag_order_stress, tests/test_gpu_edge_cases.py).
__global__ void __launch_bounds__(BLOCK) k_flag_order_stress(long long *dst, long long words, unsigned long long seq,
                                                             unsigned long long *flag) {
  for (long long w = threadIdx.x; w < words; w += BLOCK)
    __hip_atomic_store(&dst[w], (long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_waitcnt(0);   // every wave's stores into the host-mapped region have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_flag_order_stress(long long *dst_mapped, long long words, unsigned long long seq, unsigned long long *flag_mapped,
                                    hipStream_t s) {
  hipLaunchKernelGGL(k_flag_order_stress, dim3(1), dim3(BLOCK), 0, s, dst_mapped, words, seq, flag_mapped);
  return hipGetLastError();
}

}  // namespace edm
