// edm_common.h -- geometry record and the scalar index/remap arithmetic shared by
// host code and gfx950 kernels.  All of it is IEEE double evaluated in the
// reference's operation order (build with -ffp-contract=off on both sides) so
// that grid indices are bit-exact.  Citations: file:line in the reference tree.
#pragma once

#include <math.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EDM_HD __host__ __device__ __forceinline__
#else
#define EDM_HD inline
#endif

#define EDM_GAUSS_SUPPORT 8.0    // gaussian_grid.h:10
#define EDM_BC_TABLE_SIZE 65536  // gaussian_grid.h:11
#define EDM_BC_MAR 2.0           // gaussian_grid.h:12
#define EDM_GRID_TYPE 32         // grid.h:14

namespace edm {

// Everything a kernel needs to know about one grid, passed by value as a kernel
// argument.  Node records are AoS in HBM: rec = (V, dV/ds_0, .., pad) with
// `rec` doubles per node (2 in 1-D, 4 in 2-D and 3-D) so one corner of the
// interpolation stencil is one 16- or 32-byte access.
struct Geom {
  int dim;
  int interp;
  int rec;            // doubles per node record
  int has_deriv;      // 0 for plain histogram / target grids (rec == 1)
  int n[3];
  int periodic[3];    // grid periodicity
  int bper[3];        // boundary periodicity
  int msize[3];       // stencil half-width in nodes
  long long total;
  double min[3], max[3], dx[3];
  double sigma[3];    // already multiplied by sqrt(2)
  double bmin[3], bmax[3];
};

// grid.h:17-20 -- the cast binds tighter than the comparison; equals floor().
EDM_HD int ifloor(double a) {
  double r = ((int)a < 0.0) ? -ceil(fabs(a)) : floor(a);
  return (int)r;
}

// grid.h:22-26
EDM_HD double round_half(double a) { return a < 0.0 ? ceil(a - 0.5) : floor(a + 0.5); }

// gaussian_grid.h:16-32
EDM_HD double smooth_step(double t) {
  if (t < 0) return 1;
  if (t > 1) return 0;
  return 2 * t * t * t - 3 * t * t + 1;
}
EDM_HD double smooth_step_dt(double t) {
  if (t < 0) return 0;
  if (t > 1) return 0;
  return 6 * t * t - 6 * t;
}

// gaussian_grid.h:490-499 (closed interval)
template <int DIM>
EDM_HD bool in_bounds(const Geom &g, const double *x) {
#pragma unroll
  for (int d = 0; d < DIM; d++)
    if (x[d] < g.bmin[d] || x[d] > g.bmax[d]) return false;
  return true;
}

// gaussian_grid.h:504-541 -- nearest image to the grid, not minimal image
template <int DIM>
EDM_HD void remap(const Geom &g, double *x) {
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    if (x[d] < g.min[d] || x[d] > g.max[d]) {
      if (g.periodic[d]) {
        x[d] -= (g.max[d] - g.min[d]) * ifloor((x[d] - g.min[d]) / (g.max[d] - g.min[d]));
      } else if (g.bper[d]) {
        double period = g.bmax[d] - g.bmin[d];
        double s0 = round_half((g.min[d] - x[d]) / (g.bmax[d] - g.bmin[d])) * period;
        double s1 = round_half((g.max[d] - x[d]) / (g.bmax[d] - g.bmin[d])) * period;
        if (fabs(g.min[d] - x[d] - s0) < fabs(g.max[d] - x[d] - s1))
          x[d] += s0;
        else
          x[d] += s1;
      }
    }
  }
}

// grid.h:865-874
template <int DIM>
EDM_HD bool in_grid(const Geom &g, const double *x) {
#pragma unroll
  for (int d = 0; d < DIM; d++)
    if (!g.periodic[d] && (x[d] < g.min[d] || x[d] >= g.max[d] - g.dx[d])) return false;
  return true;
}

// grid.h:264-273 for one dimension; also returns the wrapped coordinate used by
// get_value_deriv (grid.h:426-430).
EDM_HD long long node_index(const Geom &g, int d, double x, double *wrapped) {
  double xi = x;
  if (g.periodic[d]) xi -= (g.max[d] - g.min[d]) * ifloor((xi - g.min[d]) / (g.max[d] - g.min[d]));
  *wrapped = xi;
  return (long long)floor((xi - g.min[d]) / g.dx[d]);
}

}  // namespace edm
