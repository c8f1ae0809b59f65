/*
 * ref_shim.cpp -- C API over the REAL reference library, for pinning the oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the builder's own code; it is
 * compiled together with the reference's lib/*.cpp *where they lie* under
 * /root/reference (see oracle/Makefile) into oracle/_ref/libedm_ref.so, which
 * is git-ignored.  No reference source is copied into the repository.
 *
 * The exported functions mirror oracle/edm_oracle.h one to one with the
 * prefix ref_ instead of ora_, so the same ctypes binding drives either.
 */
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <iterator>
#include <map>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "mpi.h"

/* the controller keeps its limiter state private; the shim needs to read it */
#define private public
#include "edm_bias.h"
#undef private
#include "gaussian_grid.h"
#include "grid.h"

using namespace EDM;

namespace {

void ensure_mpi() {
  int flag = 0;
  MPI_Initialized(&flag);
  if (!flag) MPI_Init(NULL, NULL);
}

}  // namespace

#define GRID_FIELD(h, expr)                                                                      \
  (h->dim == 1 ? (dynamic_cast<DimmedGrid<1> *>(h->g)->expr)                                     \
               : h->dim == 2 ? (dynamic_cast<DimmedGrid<2> *>(h->g)->expr)                       \
                             : (dynamic_cast<DimmedGrid<3> *>(h->g)->expr))

#define GAUSS_FIELD(h, expr)                                                                     \
  (h->dim == 1 ? (dynamic_cast<DimmedGaussGrid<1> *>(h->g)->expr)                                \
               : h->dim == 2 ? (dynamic_cast<DimmedGaussGrid<2> *>(h->g)->expr)                  \
                             : (dynamic_cast<DimmedGaussGrid<3> *>(h->g)->expr))

extern "C" {

struct ref_grid {
  int dim;
  Grid *g;
  int owned;
};

struct ref_gauss {
  int dim;
  GaussGrid *g;
  ref_grid inner;
  size_t minisize_copy[3];
};

struct ref_bias {
  EDMBias *b;
  ref_gauss gauss;
  ref_grid hist;
  std::vector<const double *> *rows_in;
  std::vector<double *> *rows_out;
};

/* ---- grid ---- */
ref_grid *ref_grid_create(int dim, const double *min, const double *max, const double *spacing,
                          const int *periodic, int b_derivatives, int b_interpolate) {
  ref_grid *h = new ref_grid;
  h->dim = dim;
  h->g = make_grid(dim, min, max, spacing, periodic, b_derivatives, b_interpolate);
  h->owned = 1;
  return h;
}
ref_grid *ref_grid_read(int dim, const char *filename, int b_interpolate) {
  ref_grid *h = new ref_grid;
  h->dim = dim;
  h->g = read_grid(dim, std::string(filename), b_interpolate);
  h->owned = 1;
  return h;
}
void ref_grid_free(ref_grid *h) {
  if (!h) return;
  if (h->owned) delete h->g;
  delete h;
}
int ref_grid_dim(const ref_grid *h) { return h->dim; }
size_t ref_grid_size(const ref_grid *h) { return h->g->get_grid_size(); }
const int *ref_grid_number(const ref_grid *h) { return GRID_FIELD(h, grid_number_); }
const double *ref_grid_dx(const ref_grid *h) { return h->g->get_dx(); }
const double *ref_grid_min(const ref_grid *h) { return h->g->get_min(); }
const double *ref_grid_max(const ref_grid *h) { return h->g->get_max(); }
const int *ref_grid_periodic(const ref_grid *h) { return GRID_FIELD(h, b_periodic_); }
int ref_grid_has_deriv(const ref_grid *h) { return GRID_FIELD(h, b_derivatives_); }
double *ref_grid_values(ref_grid *h) { return GRID_FIELD(h, grid_); }
double *ref_grid_derivs(ref_grid *h) { return GRID_FIELD(h, grid_deriv_); }
void ref_grid_set_interpolation(ref_grid *h, int b) { h->g->set_interpolation(b); }
void ref_grid_get_index(const ref_grid *h, const double *x, size_t *out) {
  switch (h->dim) {
    case 1: dynamic_cast<DimmedGrid<1> *>(h->g)->get_index(x, out); break;
    case 2: dynamic_cast<DimmedGrid<2> *>(h->g)->get_index(x, out); break;
    default: dynamic_cast<DimmedGrid<3> *>(h->g)->get_index(x, out); break;
  }
}
size_t ref_grid_multi2one(const ref_grid *h, const size_t *idx) {
  switch (h->dim) {
    case 1: return dynamic_cast<DimmedGrid<1> *>(h->g)->multi2one(idx);
    case 2: return dynamic_cast<DimmedGrid<2> *>(h->g)->multi2one(idx);
    default: return dynamic_cast<DimmedGrid<3> *>(h->g)->multi2one(idx);
  }
}
void ref_grid_one2multi(const ref_grid *h, size_t index, size_t *out) { h->g->one2multi(index, out); }
int ref_grid_in_grid(const ref_grid *h, const double *x) {
  switch (h->dim) {
    case 1: return dynamic_cast<DimmedGrid<1> *>(h->g)->in_grid(x);
    case 2: return dynamic_cast<DimmedGrid<2> *>(h->g)->in_grid(x);
    default: return dynamic_cast<DimmedGrid<3> *>(h->g)->in_grid(x);
  }
}
double ref_grid_get_value(const ref_grid *h, const double *x) { return h->g->get_value(x); }
double ref_grid_get_value_deriv(const ref_grid *h, const double *x, double *der) {
  return h->g->get_value_deriv(x, der);
}
double ref_grid_add_value(ref_grid *h, const double *x, double value) {
  if (GRID_FIELD(h, b_interpolate_)) return -1e300; /* the reference would abort */
  return h->g->add_value(x, value);
}
void ref_grid_clear(ref_grid *h) { h->g->clear(); }
double ref_grid_max_value(const ref_grid *h) { return h->g->max_value(); }
double ref_grid_min_value(const ref_grid *h) { return h->g->min_value(); }
double ref_grid_expected_bias(const ref_grid *h) { return h->g->expected_bias(); }
void ref_grid_add_grid(ref_grid *h, const ref_grid *other, double scale, double offset) {
  h->g->add(other->g, scale, offset);
}
void ref_grid_write(const ref_grid *h, const char *filename) { h->g->write(std::string(filename)); }
void ref_grid_multi_write(const ref_grid *h, const char *filename, const double *box_min,
                          const double *box_max, const int *b_periodic, int b_lammps_format) {
  ensure_mpi();
  h->g->multi_write(std::string(filename), box_min, box_max, b_periodic, b_lammps_format);
}

/* ---- gaussian grid ---- */
static void gauss_bind(ref_gauss *h) {
  h->inner.dim = h->dim;
  h->inner.owned = 0;
  switch (h->dim) {
    case 1: h->inner.g = &dynamic_cast<DimmedGaussGrid<1> *>(h->g)->grid_; break;
    case 2: h->inner.g = &dynamic_cast<DimmedGaussGrid<2> *>(h->g)->grid_; break;
    default: h->inner.g = &dynamic_cast<DimmedGaussGrid<3> *>(h->g)->grid_; break;
  }
}
ref_gauss *ref_gauss_create(int dim, const double *min, const double *max, const double *spacing,
                            const int *periodic, int b_interpolate, const double *sigma) {
  ref_gauss *h = new ref_gauss;
  h->dim = dim;
  h->g = make_gauss_grid(dim, min, max, spacing, periodic, b_interpolate, sigma);
  gauss_bind(h);
  return h;
}
ref_gauss *ref_gauss_read(int dim, const char *filename, const double *sigma) {
  ref_gauss *h = new ref_gauss;
  h->dim = dim;
  h->g = read_gauss_grid(dim, std::string(filename), sigma);
  gauss_bind(h);
  return h;
}
void ref_gauss_free(ref_gauss *h) {
  if (!h) return;
  delete h->g;
  delete h;
}
ref_grid *ref_gauss_grid(ref_gauss *h) { return &h->inner; }
void ref_gauss_set_boundary(ref_gauss *h, const double *min, const double *max, const int *periodic) {
  h->g->set_boundary(min, max, periodic);
}
double ref_gauss_add_value(ref_gauss *h, const double *x, double height) { return h->g->add_value(x, height); }
/* n calls of GaussGrid::add_value in one C call (bench.py's CPU baseline: no per-hill FFI overhead) */
double ref_gauss_add_values(ref_gauss *h, long long n, const double *x, int stride, double height) {
  double total = 0;
  for (long long i = 0; i < n; i++) total += h->g->add_value(x + i * stride, height);
  return total;
}
double ref_gauss_get_value(const ref_gauss *h, const double *x) { return h->g->get_value(x); }
double ref_gauss_get_value_deriv(const ref_gauss *h, const double *x, double *der) {
  return h->g->get_value_deriv(x, der);
}
void ref_gauss_remap(const ref_gauss *h, double *x) {
  switch (h->dim) {
    case 1: dynamic_cast<DimmedGaussGrid<1> *>(h->g)->remap(x); break;
    case 2: dynamic_cast<DimmedGaussGrid<2> *>(h->g)->remap(x); break;
    default: dynamic_cast<DimmedGaussGrid<3> *>(h->g)->remap(x); break;
  }
}
int ref_gauss_in_bounds(const ref_gauss *h, const double *x) { return h->g->in_bounds(x); }
double ref_gauss_get_volume(const ref_gauss *h) { return h->g->get_volume(); }
const double *ref_gauss_sigma(const ref_gauss *h) { return GAUSS_FIELD(h, sigma_); }
const size_t *ref_gauss_minisize(const ref_gauss *h) { return GAUSS_FIELD(h, minisize_); }
size_t ref_gauss_minisize_total(const ref_gauss *h) { return GAUSS_FIELD(h, minisize_total_); }
const double *ref_gauss_bc_table(const ref_gauss *h, int d, int deriv) {
  return deriv ? GAUSS_FIELD(h, bc_denom_deriv_table_[d]) : GAUSS_FIELD(h, bc_denom_table_[d]);
}
const double *ref_gauss_boundary_min(const ref_gauss *h) { return GAUSS_FIELD(h, boundary_min_); }
const double *ref_gauss_boundary_max(const ref_gauss *h) { return GAUSS_FIELD(h, boundary_max_); }
const int *ref_gauss_boundary_periodic(const ref_gauss *h) { return GAUSS_FIELD(h, b_periodic_boundary_); }
void ref_gauss_write(const ref_gauss *h, const char *filename) { h->g->write(std::string(filename)); }
void ref_gauss_multi_write(const ref_gauss *h, const char *filename, int b_lammps_format) {
  ensure_mpi();
  if (b_lammps_format)
    h->g->lammps_multi_write(std::string(filename));
  else
    h->g->multi_write(std::string(filename));
}

/* ---- bias controller ---- */
/* the MPI build (oracle/Makefile: ref_mpi, no -DEDM_SERIAL): the controller's constructor already asks for its
 * rank (edm_bias.cpp:63-65), so MPI must be up before it runs; run under mpiexec every process is one rank */
int ref_mpi_rank(void) {
  ensure_mpi();
  int r = 0;
  MPI_Comm_rank(MPI_COMM_WORLD, &r);
  return r;
}
int ref_mpi_size(void) {
  ensure_mpi();
  int n = 1;
  MPI_Comm_size(MPI_COMM_WORLD, &n);
  return n;
}
void ref_mpi_barrier(void) {
  ensure_mpi();
  MPI_Barrier(MPI_COMM_WORLD);
}
void ref_mpi_finalize(void) {
  int flag = 0;
  MPI_Initialized(&flag);
  if (flag) MPI_Finalize();
}
int ref_is_mpi_build(void) {
#ifdef EDM_SERIAL
  return 0;
#else
  return 1;
#endif
}

ref_bias *ref_bias_create(const char *input_filename) {
#ifndef EDM_SERIAL
  ensure_mpi();
#endif
  ref_bias *h = new ref_bias;
  /* zero the storage first: the reference leaves several members
   * (hills_added_, b_skip_hill_add_, total_volume_, overflow_buffer_)
   * uninitialised; the oracle is defined under zero-initialised storage. */
  void *mem = std::calloc(1, sizeof(EDMBias));
  h->b = new (mem) EDMBias(std::string(input_filename));
  h->gauss.g = NULL;
  h->hist.g = NULL;
  h->rows_in = new std::vector<const double *>();
  h->rows_out = new std::vector<double *>();
  return h;
}
void ref_bias_free(ref_bias *h) {
  if (!h) return;
  h->b->~EDMBias();
  std::free(h->b);
  delete h->rows_in;
  delete h->rows_out;
  delete h;
}
void ref_bias_setup(ref_bias *h, double temperature, double boltzmann) { h->b->setup(temperature, boltzmann); }
void ref_bias_subdivide(ref_bias *h, const double *sublo, const double *subhi, const double *boxlo,
                        const double *boxhi, const int *b_periodic, const double *skin) {
  double a[3] = {0, 0, 0}, b[3] = {0, 0, 0}, c[3] = {0, 0, 0}, d[3] = {0, 0, 0}, s[3] = {0, 0, 0};
  int p[3] = {0, 0, 0};
  for (unsigned i = 0; i < h->b->dim_; i++) {
    a[i] = sublo[i]; b[i] = subhi[i]; c[i] = boxlo[i]; d[i] = boxhi[i]; s[i] = skin[i]; p[i] = b_periodic[i];
  }
  h->b->subdivide(a, b, c, d, p, s);
  h->gauss.dim = (int)h->b->dim_;
  h->gauss.g = h->b->bias_;
  gauss_bind(&h->gauss);
  h->hist.dim = (int)h->b->dim_;
  h->hist.g = h->b->cv_hist_;
  h->hist.owned = 0;
}
static void make_rows(ref_bias *h, int n, const double *positions, double *forces, int stride) {
  h->rows_in->resize((size_t)n);
  h->rows_out->resize((size_t)n);
  for (int i = 0; i < n; i++) {
    (*h->rows_in)[(size_t)i] = positions + (size_t)i * stride;
    (*h->rows_out)[(size_t)i] = forces ? forces + (size_t)i * stride : NULL;
  }
}
double ref_bias_update_forces(const ref_bias *hc, int n, const double *positions, double *forces,
                              int stride, int apply_mask) {
  ref_bias *h = const_cast<ref_bias *>(hc);
  make_rows(h, n, positions, forces, stride);
  return h->b->update_forces(n, h->rows_in->data(), h->rows_out->data(), apply_mask);
}
double ref_bias_update_force(const ref_bias *h, const double *position, double *force) {
  return h->b->update_force(position, force);
}
void ref_bias_set_mask(ref_bias *h, const int *mask) { h->b->set_mask(mask); }
void ref_bias_add_hills(ref_bias *h, int n, const double *positions, int stride, const double *runiform,
                        int apply_mask) {
  make_rows(h, n, positions, NULL, stride);
  h->b->add_hills(n, h->rows_in->data(), runiform, apply_mask);
}
void ref_bias_pre_add_hill(ref_bias *h, int est) { h->b->pre_add_hill(est); }
void ref_bias_add_hill(ref_bias *h, const double *position, double runiform) { h->b->add_hill(position, runiform); }
void ref_bias_post_add_hill(ref_bias *h) { h->b->post_add_hill(); }
/* one post_force of the reference's pair fix (lammps/fix_edm_pair.cpp:173-247) driven through the REAL library in
 * the fix's own order: pre_add_hill, then per pair update_force followed by one or two add_hill, then post_add_hill */
double ref_bias_pair_loop(ref_bias *h, int n, const double *r, const int *second, const double *runiform,
                          int hill_step, int est_hill_count, double *force, int *ncalls) {
  int c = 0;
  double energy = 0;
  if (hill_step) h->b->pre_add_hill(est_hill_count);
  for (int k = 0; k < n; k++) {
    double f[1] = {0};
    energy += h->b->update_force(&r[k], f);
    force[k] = f[0];
    if (hill_step) {
      h->b->add_hill(&r[k], runiform[c++]);
      if (second[k]) h->b->add_hill(&r[k], runiform[c++]);
    }
  }
  if (hill_step) h->b->post_add_hill();
  if (ncalls) *ncalls = c;
  return energy;
}
void ref_bias_write_bias(const ref_bias *h, const char *filename) { h->b->write_bias(std::string(filename)); }
void ref_bias_write_lammps_table(const ref_bias *h, const char *filename) {
  h->b->write_lammps_table(std::string(filename));
}
void ref_bias_write_histogram(const ref_bias *h) { h->b->write_histogram(); }
void ref_bias_clear_histogram(ref_bias *h) { h->b->clear_histogram(); }
ref_gauss *ref_bias_gauss(ref_bias *h) { return &h->gauss; }
ref_grid *ref_bias_hist(ref_bias *h) { return &h->hist; }

double ref_bias_get(const ref_bias *h, const char *name) {
  const EDMBias *b = h->b;
#define G(n, expr) if (std::strcmp(name, n) == 0) return (double)(expr)
  G("dim", b->dim_);
  G("b_tempering", b->b_tempering_);
  G("b_targeting", b->b_targeting_);
  G("global_tempering", b->global_tempering_);
  G("bias_factor", b->bias_factor_);
  G("boltzmann_factor", b->boltzmann_factor_);
  G("temperature", b->temperature_);
  G("hill_prefactor", b->hill_prefactor_);
  G("bias_per_step", b->bias_per_step_);
  G("hill_density", b->hill_density_);
  G("cum_bias", b->cum_bias_);
  G("total_volume", b->total_volume_);
  G("expected_target", b->expected_target_);
  G("b_outofbounds", b->b_outofbounds_);
  G("overflow_left", b->overflow_left_i_);
  G("overflow_right", b->overflow_right_i_);
  G("b_skip_hill_add", b->b_skip_hill_add_);
  G("hills_added", b->hills_added_);
  G("steps", b->steps_);
  G("mpi_rank", b->mpi_rank_);
  G("mpi_size", b->mpi_size_);
  G("mpi_neighbor_count", b->mpi_neighbor_count_);
  G("temp_hill_cum", b->temp_hill_cum_);
#undef G
  return NAN;
}
void ref_bias_set(ref_bias *h, const char *name, double value) {
  EDMBias *b = h->b;
#define S(n, lhs, type) if (std::strcmp(name, n) == 0) { lhs = (type)value; return; }
  S("b_tempering", b->b_tempering_, int)
  S("global_tempering", b->global_tempering_, double)
  S("bias_factor", b->bias_factor_, double)
  S("hill_prefactor", b->hill_prefactor_, double)
  S("bias_per_step", b->bias_per_step_, double)
  S("hill_density", b->hill_density_, double)
  S("cum_bias", b->cum_bias_, double)
  S("total_volume", b->total_volume_, double)
#undef S
}
const double *ref_bias_array(const ref_bias *h, const char *name) {
  if (std::strcmp(name, "bias_dx") == 0) return h->b->bias_dx_;
  if (std::strcmp(name, "bias_sigma") == 0) return h->b->bias_sigma_;
  if (std::strcmp(name, "min") == 0) return h->b->min_;
  if (std::strcmp(name, "max") == 0) return h->b->max_;
  return NULL;
}

} /* extern "C" */
