// edm_host.cpp -- libedm.so: the reference's C++ API (EDM::EDMBias, EDM::GaussGrid, EDM::Grid,
// EDM::edm_error; lib/edm_bias.h, lib/gaussian_grid.h, lib/grid.h, lib/edm.h) implemented as thin
// wrappers over the C ABI of libedm_hip.so (include/edm_hip.h).  Plain host C++ (g++), no HIP
// headers: the GPU is reached only through the C ABI.  Error convention of the reference:
// a failing call prints "[EDM:<where>] <message>" and abort()s (lib/edm.cpp:4-7).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/edm/edm_bias.h"
#include "../../include/edm_hip.h"

namespace EDM {

void* pinned_alloc(size_t bytes) {
  void* p = NULL;
  if (edm_hip_host_malloc(&p, bytes) != EDM_HIP_OK) edm_error(edm_hip_last_error(), "edm_bias.h:pinned_alloc");
  return p;
}
void pinned_free(void* p) { edm_hip_host_free(p); }

void edm_error(const char* error, const char* location) {
  std::cerr << "[EDM:" << location << "] " << error << std::endl;
  abort();
}

namespace {

void check(int rc, const char* where) {
  if (rc != EDM_HIP_OK) edm_error(edm_hip_last_error(), where);
}

// a growing device allocation owned by the host layer
struct DevMem {
  void* p;
  size_t cap;
  DevMem() : p(NULL), cap(0) {}
  ~DevMem() {
    if (p) edm_hip_free(p);
  }
  void* reserve(size_t bytes) {
    if (bytes > cap) {
      if (p) edm_hip_free(p);
      p = NULL;
      size_t want = bytes + bytes / 4 + 256;
      check(edm_hip_malloc(&p, want), "edm_host:reserve");
      cap = want;
    }
    return p;
  }
};

// LAMMPS-style double** (rows of one contiguous block): returns the row stride in doubles, or 0
// when the rows are not evenly spaced and have to be packed
long row_stride(int n, const double* const* rows, unsigned int dim) {
  if (n <= 1) return (long)dim;
  const long stride = (long)(rows[1] - rows[0]);
  if (stride < (long)dim) return 0;
  if (rows[n - 1] != rows[0] + (long)(n - 1) * stride) return 0;
  return stride;
}

}  // namespace

// ================================================================================
// HipGrid
// ================================================================================
void HipGrid::refresh_geometry() {
  edm_hip_geometry g;
  check(edm_hip_grid_geometry(h_, &g), "grid.h:geometry");
  dim_ = (unsigned int)g.dim;
  grid_size_ = (size_t)g.total;
  b_derivatives_ = g.derivatives;
  b_interpolate_ = g.interpolate;
  for (int d = 0; d < 3; d++) {
    dx_[d] = g.dx[d];
    min_[d] = g.min[d];
    max_[d] = g.max[d];
    grid_number_[d] = g.n[d];
    b_periodic_[d] = g.periodic[d];
  }
}

HipGrid::HipGrid(unsigned int dim, const double* min, const double* max, const double* bin_spacing,
                 const int* b_periodic, int b_derivatives, int b_interpolate)
    : h_(NULL), owned_(true) {
  check(edm_hip_grid_create_ex(&h_, (int)dim, min, max, bin_spacing, b_periodic, b_derivatives, b_interpolate),
        "grid.h:DimmedGrid");
  refresh_geometry();
}

HipGrid::HipGrid(unsigned int dim, const std::string& input_grid, int b_interpolate) : h_(NULL), owned_(true) {
  check(edm_hip_grid_read(&h_, (int)dim, input_grid.c_str(), b_interpolate), "grid.h:read");
  refresh_geometry();
}

HipGrid::HipGrid(edm_hip_grid* borrowed) : h_(borrowed), owned_(false) { refresh_geometry(); }

HipGrid::~HipGrid() {
  if (owned_ && h_) edm_hip_grid_destroy(h_);
}

double* HipGrid::get_grid() {
  snapshot_.resize(grid_size_ ? grid_size_ : 1);
  check(edm_hip_grid_download(h_, snapshot_.data()), "grid.h:get_grid");
  return snapshot_.data();
}
const double* HipGrid::get_grid_deriv() {
  if (!b_derivatives_) return NULL;   // grid_deriv_ stays NULL without derivatives (grid.h:898)
  snapshot_deriv_.resize((grid_size_ ? grid_size_ : 1) * dim_);
  check(edm_hip_grid_download_derivs(h_, snapshot_deriv_.data()), "grid.h:get_grid_deriv");
  return snapshot_deriv_.data();
}
void HipGrid::set_grid(const double* values, const double* derivs) {
  if (derivs && b_derivatives_)
    check(edm_hip_grid_upload_derivs(h_, values, derivs), "grid.h:set_grid");
  else
    check(edm_hip_grid_upload(h_, values), "grid.h:set_grid");
}

// grid.h:865-874, :264-273, :315-325
int HipGrid::in_grid(const double* x) const {
  for (unsigned int d = 0; d < dim_; d++)
    if (!b_periodic_[d] && (x[d] < min_[d] || x[d] >= max_[d] - dx_[d])) return 0;
  return 1;
}
void HipGrid::get_index(const double* x, size_t* result) const {
  for (unsigned int d = 0; d < dim_; d++) {
    double xi = x[d];
    if (b_periodic_[d]) xi -= (max_[d] - min_[d]) * std::floor((xi - min_[d]) / (max_[d] - min_[d]));
    result[d] = (size_t)std::floor((xi - min_[d]) / dx_[d]);
  }
}
size_t HipGrid::multi2one(const size_t* index) const {
  size_t result = index[dim_ - 1];
  for (int d = (int)dim_ - 2; d >= 0; d--) result = result * (size_t)grid_number_[d] + index[d];
  return result;
}

// DimmedGrid::get_value / get_value_deriv (grid.h:343-365, :390-446) as batches on the device
void HipGrid::get_value_deriv_batch(size_t n, const double* x, int stride, double* value, double* deriv) const {
  if (n == 0) return;
  DevMem dx, dE, dD;
  dx.reserve(sizeof(double) * n * stride);
  dE.reserve(sizeof(double) * n);
  dD.reserve(sizeof(double) * n * dim_);
  check(edm_hip_memcpy_h2d(dx.p, x, sizeof(double) * n * stride), "grid.h:get_value_deriv");
  check(edm_hip_grid_get_value_deriv(h_, (long long)n, (const double*)dx.p, stride, (double*)dE.p, (double*)dD.p),
        "grid.h:get_value_deriv");
  if (value) check(edm_hip_memcpy_d2h(value, dE.p, sizeof(double) * n), "grid.h:get_value_deriv");
  if (deriv) check(edm_hip_memcpy_d2h(deriv, dD.p, sizeof(double) * n * dim_), "grid.h:get_value_deriv");
}
double HipGrid::get_value(const double* x) const {
  double v = 0;
  get_value_deriv_batch(1, x, (int)dim_, &v, NULL);
  return v;
}
double HipGrid::get_value_deriv(const double* x, double* der) const {
  double v = 0;
  get_value_deriv_batch(1, x, (int)dim_, &v, der);
  return v;
}

double HipGrid::add_value(const double* x0, double value) {
  if (b_interpolate_) edm_error("Cannot add_value when using derivatives", "grid.h:add_value");  // grid.h:371-373
  if (!in_grid(x0)) return 0;
  DevMem dx;
  dx.reserve(sizeof(double) * dim_);
  check(edm_hip_memcpy_h2d(dx.p, x0, sizeof(double) * dim_), "grid.h:add_value");
  check(edm_hip_grid_add_values(h_, 1, (const double*)dx.p, (int)dim_, NULL, value), "grid.h:add_value");
  return value;
}

void HipGrid::write(const std::string& filename) const { check(edm_hip_grid_write(h_, filename.c_str()), "grid.h:write"); }
void HipGrid::multi_write(const std::string& filename, const double* box_low, const double* box_high,
                          const int* b_periodic, int b_lammps_format) const {
  check(edm_hip_grid_multi_write(h_, filename.c_str(), box_low, box_high, b_periodic, b_lammps_format), "grid.h:multi_write");
}
void HipGrid::read(const std::string& filename) {
  check(edm_hip_grid_reread(h_, filename.c_str()), "grid.h:read");
  refresh_geometry();
}
void HipGrid::set_interpolation(int b_interpolate) {
  check(edm_hip_grid_set_interpolation(h_, b_interpolate), "grid.h:set_interpolation");
  b_interpolate_ = b_interpolate ? 1 : 0;
}
const double* HipGrid::get_dx() const { return dx_; }
const double* HipGrid::get_max() const { return max_; }
const double* HipGrid::get_min() const { return min_; }
double HipGrid::max_value() const {
  const double* v = const_cast<HipGrid*>(this)->get_grid();
  double m = v[0];
  for (size_t i = 0; i < grid_size_; i++) m = std::fmax(m, v[i]);
  return m;
}
double HipGrid::min_value() const {
  const double* v = const_cast<HipGrid*>(this)->get_grid();
  double m = v[0];
  for (size_t i = 0; i < grid_size_; i++) m = std::fmin(m, v[i]);
  return m;
}
// Grid::add (grid.h:275-290): `other` is evaluated on the device through its own get_value_deriv
void HipGrid::add(const Grid* other, double scale, double offset) {
  if (const HipGrid* g = dynamic_cast<const HipGrid*>(other))
    check(edm_hip_grid_add_grid(h_, g->handle(), scale, offset), "grid.h:add");
  else if (const HipGaussGrid* gg = dynamic_cast<const HipGaussGrid*>(other))
    check(edm_hip_grid_add_gauss(h_, gg->handle(), scale, offset), "grid.h:add");
  else
    edm_error("Grid::add needs a device-resident grid (make_grid / read_grid / make_gauss_grid)", "grid.h:add");
}
size_t HipGrid::get_grid_size() const { return grid_size_; }
void HipGrid::one2multi(size_t index, size_t* result) const {
  unsigned int d;
  for (d = 0; d < dim_ - 1; d++) {
    result[d] = index % (size_t)grid_number_[d];
    index = (index - result[d]) / (size_t)grid_number_[d];
  }
  result[d] = index;
}
double HipGrid::expected_bias() const {
  const double* v = const_cast<HipGrid*>(this)->get_grid();
  double Z = 0, offset = 0, avg = 0;
  for (size_t i = 0; i < grid_size_; i++) offset = std::fmax(offset, v[i]);
  for (size_t i = 0; i < grid_size_; i++) Z += std::exp(-v[i] - offset);
  for (size_t i = 0; i < grid_size_; i++) avg += v[i] * std::exp(-v[i] - offset);
  return avg / Z;
}
void HipGrid::clear() { check(edm_hip_grid_clear(h_), "grid.h:clear"); }

// grid.cpp:3-45
Grid* make_grid(unsigned int dim, const double* min, const double* max, const double* bin_spacing,
                const int* b_periodic, int b_derivatives, int b_interpolate) {
  if (dim < 1 || dim > 3) return NULL;
  return new HipGrid(dim, min, max, bin_spacing, b_periodic, b_derivatives, b_interpolate);
}
Grid* read_grid(unsigned int dim, const std::string& filename, int b_interpolate) {
  if (dim < 1 || dim > 3) return NULL;
  return new HipGrid(dim, filename, b_interpolate);
}
Grid* read_grid(unsigned int dim, const std::string& filename) {
  if (dim < 1 || dim > 3) return NULL;
  return new HipGrid(dim, filename, 1);
}

// ================================================================================
// HipGaussGrid
// ================================================================================
void HipGaussGrid::refresh_geometry() {
  edm_hip_geometry g;
  check(edm_hip_gauss_geometry(h_, &g), "gaussian_grid.h:geometry");
  dim_ = (unsigned int)g.dim;
  grid_size_ = (size_t)g.total;
  for (int d = 0; d < 3; d++) {
    dx_[d] = g.dx[d];
    min_[d] = g.min[d];
    max_[d] = g.max[d];
    grid_number_[d] = g.n[d];
    b_periodic_[d] = g.periodic[d];
    sigma_[d] = g.sigma[d];
    boundary_min_[d] = g.boundary_min[d];
    boundary_max_[d] = g.boundary_max[d];
    b_periodic_boundary_[d] = g.boundary_periodic[d];
    minisize_[d] = (size_t)g.minisize[d];
  }
}

HipGaussGrid::HipGaussGrid(unsigned int dim, const double* min, const double* max, const double* bin_spacing,
                           const int* b_periodic, int b_interpolate, const double* sigma)
    : h_(NULL), owned_(true) {
  check(edm_hip_gauss_create(&h_, (int)dim, min, max, bin_spacing, b_periodic, b_interpolate, sigma),
        "gaussian_grid.h:DimmedGaussGrid");
  refresh_geometry();
}
HipGaussGrid::HipGaussGrid(unsigned int dim, const std::string& filename, const double* sigma) : h_(NULL), owned_(true) {
  check(edm_hip_gauss_read(&h_, (int)dim, filename.c_str(), sigma), "gaussian_grid.h:DimmedGaussGrid");
  refresh_geometry();
}
HipGaussGrid::HipGaussGrid(edm_hip_gauss* borrowed) : h_(borrowed), owned_(false) { refresh_geometry(); }
HipGaussGrid::~HipGaussGrid() {
  if (owned_ && h_) edm_hip_gauss_destroy(h_);
}

double HipGaussGrid::get_value_deriv_batch(size_t n, const double* x, int stride, double* energy, double* deriv) const {
  if (n == 0) return 0;
  DevMem dx, dE, dD;
  dx.reserve(sizeof(double) * n * stride);
  dE.reserve(sizeof(double) * n);
  dD.reserve(sizeof(double) * n * dim_);
  check(edm_hip_memcpy_h2d(dx.p, x, sizeof(double) * n * stride), "gaussian_grid.h:get_value_deriv");
  check(edm_hip_gauss_get_value_deriv(h_, (long long)n, (const double*)dx.p, stride, (double*)dE.p, (double*)dD.p),
        "gaussian_grid.h:get_value_deriv");
  std::vector<double> e(n);
  check(edm_hip_memcpy_d2h(e.data(), dE.p, sizeof(double) * n), "gaussian_grid.h:get_value_deriv");
  if (deriv) check(edm_hip_memcpy_d2h(deriv, dD.p, sizeof(double) * n * dim_), "gaussian_grid.h:get_value_deriv");
  double sum = 0;
  for (size_t i = 0; i < n; i++) {
    if (energy) energy[i] = e[i];
    sum += e[i];
  }
  return sum;
}

double HipGaussGrid::get_value_deriv(const double* x, double* der) const {
  double e = 0;
  get_value_deriv_batch(1, x, (int)dim_, &e, der);
  return e;
}
double HipGaussGrid::get_value(const double* x) const {
  double der[3];
  return get_value_deriv(x, der);
}

void HipGaussGrid::add_values(size_t n, const double* x, int stride, const double* heights, double* added) {
  if (n == 0) return;
  DevMem dx, dh, da;
  dx.reserve(sizeof(double) * n * stride);
  dh.reserve(sizeof(double) * n);
  da.reserve(sizeof(double) * n);
  check(edm_hip_memcpy_h2d(dx.p, x, sizeof(double) * n * stride), "gaussian_grid.h:add_value");
  check(edm_hip_memcpy_h2d(dh.p, heights, sizeof(double) * n), "gaussian_grid.h:add_value");
  check(edm_hip_gauss_add_values(h_, (long long)n, (const double*)dx.p, stride, (const double*)dh.p, 0.0,
                                 added ? (double*)da.p : NULL, NULL),
        "gaussian_grid.h:add_value");
  if (added) check(edm_hip_memcpy_d2h(added, da.p, sizeof(double) * n), "gaussian_grid.h:add_value");
}

double HipGaussGrid::add_value(const double* x0, double height) {
  double added = 0;
  add_values(1, x0, (int)dim_, &height, &added);
  return added;
}

void HipGaussGrid::read(const std::string& filename) {
  check(edm_hip_gauss_reread(h_, filename.c_str()), "gaussian_grid.h:read");
  refresh_geometry();
}
void HipGaussGrid::write(const std::string& filename) const { check(edm_hip_gauss_write(h_, filename.c_str()), "gaussian_grid.h:write"); }
void HipGaussGrid::multi_write(const std::string& filename) const {
  check(edm_hip_gauss_multi_write(h_, filename.c_str(), 0), "gaussian_grid.h:multi_write");
}
void HipGaussGrid::lammps_multi_write(const std::string& filename) const {
  check(edm_hip_gauss_multi_write(h_, filename.c_str(), 1), "gaussian_grid.h:lammps_multi_write");
}
// gaussian_grid.h:160-166: the caller's box, always the PLUMED layout (the format flag is ignored there)
void HipGaussGrid::multi_write(const std::string& filename, const double* box_low, const double* box_high,
                               const int* b_periodic, int) const {
  check(edm_hip_gauss_multi_write_box(h_, filename.c_str(), box_low, box_high, b_periodic, 0), "gaussian_grid.h:multi_write");
}
void HipGaussGrid::set_interpolation(int b_interpolate) {
  check(edm_hip_gauss_set_interpolation(h_, b_interpolate), "gaussian_grid.h:set_interpolation");
}
void HipGaussGrid::set_boundary(const double* min, const double* max, const int* b_periodic) {
  check(edm_hip_gauss_set_boundary(h_, min, max, b_periodic), "gaussian_grid.h:set_boundary");
  refresh_geometry();
}
double HipGaussGrid::get_volume() const {
  double vol = 1;
  for (unsigned int d = 0; d < dim_; d++) vol *= boundary_max_[d] - boundary_min_[d];
  return vol;
}
void HipGaussGrid::one2multi(size_t index, size_t* result) const {
  unsigned int d;
  for (d = 0; d < dim_ - 1; d++) {
    result[d] = index % (size_t)grid_number_[d];
    index = (index - result[d]) / (size_t)grid_number_[d];
  }
  result[d] = index;
}
double* HipGaussGrid::get_grid() {
  snapshot_.resize(grid_size_ ? grid_size_ : 1);
  check(edm_hip_gauss_download(h_, snapshot_.data(), NULL), "gaussian_grid.h:get_grid");
  return snapshot_.data();
}
const double* HipGaussGrid::get_grid_deriv() {
  snapshot_deriv_.resize((grid_size_ ? grid_size_ : 1) * dim_);
  check(edm_hip_gauss_download(h_, NULL, snapshot_deriv_.data()), "gaussian_grid.h:get_grid_deriv");
  return snapshot_deriv_.data();
}
void HipGaussGrid::set_grid(const double* values, const double* derivs) {
  check(edm_hip_gauss_upload(h_, values, derivs), "gaussian_grid.h:set_grid");
}
const double* HipGaussGrid::get_dx() const { return dx_; }
const double* HipGaussGrid::get_min() const { return min_; }
const double* HipGaussGrid::get_max() const { return max_; }
double HipGaussGrid::max_value() const {
  const double* v = const_cast<HipGaussGrid*>(this)->get_grid();
  double m = v[0];
  for (size_t i = 0; i < grid_size_; i++) m = std::fmax(m, v[i]);
  return m;
}
double HipGaussGrid::min_value() const {
  const double* v = const_cast<HipGaussGrid*>(this)->get_grid();
  double m = v[0];
  for (size_t i = 0; i < grid_size_; i++) m = std::fmin(m, v[i]);
  return m;
}
// Grid::add through the underlying grid (gaussian_grid.h:474-476 -> grid.h:275-290)
void HipGaussGrid::add(const Grid* other, double scale, double offset) {
  if (const HipGrid* g = dynamic_cast<const HipGrid*>(other))
    check(edm_hip_gauss_add_grid(h_, g->handle(), scale, offset), "gaussian_grid.h:add");
  else if (const HipGaussGrid* gg = dynamic_cast<const HipGaussGrid*>(other))
    check(edm_hip_gauss_add_gauss(h_, gg->handle(), scale, offset), "gaussian_grid.h:add");
  else
    edm_error("Grid::add needs a device-resident grid (make_grid / read_grid / make_gauss_grid)", "gaussian_grid.h:add");
}
double HipGaussGrid::expected_bias() const {
  const double* v = const_cast<HipGaussGrid*>(this)->get_grid();
  double Z = 0, offset = 0, avg = 0;
  for (size_t i = 0; i < grid_size_; i++) offset = std::fmax(offset, v[i]);
  for (size_t i = 0; i < grid_size_; i++) Z += std::exp(-v[i] - offset);
  for (size_t i = 0; i < grid_size_; i++) avg += v[i] * std::exp(-v[i] - offset);
  return avg / Z;
}
void HipGaussGrid::clear() { check(edm_hip_gauss_clear(h_), "gaussian_grid.h:clear"); }
size_t HipGaussGrid::get_grid_size() const { return grid_size_; }
int HipGaussGrid::in_bounds(const double* x) const {
  for (unsigned int d = 0; d < dim_; d++)
    if (x[d] < boundary_min_[d] || x[d] > boundary_max_[d]) return 0;
  return 1;
}

void HipGaussGrid::remap(double* x) const {
  DevMem dx, dout;
  dx.reserve(sizeof(double) * dim_);
  dout.reserve(sizeof(double) * dim_);
  check(edm_hip_memcpy_h2d(dx.p, x, sizeof(double) * dim_), "gaussian_grid.h:remap");
  check(edm_hip_gauss_remap(h_, 1, (const double*)dx.p, (int)dim_, (double*)dout.p), "gaussian_grid.h:remap");
  check(edm_hip_memcpy_d2h(x, dout.p, sizeof(double) * dim_), "gaussian_grid.h:remap");
}

GaussGrid* make_gauss_grid(unsigned int dim, const double* min, const double* max, const double* bin_spacing,
                           const int* b_periodic, int b_interpolate, const double* sigma) {
  if (dim < 1 || dim > 3) return NULL;
  return new HipGaussGrid(dim, min, max, bin_spacing, b_periodic, b_interpolate, sigma);
}
GaussGrid* read_gauss_grid(unsigned int dim, const std::string& filename, const double* sigma) {
  if (dim < 1 || dim > 3) return NULL;
  return new HipGaussGrid(dim, filename, sigma);
}

// ================================================================================
// EDMBias
// ================================================================================
struct EDMBias::Stage {
  DevMem x, f, mask, u, r, fr;
  std::vector<double> pack_x, pack_f;
  // device-resident neighbour list of pair_list_step
  DevMem pl_x, pl_f;
  std::vector<int> pl_hi, pl_hj;
  std::vector<double> pl_hx, pl_hf;
  long long pl_npairs;
  Stage() : pl_npairs(-1) {}
};

EDMBias::EDMBias(const std::string& input_filename)
    : b_tempering_(0), b_targeting_(0), mpi_rank_(0), mpi_size_(0), dim_(0), global_tempering_(0), bias_factor_(0),
      boltzmann_factor_(0), temperature_(-1.0), hill_prefactor_(0), bias_per_step_(0), hill_density_(-1), cum_bias_(0),
      total_volume_(0), expected_target_(0), b_outofbounds_(0), bias_dx_(NULL), bias_sigma_(NULL), min_(NULL), max_(NULL),
      b_periodic_boundary_(NULL), target_(NULL), initial_bias_(NULL), bias_(NULL), mask_(NULL), mpi_neighbor_count_(0),
      mpi_neighbors_(NULL), h_(NULL), cv_hist_(NULL), serial_format_(0), st_(new Stage) {
  read_input(input_filename);
}

EDMBias::~EDMBias() {
  delete bias_;
  delete cv_hist_;
  if (h_) edm_hip_bias_destroy(h_);
  free(bias_dx_);
  free(bias_sigma_);
  free(min_);
  free(max_);
  free(b_periodic_boundary_);
  delete st_;
}

// edm_bias.cpp:986-1095.  Like the reference the constructor ignores the result; a handle that
// failed to parse reports its error on first use.
int EDMBias::read_input(const std::string& input_filename) {
  if (h_) {
    edm_hip_bias_destroy(h_);
    h_ = NULL;
  }
  const int rc = edm_hip_bias_create(&h_, input_filename.c_str());
  if (rc != EDM_HIP_OK) {
    std::cerr << edm_hip_last_error() << std::endl;
    return 0;
  }
  double d = 0;
  edm_hip_bias_get(h_, "dim", &d);
  dim_ = (unsigned int)d;
  free(bias_dx_); free(bias_sigma_); free(min_); free(max_); free(b_periodic_boundary_);
  bias_dx_ = (double*)malloc(sizeof(double) * dim_);
  bias_sigma_ = (double*)malloc(sizeof(double) * dim_);
  min_ = (double*)malloc(sizeof(double) * dim_);
  max_ = (double*)malloc(sizeof(double) * dim_);
  b_periodic_boundary_ = (int*)calloc(dim_, sizeof(int));
  edm_hip_bias_get_array(h_, "bias_dx", bias_dx_);
  edm_hip_bias_get_array(h_, "bias_sigma", bias_sigma_);
  edm_hip_bias_get_array(h_, "min", min_);
  edm_hip_bias_get_array(h_, "max", max_);
  refresh();
  return 1;
}

void EDMBias::refresh() const {
  EDMBias* s = const_cast<EDMBias*>(this);
  double v = 0;
#define PULL(name, member, type) if (edm_hip_bias_get(h_, name, &v) == EDM_HIP_OK) s->member = (type)v
  PULL("b_tempering", b_tempering_, int);
  PULL("b_targeting", b_targeting_, int);
  PULL("mpi_rank", mpi_rank_, int);
  PULL("mpi_size", mpi_size_, int);
  PULL("global_tempering", global_tempering_, double);
  PULL("bias_factor", bias_factor_, double);
  PULL("boltzmann_factor", boltzmann_factor_, double);
  PULL("temperature", temperature_, double);
  PULL("hill_prefactor", hill_prefactor_, double);
  PULL("bias_per_step", bias_per_step_, double);
  PULL("hill_density", hill_density_, double);
  PULL("cum_bias", cum_bias_, double);
  PULL("total_volume", total_volume_, double);
  PULL("expected_target", expected_target_, double);
  PULL("b_outofbounds", b_outofbounds_, int);
#undef PULL
}

void EDMBias::setup(double temperature, double boltzmann_constant) {
  check(edm_hip_bias_setup(h_, temperature, boltzmann_constant), "edm_bias.cpp:setup");
  refresh();
}

void EDMBias::subdivide(const double sublo[3], const double subhi[3], const double boxlo[3], const double boxhi[3],
                        const int b_periodic[3], const double skin[3]) {
  if (bias_ != NULL) return;  // edm_bias.cpp:121-122
  if (temperature_ < 0) edm_error("Must call setup before subdivide", "edm_bias.cpp:subdivide");
  check(edm_hip_bias_subdivide(h_, sublo, subhi, boxlo, boxhi, b_periodic, skin), "edm_bias.cpp:subdivide");
  bias_ = new HipGaussGrid(edm_hip_bias_gauss(h_));
  cv_hist_ = new HipGrid(edm_hip_bias_histogram(h_));
  HipGaussGrid* g = static_cast<HipGaussGrid*>(bias_);
  for (unsigned int d = 0; d < dim_; d++) b_periodic_boundary_[d] = g->b_periodic_boundary_[d];
  refresh();
}

void EDMBias::set_mask(const int* mask) { mask_ = mask; }

double EDMBias::update_forces(int nlocal, const double* const* positions, double** forces) const {
  return update_forces(nlocal, positions, forces, -1);
}

// edm_bias.cpp:276-295 with LAMMPS' host arrays staged through HBM
double EDMBias::update_forces(int nlocal, const double* const* positions, double** forces, int apply_mask) const {
  if (b_outofbounds_ || nlocal <= 0) return 0.0;
  return host_step(nlocal, positions, forces, NULL, apply_mask, 0);
}

double EDMBias::step(int nlocal, const double* const* positions, double** forces, const double* runiform, int apply_mask) {
  if (nlocal <= 0) {  // nothing to evaluate; the hill cycle still runs (collective under a communicator)
    add_hills(nlocal, positions, runiform, apply_mask);
    return 0.0;
  }
  const double energy = host_step(nlocal, positions, forces, runiform, apply_mask, 1);
  refresh();
  return energy;
}

// update_forces / step on LAMMPS' own atom arrays: positions up (the block is page-locked in place by the library), the
// bias-force delta down and added to `forces` on the host -- atom->f is never uploaded (edm_hip_bias_step_host)
double EDMBias::host_step(int nlocal, const double* const* positions, double** forces, const double* runiform, int apply_mask,
                          int hill_step) const {
  if (apply_mask >= 0 && mask_ == NULL) edm_error("a group mask needs set_mask", "edm_bias.cpp:update_forces");
  Stage& st = *st_;
  const size_t n = (size_t)nlocal;
  long xs = row_stride(nlocal, positions, dim_);
  long fs = row_stride(nlocal, const_cast<const double* const*>(forces), dim_);
  const double* xsrc = positions[0];
  double* fdst = forces[0];
  if (xs == 0) {  // rows are not evenly spaced: pack
    xs = (long)dim_;
    st.pack_x.resize(n * dim_);
    for (size_t i = 0; i < n; i++)
      for (unsigned int d = 0; d < dim_; d++) st.pack_x[i * dim_ + d] = positions[i][d];
    xsrc = st.pack_x.data();
  }
  const bool pack_f = (fs == 0);
  if (pack_f) {   // (the delta is added to a zeroed block and scattered below)
    fs = (long)dim_;
    st.pack_f.assign(n * dim_, 0.0);
    fdst = st.pack_f.data();
  }
  double energy = 0;
  check(edm_hip_bias_step_host(h_, nlocal, xsrc, (int)xs, fdst, (int)fs, apply_mask >= 0 ? mask_ : NULL, runiform, apply_mask,
                               hill_step, -1, &energy),
        hill_step ? "edm_bias.cpp:add_hills" : "edm_bias.cpp:update_forces");
  if (pack_f)
    for (size_t i = 0; i < n; i++)
      for (unsigned int d = 0; d < dim_; d++) forces[i][d] += st.pack_f[i * dim_ + d];
  return energy;
}

// edm_bias.cpp:297-311: a batch of one
double EDMBias::update_force(const double* positions, double* forces) const {
  if (b_outofbounds_) return 0.0;
  double der[3] = {0, 0, 0};
  const double e = bias_->get_value_deriv(positions, der);
  for (unsigned int d = 0; d < dim_; d++) forces[d] -= der[d];
  return e;
}

double EDMBias::update_pair_forces(int npairs, const double* r, double* force_r) const {
  if (npairs <= 0) return 0.0;
  Stage& st = *st_;
  const size_t bytes = sizeof(double) * (size_t)npairs;
  st.r.reserve(bytes);
  st.fr.reserve(bytes);
  check(edm_hip_memcpy_h2d(st.r.p, r, bytes), "edm_bias.cpp:update_force");
  double energy = 0;
  check(edm_hip_bias_pair_forces(h_, npairs, (const double*)st.r.p, (double*)st.fr.p, &energy), "edm_bias.cpp:update_force");
  check(edm_hip_memcpy_d2h(force_r, st.fr.p, bytes), "edm_bias.cpp:update_force");
  return energy;
}

void EDMBias::add_hills(int nlocal, const double* const* positions, const double* runiform) {
  add_hills(nlocal, positions, runiform, -1);
}

// edm_bias.cpp:401-411
void EDMBias::add_hills(int nlocal, const double* const* positions, const double* runiform, int apply_mask) {
  Stage& st = *st_;
  const size_t n = (size_t)(nlocal > 0 ? nlocal : 0);
  long xs = (long)dim_;
  if (n > 0) {
    xs = row_stride(nlocal, positions, dim_);
    const double* xsrc = positions[0];
    if (xs == 0) {
      xs = (long)dim_;
      st.pack_x.resize(n * dim_);
      for (size_t i = 0; i < n; i++)
        for (unsigned int d = 0; d < dim_; d++) st.pack_x[i * dim_ + d] = positions[i][d];
      xsrc = st.pack_x.data();
    }
    const size_t xbytes = sizeof(double) * ((n - 1) * (size_t)xs + dim_);
    st.x.reserve(xbytes);
    check(edm_hip_memcpy_h2d(st.x.p, xsrc, xbytes), "edm_bias.cpp:add_hills");
    if (runiform) {
      st.u.reserve(sizeof(double) * n);
      check(edm_hip_memcpy_h2d(st.u.p, runiform, sizeof(double) * n), "edm_bias.cpp:add_hills");
    }
    if (apply_mask >= 0) {
      if (mask_ == NULL) edm_error("add_hills with a group mask needs set_mask", "edm_bias.cpp:add_hills");
      st.mask.reserve(sizeof(int) * n);
      check(edm_hip_memcpy_h2d(st.mask.p, mask_, sizeof(int) * n), "edm_bias.cpp:add_hills");
      check(edm_hip_bias_set_mask(h_, (const int*)st.mask.p), "edm_bias.cpp:set_mask");
    }
  }
  check(edm_hip_bias_add_hills(h_, nlocal, (const double*)st.x.p, (int)xs, runiform ? (const double*)st.u.p : NULL,
                               apply_mask, -1),
        "edm_bias.cpp:add_hills");
  refresh();
}

void EDMBias::add_pair_hills(int n, const double* r, const double* runiform, int est_hill_count) {
  Stage& st = *st_;
  const size_t bytes = sizeof(double) * (size_t)(n > 0 ? n : 0);
  if (n > 0) {
    st.r.reserve(bytes);
    st.u.reserve(bytes);
    check(edm_hip_memcpy_h2d(st.r.p, r, bytes), "edm_bias.cpp:add_hill");
    check(edm_hip_memcpy_h2d(st.u.p, runiform, bytes), "edm_bias.cpp:add_hill");
  }
  check(edm_hip_bias_add_hills(h_, n, (const double*)st.r.p, 1, (const double*)st.u.p, -1, est_hill_count),
        "edm_bias.cpp:add_hill");
  refresh();
}

// One hill-depositing step of fix edm_pair on host arrays: copies and kernels queued back to back inside the library
// (distances up, force kernel, forces down while the staged samples go up, hill cycle), one wait
double EDMBias::pair_step(int npairs, const double* r, double* force_r, int n_samples, const double* sample_r,
                          const double* runiform, int est_hill_count) {
  double energy = 0;
  check(edm_hip_bias_pair_step_host(h_, npairs > 0 ? npairs : 0, r, force_r, n_samples > 0 ? n_samples : 0, sample_r,
                                    runiform, est_hill_count, &energy),
        "edm_bias.cpp:add_hill");
  refresh();
  return energy;
}

double EDMBias::pair_step_ordered(int npairs, const double* r, double* force_r, const int* first_sample, int n_samples,
                                  const double* sample_r, const double* runiform, int est_hill_count) {
  double energy = 0;
  check(edm_hip_bias_pair_step_ordered_host(h_, npairs > 0 ? npairs : 0, r, force_r, first_sample,
                                            n_samples > 0 ? n_samples : 0, sample_r, runiform, est_hill_count, &energy),
        "edm_bias.cpp:add_hill");
  refresh();
  return energy;
}

void EDMBias::set_reference_order(bool enabled) {
  check(edm_hip_bias_set(h_, "reference_order", enabled ? 1.0 : 0.0), "edm_bias.cpp:update_force");
}

double EDMBias::pair_list_step(int nlocal, int nall, const double* const* x, double** f, int inum, const int* ilist,
                               const int* numneigh, int* const* firstneigh, int neighmask, const int* type, int itype,
                               int jtype, bool list_changed, bool hill_step, int est_hill_count, int* ncalls) {
  Stage& st = *st_;
  if (ncalls) *ncalls = 0;
  if (dim_ != 1) edm_error("pair_list_step needs a 1-D CV", "edm_bias.cpp:update_force");
  if (list_changed || st.pl_npairs < 0) {
    // flatten the half list in neighbour-list order (fix_edm_pair.cpp:177-186)
    st.pl_hi.clear();
    st.pl_hj.clear();
    for (int ii = 0; ii < inum; ii++) {
      const int i = ilist[ii];
      const int* jlist = firstneigh[i];
      for (int jj = 0; jj < numneigh[i]; jj++) {
        st.pl_hi.push_back(i);
        st.pl_hj.push_back(jlist[jj] & neighmask);
      }
    }
    st.pl_npairs = (long long)st.pl_hi.size();
    check(edm_hip_bias_pair_list_upload(h_, st.pl_npairs, st.pl_hi.data(), st.pl_hj.data(), nall, type),
          "edm_bias.cpp:update_force");
  }
  const size_t n3 = (size_t)3 * (size_t)(nall > 0 ? nall : 0);
  if (n3 == 0) return 0.0;
  st.pl_x.reserve(sizeof(double) * n3);
  st.pl_f.reserve(sizeof(double) * n3);
  // positions: LAMMPS rows are one contiguous [nall][3] block; pack otherwise
  const long xs = row_stride(nall, x, 3);
  const double* xsrc = x[0];
  if (xs != 3) {
    st.pl_hx.resize(n3);
    for (int i = 0; i < nall; i++)
      for (int d = 0; d < 3; d++) st.pl_hx[(size_t)3 * i + d] = x[i][d];
    xsrc = st.pl_hx.data();
  }
  check(edm_hip_memcpy_h2d(st.pl_x.p, xsrc, sizeof(double) * n3), "edm_bias.cpp:update_force");
  double energy = 0;
  long long calls = 0;
  check(edm_hip_bias_pair_list_step(h_, nlocal, itype, jtype, (const double*)st.pl_x.p, (double*)st.pl_f.p,
                                    hill_step ? 1 : 0, est_hill_count, &energy, &calls),
        "edm_bias.cpp:add_hill");
  st.pl_hf.resize(n3);
  check(edm_hip_memcpy_d2h(st.pl_hf.data(), st.pl_f.p, sizeof(double) * n3), "edm_bias.cpp:update_force");
  for (int i = 0; i < nall; i++)
    for (int d = 0; d < 3; d++) f[i][d] += st.pl_hf[(size_t)3 * i + d];
  if (ncalls) *ncalls = (int)calls;
  if (hill_step) refresh();
  return energy;
}

void EDMBias::set_device_rng(bool enabled, unsigned long long seed) {
  check(edm_hip_bias_set_device_rng(h_, enabled ? 1 : 0, seed), "edm_bias.cpp:add_hill");
}

void EDMBias::pre_add_hill(int est_hill_count) { check(edm_hip_bias_pre_add_hill(h_, est_hill_count), "edm_bias.cpp:pre_add_hill"); }
void EDMBias::add_hill(const double* position, double runiform) {
  check(edm_hip_bias_add_hill(h_, position, runiform), "edm_bias.cpp:add_hill");
}
void EDMBias::post_add_hill() {
  check(edm_hip_bias_post_add_hill(h_), "edm_bias.cpp:post_add_hill");
  refresh();
}

void EDMBias::write_bias(const std::string& output) const {
  check(edm_hip_bias_write_bias(h_, output.c_str(), serial_format_), "edm_bias.cpp:write_bias");
}
void EDMBias::write_histogram() const { check(edm_hip_bias_write_histogram(h_, serial_format_), "edm_bias.cpp:write_histogram"); }
void EDMBias::clear_histogram() { check(edm_hip_bias_clear_histogram(h_), "edm_bias.cpp:clear_histogram"); }
void EDMBias::write_lammps_table(const std::string& output) const {
  check(edm_hip_bias_write_lammps_table(h_, output.c_str(), serial_format_), "edm_bias.cpp:write_lammps_table");
}

void EDMBias::select_device(int rank) {
  int n = 0;
  check(edm_hip_device_count(&n), "edm_bias.cpp:select_device");
  if (n < 1) edm_error("no MI355X visible: the EDM bias path has no CPU fallback", "edm_bias.cpp:select_device");
  check(edm_hip_set_device(rank % n), "edm_bias.cpp:select_device");
}
void EDMBias::make_comm_id(char id[128]) { check(edm_hip_comm_unique_id(id, 128), "edm_bias.cpp:make_comm_id"); }
void EDMBias::init_comm(const char id[128], int nranks, int rank) {
  check(edm_hip_bias_comm_init(h_, id, nranks, rank), "edm_bias.cpp:init_comm");
  refresh();
}
void EDMBias::set_hill_log(int enabled) { check(edm_hip_bias_set_hill_log(h_, enabled), "edm_bias.cpp:set_hill_log"); }

}  // namespace EDM
