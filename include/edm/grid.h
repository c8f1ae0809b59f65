// edm/grid.h -- source-compatible Grid interface of the EDM library over the MI355X
// implementation (reference: lib/grid.h:142-182 abstract Grid, :911-928 factories).
//
// Grids live in HBM behind the C ABI (include/edm_hip.h).  Every virtual of the reference's
// Grid is kept with the same signature; single-sample calls are batches of one on the device.
// get_grid() returns a HOST SNAPSHOT of the node values (refreshed by every call to it).
#ifndef EDM_GRID_H_
#define EDM_GRID_H_

#include <cstddef>
#include <string>
#include <vector>

#include "edm.h"

struct edm_hip_grid;
struct edm_hip_gauss;

namespace EDM {

class Grid {
 public:
  virtual double get_value(const double* x) const = 0;
  virtual double add_value(const double* x0, double value) = 0;
  virtual ~Grid() {}
  virtual double get_value_deriv(const double* x, double* der) const = 0;
  virtual void write(const std::string& filename) const = 0;
  virtual void multi_write(const std::string& filename, const double* box_low, const double* box_high,
                           const int* b_periodic, int b_lammps_format) const = 0;
  virtual void read(const std::string& filename) = 0;
  virtual void set_interpolation(int b_interpolate) = 0;
  virtual double* get_grid() = 0;
  virtual const double* get_dx() const = 0;
  virtual const double* get_max() const = 0;
  virtual const double* get_min() const = 0;
  virtual double max_value() const = 0;
  virtual double min_value() const = 0;
  virtual void add(const Grid* other, double scale, double offset) = 0;
  virtual size_t get_grid_size() const = 0;
  virtual void one2multi(size_t index, size_t* result) const = 0;
  virtual double expected_bias() const = 0;
  virtual void clear() = 0;
};

// Device-resident DimmedGrid<DIM> (grid.h:184-905): the CV histogram and target flavour (no derivatives, no
// interpolation, make_grid(dim, ..., 0, 0), edm_bias.cpp:163) as well as grids with derivative records and
// cubic-Hermite interpolation (what read_grid returns for a PLUMED file with FORCE 1).
class HipGrid : public Grid {
 public:
  HipGrid(unsigned int dim, const double* min, const double* max, const double* bin_spacing, const int* b_periodic,
          int b_derivatives = 0, int b_interpolate = 0);
  HipGrid(unsigned int dim, const std::string& input_grid, int b_interpolate = 1);  // DimmedGrid(filename[, b_interpolate])
  explicit HipGrid(edm_hip_grid* borrowed);  // view of a grid owned by an EDMBias
  ~HipGrid();
  double get_value(const double* x) const;
  double add_value(const double* x0, double value);
  double get_value_deriv(const double* x, double* der) const;
  void write(const std::string& filename) const;
  void multi_write(const std::string& filename, const double* box_low, const double* box_high,
                   const int* b_periodic, int b_lammps_format) const;
  void read(const std::string& filename);
  void set_interpolation(int b_interpolate);
  double* get_grid();
  const double* get_dx() const;
  const double* get_max() const;
  const double* get_min() const;
  double max_value() const;
  double min_value() const;
  void add(const Grid* other, double scale, double offset);
  size_t get_grid_size() const;
  void one2multi(size_t index, size_t* result) const;
  double expected_bias() const;
  void clear();

  // index helpers of DimmedGrid (grid.h:264-273, :315-325, :865-874), host arithmetic in the reference's order
  void get_index(const double* x, size_t* result) const;
  size_t multi2one(const size_t* index) const;
  int in_grid(const double* x) const;
  // host snapshot of grid_deriv_ [grid_size][dim] (grids with derivatives)
  const double* get_grid_deriv();
  // overwrite node values / derivatives from host arrays (what writing grid_[i] / grid_deriv_[i] does in the reference)
  void set_grid(const double* values, const double* derivs);
  // batched lookup on host arrays: rows [n][stride]; value / deriv [n][dim] may be NULL
  void get_value_deriv_batch(size_t n, const double* x, int stride, double* value, double* deriv) const;

  // DimmedGrid's public members (grid.h:876-885)
  size_t grid_size_;
  int b_derivatives_, b_interpolate_;
  unsigned int dim_;
  double dx_[3], min_[3], max_[3];
  int grid_number_[3], b_periodic_[3];
  edm_hip_grid* handle() const { return h_; }

 private:
  void refresh_geometry();
  edm_hip_grid* h_;
  bool owned_;
  mutable std::vector<double> snapshot_, snapshot_deriv_;
};

// make_grid / read_grid (grid.h:911-928, grid.cpp:3-45)
Grid* make_grid(unsigned int dim, const double* min, const double* max, const double* bin_spacing,
                const int* b_periodic, int b_derivatives, int b_interpolate);
Grid* read_grid(unsigned int dim, const std::string& filename, int b_interpolate);
Grid* read_grid(unsigned int dim, const std::string& filename);

}  // namespace EDM
#endif
