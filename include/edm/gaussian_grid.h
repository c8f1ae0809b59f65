// edm/gaussian_grid.h -- source-compatible GaussGrid interface over the MI355X implementation
// (reference: lib/gaussian_grid.h:41-56 abstract GaussGrid, :636-647 factories).
#ifndef EDM_GAUSS_GRID_H_
#define EDM_GAUSS_GRID_H_

#include <string>
#include <vector>

#include "edm.h"
#include "grid.h"

#define GAUSS_SUPPORT 8.0    // sigma^2 considered for gaussian (gaussian_grid.h:10)
#define BC_TABLE_SIZE 65536  // boundary correction function look up size (gaussian_grid.h:11)
#define BC_MAR 2.0

namespace EDM {

class GaussGrid : public Grid {
 public:
  virtual ~GaussGrid() {}
  virtual double add_value(const double* x, double height) = 0;
  virtual void set_boundary(const double* min, const double* max, const int* b_periodic) = 0;
  virtual double get_volume() const = 0;
  virtual int in_bounds(const double* x) const = 0;
  virtual void multi_write(const std::string& filename) const = 0;
  virtual void lammps_multi_write(const std::string& filename) const = 0;
  using Grid::multi_write;
};

// Device-resident DimmedGaussGrid<DIM> (gaussian_grid.h:58-631).
class HipGaussGrid : public GaussGrid {
 public:
  HipGaussGrid(unsigned int dim, const double* min, const double* max, const double* bin_spacing,
               const int* b_periodic, int b_interpolate, const double* sigma);
  HipGaussGrid(unsigned int dim, const std::string& filename, const double* sigma);  // DimmedGaussGrid(filename, sigma)
  explicit HipGaussGrid(edm_hip_gauss* borrowed);  // view of the grid owned by an EDMBias
  ~HipGaussGrid();

  double get_value(const double* x) const;
  double get_value_deriv(const double* x, double* der) const;
  double add_value(const double* x0, double height);
  void read(const std::string& filename);
  void write(const std::string& filename) const;
  void multi_write(const std::string& filename) const;
  void lammps_multi_write(const std::string& filename) const;
  void multi_write(const std::string& filename, const double* box_low, const double* box_high,
                   const int* b_periodic, int b_lammps_format) const;
  void set_interpolation(int b_interpolate);
  void set_boundary(const double* min, const double* max, const int* b_periodic);
  double get_volume() const;
  void one2multi(size_t index, size_t* result) const;
  double* get_grid();
  const double* get_dx() const;
  const double* get_min() const;
  const double* get_max() const;
  double max_value() const;
  double min_value() const;
  void add(const Grid* other, double scale, double offset);
  double expected_bias() const;
  void clear();
  size_t get_grid_size() const;
  int in_bounds(const double* x) const;
  // DimmedGaussGrid::remap (gaussian_grid.h:504-541), evaluated by the device code the lookups use; x updated in place
  void remap(double* x) const;

  // ---- batched entry points (host arrays; staged through HBM) ----
  // positions/derivs row-major [n][stride]; returns sum of values
  double get_value_deriv_batch(size_t n, const double* x, int stride, double* energy, double* deriv) const;
  // applies n hills in order, writes each hill's integrated bias to added (may be NULL)
  void add_values(size_t n, const double* x, int stride, const double* heights, double* added);
  // host snapshot of the derivative array [grid_size][dim] (grid_.grid_deriv_ of the reference)
  const double* get_grid_deriv();
  // overwrite node values / derivatives from host arrays (what writing grid_.grid_[i] does in the reference)
  void set_grid(const double* values, const double* derivs);

  // public geometry members of DimmedGaussGrid / its DimmedGrid (gaussian_grid.h:544-549)
  unsigned int dim_;
  size_t minisize_[3];
  double sigma_[3];
  double boundary_min_[3], boundary_max_[3];
  int b_periodic_boundary_[3];
  double dx_[3], min_[3], max_[3];
  int grid_number_[3], b_periodic_[3];
  size_t grid_size_;
  edm_hip_gauss* handle() const { return h_; }

 private:
  void refresh_geometry();
  edm_hip_gauss* h_;
  bool owned_;
  mutable std::vector<double> snapshot_, snapshot_deriv_;
};

GaussGrid* make_gauss_grid(unsigned int dim, const double* min, const double* max, const double* bin_spacing,
                           const int* b_periodic, int b_interpolate, const double* sigma);
// read_gauss_grid (gaussian_grid.h:647, gaussian_grid.cpp:23-33)
GaussGrid* read_gauss_grid(unsigned int dim, const std::string& filename, const double* sigma);

}  // namespace EDM
#endif
