// edm/edm_bias.h -- source-compatible EDMBias of the EDM library over the MI355X implementation
// (reference: lib/edm_bias.h:29-225).  The LAMMPS fixes compile against this header unchanged:
//   #include <edm/edm_bias.h>   and   -ledm
// All heavy lifting happens in libedm_hip.so behind include/edm_hip.h; this class keeps the
// reference's method signatures and public data members and stages LAMMPS' host arrays
// (atom->x, atom->f, atom->mask, uniform randoms) through HBM for each batched call.
#ifndef EDM_BIAS_H_
#define EDM_BIAS_H_

#include <string>
#include <vector>

#include "edm.h"
#include "gaussian_grid.h"
#include "grid.h"

#define BIAS_CLAMP 1.0
#define BIAS_BUFFER_SIZE 2048
#define BIAS_BUFFER_DBLS 8192
#define NO_COMM_PARTNER -1
#define INTERPOLATE 1

#define NEIGH_HILL 'n'
#define BUFF_HILL 'b'
#define BUFF_UNDO_HILL 'v'
#define ADD_HILL 'h'
#define ADD_UNDO_HILL 'u'
#define BUFF_ZERO_HILL 'z'

struct edm_hip_bias;

namespace EDM {

// Page-locked host memory for the arrays a fix hands to the batched calls every step (pair distances, pair forces,
// staged add_hill samples): std::vector<double, EDM::PinnedAllocator<double> >.  Such arrays travel to and from the
// GPU by DMA, queued around the kernels (EDMBias::pair_step); ordinary arrays work too, staged by the runtime.
void* pinned_alloc(size_t bytes);
void pinned_free(void* p);
template <class T>
struct PinnedAllocator {
  typedef T value_type;
  PinnedAllocator() {}
  template <class U> PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(pinned_alloc(n * sizeof(T))); }
  void deallocate(T* p, size_t) { pinned_free(p); }
  template <class U> bool operator==(const PinnedAllocator<U>&) const { return true; }
  template <class U> bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
typedef std::vector<double, PinnedAllocator<double> > pinned_vector;

class EDMBias {
 public:
  EDMBias(const std::string& input_filename);
  ~EDMBias();

  void subdivide(const double sublo[3], const double subhi[3], const double boxlo[3], const double boxhi[3],
                 const int b_periodic[3], const double skin[3]);
  void setup(double temperature, double boltzmann_constant);
  int read_input(const std::string& input_filename);

  double update_forces(int nlocal, const double* const* positions, double** forces, int apply_mask) const;
  double update_forces(int nlocal, const double* const* positions, double** forces) const;
  double update_force(const double* positions, double* forces) const;
  void set_mask(const int* mask);
  void add_hills(int nlocal, const double* const* positions, const double* runiform);
  void add_hills(int nlocal, const double* const* positions, const double* runiform, int apply_mask);
  void pre_add_hill(int est_hill_count);
  void add_hill(const double* position, double runiform);
  void post_add_hill();

  void write_bias(const std::string& output) const;
  void write_histogram() const;
  void clear_histogram();
  void write_lammps_table(const std::string& output) const;

  // ---- batched additions used by the rewritten USER-EDM fixes ----
  // fix_edm_pair's inner loop (fix_edm_pair.cpp:215-217) over an array of pair distances:
  // force_r[i] = edm_force[0] after update_force(&r[i], edm_force) on a zeroed accumulator.
  double update_pair_forces(int npairs, const double* r, double* force_r) const;
  // one hill cycle over a flat distance array: pre_add_hill(est); add_hill(&r[i], runiform[i]); post
  void add_pair_hills(int n, const double* r, const double* runiform, int est_hill_count);
  // fast mode of the random numbers: add_hills / step / pair_step called with runiform == NULL draw their
  // uniforms from a counter-based stream on the device (see edm_hip_bias_set_device_rng); use seed + rank
  void set_device_rng(bool enabled, unsigned long long seed);
  // one hill-depositing fix_edm step in one call: update_forces(nlocal, positions, forces, apply_mask) then
  // add_hills(nlocal, positions, runiform, apply_mask); positions and mask cross PCIe once, one device wait
  double step(int nlocal, const double* const* positions, double** forces, const double* runiform, int apply_mask);
  // one hill-depositing fix_edm_pair step in one call (one device round trip): pre_add_hill(est);
  // force_r = update_pair_forces(npairs, r); add_hill(&sample_r[i], runiform[i]) for i < n_samples;
  // post_add_hill().  Returns the bias energy of the pairs.
  double pair_step(int npairs, const double* r, double* force_r, int n_samples, const double* sample_r,
                   const double* runiform, int est_hill_count);
  // the same step in the REFERENCE'S OWN ORDER (lammps/fix_edm_pair.cpp:177-238): force_r[k] is read from the bias as it
  // stood when the reference's loop reached pair k -- after the add_hill calls of pairs 0..k-1 of this step.
  // first_sample[k] = number of add_hill calls (samples) issued before pair k.  Hills, grid, histogram and limiter
  // state are those of pair_step; only the forces and the energy of a hill step differ.
  double pair_step_ordered(int npairs, const double* r, double* force_r, const int* first_sample, int n_samples,
                           const double* sample_r, const double* runiform, int est_hill_count);
  // pair_list_step in the reference's order too (default off: every force on the bias as it stands after pre_add_hill)
  void set_reference_order(bool enabled);
  // fix edm_pair with the neighbour list resident on the GPU (needs set_device_rng): LAMMPS' half list
  // (ilist/numneigh/firstneigh, j masked with neighmask) is flattened and uploaded when list_changed, this
  // step's positions x[nall][3] go in, the bias forces come back ADDED to f[nall][3] (i always, j iff
  // j < nlocal), the return value is the bias energy.  hill_step: pre_add_hill(est_hill_count) first, then
  // one add_hill per list entry of the right types and a second iff j is owned; *ncalls = calls made.
  double pair_list_step(int nlocal, int nall, const double* const* x, double** f, int inum, const int* ilist,
                        const int* numneigh, int* const* firstneigh, int neighmask, const int* type, int itype,
                        int jtype, bool list_changed, bool hill_step, int est_hill_count, int* ncalls);
  // one rank per GPU: id is an ncclUniqueId made by make_comm_id() on rank 0 and broadcast by the
  // caller (MPI_Bcast in LAMMPS); must be called before subdivide
  // binds this process to GPU (rank % visible devices); call before constructing an EDMBias
  static void select_device(int rank);
  static void make_comm_id(char id[128]);
  void init_comm(const char id[128], int nranks, int rank);
  // serial_format = 1 reproduces the reference's EDM_SERIAL writers (all plain PLUMED grids),
  // 0 the MPI build's multi_write / LAMMPS table (default: the MPI build, like an installed LAMMPS)
  void set_serial_format(int serial_format) { serial_format_ = serial_format; }
  void set_hill_log(int enabled);
  edm_hip_bias* handle() const { return h_; }

  // public data members (edm_bias.h:118-157); refreshed from the controller after every call
  int b_tempering_;
  int b_targeting_;
  int mpi_rank_;
  int mpi_size_;
  unsigned int dim_;
  double global_tempering_;
  double bias_factor_;
  double boltzmann_factor_;
  double temperature_;
  double hill_prefactor_;
  double bias_per_step_;
  double hill_density_;
  double cum_bias_;
  double total_volume_;
  double expected_target_;
  int b_outofbounds_;
  double* bias_dx_;
  double* bias_sigma_;
  double* min_;
  double* max_;
  int* b_periodic_boundary_;
  Grid* target_;
  Grid* initial_bias_;
  GaussGrid* bias_;
  const int* mask_;
  unsigned int mpi_neighbor_count_;
  int* mpi_neighbors_;

 private:
  EDMBias(const EDMBias& that);  // just disable copy constructor
  void refresh() const;
  double host_step(int nlocal, const double* const* positions, double** forces, const double* runiform, int apply_mask,
                   int hill_step) const;
  edm_hip_bias* h_;
  Grid* cv_hist_;
  int serial_format_;
  // device staging of host arrays
  struct Stage;
  Stage* st_;
};

}  // namespace EDM
#endif  // EDM_BIAS_H_
