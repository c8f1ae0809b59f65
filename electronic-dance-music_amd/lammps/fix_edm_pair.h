/* fix edm_pair -- pairwise-distance EDM bias for LAMMPS (USER-EDM), MI355X build.
   Same fix style, arguments and semantics as the reference's lammps/fix_edm_pair.{h,cpp};
   the per-pair update_force/add_hill calls are batched per timestep onto the GPU. */
#ifdef FIX_CLASS

FixStyle(edm_pair,FixEDMPair)

#else

#ifndef LMP_FIX_EDM_PAIR_H
#define LMP_FIX_EDM_PAIR_H

#include "fix.h"
#include <edm/edm_bias.h>
#include <vector>

namespace LAMMPS_NS {

class FixEDMPair : public Fix {
 public:
  FixEDMPair(class LAMMPS *, int, char **);
  ~FixEDMPair();
  int setmask();
  void init();
  void setup(int);
  void min_setup(int);
  void post_force(int);
  void post_force_respa(int, int, int);
  void min_post_force(int);
  void init_list(int, class NeighList *);
  bool list_rebuilt();
  double compute_scalar();

 private:
  class EDM::EDMBias *bias;
  char bias_file[512];
  char lammps_table_file[512];
  double temperature;
  double edm_energy;
  int stride;
  int write_stride;
  class RanMars *random;
  class NeighList *list;  // half neighbor list
  unsigned int seed;
  bool device_rng;   // optional trailing keyword "device_rng" (see fix_edm.h)
  bool gpu_list;     // optional trailing keyword "gpu_list" (implies device_rng): neighbour list resident on the GPU
  bool batch_order;  // optional trailing keyword "batch_order": every force of a hill step on the bias as it stands after
                     // pre_add_hill (default: the reference's order, pair k behind the hills of pairs 0..k-1)
  int last_list_size;  // total entries of the list last uploaded (re-upload when LAMMPS rebuilt it)
  int nlevels_respa;
  int last_calls;  // an estimate of the number of add_hill calls on this processor
  int ipair, jpair;
  // per-step batch of pair records (host staging, reused across steps)
  // (page-locked: they cross PCIe every step, queued around the kernels by EDMBias::pair_step)
  EDM::pinned_vector pair_r, pair_f;
  std::vector<double> pair_del;
  EDM::pinned_vector hill_r, hill_u;  // staged add_hill(r, uniform) calls of a hill step, in call order
  std::vector<int> pair_i, pair_j;
  std::vector<int> pair_first;  // hill steps: number of add_hill calls issued before each pair's update_force
};

}

#endif
#endif
