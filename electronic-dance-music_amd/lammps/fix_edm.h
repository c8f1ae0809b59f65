/* fix edm -- coordinate-CV EDM bias for LAMMPS (USER-EDM), MI355X build.
   Same fix style, arguments and semantics as the reference's lammps/fix_edm.{h,cpp}. */
#ifdef FIX_CLASS

FixStyle(edm,FixEDM)

#else

#ifndef LMP_FIX_EDM_H
#define LMP_FIX_EDM_H

#include "fix.h"
#include <edm/edm_bias.h>

namespace LAMMPS_NS {

class FixEDM : public Fix {
 public:
  FixEDM(class LAMMPS *, int, char **);
  ~FixEDM();
  int setmask();
  void init();
  void setup(int);
  void min_setup(int);
  void post_force(int);
  void post_force_respa(int, int, int);
  void min_post_force(int);
  double compute_scalar();

 private:
  class EDM::EDMBias *bias;
  char bias_file[256];
  double temperature;
  double edm_energy;
  int stride;
  int write_stride;
  double *random_numbers;
  int random_cap;
  class RanMars *random;
  unsigned int seed;
  bool device_rng;   // optional trailing keyword "device_rng": uniforms drawn on the GPU (fast mode, not RanMars)
  int nlevels_respa;
};

}

#endif
#endif
