/* fix [ID] [group-ID] edm [temperature] [input_file] [add hill stride] [write bias stride] [bias file] [seed]

   Same command line and behaviour as the reference fix (lammps/fix_edm.cpp:35-60, :134-162).
   EDMBias::update_forces / add_hills already take LAMMPS' per-atom arrays, so the fix body is the
   reference's: the batching happens behind those two calls.  Deviation from the reference,
   documented in DESIGN.md: the bias grid is replicated on every GPU over the whole box instead of
   being cut into per-rank sub-grids with a skin (edm_bias.cpp:142-161), so every rank passes the
   box bounds as its sub-domain. */

#include "fix_edm.h"

#include <cstdlib>
#include <cstring>

#include "atom.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "neighbor.h"
#include "random_mars.h"
#include "respa.h"
#include "update.h"

using namespace LAMMPS_NS;
using namespace FixConst;

FixEDM::FixEDM(LAMMPS *lmp, int narg, char **arg) : Fix(lmp, narg, arg), bias(NULL), random_numbers(NULL), random_cap(0), random(NULL)
{
  int me, size;
  if (narg < 9) error->all(FLERR, "Illegal fix EDM command");
  MPI_Comm_rank(world, &me);
  MPI_Comm_size(world, &size);
  if (!atom->tag_enable) error->all(FLERR, "fix EDM requires atom tags");

  temperature = atof(arg[3]);
  stride = atoi(arg[5]);
  write_stride = atoi(arg[6]);
  strncpy(bias_file, arg[7], sizeof(bias_file) - 1);
  bias_file[sizeof(bias_file) - 1] = '\0';
  seed = atoi(arg[8]);
  if (stride < 0) error->all(FLERR, "Illegal stride given to EDM command");
  if (write_stride < 0) error->all(FLERR, "Illegal write bias stride given to EDM command");

  EDM::EDMBias::select_device(me);
  bias = new EDM::EDMBias(arg[4]);
  if (size > 1) {
    char id[128];
    if (me == 0) EDM::EDMBias::make_comm_id(id);
    MPI_Bcast(id, 128, MPI_CHAR, 0, world);
    bias->init_comm(id, size, me);
  }
  // extension: "... seed device_rng" draws the acceptance uniforms on the GPU from a counter-based stream keyed
  // by seed + rank (no RanMars calls, nothing to upload); without the keyword the reference's RNG order is kept
  device_rng = (narg > 9 && strcmp(arg[9], "device_rng") == 0);
  if (device_rng) bias->set_device_rng(true, (unsigned long long) seed + (unsigned long long) me);
  thermo_energy = 1;
  random = new RanMars(lmp, seed + me);
  edm_energy = 0;
  nlevels_respa = 0;
}

FixEDM::~FixEDM()
{
  delete bias;
  delete random;
  free(random_numbers);
}

int FixEDM::setmask()
{
  int mask = 0;
  mask |= POST_FORCE;
  mask |= THERMO_ENERGY;
  mask |= POST_FORCE_RESPA;
  mask |= MIN_POST_FORCE;
  return mask;
}

void FixEDM::init()
{
  if (strcmp(update->integrate_style, "respa") == 0) nlevels_respa = ((Respa *) update->integrate)->nlevels;
  bias->setup(temperature, force->boltz);
  double skin[3];
  skin[0] = skin[1] = skin[2] = neighbor->skin;
  // replicated grid: this rank's "sub-domain" is the whole box
  bias->subdivide(domain->boxlo, domain->boxhi, domain->boxlo, domain->boxhi, domain->periodicity, skin);
  edm_energy = 0;
}

void FixEDM::setup(int vflag)
{
  if (strcmp(update->integrate_style, "verlet") == 0)
    post_force(vflag);
  else {
    ((Respa *) update->integrate)->copy_flevel_f(nlevels_respa - 1);
    post_force_respa(vflag, nlevels_respa - 1, 0);
    ((Respa *) update->integrate)->copy_f_flevel(nlevels_respa - 1);
  }
}

void FixEDM::min_setup(int vflag) { post_force(vflag); }

void FixEDM::post_force(int /*vflag*/)
{
  bias->set_mask(atom->mask);  // re-fetched every call: LAMMPS may reallocate atom->mask
  if (update->ntimestep % stride == 0) {
    if (random_cap < atom->nmax) {  // the bias is paid in uniform random numbers (fix_edm.cpp:145-151)
      free(random_numbers);
      random_cap = atom->nmax;
      random_numbers = (double *) malloc(sizeof(double) * (size_t) (random_cap > 0 ? random_cap : 1));
    }
    if (!device_rng)
      for (int i = 0; i < atom->nlocal; i++) random_numbers[i] = random->uniform();
    // update_forces + add_hills in one call: positions and the group mask cross PCIe once, one device wait
    edm_energy = bias->step(atom->nlocal, atom->x, atom->f, device_rng ? NULL : random_numbers, groupbit);
  } else {
    edm_energy = bias->update_forces(atom->nlocal, atom->x, atom->f, groupbit);
  }

  if (update->ntimestep % write_stride == 0) {
    bias->write_bias(bias_file);
    bias->write_histogram();
    bias->clear_histogram();
  }
}

void FixEDM::post_force_respa(int vflag, int ilevel, int /*iloop*/)
{
  if (ilevel == nlevels_respa - 1) post_force(vflag);
}

void FixEDM::min_post_force(int vflag) { post_force(vflag); }

double FixEDM::compute_scalar() { return edm_energy; }
