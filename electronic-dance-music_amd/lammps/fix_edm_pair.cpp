/* fix [ID] [group-ID] edm_pair [temperature] [input_file] [add hill stride] [write bias stride]
       [bias file] [seed] [itype] [jtype]

   Same command line and behaviour as the reference fix (lammps/fix_edm_pair.cpp:20-60, :139-256).
   What changed is HOW the work reaches the bias: the reference calls EDMBias::update_force and
   add_hill once per neighbour-list entry; here one pass over the list collects the pair
   distances and, on hill steps, the add_hill samples in call order; ONE batched call applies the
   step's hills and evaluates energy and dV/dr of every pair on the GPU; a second pass applies
   the forces in the reference's order.  Within a hill step the reference lets the hills added
   for pairs 0..k-1 bias pair k's force (fix_edm_pair.cpp:215-237): the default reproduces that
   (EDMBias::pair_step_ordered).  Keyword batch_order evaluates every force of a hill step on
   the bias as it stands after pre_add_hill instead (faster; INTEGRATION.md has the deviation). */

#include "fix_edm_pair.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "atom.h"
#include "error.h"
#include "force.h"
#include "neigh_list.h"
#include "neigh_request.h"
#include "neighbor.h"
#include "pair.h"
#include "random_mars.h"
#include "respa.h"
#include "update.h"

using namespace LAMMPS_NS;
using namespace FixConst;

FixEDMPair::FixEDMPair(LAMMPS *lmp, int narg, char **arg) : Fix(lmp, narg, arg), bias(NULL), random(NULL), list(NULL)
{
  int me, size;
  if (narg < 11) error->all(FLERR, "Illegal fix edm_pair command");
  MPI_Comm_rank(world, &me);
  MPI_Comm_size(world, &size);
  if (!atom->tag_enable) error->all(FLERR, "fix edm_pair requires atom tags");

  temperature = atof(arg[3]);
  stride = atoi(arg[5]);
  write_stride = atoi(arg[6]);
  strncpy(bias_file, arg[7], sizeof(bias_file) - 1);
  bias_file[sizeof(bias_file) - 1] = '\0';
  snprintf(lammps_table_file, sizeof(lammps_table_file), "%s.ltab", arg[7]);
  seed = atoi(arg[8]);
  if (stride < 0) error->all(FLERR, "Illegal stride given to edm_pair command");
  if (write_stride < 0) error->all(FLERR, "Illegal write bias stride given to edm_pair command");
  ipair = atoi(arg[9]);
  jpair = atoi(arg[10]);
  if (!ipair || !jpair) error->all(FLERR, "Illegeal EDM command, invalid types");

  thermo_energy = 1;  // by default calculate energy
  EDM::EDMBias::select_device(me);  // one rank per GPU of the node
  bias = new EDM::EDMBias(arg[4]);
  if (bias->dim_ != 1) error->all(FLERR, "Pairwise distance must be 1 dimension in EDM input file");
  if (size > 1) {  // RCCL over xGMI replaces the reference's MPI hill exchange
    char id[128];
    if (me == 0) EDM::EDMBias::make_comm_id(id);
    MPI_Bcast(id, 128, MPI_CHAR, 0, world);
    bias->init_comm(id, size, me);
  }
  // extensions: "... jtype device_rng" draws the acceptance uniforms on the GPU; "... jtype gpu_list" also keeps the
  // neighbour list on the GPU: per step the positions go in and the bias forces come out (24 B per atom each way)
  // instead of one distance and one force per PAIR
  // "... jtype batch_order": every force of a hill step on the bias as it stands after pre_add_hill (the reference
  // reads pair k's force behind the hills of pairs 0..k-1 -- the default here too)
  gpu_list = device_rng = batch_order = false;
  for (int a = 11; a < narg; a++) {
    if (strcmp(arg[a], "gpu_list") == 0) gpu_list = true;
    else if (strcmp(arg[a], "device_rng") == 0) device_rng = true;
    else if (strcmp(arg[a], "batch_order") == 0) batch_order = true;
    else error->all(FLERR, "Illegal fix edm_pair keyword (gpu_list, device_rng, batch_order)");
  }
  last_list_size = -1;
  if (gpu_list) device_rng = true;
  // (with more than one rank a rank's pairs see ITS OWN hills of the step, like the reference's, whose ranks replay the
  //  other ranks' hills in post_add_hill only)
  if (device_rng) bias->set_device_rng(true, (unsigned long long) seed + (unsigned long long) me);
  bias->set_reference_order(!batch_order);
  random = new RanMars(lmp, seed + me);
  edm_energy = 0;
  last_calls = 0;
  nlevels_respa = 0;
}

FixEDMPair::~FixEDMPair()
{
  delete bias;
  delete random;
}

int FixEDMPair::setmask()
{
  int mask = 0;
  mask |= POST_FORCE;
  mask |= THERMO_ENERGY;
  mask |= POST_FORCE_RESPA;
  mask |= MIN_POST_FORCE;
  return mask;
}

void FixEDMPair::init()
{
  if (strcmp(update->integrate_style, "respa") == 0) nlevels_respa = ((Respa *) update->integrate)->nlevels;
  bias->setup(temperature, force->boltz);

  // the bounds for every rank are the bounds of the pairwise force (fix_edm_pair.cpp:95-104)
  double skin[3] = {neighbor->skin, 0, 0};
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  hi[0] = force->pair->cutforce + neighbor->skin;  // neighbor->cutneighmax is not yet known
  int p[3] = {0, 0, 0};
  bias->subdivide(lo, hi, lo, hi, p, skin);
  last_calls = atom->nmax;  // very conservative first estimate of the number of pairs

#ifdef EDM_LAMMPS_LEGACY_NEIGH
  int irequest = neighbor->request((void *) this);
  neighbor->requests[irequest]->pair = 0;
  neighbor->requests[irequest]->fix = 1;
#else
  neighbor->add_request(this);  // half list, built as a fix list
#endif
  edm_energy = 0;
}

void FixEDMPair::setup(int vflag)
{
  if (strcmp(update->integrate_style, "verlet") == 0)
    post_force(vflag);
  else {
    ((Respa *) update->integrate)->copy_flevel_f(nlevels_respa - 1);
    post_force_respa(vflag, nlevels_respa - 1, 0);
    ((Respa *) update->integrate)->copy_f_flevel(nlevels_respa - 1);
  }
}

void FixEDMPair::min_setup(int vflag) { post_force(vflag); }

void FixEDMPair::post_force(int /*vflag*/)
{
  double **x = atom->x;
  double **f = atom->f;
  int *type = atom->type;
  const int nlocal = atom->nlocal;
  const int newton_pair = force->newton_pair;
  if (newton_pair)
    error->all(FLERR, "fix edm_pair requires 'newton off' to be declared in the lammps input script");

  const int inum = list->inum;
  int *ilist = list->ilist;
  int *numneigh = list->numneigh;
  int **firstneigh = list->firstneigh;
  const bool hill_step = (update->ntimestep % stride == 0);

  edm_energy = 0;

  if (gpu_list) {
    // the list is re-uploaded when LAMMPS rebuilt it (neighbor->ago == 0 on those steps; the mock and very old
    // versions lack the field, so a changed total size or atom count also triggers it)
    int total = 0;
    for (int ii = 0; ii < inum; ii++) total += numneigh[ilist[ii]];
    const int nall = atom->nlocal + atom->nghost;
    const bool changed = (total != last_list_size) || list_rebuilt();
    last_list_size = total;
    int ncalls = 0;
    edm_energy = bias->pair_list_step(nlocal, nall, x, f, inum, ilist, numneigh, firstneigh, NEIGHMASK, type, ipair, jpair,
                                      changed, hill_step, last_calls, &ncalls);
    if (hill_step) last_calls = ncalls;  // next step's estimate (fix_edm_pair.cpp:245)
    if (update->ntimestep % write_stride == 0) {
      bias->write_bias(bias_file);
      bias->write_lammps_table(lammps_table_file);
      bias->write_histogram();
      bias->clear_histogram();
    }
    return;
  }

  // pass 1: collect the pair records in neighbour-list order (filters of fix_edm_pair.cpp:181-202)
  // and, on hill steps, stage the hill samples with their uniforms in the reference's call order
  // (one add_hill per list entry, a second one iff j is owned, :230-237)
  pair_r.clear(); pair_del.clear(); pair_i.clear(); pair_j.clear();
  hill_r.clear(); hill_u.clear(); pair_first.clear();
  for (int ii = 0; ii < inum; ii++) {
    const int i = ilist[ii];
    const int itype = type[i];
    int type_ind;
    if (itype == ipair) type_ind = 1;
    else if (itype == jpair) type_ind = 0;
    else continue;
    const double xtmp = x[i][0], ytmp = x[i][1], ztmp = x[i][2];
    int *jlist = firstneigh[i];
    const int jnum = numneigh[i];
    for (int jj = 0; jj < jnum; jj++) {
      int j = jlist[jj];
      j &= NEIGHMASK;
      const int jtype = type[j];
      if (type_ind && jtype != jpair) continue;
      else if (!type_ind && jtype != ipair) continue;
      double delx = xtmp - x[j][0], dely = ytmp - x[j][1], delz = ztmp - x[j][2];
      const double r = sqrt(delx * delx + dely * dely + delz * delz);
      const double rinv = 1.0 / r;
      delx *= rinv; dely *= rinv; delz *= rinv;
      pair_r.push_back(r);
      pair_del.push_back(delx); pair_del.push_back(dely); pair_del.push_back(delz);
      pair_i.push_back(i);
      pair_j.push_back(j);
      if (hill_step) {
        pair_first.push_back((int) hill_r.size());   // add_hill calls issued before this pair's update_force
        hill_r.push_back(r);
        if (!device_rng) hill_u.push_back(random->uniform());
        if (newton_pair || j < nlocal) {
          hill_r.push_back(r);
          if (!device_rng) hill_u.push_back(random->uniform());
        }
      }
    }
  }

  // one device round trip per step: the batched bias evaluation for all pairs of this rank and, on
  // hill steps, pre_add_hill(last_calls) (overflow flush) before it and the hill cycle behind it
  const int npairs = (int) pair_r.size();
  pair_f.resize(pair_r.size());
  if (hill_step) {
    const int ncalls = (int) hill_r.size();
    if (batch_order)
      edm_energy = bias->pair_step(npairs, pair_r.data(), pair_f.data(), ncalls, hill_r.data(),
                                   device_rng ? NULL : hill_u.data(), last_calls);
    else
      edm_energy = bias->pair_step_ordered(npairs, pair_r.data(), pair_f.data(), pair_first.data(), ncalls, hill_r.data(),
                                           device_rng ? NULL : hill_u.data(), last_calls);
    last_calls = ncalls;  // next step's estimate (fix_edm_pair.cpp:245)
  } else {
    edm_energy = bias->update_pair_forces(npairs, pair_r.data(), pair_f.data());
  }

  // pass 2: apply the pair forces
  for (int k = 0; k < npairs; k++) {
    const int i = pair_i[k], j = pair_j[k];
    const double fr = pair_f[k];
    const double *del = &pair_del[3 * (size_t) k];
    f[i][0] += del[0] * fr;
    f[i][1] += del[1] * fr;
    f[i][2] += del[2] * fr;
    if (newton_pair || j < nlocal) {
      f[j][0] -= del[0] * fr;
      f[j][1] -= del[1] * fr;
      f[j][2] -= del[2] * fr;
    }
  }
  if (update->ntimestep % write_stride == 0) {
    bias->write_bias(bias_file);
    bias->write_lammps_table(lammps_table_file);
    bias->write_histogram();
    bias->clear_histogram();
  }
}

bool FixEDMPair::list_rebuilt()
{
#ifdef EDM_LAMMPS_NO_NEIGHBOR_AGO
  return true;    // no way to tell: upload every step
#else
  return neighbor->ago == 0;
#endif
}

void FixEDMPair::post_force_respa(int vflag, int ilevel, int /*iloop*/)
{
  if (ilevel == nlevels_respa - 1) post_force(vflag);
}

void FixEDMPair::min_post_force(int vflag) { post_force(vflag); }

void FixEDMPair::init_list(int /*id*/, NeighList *ptr) { list = ptr; }

double FixEDMPair::compute_scalar() { return edm_energy; }
