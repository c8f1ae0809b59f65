// Drives the rewritten USER-EDM fixes against the mock LAMMPS classes: builds a small periodic LJ-like
// configuration with a half neighbour list, runs a few "timesteps" of fix edm_pair and fix edm, and
// prints energies / force checksums so the pytest wrapper can compare with the oracle's per-pair loop.
//   usage: drive_fixes <pair.edm> <coord.edm> <outfile>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lammps.h"
#include "fix_edm.h"
#include "fix_edm_pair.h"

using namespace LAMMPS_NS;

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  LAMMPS lmp;
  Error err;
  Memory mem;
  Atom atom;
  Update upd;
  Neighbor nb;
  Force frc;
  Domain dom;
  Pair pair;
  Respa respa;
  lmp.error = &err; lmp.memory = &mem; lmp.atom = &atom; lmp.update = &upd; lmp.neighbor = &nb; lmp.force = &frc;
  lmp.domain = &dom; lmp.world = MPI_COMM_WORLD;
  const int n = 512;
  const double box = 8.0;
  std::vector<double> xb(n * 3), fb(n * 3, 0.0);
  std::vector<double *> xr(n), fr(n);
  std::vector<int> type(n, 1), mask(n, 1);
  unsigned long long s = 99;
  for (int i = 0; i < n; i++) {
    xr[i] = &xb[3 * i];
    fr[i] = &fb[3 * i];
    for (int d = 0; d < 3; d++) {
      s = s * 6364136223846793005ULL + 1442695040888963407ULL;
      xb[3 * i + d] = box * ((s >> 11) * (1.0 / 9007199254740992.0));
    }
  }
  atom.tag_enable = 1; atom.nlocal = n; atom.nghost = 0; atom.nmax = n; atom.x = xr.data(); atom.f = fr.data();
  atom.type = type.data(); atom.mask = mask.data();
  upd.ntimestep = 0; upd.integrate_style = "verlet"; upd.integrate = &respa; respa.nlevels = 1;
  pair.cutforce = 2.5; frc.boltz = 1.0; frc.newton_pair = 0; frc.pair = &pair;
  for (int d = 0; d < 3; d++) { dom.boxlo[d] = dom.sublo[d] = 0; dom.boxhi[d] = dom.subhi[d] = box; dom.periodicity[d] = 1; }
  // half neighbour list, no periodic images (open cluster), cutoff + skin
  const double rc = pair.cutforce + nb.skin;
  std::vector<std::vector<int> > neigh(n);
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      double d2 = 0;
      for (int d = 0; d < 3; d++) d2 += (xb[3 * i + d] - xb[3 * j + d]) * (xb[3 * i + d] - xb[3 * j + d]);
      if (d2 < rc * rc) neigh[i].push_back(j);
    }
  std::vector<int> ilist(n), numneigh(n);
  std::vector<int *> firstneigh(n);
  for (int i = 0; i < n; i++) { ilist[i] = i; numneigh[i] = (int) neigh[i].size(); firstneigh[i] = neigh[i].data(); }
  NeighList list;
  list.inum = n; list.ilist = ilist.data(); list.numneigh = numneigh.data(); list.firstneigh = firstneigh.data();

  FILE *out = std::fopen(argv[3], "w");
  {
    char a0[] = "1", a1[] = "all", a2[] = "edm_pair", a3[] = "1.0", a5[] = "2", a6[] = "1000000", a8[] = "7", a9[] = "1", a10[] = "1";
    std::string bf = std::string(argv[3]) + ".pairbias";
    // argv[4..]: trailing keywords of fix edm_pair (device_rng, gpu_list, batch_order)
    std::vector<char *> args = {a0, a1, a2, a3, argv[1], a5, a6, &bf[0], a8, a9, a10};
    for (int a = 4; a < argc; a++) args.push_back(argv[a]);
    FixEDMPair fix(&lmp, (int) args.size(), args.data());
    std::fprintf(out, "pair_mask %d\n", fix.setmask());
    fix.init();
    fix.init_list(0, &list);
    long npairs = 0;
    for (int i = 0; i < n; i++) npairs += numneigh[i];
    std::fprintf(out, "pairs %ld\n", npairs);
    for (int step = 0; step < 6; step++) {
      upd.ntimestep = step;
      for (size_t k = 0; k < fb.size(); k++) fb[k] = 0;
      fix.post_force(0);
      double fsum = 0, fabs_sum = 0, fw = 0;
      for (size_t k = 0; k < fb.size(); k++) { fsum += fb[k]; fabs_sum += std::fabs(fb[k]); fw += fb[k] * std::cos(0.37 * (double) k); }
      std::fprintf(out, "pair_step %d E %.12e fsum %.6e fabs %.12e fw %.12e\n", step, fix.compute_scalar(), fsum, fabs_sum, fw);
    }
  }
  {
    char a0[] = "2", a1[] = "all", a2[] = "edm", a3[] = "1.0", a5[] = "2", a6[] = "1000000", a8[] = "11";
    std::string bf = std::string(argv[3]) + ".coordbias";
    char kw[] = "device_rng";
    bool fast_rng = false;
    for (int a = 4; a < argc; a++) fast_rng |= (std::string(argv[a]) == "device_rng");
    char *args[] = {a0, a1, a2, a3, argv[2], a5, a6, &bf[0], a8, kw};
    FixEDM fix(&lmp, fast_rng ? 10 : 9, args);
    std::fprintf(out, "coord_mask %d\n", fix.setmask());
    fix.init();
    for (int step = 0; step < 4; step++) {
      upd.ntimestep = step;
      for (size_t k = 0; k < fb.size(); k++) fb[k] = 0;
      fix.post_force(0);
      double fabs_sum = 0;
      for (size_t k = 0; k < fb.size(); k++) fabs_sum += std::fabs(fb[k]);
      std::fprintf(out, "coord_step %d E %.12e fabs %.12e\n", step, fix.compute_scalar(), fabs_sum);
    }
  }
  std::fclose(out);
  return 0;
}
