// Mock of the handful of LAMMPS classes the USER-EDM fixes touch.  NOT LAMMPS: just enough
// surface (names, members, signatures of the Fix plugin API) to compile and drive the fixes.
#ifndef MOCK_LAMMPS_H
#define MOCK_LAMMPS_H
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "mpi.h"

#define FLERR __FILE__, __LINE__
#define NEIGHMASK 0x3FFFFFFF

namespace LAMMPS_NS {
typedef int64_t bigint;

class Error {
 public:
  void all(const char *file, int line, const char *msg) {
    std::fprintf(stderr, "ERROR: %s (%s:%d)\n", msg, file, line);
    std::exit(1);
  }
};
class Memory {};
class Atom {
 public:
  int tag_enable, nlocal, nghost, nmax;
  double **x, **f;
  int *type, *mask;
};
class Integrate {};
class Respa : public Integrate {
 public:
  int nlevels;
  void copy_flevel_f(int) {}
  void copy_f_flevel(int) {}
};
class Update {
 public:
  bigint ntimestep;
  const char *integrate_style;
  Integrate *integrate;
};
class Pair {
 public:
  double cutforce;
};
class Force {
 public:
  double boltz;
  int newton_pair;
  Pair *pair;
};
class Domain {
 public:
  double boxlo[3], boxhi[3], sublo[3], subhi[3];
  int periodicity[3];
};
class NeighRequest {
 public:
  int pair, fix;
};
class NeighList {
 public:
  int inum;
  int *ilist, *numneigh;
  int **firstneigh;
};
class Neighbor {
 public:
  double skin;
  int ago;   // steps since the lists were last rebuilt
  NeighRequest **requests;
  int nrequest;
  void *last_requestor;
  Neighbor() : skin(0.3), ago(0), requests(NULL), nrequest(0), last_requestor(NULL) {}
  int request(void *who) {  // 2014-era API
    last_requestor = who;
    requests = (NeighRequest **) std::realloc(requests, sizeof(NeighRequest *) * (nrequest + 1));
    requests[nrequest] = new NeighRequest();
    return nrequest++;
  }
  NeighRequest *add_request(void *who, int = 0) {  // current API
    last_requestor = who;
    return NULL;
  }
};

class LAMMPS {
 public:
  Memory *memory;
  Error *error;
  Atom *atom;
  Update *update;
  Neighbor *neighbor;
  Force *force;
  Domain *domain;
  MPI_Comm world;
};

class Pointers {
 public:
  Pointers(LAMMPS *ptr)
      : lmp(ptr), memory(ptr->memory), error(ptr->error), atom(ptr->atom), update(ptr->update), neighbor(ptr->neighbor),
        force(ptr->force), domain(ptr->domain), world(ptr->world) {}
  virtual ~Pointers() {}

 protected:
  LAMMPS *lmp;
  Memory *&memory;
  Error *&error;
  Atom *&atom;
  Update *&update;
  Neighbor *&neighbor;
  Force *&force;
  Domain *&domain;
  MPI_Comm &world;
};

namespace FixConst {
static const int POST_FORCE = 1 << 6;
static const int THERMO_ENERGY = 1 << 9;
static const int POST_FORCE_RESPA = 1 << 12;
static const int MIN_POST_FORCE = 1 << 14;
}  // namespace FixConst

class Fix : protected Pointers {
 public:
  int thermo_energy, groupbit;
  Fix(LAMMPS *lmp, int, char **) : Pointers(lmp), thermo_energy(0), groupbit(1) {}
  virtual ~Fix() {}
  virtual int setmask() = 0;
  virtual void init() {}
  virtual void init_list(int, NeighList *) {}
  virtual void setup(int) {}
  virtual void min_setup(int) {}
  virtual void post_force(int) {}
  virtual void post_force_respa(int, int, int) {}
  virtual void min_post_force(int) {}
  virtual double compute_scalar() { return 0.0; }
};

// uniform generator with RanMars' interface (NOT Marsaglia's sequence: fix-level parity with a
// real LAMMPS run is unpinned)
class RanMars : protected Pointers {
 public:
  RanMars(LAMMPS *lmp, int seed) : Pointers(lmp), s((uint64_t) seed * 0x9E3779B97F4A7C15ULL + 1) {}
  double uniform() {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double) (z >> 11) * (1.0 / 9007199254740992.0);
  }

 private:
  uint64_t s;
};
}  // namespace LAMMPS_NS
#endif
