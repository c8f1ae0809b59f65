#include "lammps.h"
