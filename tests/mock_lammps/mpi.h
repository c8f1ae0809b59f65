// Minimal serial MPI stand-in (like LAMMPS' own src/STUBS/mpi.h) for the mock-LAMMPS build of the fixes.
#ifndef MOCK_MPI_H
#define MOCK_MPI_H
typedef int MPI_Comm;
typedef int MPI_Datatype;
#define MPI_COMM_WORLD 0
#define MPI_CHAR 1
inline int MPI_Comm_rank(MPI_Comm, int *r) { *r = 0; return 0; }
inline int MPI_Comm_size(MPI_Comm, int *s) { *s = 1; return 0; }
inline int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm) { return 0; }
#endif
