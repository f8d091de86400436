// The MPI calls the driver makes (init/finalize, rank/size, three
// tiny all-reduces, barriers) expressed over the library's rank/GPU communicator.
// No MPI library is involved: processes are started one per GPU by any launcher
// that exports RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.
#pragma once
#include <cstdlib>
#include <cstring>

#include "HYPRE_mi_ext.h"

enum MiMpiType { MPI_INT = 2, MPI_LONG_LONG_INT = 1, MPI_DOUBLE = 0 };
enum MiMpiOp { MPI_SUM = 0, MPI_MIN = 1, MPI_MAX = 2 };

inline int MPI_Init(int *, char ***) { return 0; }  // the communicator is bound after hipSetDevice (main.cpp)
inline int MPI_Finalize() { return HYPRE_MI_CommFinalize(); }
inline int MPI_Comm_rank(MPI_Comm, int *r) { return HYPRE_MI_CommRank(r); }
inline int MPI_Comm_size(MPI_Comm, int *s) { return HYPRE_MI_CommSize(s); }
inline int MPI_Barrier(MPI_Comm) { return HYPRE_MI_CommBarrier(); }
inline size_t mi_mpi_size(MiMpiType t) { return t == MPI_INT ? 4 : 8; }
inline int MPI_Allreduce(const void *send, void *recv, int count, MiMpiType t, MiMpiOp op, MPI_Comm) {
  if (send != recv) memcpy(recv, send, mi_mpi_size(t) * (size_t)count);
  return HYPRE_MI_CommAllreduce(recv, (size_t)count, (int)t, (int)op);
}
// every rank receives the result (a superset of MPI_Reduce's contract)
inline int MPI_Reduce(const void *send, void *recv, int count, MiMpiType t, MiMpiOp op, int, MPI_Comm c) {
  return MPI_Allreduce(send, recv, count, t, op, c);
}
inline int mi_env_int(const char *name, int dflt) { return getenv(name) ? atoi(getenv(name)) : dflt; }
