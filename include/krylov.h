/* krylov.h -- src/HypreSystem.h:23 includes it for the Krylov typedefs. */
#ifndef hypre_KRYLOV_HEADER
#define hypre_KRYLOV_HEADER
#include "HYPRE_krylov.h"
#endif
