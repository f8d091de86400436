/* _hypre_parcsr_mv.h -- src/HypreSystem.h:22. */
#ifndef hypre_PARCSR_MV_HEADER
#define hypre_PARCSR_MV_HEADER
#include "_hypre_utilities.h"
#include "HYPRE_parcsr_mv.h"
#include "HYPRE_IJ_mv.h"
#endif
