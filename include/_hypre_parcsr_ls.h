/* _hypre_parcsr_ls.h -- src/HypreSystem.h:21.  The driver reaches into
 * hypre_ParAMGData to dump the level operators (src/HypreSystem.cpp:705-711);
 * the two accessors it uses are provided as functions over the opaque handle. */
#ifndef hypre_PARCSR_LS_HEADER
#define hypre_PARCSR_LS_HEADER
#include "_hypre_parcsr_mv.h"
#include "HYPRE_parcsr_ls.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct hypre_Solver_struct hypre_ParAMGData;
/* array of num_levels borrowed level operators, owned by the AMG object */
hypre_ParCSRMatrix **hypre_ParAMGDataAArray(hypre_ParAMGData *amg_data);
HYPRE_Int hypre_ParAMGDataNumLevels(hypre_ParAMGData *amg_data);
#ifdef __cplusplus
}
#endif
#endif
