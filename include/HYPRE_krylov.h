/* HYPRE_krylov.h -- Krylov solver handles (ParCSR flavour only; the driver uses
 * nothing else, src/HypreSystem.cpp:372-455). */
#ifndef HYPRE_KRYLOV_HEADER
#define HYPRE_KRYLOV_HEADER
#include "HYPRE_parcsr_mv.h"
#ifdef __cplusplus
extern "C" {
#endif

struct hypre_Solver_struct;
typedef struct hypre_Solver_struct *HYPRE_Solver;

/* the shape shared by every Setup/Solve, solver and preconditioner alike
 * (src/HypreSystem.h:265-277) */
typedef HYPRE_Int (*HYPRE_PtrToParSolverFcn)(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector);

#ifdef __cplusplus
}
#endif
#endif
