/* HYPRE_parcsr_mv.h -- ParCSR matrix / ParVector kernels. */
#ifndef HYPRE_PARCSR_MV_HEADER
#define HYPRE_PARCSR_MV_HEADER
#include "HYPRE_utilities.h"
#ifdef __cplusplus
extern "C" {
#endif

struct hypre_ParCSRMatrix_struct;
typedef struct hypre_ParCSRMatrix_struct *HYPRE_ParCSRMatrix;
struct hypre_ParVector_struct;
typedef struct hypre_ParVector_struct *HYPRE_ParVector;
typedef struct hypre_ParCSRMatrix_struct hypre_ParCSRMatrix;
typedef struct hypre_ParVector_struct hypre_ParVector;

/* y = alpha*A*x + beta*y : pack halo -> neighbour exchange -> diag SpMV -> offd SpMV */
HYPRE_Int HYPRE_ParCSRMatrixMatvec(HYPRE_Complex alpha, HYPRE_ParCSRMatrix A, HYPRE_ParVector x, HYPRE_Complex beta,
                                   HYPRE_ParVector y);
HYPRE_Int HYPRE_ParCSRMatrixGetDims(HYPRE_ParCSRMatrix A, HYPRE_BigInt *M, HYPRE_BigInt *N);
HYPRE_Int HYPRE_ParCSRMatrixGetLocalRange(HYPRE_ParCSRMatrix A, HYPRE_BigInt *row_start, HYPRE_BigInt *row_end,
                                          HYPRE_BigInt *col_start, HYPRE_BigInt *col_end);
/* src/HypreSystem.cpp:711 (AMG level dump) -- file <name>.%05d, IJ text dialect */
HYPRE_Int hypre_ParCSRMatrixPrintIJ(const hypre_ParCSRMatrix *A, HYPRE_Int base_i, HYPRE_Int base_j,
                                    const char *filename);

HYPRE_Int HYPRE_ParVectorSetConstantValues(HYPRE_ParVector v, HYPRE_Complex value); /* src/HypreSystem.cpp:579-580 */
HYPRE_Int HYPRE_ParVectorInnerProd(HYPRE_ParVector x, HYPRE_ParVector y, HYPRE_Real *prod);
HYPRE_Int HYPRE_ParVectorAxpy(HYPRE_Complex alpha, HYPRE_ParVector x, HYPRE_ParVector y);
HYPRE_Int HYPRE_ParVectorScale(HYPRE_Complex alpha, HYPRE_ParVector y);
HYPRE_Int HYPRE_ParVectorCopy(HYPRE_ParVector x, HYPRE_ParVector y);

#ifdef __cplusplus
}
#endif
#endif
