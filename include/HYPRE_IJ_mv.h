/* HYPRE_IJ_mv.h -- IJ matrix/vector assembly boundary (SURVEY.md 8b rows
 * "IJ matrix", "IJ vector").  Array arguments may be HOST or DEVICE pointers;
 * the library queries the pointer (the reference passes device pointers when
 * HYPRE_USING_GPU, src/HypreSystem.cpp:905-947, :973-1007). */
#ifndef HYPRE_IJ_MV_HEADER
#define HYPRE_IJ_MV_HEADER
#include "HYPRE_utilities.h"
#ifdef __cplusplus
extern "C" {
#endif

#define HYPRE_PARCSR 5555

struct hypre_IJMatrix_struct;
typedef struct hypre_IJMatrix_struct *HYPRE_IJMatrix;
struct hypre_IJVector_struct;
typedef struct hypre_IJVector_struct *HYPRE_IJVector;

/* src/HypreSystem.cpp:552 -- rows [ilower,iupper], diag-block columns [jlower,jupper] (inclusive) */
HYPRE_Int HYPRE_IJMatrixCreate(MPI_Comm comm, HYPRE_BigInt ilower, HYPRE_BigInt iupper, HYPRE_BigInt jlower,
                               HYPRE_BigInt jupper, HYPRE_IJMatrix *matrix);
HYPRE_Int HYPRE_IJMatrixDestroy(HYPRE_IJMatrix matrix);                            /* :501 */
HYPRE_Int HYPRE_IJMatrixSetObjectType(HYPRE_IJMatrix matrix, HYPRE_Int type);      /* :553 */
HYPRE_Int HYPRE_IJMatrixInitialize(HYPRE_IJMatrix matrix);                         /* :554 */
/* :555, :606 -- borrowed HYPRE_ParCSRMatrix, valid before and after Assemble */
HYPRE_Int HYPRE_IJMatrixGetObject(HYPRE_IJMatrix matrix, void **object);
HYPRE_Int HYPRE_IJMatrixSetConstantValues(HYPRE_IJMatrix matrix, HYPRE_Complex value); /* :556 */
/* :942, :945, :1567 -- ncols == NULL means one entry per row; row_indexes == NULL
 * means packed; a later Set of the same (row,col) overwrites */
HYPRE_Int HYPRE_IJMatrixSetValues2(HYPRE_IJMatrix matrix, HYPRE_Int nrows, HYPRE_Int *ncols, const HYPRE_BigInt *rows,
                                   const HYPRE_Int *row_indexes, const HYPRE_BigInt *cols,
                                   const HYPRE_Complex *values);
/* :1572 -- accumulates */
HYPRE_Int HYPRE_IJMatrixAddToValues2(HYPRE_IJMatrix matrix, HYPRE_Int nrows, HYPRE_Int *ncols,
                                     const HYPRE_BigInt *rows, const HYPRE_Int *row_indexes,
                                     const HYPRE_BigInt *cols, const HYPRE_Complex *values);
HYPRE_Int HYPRE_IJMatrixSetValues(HYPRE_IJMatrix matrix, HYPRE_Int nrows, HYPRE_Int *ncols, const HYPRE_BigInt *rows,
                                  const HYPRE_BigInt *cols, const HYPRE_Complex *values);
HYPRE_Int HYPRE_IJMatrixAddToValues(HYPRE_IJMatrix matrix, HYPRE_Int nrows, HYPRE_Int *ncols,
                                    const HYPRE_BigInt *rows, const HYPRE_BigInt *cols, const HYPRE_Complex *values);
HYPRE_Int HYPRE_IJMatrixAssemble(HYPRE_IJMatrix matrix);                           /* :605 (collective) */
HYPRE_Int HYPRE_IJMatrixPrint(HYPRE_IJMatrix matrix, const char *filename);        /* :746 */
HYPRE_Int HYPRE_IJMatrixGetLocalRange(HYPRE_IJMatrix matrix, HYPRE_BigInt *ilower, HYPRE_BigInt *iupper,
                                      HYPRE_BigInt *jlower, HYPRE_BigInt *jupper);  /* :1097 (dead code) */
/* sizing hints of the dead fast-assemble path (:933-940): accepted, ignored */
HYPRE_Int HYPRE_IJMatrixSetMaxOnProcElmts(HYPRE_IJMatrix matrix, HYPRE_Int max_on_proc_elmts);
HYPRE_Int HYPRE_IJMatrixSetOffProcSendElmts(HYPRE_IJMatrix matrix, HYPRE_Int n);
HYPRE_Int HYPRE_IJMatrixSetOffProcRecvElmts(HYPRE_IJMatrix matrix, HYPRE_Int n);
HYPRE_Int HYPRE_IJMatrixRead(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJMatrix *matrix);

HYPRE_Int HYPRE_IJVectorCreate(MPI_Comm comm, HYPRE_BigInt jlower, HYPRE_BigInt jupper, HYPRE_IJVector *vector); /* :567 */
HYPRE_Int HYPRE_IJVectorDestroy(HYPRE_IJVector vector);                            /* :504 */
HYPRE_Int HYPRE_IJVectorSetObjectType(HYPRE_IJVector vector, HYPRE_Int type);      /* :568 */
HYPRE_Int HYPRE_IJVectorSetNumComponents(HYPRE_IJVector vector, HYPRE_Int num_components); /* :569 */
HYPRE_Int HYPRE_IJVectorSetComponent(HYPRE_IJVector vector, HYPRE_Int component);  /* :967 */
HYPRE_Int HYPRE_IJVectorInitialize(HYPRE_IJVector vector);                         /* :570 */
HYPRE_Int HYPRE_IJVectorGetObject(HYPRE_IJVector vector, void **object);           /* :571 */
HYPRE_Int HYPRE_IJVectorSetValues(HYPRE_IJVector vector, HYPRE_Int nvalues, const HYPRE_BigInt *indices,
                                  const HYPRE_Complex *values);                    /* :1002, :1596 */
HYPRE_Int HYPRE_IJVectorAddToValues(HYPRE_IJVector vector, HYPRE_Int nvalues, const HYPRE_BigInt *indices,
                                    const HYPRE_Complex *values);
/* :804 -- indices == NULL means the local range in order */
HYPRE_Int HYPRE_IJVectorGetValues(HYPRE_IJVector vector, HYPRE_Int nvalues, const HYPRE_BigInt *indices,
                                  HYPRE_Complex *values);
HYPRE_Int HYPRE_IJVectorAssemble(HYPRE_IJVector vector);                           /* :609 */
HYPRE_Int HYPRE_IJVectorPrint(HYPRE_IJVector vector, const char *filename);        /* :749 */
HYPRE_Int HYPRE_IJVectorSetMaxOnProcElmts(HYPRE_IJVector vector, HYPRE_Int n);
HYPRE_Int HYPRE_IJVectorSetOffProcSendElmts(HYPRE_IJVector vector, HYPRE_Int n);
HYPRE_Int HYPRE_IJVectorSetOffProcRecvElmts(HYPRE_IJVector vector, HYPRE_Int n);
HYPRE_Int HYPRE_IJVectorRead(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJVector *vector);

#ifdef __cplusplus
}
#endif
#endif
