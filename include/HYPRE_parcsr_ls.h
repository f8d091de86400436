/* HYPRE_parcsr_ls.h -- BoomerAMG, ParCSR GMRES / BiCGSTAB and the solver families
 * the driver can name but the north-star path does not include (stubs that
 * report HYPRE_ERROR_GENERIC).  SURVEY.md 8a rows a2-a7, Appendix C. */
#ifndef HYPRE_PARCSR_LS_HEADER
#define HYPRE_PARCSR_LS_HEADER
#include "HYPRE_krylov.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- BoomerAMG
 * src/HypreSystem.cpp:119-326 (preconditioner) and :91-117 (solver) */
HYPRE_Int HYPRE_BoomerAMGCreate(HYPRE_Solver *solver);                                  /* :122 */
HYPRE_Int HYPRE_BoomerAMGDestroy(HYPRE_Solver solver);                                  /* :325 */
HYPRE_Int HYPRE_BoomerAMGSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x); /* :323 */
/* one call = max_iter cycles from the initial guess x; as a preconditioner
 * max_iter 1 / tol 0 (:154-155) = exactly one V-cycle, no norms */
HYPRE_Int HYPRE_BoomerAMGSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x); /* :324 */
HYPRE_Int HYPRE_BoomerAMGSetPrintLevel(HYPRE_Solver solver, HYPRE_Int print_level);     /* :123 */
HYPRE_Int HYPRE_BoomerAMGSetDebugFlag(HYPRE_Solver solver, HYPRE_Int debug_flag);       /* :124 */
HYPRE_Int HYPRE_BoomerAMGSetCoarsenType(HYPRE_Solver solver, HYPRE_Int coarsen_type);   /* :125; 8 PMIS, 10 HMIS / 11 (one-pass RS), 6 Falgout / 1 / 3 (two-pass RS); any other type is refused at Setup */
HYPRE_Int HYPRE_BoomerAMGSetCycleType(HYPRE_Solver solver, HYPRE_Int cycle_type);       /* :127; 1 V, 2 W */
HYPRE_Int HYPRE_BoomerAMGSetRelaxType(HYPRE_Solver solver, HYPRE_Int relax_type);       /* :138; sets down/up, coarsest = 9 */
HYPRE_Int HYPRE_BoomerAMGSetCycleRelaxType(HYPRE_Solver solver, HYPRE_Int relax_type, HYPRE_Int k); /* :131-136; k 1 down 2 up 3 coarsest */
HYPRE_Int HYPRE_BoomerAMGSetNumSweeps(HYPRE_Solver solver, HYPRE_Int num_sweeps);       /* :150; down/up = n, coarsest = 1 */
HYPRE_Int HYPRE_BoomerAMGSetCycleNumSweeps(HYPRE_Solver solver, HYPRE_Int num_sweeps, HYPRE_Int k); /* :143-148 */
HYPRE_Int HYPRE_BoomerAMGSetSmoothNumSweeps(HYPRE_Solver solver, HYPRE_Int n);          /* :152 (stored, unused) */
HYPRE_Int HYPRE_BoomerAMGSetTol(HYPRE_Solver solver, HYPRE_Real tol);                   /* :154 */
HYPRE_Int HYPRE_BoomerAMGSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);           /* :155 */
HYPRE_Int HYPRE_BoomerAMGSetRelaxOrder(HYPRE_Solver solver, HYPRE_Int relax_order);     /* :156; 1 = C/F */
HYPRE_Int HYPRE_BoomerAMGSetMaxLevels(HYPRE_Solver solver, HYPRE_Int max_levels);       /* :157 */
HYPRE_Int HYPRE_BoomerAMGSetStrongThreshold(HYPRE_Solver solver, HYPRE_Real theta);     /* :158 */
HYPRE_Int HYPRE_BoomerAMGSetMaxRowSum(HYPRE_Solver solver, HYPRE_Real max_row_sum);
HYPRE_Int HYPRE_BoomerAMGSetInterpType(HYPRE_Solver solver, HYPRE_Int interp_type);     /* :196; 6 ext+i, 3 direct, 0 classical */
HYPRE_Int HYPRE_BoomerAMGSetTruncFactor(HYPRE_Solver solver, HYPRE_Real trunc_factor);  /* :231 */
HYPRE_Int HYPRE_BoomerAMGSetPMaxElmts(HYPRE_Solver solver, HYPRE_Int pmax);
HYPRE_Int HYPRE_BoomerAMGSetMinCoarseSize(HYPRE_Solver solver, HYPRE_Int n);            /* :201 */
HYPRE_Int HYPRE_BoomerAMGSetMaxCoarseSize(HYPRE_Solver solver, HYPRE_Int n);            /* :206 */
/* levels with at most this many (global) rows are kept whole on every rank and cycled redundantly, without
 * halo exchanges (HYPRE: the coarse problem is solved sequentially below seq_threshold).  0 = off; the
 * library default on more than one rank is 200000 (MI_HYPRE_REDUNDANT_ROWS). */
HYPRE_Int HYPRE_BoomerAMGSetSeqThreshold(HYPRE_Solver solver, HYPRE_Int seq_threshold);
HYPRE_Int HYPRE_BoomerAMGSetRelaxWt(HYPRE_Solver solver, HYPRE_Real w);
HYPRE_Int HYPRE_BoomerAMGSetOuterWt(HYPRE_Solver solver, HYPRE_Real w);
HYPRE_Int HYPRE_BoomerAMGGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_BoomerAMGGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *rel_resid_norm);
/* aggressive coarsening on levels < n: second coarsening on the second-generation strength graph (A2),
 * multipass interpolation (agg_interp_type 4; any other type is refused at Setup) with its own truncation */
HYPRE_Int HYPRE_BoomerAMGSetAggNumLevels(HYPRE_Solver solver, HYPRE_Int n);             /* :216 */
HYPRE_Int HYPRE_BoomerAMGSetAggInterpType(HYPRE_Solver solver, HYPRE_Int t);            /* :221 */
HYPRE_Int HYPRE_BoomerAMGSetAggPMaxElmts(HYPRE_Solver solver, HYPRE_Int n);             /* :212, :226 */
HYPRE_Int HYPRE_BoomerAMGSetAggTruncFactor(HYPRE_Solver solver, HYPRE_Real f);
/* accepted; settings that would change the result are reported once on stderr and ignored (:161-321) */
HYPRE_Int HYPRE_BoomerAMGSetKeepTranspose(HYPRE_Solver solver, HYPRE_Int keep);         /* :191 (R = P^T is always stored) */
HYPRE_Int HYPRE_BoomerAMGSetRAP2(HYPRE_Solver solver, HYPRE_Int rap2);                  /* :186 (always R*(A*P)) */
HYPRE_Int HYPRE_BoomerAMGSetVariant(HYPRE_Solver solver, HYPRE_Int variant);            /* :181 */
HYPRE_Int HYPRE_BoomerAMGSetNonGalerkinTol(HYPRE_Solver solver, HYPRE_Real tol);        /* :163 */
HYPRE_Int HYPRE_BoomerAMGSetLevelNonGalerkinTol(HYPRE_Solver solver, HYPRE_Real tol, HYPRE_Int level); /* :175 */
HYPRE_Int HYPRE_BoomerAMGSetSmoothType(HYPRE_Solver solver, HYPRE_Int t);               /* :238 */
HYPRE_Int HYPRE_BoomerAMGSetSmoothNumLevels(HYPRE_Solver solver, HYPRE_Int n);          /* :246 */
HYPRE_Int HYPRE_BoomerAMGSetILUType(HYPRE_Solver solver, HYPRE_Int v);                  /* :252 */
HYPRE_Int HYPRE_BoomerAMGSetILULevel(HYPRE_Solver solver, HYPRE_Int v);                 /* :256 */
HYPRE_Int HYPRE_BoomerAMGSetILULocalReordering(HYPRE_Solver solver, HYPRE_Int v);       /* :260 */
HYPRE_Int HYPRE_BoomerAMGSetILUMaxRowNnz(HYPRE_Solver solver, HYPRE_Int v);             /* :264 */
HYPRE_Int HYPRE_BoomerAMGSetILUMaxIter(HYPRE_Solver solver, HYPRE_Int v);               /* :268 */
HYPRE_Int HYPRE_BoomerAMGSetILUDroptol(HYPRE_Solver solver, HYPRE_Real v);              /* :272 */
HYPRE_Int HYPRE_BoomerAMGSetILUIterSetupType(HYPRE_Solver solver, HYPRE_Int v);         /* :284 */
HYPRE_Int HYPRE_BoomerAMGSetILUIterSetupOption(HYPRE_Solver solver, HYPRE_Int v);       /* :289 */
HYPRE_Int HYPRE_BoomerAMGSetILUIterSetupMaxIter(HYPRE_Solver solver, HYPRE_Int v);      /* :294 */
HYPRE_Int HYPRE_BoomerAMGSetILUIterSetupTolerance(HYPRE_Solver solver, HYPRE_Real v);   /* :300 */
HYPRE_Int HYPRE_BoomerAMGSetILUTriSolve(HYPRE_Solver solver, HYPRE_Int v);              /* :311 */
HYPRE_Int HYPRE_BoomerAMGSetILULowerJacobiIters(HYPRE_Solver solver, HYPRE_Int v);      /* :316 */
HYPRE_Int HYPRE_BoomerAMGSetILUUpperJacobiIters(HYPRE_Solver solver, HYPRE_Int v);      /* :320 */

/* ---------------------------------------------------------------- ParCSR GMRES
 * src/HypreSystem.cpp:390-404; right-preconditioned restarted GMRES(k), MGS */
HYPRE_Int HYPRE_ParCSRGMRESCreate(MPI_Comm comm, HYPRE_Solver *solver);                 /* :392 */
HYPRE_Int HYPRE_ParCSRGMRESDestroy(HYPRE_Solver solver);                                /* :400 */
HYPRE_Int HYPRE_ParCSRGMRESSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x); /* :401, called :692 */
HYPRE_Int HYPRE_ParCSRGMRESSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x); /* :403, called :723 */
HYPRE_Int HYPRE_ParCSRGMRESSetPrecond(HYPRE_Solver solver, HYPRE_PtrToParSolverFcn precond,
                                      HYPRE_PtrToParSolverFcn precond_setup, HYPRE_Solver precond_solver); /* :402, called :687 */
HYPRE_Int HYPRE_ParCSRGMRESSetTol(HYPRE_Solver solver, HYPRE_Real tol);                 /* :393 */
HYPRE_Int HYPRE_ParCSRGMRESSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_ParCSRGMRESSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);         /* :394 */
HYPRE_Int HYPRE_ParCSRGMRESSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_ParCSRGMRESSetKDim(HYPRE_Solver solver, HYPRE_Int k_dim);               /* :396 */
HYPRE_Int HYPRE_ParCSRGMRESSetPrintLevel(HYPRE_Solver solver, HYPRE_Int print_level);   /* :397 */
HYPRE_Int HYPRE_ParCSRGMRESSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
HYPRE_Int HYPRE_ParCSRGMRESSetCGS(HYPRE_Solver solver, HYPRE_Int cgs);                  /* :398 (commented out in the driver) */
/* never called by the driver; the harness needs them for iterations/s */
HYPRE_Int HYPRE_ParCSRGMRESGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_ParCSRGMRESGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);

/* ---------------------------------------------------------------- ParCSR BiCGSTAB
 * src/HypreSystem.cpp:423-438 */
HYPRE_Int HYPRE_ParCSRBiCGSTABCreate(MPI_Comm comm, HYPRE_Solver *solver);
HYPRE_Int HYPRE_ParCSRBiCGSTABDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRBiCGSTABSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetPrecond(HYPRE_Solver solver, HYPRE_PtrToParSolverFcn precond,
                                         HYPRE_PtrToParSolverFcn precond_setup, HYPRE_Solver precond_solver);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetKDim(HYPRE_Solver solver, HYPRE_Int k_dim);            /* :431 (commented out) */
HYPRE_Int HYPRE_ParCSRBiCGSTABSetPrintLevel(HYPRE_Solver solver, HYPRE_Int print_level);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
HYPRE_Int HYPRE_ParCSRBiCGSTABGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_ParCSRBiCGSTABGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);

/* ---------------------------------------------------------------- further Krylov families;
 * all implemented; HYPRE_ILU further down covers block-Jacobi ILU(k) */
#define MI_HYPRE_DECLARE_KRYLOV_FAMILY(NAME)                                                                           \
  HYPRE_Int HYPRE_ParCSR##NAME##Create(MPI_Comm comm, HYPRE_Solver *solver);                                         \
  HYPRE_Int HYPRE_ParCSR##NAME##Destroy(HYPRE_Solver solver);                                                        \
  HYPRE_Int HYPRE_ParCSR##NAME##Setup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x); \
  HYPRE_Int HYPRE_ParCSR##NAME##Solve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x); \
  HYPRE_Int HYPRE_ParCSR##NAME##SetPrecond(HYPRE_Solver solver, HYPRE_PtrToParSolverFcn precond,                     \
                                           HYPRE_PtrToParSolverFcn precond_setup, HYPRE_Solver precond_solver);      \
  HYPRE_Int HYPRE_ParCSR##NAME##SetTol(HYPRE_Solver solver, HYPRE_Real tol);                                         \
  HYPRE_Int HYPRE_ParCSR##NAME##SetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);                                 \
  HYPRE_Int HYPRE_ParCSR##NAME##SetKDim(HYPRE_Solver solver, HYPRE_Int k_dim);                                       \
  HYPRE_Int HYPRE_ParCSR##NAME##SetPrintLevel(HYPRE_Solver solver, HYPRE_Int print_level);
/* implemented: FlexGMRES = GMRES that keeps z_j = M^-1 p_j (krylov/flexgmres.c),
 * PCG = preconditioned conjugate gradients (krylov/pcg.c), COGMRES = the GMRES skeleton with classical
 * Gram-Schmidt in block form: one block of inner products (one all-reduce) + one block update per pass
 * (krylov/cogmres.c); SetCGS(cgs): cgs <= 1 one pass, cgs >= 2 two passes */
MI_HYPRE_DECLARE_KRYLOV_FAMILY(COGMRES)  /* src/HypreSystem.cpp:372-388 */
MI_HYPRE_DECLARE_KRYLOV_FAMILY(FlexGMRES) /* :406-421 */
MI_HYPRE_DECLARE_KRYLOV_FAMILY(PCG)      /* :440-455 */
HYPRE_Int HYPRE_ParCSRFlexGMRESSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_ParCSRFlexGMRESSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_ParCSRFlexGMRESSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
HYPRE_Int HYPRE_ParCSRFlexGMRESGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_ParCSRFlexGMRESGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
HYPRE_Int HYPRE_ParCSRPCGSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_ParCSRPCGSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_ParCSRPCGSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
HYPRE_Int HYPRE_ParCSRPCGSetTwoNorm(HYPRE_Solver solver, HYPRE_Int two_norm);
HYPRE_Int HYPRE_ParCSRPCGGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_ParCSRPCGGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
HYPRE_Int HYPRE_ParCSRCOGMRESSetCGS(HYPRE_Solver solver, HYPRE_Int cgs);                /* :382 */
HYPRE_Int HYPRE_ParCSRCOGMRESSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_ParCSRCOGMRESSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_ParCSRCOGMRESSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
HYPRE_Int HYPRE_ParCSRCOGMRESGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_ParCSRCOGMRESGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);

/* ILU (src/HypreSystem.cpp:328-370, :457-497): type 0 (block Jacobi) with level of fill k >= 0 -- ILU(k) of the rank's
 * diagonal block, exact (level-scheduled) or Jacobi-iterated triangular solves; other types / fill levels and the
 * iterative setup report an error at Setup */
HYPRE_Int HYPRE_ILUCreate(HYPRE_Solver *solver);
HYPRE_Int HYPRE_ILUGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_ILUGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
HYPRE_Int HYPRE_ILUDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_ILUSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ILUSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ILUSetType(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetMaxIter(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetTol(HYPRE_Solver solver, HYPRE_Real v);
HYPRE_Int HYPRE_ILUSetLocalReordering(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetPrintLevel(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetLevelOfFill(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetMaxNnzPerRow(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetDropThreshold(HYPRE_Solver solver, HYPRE_Real v);
HYPRE_Int HYPRE_ILUSetIterativeSetupType(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetIterativeSetupOption(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetIterativeSetupMaxIter(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetIterativeSetupTolerance(HYPRE_Solver solver, HYPRE_Real v);
HYPRE_Int HYPRE_ILUSetTriSolve(HYPRE_Solver solver, HYPRE_Int v);                       /* also called on an AMG handle, :306 */
HYPRE_Int HYPRE_ILUSetLowerJacobiIters(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_ILUSetUpperJacobiIters(HYPRE_Solver solver, HYPRE_Int v);

#ifdef __cplusplus
}
#endif
#endif
