/* HYPRE_mi_ext.h -- entry points that HYPRE does not have: rank/GPU binding over
 * RCCL, hierarchy inspection for the parity tests, HIP-event kernel timing for
 * bench.py's roofline figure, and the synthetic problem generator. */
#ifndef HYPRE_MI_EXT_HEADER
#define HYPRE_MI_EXT_HEADER
#include "HYPRE_parcsr_ls.h"
#include "HYPRE_IJ_mv.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- communicator (one rank per GPU).  Replaces the MPI_Comm that libHYPRE
 * takes from src/main.cpp:33-35 / src/HypreSystem.cpp:12-13. */
#define HYPRE_MI_UNIQUE_ID_BYTES 128
HYPRE_Int HYPRE_MI_CommGetUniqueId(void *id128);               /* rank 0, then broadcast by the launcher */
HYPRE_Int HYPRE_MI_CommInitRCCL(const void *id128, HYPRE_Int rank, HYPRE_Int size);
/* RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env); the id travels over TCP */
HYPRE_Int HYPRE_MI_CommInitFromEnv(void);
/* host-staged transport supplied by the caller (tests: torch.distributed gloo) */
typedef void (*HYPRE_MI_AllreduceFn)(void *ctx, void *buf, size_t count, int dtype /*0 f64,1 i64,2 i32,3 u8*/,
                                     int op /*0 sum,1 min,2 max*/);
typedef void (*HYPRE_MI_AllgatherFn)(void *ctx, const void *send, void *recv, size_t bytes_per_rank);
typedef void (*HYPRE_MI_ExchangeFn)(void *ctx, int nsend, const int *send_peers, void *const *send_ptrs,
                                    const size_t *send_bytes, int nrecv, const int *recv_peers, void *const *recv_ptrs,
                                    const size_t *recv_bytes);
HYPRE_Int HYPRE_MI_CommInitCallbacks(void *ctx, HYPRE_MI_AllreduceFn ar, HYPRE_MI_AllgatherFn ag,
                                     HYPRE_MI_ExchangeFn ex, HYPRE_Int rank, HYPRE_Int size);
/* Neighbour exchange (the halo updates) by peer stores instead of ncclSend/ncclRecv groups, on top of the
 * communicator set up before (RCCL or callbacks; it keeps the reductions and gathers): every rank exports a mailbox
 * arena with hipIpcGetMemHandle, maps the others' (xGMI peer memory; the same device when ranks share a GPU), and one
 * kernel launch per exchange stores the packed halo into the receiver's slot, publishes a sequence number and takes
 * the incoming messages out of its own slots.  slot_bytes: mailbox slot per directed pair and parity (0 = env
 * MI_HYPRE_IPC_SLOT_BYTES or 4 MiB; larger messages travel in parts).  Collective.  Waits are bounded
 * (MI_HYPRE_IPC_TIMEOUT_MS, default 20 000): HYPRE_MI_CommCheck reports a message that never arrived. */
HYPRE_Int HYPRE_MI_CommEnablePeerStoreExchange(HYPRE_BigInt slot_bytes);
HYPRE_Int HYPRE_MI_CommCheck(void);
/* sum all-reduce of `count` doubles in DEVICE memory through the current transport (what an inner product of the
 * Krylov loops does; completed on return).  On the peer-store transport up to 8 doubles take one launch: every rank
 * stores its values into every peer's mailbox and adds the contributions in rank order (identical bits everywhere). */
HYPRE_Int HYPRE_MI_CommAllreduceDevice(HYPRE_Real *dev_buf, HYPRE_Int count);
/* one neighbour exchange of DEVICE buffers through the current transport, on the library stream, completed on return
 * (what a halo update does; for transport tests) */
HYPRE_Int HYPRE_MI_CommExchangeDevice(HYPRE_Int nsend, const HYPRE_Int *send_peers, void *const *send_ptrs,
                                      const size_t *send_bytes, HYPRE_Int nrecv, const HYPRE_Int *recv_peers,
                                      void *const *recv_ptrs, const size_t *recv_bytes);
HYPRE_Int HYPRE_MI_CommName(char *name, HYPRE_Int max_len);
/* one-rank exercise of the RCCL transport (dlopen, init, all-reduce, all-gather, send/recv) */
HYPRE_Int HYPRE_MI_CommSelfTestRCCL(void);
HYPRE_Int HYPRE_MI_CommFinalize(void);
HYPRE_Int HYPRE_MI_CommRank(HYPRE_Int *rank);
HYPRE_Int HYPRE_MI_CommSize(HYPRE_Int *size);
HYPRE_Int HYPRE_MI_CommBarrier(void);
/* blocking all-reduce of a HOST buffer (loaders: src/HypreSystem.cpp:1166-1167, :829) */
HYPRE_Int HYPRE_MI_CommAllreduce(void *buf, size_t count, int dtype, int op);

/* ---- host-only halves of Assemble / Setup (no device needed): the N>1 host logic
 * -- row partition, halo plan, rank-local coarsening, P-row exchange, Galerkin
 * product -- is exercised by world_size-2 gloo tests on CPU through these. */
HYPRE_Int HYPRE_MI_IJMatrixAssembleHostOnly(HYPRE_IJMatrix matrix);
HYPRE_Int HYPRE_MI_BoomerAMGSetupHostOnly(HYPRE_Solver solver, HYPRE_ParCSRMatrix A);
/* arrays may be NULL to query the counts; send_starts/recv_starts hold npeers+1 offsets */
HYPRE_Int HYPRE_MI_ParCSRGetHaloPlan(HYPRE_ParCSRMatrix A, HYPRE_Int *nsend_peers, HYPRE_Int *send_peers,
                                     HYPRE_Int *send_starts, HYPRE_Int *send_map, HYPRE_Int *nrecv_peers,
                                     HYPRE_Int *recv_peers, HYPRE_Int *recv_starts);

/* ---- device / stream */
HYPRE_Int HYPRE_MI_GetStream(void **hip_stream);
HYPRE_Int HYPRE_MI_StreamSynchronize(void);
HYPRE_Int HYPRE_MI_SetGSChunk(HYPRE_Int rows_per_chunk);   /* hybrid-GS "thread" size, default 8 */
HYPRE_Int HYPRE_MI_GetGSChunk(HYPRE_Int *rows_per_chunk);
/* How the first relaxation sweep of a cycle's down leg (zero guess on every level) is run: 0 like any other
 * sweep, 1 without gathering the known zeros, 2 also on the level's zero-guess sub-operator, built by
 * BoomerAMGSetup: the rows' in-chunk entries plus the F rows' C columns -- everything else multiplies zeros; 3
 * (default, env MI_HYPRE_GS_ZERO_SKIP) the residual that follows also reuses the F pass's product with the C values
 * and reads the F rows without their C columns.  Same result up to summation order. */
HYPRE_Int HYPRE_MI_SetZeroGuessMode(HYPRE_Int mode);
/* test hook: fill every level's solution / scratch vectors with NaN (a zero-guess cycle that skips its zero-fills must
 * not read them) */
HYPRE_Int HYPRE_MI_BoomerAMGPoisonWorkVectors(HYPRE_Solver solver);
/* test hook: the tile schedule of an operator with the given row pointers (host, n + 1 entries), built by the device
 * routine of the setup (bisection per tile) and by the host routine (the plain greedy loop); *mismatch = 0 when the two
 * agree tile by tile and in the chunk-alignment flag, else 1 + the index of the first tile that differs; *ntiles = the
 * host's tile count.  row_cap: 256 (operators a Gauss-Seidel kernel sweeps) ... 1024 (SpMV only); tile_entries: 2048 / 4096 */
HYPRE_Int HYPRE_MI_TileScheduleCheck(HYPRE_Int n, const HYPRE_BigInt *row_ptr, HYPRE_Int row_cap, HYPRE_Int tile_entries,
                                     HYPRE_Int *ntiles, HYPRE_Int *mismatch);
/* test hook: a seeded storm of device allocations / releases of every size through the library's allocator (arena, block
 * cache or plain, whatever MI_HYPRE_POOL says), every block pattern-filled and checked before its release */
HYPRE_Int HYPRE_MI_ArenaSelfTest(HYPRE_Int seed, HYPRE_Int rounds, HYPRE_BigInt max_block_bytes, HYPRE_BigInt *verified,
                                 HYPRE_BigInt *peak_bytes);
/* Value dictionary of operators with at most 256 distinct values (constant-coefficient stencils such as the
 * reference generator's 26 / -1, /root/reference/src/laplace_3d_weak_scaling.hpp:558,600): one byte per entry instead
 * of the 8-byte value in the matrix stream; same doubles, same results.  on = 0 keeps the plain stream -- what a
 * general (variable-coefficient) operator gets anyway; applies to hierarchies set up afterwards (env
 * MI_HYPRE_VALUE_DICT).  bench.py uses it to report the general-operator roofline beside the headline. */
HYPRE_Int HYPRE_MI_SetValueDictionary(HYPRE_Int on);
/* Counters of the multi-rank choreography since the library was loaded: "matvec_overlapped" (SpMVs whose
 * neighbour exchange ran beside the diag-block product), "gs_overlapped" / "gs_in_order" (relaxation passes that
 * swept their halo-free rows while the halo travelled / that waited for it first); collectives of the solve phase
 * on N > 1 ranks: "allreduce" (one per inner product; modified Gram-Schmidt needs i + 1 of them at Arnoldi step i --
 * each coefficient depends on the previous update -- the block classical Gram-Schmidt of COGMRES 2 per step),
 * "halo_exchange" (neighbour send/recv groups), "allgather" (coarsest / redundant levels); the distributed setup:
 * "setup_distributed" (count), "setup_ext_rows_max" (largest per-rank extended sub-problem, rows),
 * "setup_global_rows_gathered" (rows gathered on every rank: the redundant tail only), "setup_device_levels" (levels
 * of the distributed setup whose per-rank pieces were built on the device). */
HYPRE_Int HYPRE_MI_GetCounter(const char *name, long long *value);
/* One rank's block keeps 32-bit local row ids in the solve format; the entry offsets of its diagonal block are 64-bit
 * (the reference's 27-point operator at 512^3 -- 3.6e9 entries, /root/reference/src/laplace_3d_weak_scaling.hpp:558,600
 * -- assembles and solves on ONE rank, as with HYPRE's bigint / mixedint builds, /root/reference/src/HypreSystem.h:174-219).
 * HYPRE_IJMatrixAssemble refuses (HYPRE_ERROR_ARG + message, never a wrap-around) >= 2147483000 local rows, a halo
 * block or a single row with that many entries.  This applies the same check to a row-pointer array. */
HYPRE_Int HYPRE_MI_CheckBlockRowPointers(HYPRE_BigInt nrows, const HYPRE_BigInt *row_ptr);

/* ---- results the driver never asks HYPRE for */
HYPRE_Int HYPRE_MI_KrylovGetResidualHistory(HYPRE_Solver solver, HYPRE_Real *norms, HYPRE_Int max_n, HYPRE_Int *n);
HYPRE_Int HYPRE_MI_KrylovGetSolveSeconds(HYPRE_Solver solver, HYPRE_Real *seconds);

/* ---- the setup phase's device sparse kernels on caller (host) CSR arrays: op 0 C = A*B (B rows sorted),
 * 1 C = A^T, 2 C = rows of A in perm order (perm[new] = old, NULL = identity) with columns mapped through
 * colpos (NULL = identity) and re-sorted.  Bit-identical to the host/oracle arithmetic.  Results are
 * malloc'ed: release with HYPRE_MI_Free. */
HYPRE_Int HYPRE_MI_CSRDeviceOp(HYPRE_Int op, HYPRE_Int a_nrows, HYPRE_Int a_ncols, const HYPRE_BigInt *a_ia,
                               const HYPRE_Int *a_ja, const HYPRE_Complex *a_a, HYPRE_Int b_nrows, HYPRE_Int b_ncols,
                               const HYPRE_BigInt *b_ia, const HYPRE_Int *b_ja, const HYPRE_Complex *b_a,
                               const HYPRE_Int *perm, const HYPRE_Int *colpos, HYPRE_Int *c_nrows, HYPRE_Int *c_ncols,
                               HYPRE_BigInt **c_ia, HYPRE_Int **c_ja, HYPRE_Complex **c_a);

/* ---- hierarchy inspection (parity tests; mirrors hypre_ParAMGDataAArray) */
HYPRE_Int HYPRE_MI_BoomerAMGGetNumLevels(HYPRE_Solver solver, HYPRE_Int *num_levels);
HYPRE_Int HYPRE_MI_BoomerAMGGetOperatorComplexity(HYPRE_Solver solver, HYPRE_Real *cx);
HYPRE_Int HYPRE_MI_BoomerAMGGetSetupSeconds(HYPRE_Solver solver, HYPRE_Real *seconds);
/* which: 0 A diag block, 1 A offd block, 2 P diag, 3 R diag, 4 P offd (halo columns), 5 R offd.
 * P (rows: this level, columns: next level) and R = P^T are rectangular ParCSR operators.
 * GetLevelCSRSize also takes which = 6: the level's zero-guess sub-operator (the entries of the diag block a
 * first sweep on a zero guess can meet with a non-zero; 0 x 0 when the level has none), and which = 7: the x cache
 * of the level's diag block (nrows = number of tiles, nnz = unique columns summed over the tiles); which = 8: the
 * operator of the residual after a zero-guess sweep (0 x 0 when the level has none); which = 9: the C rows of the diag
 * block (nrows = number of C points, nnz = their entries: what a C pass of the relaxation streams). */
HYPRE_Int HYPRE_MI_BoomerAMGGetLevelCSRSize(HYPRE_Solver solver, HYPRE_Int level, HYPRE_Int which, HYPRE_Int *nrows,
                                            HYPRE_Int *ncols, HYPRE_BigInt *nnz);
HYPRE_Int HYPRE_MI_BoomerAMGGetLevelCSR(HYPRE_Solver solver, HYPRE_Int level, HYPRE_Int which, HYPRE_BigInt *ia,
                                        HYPRE_Int *ja, HYPRE_Complex *a);
HYPRE_Int HYPRE_MI_BoomerAMGGetLevelCF(HYPRE_Solver solver, HYPRE_Int level, HYPRE_Int *cf);
/* the level's C-first ordering: perm[new local row] = old local row (level matrices, P, R and the
 * C/F marker are reported in the NEW ordering; level 0 is a renumbered copy of the caller's matrix) */
HYPRE_Int HYPRE_MI_BoomerAMGGetLevelPerm(HYPRE_Solver solver, HYPRE_Int level, HYPRE_Int *perm);
/* Internal locality numbering of the input (one rank, large systems; MI_HYPRE_LOCALITY_ORDER = 0 off / 1 on, unset:
 * at least MI_HYPRE_LOCALITY_MIN_ROWS = 1e6 rows): the hierarchy is built on Q A Q^T where Q groups rows into
 * graph-compact clusters (graph Voronoi cells of ~512 rows, ranked in sweep order) so that tiles of consecutive rows gather few distinct
 * columns.  order[new] = caller's local row (identity and *applied = 0 when not in use); GetLevelPerm(0) already
 * includes it (level-0 row -> caller row). */
HYPRE_Int HYPRE_MI_BoomerAMGGetInputOrdering(HYPRE_Solver solver, HYPRE_Int *applied, HYPRE_Int *order);
HYPRE_Int HYPRE_MI_BoomerAMGGetLevelColMap(HYPRE_Solver solver, HYPRE_Int level, HYPRE_BigInt *col_map_offd,
                                           HYPRE_BigInt *row_start);
/* sorted global column ids of an offd block (which: 1 A, 4 P, 5 R; length = that block's ncols) */
HYPRE_Int HYPRE_MI_BoomerAMGGetLevelOffdColMap(HYPRE_Solver solver, HYPRE_Int level, HYPRE_Int which,
                                               HYPRE_BigInt *col_map_offd);
/* one relaxation call / one cycle on HOST arrays of the level's local length */
HYPRE_Int HYPRE_MI_BoomerAMGRelaxLevel(HYPRE_Solver solver, HYPRE_Int level, HYPRE_Int relax_type, HYPRE_Int points,
                                       const HYPRE_Real *f_host, HYPRE_Real *u_host);

/* ---- HIP-event timing of kernel classes on the library stream.
 * id: 0 level-0 SpMV of the Krylov loop, 1 level-0 relaxation, 2 dot, 3 axpy; per AMG level l < 16:
 * 4 + l residual SpMV of the cycle, 20 + l relaxation passes, 36 + l restriction, 52 + l prolongation */
HYPRE_Int HYPRE_MI_ProfileEnable(HYPRE_Int id, HYPRE_Int capacity);
HYPRE_Int HYPRE_MI_ProfileReset(void);
/* instantiation (template flags included, as rocprofv3's kernel statistics spell it) of the kernel last launched
 * under the class; empty when none was */
HYPRE_Int HYPRE_MI_ProfileKernelName(HYPRE_Int id, char *name, HYPRE_Int max_len);
HYPRE_Int HYPRE_MI_ProfileGet(HYPRE_Int id, long long *launches, double *total_ms, double *min_ms);

/* ---- synthetic problem: n^3-type Laplacian, true lexicographic global numbering
 * (7-point: diag 6 / off -1; 27-point: diag 26 / off -1 as
 * src/laplace_3d_weak_scaling.hpp:558,600; rhs = row sum, :321).  Fills COO
 * triples for global rows [ilower, iupper]; buffers come from malloc and are
 * released with HYPRE_MI_Free. */
HYPRE_Int HYPRE_MI_Laplace3D(HYPRE_Int nx, HYPRE_Int ny, HYPRE_Int nz, HYPRE_Int stencil, HYPRE_BigInt ilower,
                             HYPRE_BigInt iupper, HYPRE_BigInt *nnz, HYPRE_BigInt **rows, HYPRE_BigInt **cols,
                             HYPRE_Complex **vals, HYPRE_Complex **rhs);
void HYPRE_MI_Free(void *p);

#ifdef __cplusplus
}
#endif
#endif
