/* HYPRE_utilities.h -- types, init/finalize, error flag, memory helpers.
 *
 * C ABI of the MI355X-native replacement for the part of libHYPRE that
 * Exawind/hypre-mini-app drives.  Every entry point cites the reference line
 * that calls it (paths relative to /root/reference).  Plain pointers and
 * sizes only; all functions return HYPRE_Int (0 = success) and accumulate into
 * a global error flag exactly like HYPRE. */
#ifndef HYPRE_UTILITIES_HEADER
#define HYPRE_UTILITIES_HEADER

#include "HYPRE_config.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int HYPRE_Int;             /* src/HypreSystem.h:196-219 */
typedef long long HYPRE_BigInt;    /* src/HypreSystem.h:174-195 */
typedef double HYPRE_Real;
typedef double HYPRE_Complex;      /* src/HypreSystem.h:223-227 */

/* MPI_Comm: the library never calls MPI.  Ranks are bound to GPUs by
 * HYPRE_MI_Comm* (HYPRE_mi_ext.h); the comm argument of the Create calls is
 * accepted and ignored, as in HYPRE's own --without-MPI build. */
#if !defined(MPI_VERSION) && !defined(MI_HYPRE_HAVE_MPI_COMM)
#define MI_HYPRE_HAVE_MPI_COMM 1
typedef int MPI_Comm;
#ifndef MPI_COMM_WORLD
#define MPI_COMM_WORLD 0
#endif
#endif

#define HYPRE_ERROR_GENERIC 1
#define HYPRE_ERROR_MEMORY 2
#define HYPRE_ERROR_ARG 4
#define HYPRE_ERROR_CONV 256

typedef enum { HYPRE_MEMORY_UNDEFINED = -1, HYPRE_MEMORY_HOST = 0, HYPRE_MEMORY_DEVICE = 1 } HYPRE_MemoryLocation;
typedef enum { HYPRE_EXEC_UNDEFINED = -1, HYPRE_EXEC_HOST = 0, HYPRE_EXEC_DEVICE = 1 } HYPRE_ExecutionPolicy;

/* src/main.cpp:82 (after hipSetDevice, :65) / :218.  Fails when no HIP device is
 * present: there is no CPU path. */
HYPRE_Int HYPRE_Initialize(void);
HYPRE_Int HYPRE_Init(void);
HYPRE_Int HYPRE_Finalize(void);
HYPRE_Int HYPRE_Initialized(void);

HYPRE_Int HYPRE_GetError(void);
HYPRE_Int HYPRE_ClearAllErrors(void);
/* HYPRE's own error text call (utilities/error.c): which flags `errorcode` holds, written into descr (the caller
 * provides >= 256 bytes); this library appends the message of the last failure */
void HYPRE_DescribeError(HYPRE_Int errorcode, char *descr);
/* text of the last failure (not part of HYPRE; valid until the next call) */
const char *HYPRE_MI_LastErrorMessage(void);

/* src/main.cpp:117-125: only DEVICE is implemented; HOST requests are refused */
HYPRE_Int HYPRE_SetMemoryLocation(HYPRE_MemoryLocation loc);
HYPRE_Int HYPRE_SetExecutionPolicy(HYPRE_ExecutionPolicy pol);
/* accepted and ignored (src/main.cpp:100-114, :127-155, :169) */
HYPRE_Int HYPRE_SetGPUMemoryPoolSize(HYPRE_Int bin_growth, HYPRE_Int min_bin, HYPRE_Int max_bin, size_t max_bytes);
HYPRE_Int hypre_SetCubMemPoolSize(unsigned bin_growth, unsigned min_bin, unsigned max_bin, size_t max_bytes);
HYPRE_Int HYPRE_SetUmpireDevicePoolName(const char *name);
/* src/main.cpp:107-114 (YAML `umpire_device_pool_mbs`): the initial size of the device pool -- this library's device
 * arena maps that many bytes ahead of demand, in the background (after HYPRE_Init) */
HYPRE_Int HYPRE_SetUmpireDevicePoolSize(size_t nbytes);
HYPRE_Int HYPRE_SetSpGemmUseVendor(HYPRE_Int use_vendor);
HYPRE_Int HYPRE_SetSpMVUseVendor(HYPRE_Int use_vendor);
HYPRE_Int HYPRE_SetSpTransUseVendor(HYPRE_Int use_vendor);
HYPRE_Int hypre_ResetDeviceRandGenerator(unsigned long long seed, unsigned long long offset);

/* hypre_TAlloc / hypre_TFree / hypre_TMemcpy as the driver uses them
 * (src/HypreSystem.cpp:516-522, :793-810, :907-926) */
void *hypre_MAlloc(size_t bytes, HYPRE_MemoryLocation loc);
void *hypre_CAlloc(size_t count, size_t elt, HYPRE_MemoryLocation loc);
void hypre_Free(void *ptr, HYPRE_MemoryLocation loc);
void hypre_Memcpy(void *dst, const void *src, size_t bytes, HYPRE_MemoryLocation ldst, HYPRE_MemoryLocation lsrc);
#define hypre_TAlloc(type, count, location) ((type *)hypre_MAlloc((size_t)(sizeof(type) * (count)), location))
#define hypre_CTAlloc(type, count, location) ((type *)hypre_CAlloc((size_t)(count), (size_t)sizeof(type), location))
#define hypre_TFree(ptr, location) (hypre_Free((void *)ptr, location), ptr = NULL)
#define hypre_TMemcpy(dst, src, type, count, locdst, locsrc) \
  (hypre_Memcpy((void *)(dst), (void *)(src), (size_t)(sizeof(type) * (count)), locdst, locsrc))
#define HYPRE_MPI_INT MPI_INT
#define HYPRE_MPI_BIG_INT MPI_LONG_LONG_INT

#ifdef __cplusplus
}
#endif
#endif
