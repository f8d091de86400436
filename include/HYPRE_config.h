/* Build configuration of the MI355X-native HYPRE-shaped library.
 * Mirrors the macros the reference driver keys on
 * (/root/reference/src/HypreSystem.h:174-227, src/main.cpp:59-156). */
#ifndef HYPRE_CONFIG_H
#define HYPRE_CONFIG_H
#define HYPRE_RELEASE_NAME "mi_hypre"
#define HYPRE_RELEASE_VERSION "0.1.0-gfx950"
#define HYPRE_MIXEDINT 1      /* HYPRE_Int = int32, HYPRE_BigInt = int64 (etc/build_script_tmpl.sh:20 asks for bigint ids) */
#define HYPRE_USING_GPU 1
#define HYPRE_USING_HIP 1
#define HYPRE_USING_DEVICE_MEMORY 1
#endif
