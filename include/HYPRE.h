/* HYPRE.h -- umbrella header (src/HypreSystem.h:18). */
#ifndef HYPRE_HEADER
#define HYPRE_HEADER
#include "HYPRE_utilities.h"
#endif
