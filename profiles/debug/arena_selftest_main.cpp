// the arena self-test without python: HYPRE_Init + HYPRE_MI_ArenaSelfTest(seed, rounds, max block)
#include <cstdio>
#include <cstdlib>
#include "HYPRE.h"
#include "HYPRE_mi_ext.h"
extern "C" const char *HYPRE_MI_LastErrorMessage(void);
int main(int argc, char **argv) {
  HYPRE_Init();
  HYPRE_BigInt ok = 0, peak = 0;
  const int seed = argc > 1 ? atoi(argv[1]) : 3, rounds = argc > 2 ? atoi(argv[2]) : 500;
  const long long mx = argc > 3 ? atoll(argv[3]) : (1ll << 30);
  int rc = HYPRE_MI_ArenaSelfTest(seed, rounds, mx, &ok, &peak);
  printf("rc %d verified %lld peak %.2f GiB  %s\n", rc, (long long)ok, peak / 1073741824.0, rc ? HYPRE_MI_LastErrorMessage() : "");
  HYPRE_Finalize();
  return rc;
}
