/* Writes an n^3 7-point Laplacian (diag 6 / off -1, lexicographic rows) with rhs = A*1 and the exact
 * solution as MatrixMarket files -- a stand-in of configurable size for the nalu-wind MatrixMarket dumps
 * the loaders are built for (config 4 of BASELINE.json; /root/reference/src/HypreSystem.cpp:1717-1850).
 *   gen_mm N OUTDIR   ->   OUTDIR/mat.mm  OUTDIR/rhs.mm  OUTDIR/sln.mm
 */
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s N OUTDIR\n", argv[0]);
    return 2;
  }
  const long n = atol(argv[1]);
  const long N = n * n * n;
  char path[4096];
  snprintf(path, sizeof(path), "%s/mat.mm", argv[2]);
  FILE *fm = fopen(path, "w");
  snprintf(path, sizeof(path), "%s/rhs.mm", argv[2]);
  FILE *fr = fopen(path, "w");
  snprintf(path, sizeof(path), "%s/sln.mm", argv[2]);
  FILE *fs = fopen(path, "w");
  if (!fm || !fr || !fs) {
    perror("fopen");
    return 1;
  }
  static char bm[1 << 22], br[1 << 20], bs[1 << 20];
  setvbuf(fm, bm, _IOFBF, sizeof(bm));
  setvbuf(fr, br, _IOFBF, sizeof(br));
  setvbuf(fs, bs, _IOFBF, sizeof(bs));
  const long nnz = 7 * N - 6 * n * n;
  fprintf(fm, "%%%%MatrixMarket matrix coordinate real general\n%ld %ld %ld\n", N, N, nnz);
  fprintf(fr, "%%%%MatrixMarket matrix array real general\n%ld 1\n", N);
  fprintf(fs, "%%%%MatrixMarket matrix array real general\n%ld 1\n", N);
  for (long z = 0; z < n; z++)
    for (long y = 0; y < n; y++)
      for (long x = 0; x < n; x++) {
        const long r = x + n * (y + n * z) + 1;
        int nb = 0;
        if (z > 0) fprintf(fm, "%ld %ld -1.0\n", r, r - n * n), nb++;
        if (y > 0) fprintf(fm, "%ld %ld -1.0\n", r, r - n), nb++;
        if (x > 0) fprintf(fm, "%ld %ld -1.0\n", r, r - 1), nb++;
        fprintf(fm, "%ld %ld 6.0\n", r, r);
        if (x < n - 1) fprintf(fm, "%ld %ld -1.0\n", r, r + 1), nb++;
        if (y < n - 1) fprintf(fm, "%ld %ld -1.0\n", r, r + n), nb++;
        if (z < n - 1) fprintf(fm, "%ld %ld -1.0\n", r, r + n * n), nb++;
        fprintf(fr, "%d.0\n", 6 - nb);
        fputs("1.0\n", fs);
      }
  fclose(fm);
  fclose(fr);
  fclose(fs);
  return 0;
}
