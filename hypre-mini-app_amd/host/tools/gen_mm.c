/* Writes an n^3 7-point Laplacian (diag 6 / off -1, lexicographic rows) with rhs = A*1 and the exact
 * solution as MatrixMarket files -- a stand-in of configurable size for the nalu-wind MatrixMarket dumps
 * the loaders are built for (config 4 of BASELINE.json; /root/reference/src/HypreSystem.cpp:1717-1850).
 *   gen_mm N OUTDIR [var]   ->   OUTDIR/mat.mm  OUTDIR/rhs.mm  OUTDIR/sln.mm
 * var: variable-coefficient diffusion instead -- every cell face carries a conductivity k in [0.5, 1.5] (a smooth
 * field, multiples of 1/256 so that the printed decimals are exact), row = sum of its six face conductivities on the
 * diagonal (boundary faces included: Dirichlet), -k for the neighbours inside; rhs = A*1 = the boundary faces' k.
 * Symmetric M-matrix with 257 distinct off-diagonal values: what the unstructured pressure-Poisson dumps look like
 * to the kernels (no value dictionary applies).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* conductivity of the face between cell (x,y,z) and its +1 neighbour along axis (x may be -1: the low boundary face) */
static double face_k(long x, long y, long z, int axis) {
  const double s = sin(0.11 * (double)x + 0.07 * (double)y + 0.05 * (double)z + 1.3 * (double)axis);
  return 1.0 + floor(128.0 * s + 0.5) / 256.0;
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s N OUTDIR\n", argv[0]);
    return 2;
  }
  const long n = atol(argv[1]);
  const int var = argc > 3 && strcmp(argv[3], "var") == 0;
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
        if (var) {
          const double kzm = face_k(x, y, z - 1, 2), kym = face_k(x, y - 1, z, 1), kxm = face_k(x - 1, y, z, 0);
          const double kxp = face_k(x, y, z, 0), kyp = face_k(x, y, z, 1), kzp = face_k(x, y, z, 2);
          double bnd = 0.0;
          if (z > 0) fprintf(fm, "%ld %ld %.10g\n", r, r - n * n, -kzm); else bnd += kzm;
          if (y > 0) fprintf(fm, "%ld %ld %.10g\n", r, r - n, -kym); else bnd += kym;
          if (x > 0) fprintf(fm, "%ld %ld %.10g\n", r, r - 1, -kxm); else bnd += kxm;
          fprintf(fm, "%ld %ld %.10g\n", r, r, kzm + kym + kxm + kxp + kyp + kzp);
          if (x < n - 1) fprintf(fm, "%ld %ld %.10g\n", r, r + 1, -kxp); else bnd += kxp;
          if (y < n - 1) fprintf(fm, "%ld %ld %.10g\n", r, r + n, -kyp); else bnd += kyp;
          if (z < n - 1) fprintf(fm, "%ld %ld %.10g\n", r, r + n * n, -kzp); else bnd += kzp;
          fprintf(fr, "%.10g\n", bnd);
          fputs("1.0\n", fs);
          continue;
        }
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
