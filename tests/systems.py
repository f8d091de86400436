"""Seeded test systems shared by the parity tests and tests/golden/make_golden.py."""
import numpy as np
import scipy.sparse as sp


def convection_diffusion_3d(n, seed=1234, peclet=0.6):
    """Stand-in for the nalu-wind momentum system of BASELINE.json config 5 (no dump exists offline):
    7-point diffusion (diag 6, off -1) plus first-order UPWIND convection with a seeded, spatially varying
    velocity field -- a diagonally dominant, non-symmetric M-matrix on the n^3 grid, lexicographic numbering."""
    rng = np.random.default_rng(seed)
    N = n ** 3
    idx = np.arange(N).reshape(n, n, n)  # [z, y, x]
    vel = peclet * rng.uniform(-1.0, 1.0, size=(3, n, n, n))  # z, y, x components (cell Peclet numbers)
    rows, cols, vals = [], [], []
    diag = np.full((n, n, n), 6.0)
    for axis in range(3):
        v = vel[axis]
        for sgn in (-1, +1):
            src = [slice(None)] * 3
            dst = [slice(None)] * 3
            if sgn < 0:
                src[axis], dst[axis] = slice(1, None), slice(None, -1)   # neighbour at -1
            else:
                src[axis], dst[axis] = slice(None, -1), slice(1, None)   # neighbour at +1
            r = idx[tuple(src)].ravel()
            c = idx[tuple(dst)].ravel()
            vv = v[tuple(src)].ravel()
            # upwind: flow in +axis direction takes from the -1 neighbour
            conv = np.where(sgn < 0, np.maximum(vv, 0.0), np.maximum(-vv, 0.0))
            rows.append(r)
            cols.append(c)
            vals.append(-1.0 - conv)
        diag += np.abs(v)  # the upwind contributions of both directions sum to |v| on the diagonal
    rows.append(idx.ravel())
    cols.append(idx.ravel())
    vals.append(diag.ravel())
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))
    A.sort_indices()
    return A


def three_component_rhs(A, seed=99):
    """Three right-hand sides b_c = A x_c with smooth-plus-random exact solutions (component-major (3, N))."""
    rng = np.random.default_rng(seed)
    N = A.shape[0]
    t = np.linspace(0.0, 1.0, N)
    xs = np.stack([1.0 + 0.0 * t, np.sin(6.0 * t) + 0.1 * rng.standard_normal(N), rng.standard_normal(N)])
    return np.stack([A @ x for x in xs]), xs
