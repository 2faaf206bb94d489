"""Largest eigenvalue of the normalised Laplacian on the device.

The reference asks ARPACK on the host for it (``eigsh(laplacian, 1,
which='LM')``, reference gcn/utils.py:158) before rescaling the Laplacian for
the Chebyshev recurrence (:159).  At 10^7..10^8 rows the host cannot hold the
matrix, so here the same quantity comes from a Lanczos iteration whose only
matrix operation is the HIP SpMM on the resident CSR of
``S = D^-1/2 A D^-1/2``: ``L v = v - S v``.  Vectors stay in device memory;
the inner products and the recurrence coefficients are carried in fp64, every
new vector is orthogonalised against all earlier ones (twice), and the small
tridiagonal problem is solved on the host.

The SpMM is fp32, so the result carries about 1e-6 relative error against
ARPACK's fp64 answer; ``chebyshev_polynomials(adj, k, largest_eigval=...)``
accepts it.
"""
import numpy as np
import torch

from . import ops


def largest_laplacian_eigval(s, max_iters=96, tol=1e-7, seed=0):
    """lmax of ``I - S`` for a device ``Csr`` ``s`` (square, symmetric).

    Returns ``(lmax, iterations, residual)``; ``residual`` bounds
    ``|lmax - true eigenvalue|`` for the Krylov space built (Lanczos' own
    estimate ``|beta_m * last component of the Ritz vector|``)."""
    if s.n_rows != s.n_cols:
        raise ValueError("largest_laplacian_eigval needs a square matrix, got %d x %d" % (s.n_rows, s.n_cols))
    n = s.n_rows
    dev = s.val.device
    m = int(min(max_iters, n))
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    basis = torch.empty((m + 1, n), dtype=torch.float32, device=dev)
    v = torch.rand((n,), generator=gen, device=dev, dtype=torch.float32) - 0.5
    v /= torch.linalg.vector_norm(v.double()).float()
    basis[0] = v
    alphas, betas = [], []
    sv = torch.empty((n, 1), dtype=torch.float32, device=dev)
    lmax, resid, it = 0.0, float("inf"), 0
    for j in range(m):
        vj = basis[j]
        ops.spmm(s, vj.view(n, 1), out=sv)                 # S v
        w = vj.double() - sv.view(n).double()              # L v = v - S v
        alpha = float(torch.dot(w, vj.double()))
        alphas.append(alpha)
        q = basis[:j + 1].double()
        for _ in range(2):                                 # full re-orthogonalisation, twice
            w -= q.t().mv(q.mv(w))
        beta = float(torch.linalg.vector_norm(w))
        it = j + 1
        t = np.diag(alphas) + np.diag(betas, 1) + np.diag(betas, -1)
        evals, evecs = np.linalg.eigh(t)
        lmax = float(evals[-1])
        resid = abs(beta * evecs[-1, -1])
        if resid <= tol * max(abs(lmax), 1.0) or beta <= 1e-12 or j + 1 == m:
            break
        betas.append(beta)
        basis[j + 1] = (w / beta).float()
    return lmax, it, float(resid)


def largest_eigval_of_adjacency(adj, **kw):
    """Host convenience: scipy adjacency -> lmax of its normalised Laplacian,
    computed on the device (the value ``chebyshev_polynomials`` obtains from
    ARPACK, reference gcn/utils.py:155-158)."""
    from .sparse import Csr
    from .utils import normalize_adj
    s = Csr.from_scipy(normalize_adj(adj).tocsr(), device="cuda")
    return largest_laplacian_eigval(s, **kw)[0]
