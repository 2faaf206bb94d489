"""Shared helpers of the parity tests."""
import numpy as np

# fp32 parity bar of the propagation path: 1e-5 relative.  "Relative" is taken
# against the size of the sum's terms (sum_k |a_ik * b_kj|, computed in fp64 by
# the oracle), which is the quantity fp32 rounding error actually scales with;
# a plain per-element ratio is meaningless where a sum cancels to ~0.
RTOL = 1e-5


def assert_close_to_f64(got, ref64, mag64, rtol=RTOL, what=""):
    got = np.asarray(got, dtype=np.float64)
    err = np.abs(got - ref64)
    bound = rtol * mag64 + 1e-30
    bad = err > bound
    if bad.any():
        i = np.unravel_index(np.argmax(err / bound), err.shape)
        raise AssertionError("%s: %d elements off; worst at %s: got %r ref %r |terms| %r" %
                             (what, int(bad.sum()), i, got[i], ref64[i], mag64[i]))


def rel_maxnorm(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = np.abs(ref).max()
    return float(np.abs(got - ref).max() / den) if den > 0 else float(np.abs(got).max())


def random_coo(rng, n_rows, n_cols, nnz, hub_rows=(), empty_rows=(), dup_frac=0.05):
    """Unsorted COO with duplicates, optional hub rows and forced-empty rows."""
    rows = rng.randint(0, n_rows, size=nnz).astype(np.int32)
    cols = rng.randint(0, n_cols, size=nnz).astype(np.int32)
    for h, cnt in hub_rows:
        idx = rng.choice(nnz, size=min(cnt, nnz), replace=False)
        rows[idx] = h
    ndup = int(nnz * dup_frac)
    if ndup and nnz > 1:
        src = rng.randint(0, nnz, size=ndup)
        dst = rng.randint(0, nnz, size=ndup)
        rows[dst] = rows[src]
        cols[dst] = cols[src]
    if len(empty_rows):
        keep = ~np.isin(rows, np.asarray(empty_rows))
        rows, cols = rows[keep], cols[keep]
    vals = rng.standard_normal(len(rows))
    return rows, cols, vals
