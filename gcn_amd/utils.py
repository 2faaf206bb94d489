"""Host-side preprocessing with the reference's names and return formats
(reference gcn/utils.py).  Everything here is numpy / scipy on the host and is
a bit-exact target: the sparse tuples ``(coords int32 [nnz,2], values, shape)``
produced here are what the device CSR builder consumes.

Differences from the reference that do not change results:
  * dataset files are read through a non-executing decoder / ``.npz``
    containers instead of ``pickle.load``;
  * the adjacency matrix is assembled directly from the neighbour lists
    instead of through networkx (same node order, same 0/1 entries);
  * ``chebyshev_polynomials`` accepts an optional ``largest_eigval`` so tests
    can pin the otherwise unseeded ARPACK result.
"""
import os

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import eigsh

from . import planetoid

_PKG_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def parse_index_file(filename):
    """One integer per line (reference gcn/utils.py:9-14)."""
    with open(filename) as fh:
        return [int(tok.strip()) for tok in fh]


def sample_mask(idx, l):
    """Boolean vector of length ``l`` that is True at ``idx``
    (reference gcn/utils.py:17-21)."""
    m = np.zeros(l)
    m[idx] = 1
    return m.astype(np.bool_)


def _find_dataset(dataset_str, data_dir):
    """Locate the dataset: raw ``ind.*`` files or the shipped ``.npz``.
    Search order: explicit ``data_dir``, ``./data`` (the reference opens
    ``data/ind...`` relative to the cwd, gcn/utils.py:47), package data."""
    dirs = [data_dir] if data_dir else ["data", _PKG_DATA]
    for d in dirs:
        if os.path.exists(os.path.join(d, "ind.%s.graph" % dataset_str)):
            return planetoid.read_ind_files(d, dataset_str)
        npz = os.path.join(d, "ind.%s.npz" % dataset_str)
        if os.path.exists(npz):
            return planetoid.load_npz(npz)
    raise FileNotFoundError("dataset %r not found in %s" % (dataset_str, dirs))


def adjacency_from_dict_of_lists(graph):
    """0/1 symmetric adjacency (int64 CSR, canonical) of the undirected simple
    graph described by ``{node: [neighbours]}`` with nodes numbered in order of
    first appearance: keys first, then unseen neighbours.  This is what
    ``nx.adjacency_matrix(nx.from_dict_of_lists(graph))`` yields
    (reference gcn/utils.py:70): duplicates collapse, a self-loop is a single 1
    on the diagonal."""
    order = {}
    for k in graph:
        order.setdefault(k, len(order))
    for nbrs in graph.values():
        for v in nbrs:
            order.setdefault(v, len(order))
    n = len(order)
    src = np.fromiter((order[u] for u, nb in graph.items() for _ in nb), dtype=np.int64)
    dst = np.fromiter((order[v] for nb in graph.values() for v in nb), dtype=np.int64)
    r = np.concatenate([src, dst])
    c = np.concatenate([dst, src])
    key = np.unique(r * n + c)
    rows, cols = key // n, key % n
    a = sp.csr_matrix((np.ones(len(key), dtype=np.int64), (rows, cols)), shape=(n, n))
    a.sort_indices()
    return a


def load_data(dataset_str, data_dir=None):
    """Planetoid loader (reference gcn/utils.py:24-90).  Returns
    ``adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask``
    with the reference's types: ``adj`` int64 CSR, ``features`` LIL, label
    matrices float64 [N,C] zeroed outside their mask, masks ``bool``."""
    p = _find_dataset(dataset_str, data_dir)
    if "allx" not in p:
        raise FileNotFoundError(
            "ind.%s.allx is not available (not distributed with the reference "
            "tree); only the graph of this dataset can be loaded, see "
            "load_graph()" % dataset_str)
    x, y, tx, ty, allx, ally = p["x"], p["y"], p["tx"], p["ty"], p["allx"], p["ally"]
    reorder = np.asarray(p["test_index"])
    sorted_idx = np.sort(reorder)

    if dataset_str == "citeseer":
        # Citeseer's test block has gaps (isolated nodes): widen tx/ty to the
        # full index range with zero rows (reference gcn/utils.py:57-66).
        lo, hi = int(reorder.min()), int(reorder.max())
        wide_tx = sp.lil_matrix((hi - lo + 1, x.shape[1]))
        wide_tx[sorted_idx - lo, :] = tx
        tx = wide_tx
        wide_ty = np.zeros((hi - lo + 1, y.shape[1]))
        wide_ty[sorted_idx - lo, :] = ty
        ty = wide_ty

    features = sp.vstack((allx, tx)).tolil()
    features[reorder, :] = features[sorted_idx, :]
    adj = adjacency_from_dict_of_lists(p["graph"])

    labels = np.vstack((ally, ty))
    labels[reorder, :] = labels[sorted_idx, :]

    n = labels.shape[0]
    n_train = len(y)
    train_mask = sample_mask(range(n_train), n)
    val_mask = sample_mask(range(n_train, n_train + 500), n)
    test_mask = sample_mask(sorted_idx.tolist(), n)

    def keep(mask):
        out = np.zeros(labels.shape)
        out[mask, :] = labels[mask, :]
        return out

    return (adj, features, keep(train_mask), keep(val_mask), keep(test_mask),
            train_mask, val_mask, test_mask)


def load_graph(dataset_str, data_dir=None):
    """Adjacency only (works for Pubmed, whose feature blob is absent)."""
    return adjacency_from_dict_of_lists(_find_dataset(dataset_str, data_dir)["graph"])


def load_pubmed_synthetic_features(data_dir=None, nnz_per_row=50, n_features=500, seed=123):
    """Pubmed with its REAL graph, labels and splits and shape-matched SYNTHETIC
    features: ``ind.pubmed.allx`` is not distributed with the reference tree, so
    the bag-of-words matrix [19717, 500] is replaced by ``nnz_per_row`` ones per
    row at seeded random columns (duplicates add up).  Same return tuple as
    :func:`load_data`.  Timing and kernel parity only: accuracy on these features
    means nothing."""
    p = _find_dataset("pubmed", data_dir)
    reorder = np.asarray(p["test_index"])
    sorted_idx = np.sort(reorder)
    labels = np.vstack((p["ally"], p["ty"])).astype(np.float64)
    labels[reorder, :] = labels[sorted_idx, :]
    n = labels.shape[0]
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n), nnz_per_row)
    cols = rng.randint(0, n_features, size=n * nnz_per_row)
    feats = sp.csr_matrix((np.ones(n * nnz_per_row, dtype=np.float32), (rows, cols)), shape=(n, n_features))
    feats.data[:] = 1.0
    n_train = len(p["y"])
    train_mask = sample_mask(range(n_train), n)
    val_mask = sample_mask(range(n_train, n_train + 500), n)
    test_mask = sample_mask(sorted_idx.tolist(), n)
    keep = lambda m: np.where(m[:, None], labels, 0.0)
    return (adjacency_from_dict_of_lists(p["graph"]), feats.tolil(), keep(train_mask), keep(val_mask),
            keep(test_mask), train_mask, val_mask, test_mask)


def _one_tuple(mx):
    if not sp.isspmatrix_coo(mx):
        mx = mx.tocoo()
    coords = np.vstack((mx.row, mx.col)).transpose()
    return coords, mx.data, mx.shape


def sparse_to_tuple(sparse_mx):
    """``(coords[nnz,2], values[nnz], shape)``; a list is converted in place,
    element by element (reference gcn/utils.py:93-109)."""
    if isinstance(sparse_mx, list):
        for i, m in enumerate(sparse_mx):
            sparse_mx[i] = _one_tuple(m)
        return sparse_mx
    return _one_tuple(sparse_mx)


def preprocess_features(features):
    """Scale each row to unit sum, in the dtype of ``features``; all-zero rows
    stay zero (reference gcn/utils.py:112-119)."""
    rowsum = np.array(features.sum(1))
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.
    return sparse_to_tuple(sp.diags(r_inv).dot(features))


def normalize_adj(adj):
    """D^-1/2 A D^-1/2 in fp64 as COO; entry (r,c) is
    ``fl(fl(a[c,r] * d[r]) * d[c])`` with ``d = rowsum ** -0.5`` and zero-degree
    rows mapped to 0 (reference gcn/utils.py:122-129)."""
    adj = sp.coo_matrix(adj)
    rowsum = np.array(adj.sum(1))
    with np.errstate(divide="ignore"):
        d = np.power(rowsum, -0.5).flatten()
    d[np.isinf(d)] = 0.
    dm = sp.diags(d)
    return adj.dot(dm).transpose().dot(dm).tocoo()


def preprocess_adj(adj):
    """Sparse tuple of normalize_adj(A + I) (reference gcn/utils.py:132-135)."""
    return sparse_to_tuple(normalize_adj(adj + sp.eye(adj.shape[0])))


def construct_feed_dict(features, support, labels, labels_mask, placeholders):
    """Map placeholder handles to this step's inputs
    (reference gcn/utils.py:138-146)."""
    fd = {placeholders['labels']: labels,
          placeholders['labels_mask']: labels_mask,
          placeholders['features']: features}
    for ph, s in zip(placeholders['support'], support):
        fd[ph] = s
    fd[placeholders['num_features_nonzero']] = features[1].shape
    return fd


def chebyshev_polynomials(adj, k, largest_eigval=None):
    """Sparse tuples of T_0..T_k of the rescaled Laplacian
    L~ = (2/lmax)(I - D^-1/2 A D^-1/2) - I, with T_i = 2 L~ T_{i-1} - T_{i-2}
    formed as sparse-sparse products (reference gcn/utils.py:149-169).
    ``largest_eigval`` overrides the ARPACK estimate of lmax."""
    print("Calculating Chebyshev polynomials up to order {}...".format(k))
    n = adj.shape[0]
    lap = sp.eye(n) - normalize_adj(adj)
    if largest_eigval is None:
        largest_eigval = eigsh(lap, 1, which='LM')[0][0]
    scaled = (2. / largest_eigval) * lap - sp.eye(n)

    polys = [sp.eye(n), scaled]
    s_lap = sp.csr_matrix(scaled, copy=True)
    for _ in range(2, k + 1):
        polys.append(2 * s_lap.dot(polys[-1]) - polys[-2])
    return sparse_to_tuple(polys)
