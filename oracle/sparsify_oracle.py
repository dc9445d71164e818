"""ORACLE — test infrastructure only (see oracle/README.md).

numpy/scipy restatement of ``sparsify_using_global_cutoff``
(annealing_sign_problem/common.py:634-692) on plain arrays, with the very scipy calls the
reference makes (csr + csr.transpose(), eliminate_zeros, csgraph.connected_components, fancy
indexing), so the HIP implementation is checked against scipy's semantics and not a re-derivation.
Pinned by tests/golden/make_ising_kagome16_cluster.npz (sp_* arrays written by the reference's
own function).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse
from scipy.sparse.csgraph import connected_components


def sparsify_component(exchange, is_frozen, reltol: float, anchor: int):
    """``(keep bool[K], block csr)``: spins connected to ``anchor`` after the cutoff and the
    un-pruned block of ``exchange`` on them.  Raises AssertionError like the reference when a
    frozen spin falls outside."""
    full = scipy.sparse.csr_matrix(exchange)
    full.sort_indices()
    is_frozen = np.asarray(is_frozen, dtype=bool)
    rows = np.repeat(np.arange(full.shape[0]), np.diff(full.indptr))
    data = full.data.copy()
    if data.size:                                             # common.py:634-643
        threshold = reltol * np.max(np.abs(data))
        weak = (np.abs(data) < threshold) & ~(is_frozen[rows] & is_frozen[full.indices])
        data[weak] = 0
    pruned = scipy.sparse.csr_matrix((data, full.indices, full.indptr), shape=full.shape)
    pruned = 0.5 * (pruned + pruned.transpose())             # common.py:660-662
    pruned.eliminate_zeros()
    _, component = connected_components(pruned, directed=False)   # common.py:664
    wanted = component[anchor]
    assert np.all(component[is_frozen] == wanted)             # common.py:666
    keep = component == wanted
    return keep, full[keep][:, keep]                          # common.py:674
