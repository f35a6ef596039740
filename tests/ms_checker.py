"""CHECKER (test infrastructure): the multisample reduction of the reference (bam_utils_multisample.py:503-636, 737-791) from
the (EC, cell, file) triples, in numpy -- what libecb's ecb_ms_filter must produce."""
import numpy as np


def reduce_triples(tr, n_ecs, n_cells, minimum_count):
    """Triples -> (kept cell ids in sample order, EC keep mask, CSC N) -- ``bam_utils_multisample.py:503-636, 737-791``."""
    ec, cell, fil, cnt, first = tr["ec"], tr["cell"], tr["file"], tr["count"], tr["first"]
    # cr_totals insertion order: files in order; within a file ECs by first appearance; within an EC cells by first appearance
    fe = fil * n_ecs + ec
    _, inv = np.unique(fe, return_inverse=True)
    fec = np.full(inv.max() + 1 if len(inv) else 0, np.iinfo(np.int64).max)
    np.minimum.at(fec, inv, first)
    order = np.lexsort((first, fec[inv], fil))
    seq = cell[order]
    _, idx = np.unique(seq, return_index=True)
    cr_order = seq[np.sort(idx)]
    totals = np.bincount(cell, weights=cnt, minlength=n_cells).astype(np.int64)
    if minimum_count <= 0:
        minimum_count = 1                                             # :596-597
    kept_cells = [int(c) for c in cr_order if totals[c] >= minimum_count]
    new_cell = np.full(n_cells, -1, dtype=np.int64)
    new_cell[kept_cells] = np.arange(len(kept_cells))
    sel = new_cell[cell] >= 0
    ec_keep = np.zeros(n_ecs, dtype=bool)
    ec_keep[ec[sel]] = True                                           # ECs left empty are dropped, the rest re-ranked (:616-636)
    new_rank = np.cumsum(ec_keep) - 1
    S = len(kept_cells)
    key = new_cell[cell[sel]] * int(ec_keep.sum()) + new_rank[ec[sel]]          # column-major: CSC order
    uk, kinv = np.unique(key, return_inverse=True)
    data = np.bincount(kinv, weights=cnt[sel]).astype(np.int64)
    E2 = int(ec_keep.sum())
    cols, rows = uk // max(E2, 1), uk % max(E2, 1)
    indptr = np.zeros(S + 1, dtype=np.int64)
    np.add.at(indptr, cols + 1, 1)
    indptr = np.cumsum(indptr)
    return kept_cells, ec_keep, (indptr, rows, data)


def select_rows(indptr, indices, data, keep):
    lens = np.diff(indptr)
    rowsel = np.repeat(keep, lens)
    new_ptr = np.concatenate([[0], np.cumsum(lens[keep])])
    return new_ptr, indices[rowsel], data[rowsel]


