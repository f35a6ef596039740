"""CHECKER (test infrastructure): the multisample reduction of the reference (bam_utils_multisample.py:503-636, 737-791) from
the (EC, cell, file) triples, in numpy -- what libecb's ecb_ms_filter must produce."""
import numpy as np


def reduce_triples(tr, n_ecs, n_cells, minimum_count):
    """Triples -> (kept cell ids in sample order, EC keep mask, CSC N) -- ``bam_utils_multisample.py:503-636, 737-791``."""
    ec, cell, fil, cnt, first = tr["ec"], tr["cell"], tr["file"], tr["count"], tr["first"]
    # cr_totals insertion order: files in order; within a file ECs by first appearance; within an EC cells by first appearance
    ec, cell, fil, cnt, first = (np.asarray(a, dtype=np.int64) for a in (ec, cell, fil, cnt, first))
    fe = fil * n_ecs + ec
    # first appearance of every EC in every file = the smallest first read among the (file, EC)'s triples
    # (sorted groups rather than ufunc.at, one 64-bit key where the numbers fit: the full-size GPU tests bring 10^8 triples)
    if len(fe) and int(fe.max()) < (1 << 31) and int(first.max()) < (1 << 32):
        o = np.argsort((fe << 32) | first, kind="stable")
    else:
        o = np.lexsort((first, fe))
    fe_s = fe[o]
    head = np.ones(len(o), dtype=bool)
    head[1:] = fe_s[1:] != fe_s[:-1]
    gid = np.cumsum(head) - 1
    fec_t = np.empty(len(o), dtype=np.int64)
    fec_t[o] = first[o][head][gid] if len(o) else 0
    del o, fe_s, head, gid
    # cr_totals' order = the order in which the cells first turn up when the triples are walked by (file, the EC's first appearance
    # in the file, first read): per cell the smallest such key, then the cells sorted by it
    oc = np.argsort(cell.astype(np.uint32 if n_cells > 65535 else np.uint16), kind="stable")
    cs = cell[oc]
    starts = np.flatnonzero(np.concatenate([[True], cs[1:] != cs[:-1]])) if len(cs) else np.zeros(0, dtype=np.int64)
    present = cs[starts]
    hi, lo = (fil[oc] << 40) | fec_t[oc], first[oc]                   # (fewer than 2^23 files; first reads below 2^40)
    min_hi = np.minimum.reduceat(hi, starts) if len(starts) else hi[:0]
    rep = np.repeat(min_hi, np.diff(np.concatenate([starts, [len(cs)]])))
    min_lo = np.minimum.reduceat(np.where(hi == rep, lo, np.iinfo(np.int64).max), starts) if len(starts) else lo[:0]
    cr_order = present[np.lexsort((min_lo, min_hi))]
    del oc, cs, hi, lo, rep
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
    indptr = np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=S))]).astype(np.int64)
    return kept_cells, ec_keep, (indptr, rows, data)


def select_rows(indptr, indices, data, keep):
    lens = np.diff(indptr)
    rowsel = np.repeat(keep, lens)
    new_ptr = np.concatenate([[0], np.cumsum(lens[keep])])
    return new_ptr, indices[rowsel], data[rowsel]


