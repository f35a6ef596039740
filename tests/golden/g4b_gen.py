# -*- coding: utf-8 -*-
"""The records of the ``g4b`` multisample fixture (6 BAM files, 21 000 reads, 260 cells of very different sizes, 30 genes x 2
haplotypes), regenerated from a seed by the golden script (which feeds them to the reference) and by the tests (which feed them
to the oracle and to the checkers): the fixture itself holds only what the reference made of them.  Test scaffolding, own code."""
import numpy as np

N_FILES, READS_PER_FILE, N_CELLS, N_GENES = 6, 3500, 260, 30
FILE_NAMES = ["m_%s.bam" % c for c in "dbfaec"]          # (glob order is whatever the file system says: the fixture records it)
MINCOUNTS = (-1, 40, 160)


def references():
    refs = []
    for g in range(N_GENES):
        refs.append(("T%03d_A" % g, 600 + 37 * g))
        refs.append(("T%03d_B" % g, 605 + 37 * g))
    return refs


def _qname(read, cell):
    f = ["x%d" % i for i in range(15)]
    f[0], f[14] = read, cell
    return "|||".join(f)


def files(seed=20260104):
    """-> {file name: [(qname, flag, tid, pos, next_tid, next_pos), ...]}"""
    rng = np.random.RandomState(seed)
    refs = references()
    w = 1.0 / np.arange(1, N_CELLS + 1) ** 1.2                  # a few big cells, a long tail of tiny ones
    w /= w.sum()
    gw = 1.0 / np.arange(1, N_GENES + 1) ** 0.9
    gw /= gw.sum()
    out = {}
    for fi, fname in enumerate(FILE_NAMES):
        recs = []
        cells = rng.choice(N_CELLS, size=READS_PER_FILE, p=np.roll(w, 7 * fi))     # every file has its own big cells
        for r in range(READS_PER_FILE):
            name = _qname("f%dr%05d" % (fi, r), "BC%04d" % cells[r])
            if r % 97 == 13:
                name += " 1:N:0"                                 # a name with a space: tracked untrimmed after a switch
            u = rng.rand()
            if u < 0.04:
                recs.append((name, 4, -1, -1, -1, -1))           # unmapped
                continue
            g = int(rng.choice(N_GENES, p=gw))
            tids = [2 * g] if rng.rand() < 0.25 else ([2 * g + 1] if rng.rand() < 0.2 else [2 * g, 2 * g + 1])
            if rng.rand() < 0.35:                                # a second gene, near the first
                g2 = (g + 1 + int(rng.randint(3))) % N_GENES
                tids += [2 * g2 + int(rng.randint(2))] if rng.rand() < 0.5 else [2 * g2, 2 * g2 + 1]
            if rng.rand() < 0.08:
                tids.append(tids[0])                             # a duplicate (read, target) alignment
            if rng.rand() < 0.3:
                tids = [tids[i] for i in rng.permutation(len(tids))]
            for t in tids:
                recs.append((name, 16 if rng.rand() < 0.5 else 0, int(t), int(rng.randint(refs[t][1])), -1, -1))
        out[fname] = recs
    return out
