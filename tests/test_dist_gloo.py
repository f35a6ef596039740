"""The multi-GPU exchange (alntools_amd/dist.py) at world sizes 2 and 3 on CPU / gloo, with an oracle-backed engine in
place of libecb: contiguous read shards -> per-rank EC tables -> cut by key range, exchanged, merged per range ->
gathered and adopted on the root must equal the single-process oracle on the whole stream (EC order = global first
appearance).  The simpler whole-table-to-root protocol is covered as well."""
import hashlib
import os
import socket
import struct

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alntools_amd import dist as ecdist
from alntools_amd import synth
from oracle import ec_oracle as orc


class OracleEngine(object):
    """EC table of one shard, built with the oracle; speaks the same table protocol as libecb."""
    device = torch.device("cpu")

    def __init__(self, n_haps):
        self.n_haps = n_haps
        self.table = {}              # (lo, hi) -> [count, first, pairs]
        self.n_all = self.n_valid = self.n_reads = 0
        self.read_ident, self.meta, self.triples = [], None, None     # multisample: EC identity and cell|file<<22 per read

    @staticmethod
    def _ident(pairs):
        d = hashlib.md5(np.asarray(pairs, dtype=np.int64).tobytes()).digest()
        lo, hi = struct.unpack("<qq", d)
        return lo, hi

    def build(self, t):
        valid = orc.tuples_valid(t["hapflag"])
        rid = t["read_id"].astype(np.int64)
        loc = t["locus"].astype(np.int64)
        hap = (t["hapflag"].astype(np.int64) >> 16) & 0xFF
        cur, row = None, None

        def close():
            pairs = tuple(sorted(row.items()))
            e = self.table.setdefault(self._ident(pairs), [0, cur, pairs])
            self.read_ident.append(self._ident(pairs))
            e[0] += 1
            e[1] = min(e[1], cur)

        for i in np.nonzero(valid)[0]:
            if rid[i] != cur:
                if cur is not None:
                    close()
                cur, row = int(rid[i]), {}
                self.n_reads += 1
            row[int(loc[i])] = row.get(int(loc[i]), 0) | (1 << int(hap[i]))
        if cur is not None:
            close()
        self.n_all, self.n_valid = len(rid), int(valid.sum())

    # --- protocol -------------------------------------------------------------------------
    def table_sizes(self):
        return len(self.table), sum(len(v[2]) for v in self.table.values()), self.n_reads

    def counters(self):
        return self.n_all, self.n_valid, self.n_reads

    def table_export(self, read_base):
        ent, prs = [], []
        for (lo, hi), (count, first, pairs) in self.table.items():
            first_inv = (~(first + read_base)) & 0xFFFFFFFF
            w2 = count | (first_inv << 32)
            w3 = len(prs) | (len(pairs) << 32)
            ent += [lo, hi, w2 - (1 << 64) if w2 >= (1 << 63) else w2, w3]
            prs += [l | (m << 32) for l, m in pairs]
        return (torch.tensor(ent or [0] * 4, dtype=torch.int64), torch.tensor(prs or [0], dtype=torch.int64))

    def table_export_parts(self, read_base, n_parts):
        ent, prs, eoff, poff = [], [], [0], [0]
        for q in range(n_parts):
            for (lo, hi), (count, first, pairs) in self.table.items():
                if ((lo & 0xFFFFFFFFFFFFFFFF) >> 40) % n_parts != q:
                    continue
                first_inv = (~(first + read_base)) & 0xFFFFFFFF
                w2 = count | (first_inv << 32)
                w3 = (len(prs) - poff[q]) | (len(pairs) << 32)          # off relative to the part's pairs
                ent += [lo, hi, w2 - (1 << 64) if w2 >= (1 << 63) else w2, w3]
                prs += [l | (m << 32) for l, m in pairs]
            eoff.append(len(ent) // 4)
            poff.append(len(prs))
        return (torch.tensor(ent or [0] * 4, dtype=torch.int64), torch.tensor(prs or [0], dtype=torch.int64), eoff, poff)

    def table_rebase(self, ent, n_entries, read_base):
        e = ent.numpy().view(np.uint64).reshape(-1, 4)[:n_entries]        # (in place: a view of the tensor's memory)
        first = (~(e[:, 2] >> np.uint64(32))) & np.uint64(0xFFFFFFFF)
        assert int(first.max(initial=0)) + read_base < 0xFFFFFFFF
        e[:, 2] = (e[:, 2] & np.uint64(0xFFFFFFFF)) | (((~(first + np.uint64(read_base))) & np.uint64(0xFFFFFFFF)) << np.uint64(32))
        return ent

    def table_adopt(self, ent, n_entries, prs, n_pairs):
        before = len(self.table)
        self.table_merge(ent, n_entries, prs, n_pairs)
        assert len(self.table) == before + n_entries, "adopted parts must hold distinct ECs"

    # multisample across ranks
    def ec_keys(self, n_ecs):
        c = self.csr()
        if getattr(self, "assembled", None) is not None:      # no table behind an assembled result: an EC's identity is a function of its key
            ip, ix, da = c["indptr"], c["indices"], c["data"]
            rows = [(self._ident(tuple((int(ix[i]), int(da[i])) for i in range(ip[e], ip[e + 1]))), None) for e in range(len(ip) - 1)]
        else:
            rows = sorted(self.table.items(), key=lambda kv: kv[1][1])
        assert len(rows) == n_ecs
        packed = ecdist.pack_ec_keys(torch.tensor([lo for (lo, hi), _ in rows], dtype=torch.int64),
                                     torch.tensor(c["indptr"], dtype=torch.int32), torch.tensor(c["indices"], dtype=torch.int32),
                                     torch.tensor(c["data"], dtype=torch.int32))
        return packed, len(c["indices"])

    def ms_local_triples(self, packed, n_ecs, nnz, read_base):
        keys, ip, ix, da = (t.tolist() for t in ecdist.unpack_ec_keys(packed, n_ecs, nnz))
        rank_of_key = {tuple(zip(ix[ip[e]:ip[e + 1]], da[ip[e]:ip[e + 1]])): e for e in range(n_ecs)}     # exact: the key itself
        rank_of = {ident: rank_of_key[v[2]] for ident, v in self.table.items()}
        assert all(keys[rank_of[ident]] == ident[0] for ident in self.table)
        acc = {}
        for i, ident in enumerate(self.read_ident):
            t = acc.setdefault((rank_of[ident] << 32) | int(self.meta[i]), [0, i + read_base])
            t[0] += 1
        ks = sorted(acc)
        return (torch.tensor(ks or [0], dtype=torch.int64), torch.tensor([acc[x][0] for x in ks] or [0], dtype=torch.int32),
                torch.tensor([acc[x][1] for x in ks] or [0], dtype=torch.int32), len(ks))

    def ms_adopt_triples(self, tables):
        acc = {}
        for key, cnt, first, n in tables:
            for x, c, f in zip(key[:n].tolist(), cnt[:n].tolist(), first[:n].tolist()):
                t = acc.setdefault(x, [0, f])
                t[0] += c
                t[1] = min(t[1], f)
        self.triples = sorted((x >> 32, x & 0xFFFFFFFF, c, f) for x, (c, f) in acc.items())
        return len(self.triples)

    def table_merge_many(self, tables):
        for t in tables:
            self.table_merge(*t)

    def table_adopt_many(self, tables):
        for t in tables:
            self.table_adopt(*t)

    def table_merge(self, ent, n_entries, prs, n_pairs):
        e = ent.numpy().astype(np.uint64).reshape(-1, 4)[:n_entries]
        p = prs.numpy().astype(np.uint64)
        for lo, hi, w2, w3 in e.tolist():
            lo, hi = (x - (1 << 64) if x >= (1 << 63) else x for x in (lo, hi))      # keys stay signed 64-bit
            count, first = w2 & 0xFFFFFFFF, (~(w2 >> 32)) & 0xFFFFFFFF
            off, n = w3 & 0xFFFFFFFF, w3 >> 32
            pairs = tuple((int(x) & 0xFFFFFFFF, int(x) >> 32) for x in p[off:off + n])
            t = self.table.setdefault((lo, hi), [0, first, pairs])
            assert t[2] == pairs
            t[0] += count
            t[1] = min(t[1], first)

    def add_counters(self, a, v, r):
        self.n_all += a
        self.n_valid += v
        self.n_reads += r

    # finalize per key range
    def finalize_range(self, n_all, n_valid, n_reads):
        rows = sorted(self.table.values(), key=lambda v: v[1])
        assert all(0 <= r[1] < n_reads for r in rows)
        c = self.csr()
        n_ecs, nnz = len(rows), len(c["indices"])
        if not n_ecs:
            return torch.zeros(1, dtype=torch.int32), 0, 0
        packed = torch.empty(ecdist.piece_words(n_ecs, nnz), dtype=torch.int32)
        ip, cn, fi, ix, da = ecdist.unpack_piece(packed, n_ecs, nnz)
        ip[:] = torch.tensor(c["indptr"], dtype=torch.int32)
        cn[:] = torch.tensor(c["count"], dtype=torch.int32)
        fi[:] = torch.tensor([r[1] for r in rows], dtype=torch.int64).to(torch.int32)
        ix[:] = torch.tensor(c["indices"], dtype=torch.int32)
        da[:] = torch.tensor(c["data"], dtype=torch.int32)
        return packed, n_ecs, nnz

    def assemble_ranges(self, pieces, n_all, n_valid, n_reads):
        assert not self.table
        rows = []
        for packed, n_ecs, nnz in pieces:
            ip, cn, fi, ix, da = (t.tolist() for t in ecdist.unpack_piece(packed, n_ecs, nnz))
            rows += [(fi[e] & 0xFFFFFFFF, cn[e], ix[ip[e]:ip[e + 1]], da[ip[e]:ip[e + 1]]) for e in range(n_ecs)]
        rows.sort(key=lambda r: r[0])
        assert len(set(r[0] for r in rows)) == len(rows), "two ECs with one first read"
        indptr = np.cumsum([0] + [len(r[2]) for r in rows])
        self.assembled = dict(indptr=indptr, indices=np.array([x for r in rows for x in r[2]]),
                              data=np.array([x for r in rows for x in r[3]]), count=np.array([r[1] for r in rows]))
        self.add_counters(n_all, n_valid, n_reads)

    def csr(self):
        if getattr(self, "assembled", None) is not None:
            return self.assembled
        rows = sorted(self.table.values(), key=lambda v: v[1])
        indptr, indices, data = [0], [], []
        for count, first, pairs in rows:
            indices += [l for l, _ in pairs]
            data += [m for _, m in pairs]
            indptr.append(len(indices))
        return dict(indptr=np.array(indptr), indices=np.array(indices), data=np.array(data),
                    count=np.array([r[0] for r in rows]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, spec_args, out_path, protocol):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = synth.SynthSpec(**spec_args)
    R = spec.n_reads
    t = synth.generate(spec, rank * R // world, (rank + 1) * R // world)      # local read ids from 0
    eng = OracleEngine(spec.n_haps)
    eng.build(t)
    fresh = lambda: OracleEngine(spec.n_haps)
    if protocol == "ranges":
        merged = ecdist.exchange_and_merge(eng, fresh, fresh, root=0)
    elif protocol == "finalize_ranges":
        merged = ecdist.exchange_and_merge(eng, fresh, fresh, root=0, finalize_ranges=True)
    else:
        merged = ecdist.exchange_and_merge_on_root(eng, fresh, root=0)
    if rank == 0:
        c = merged.csr()
        np.savez(out_path, n_all=merged.n_all, n_valid=merged.n_valid, n_reads=merged.n_reads, **c)
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,paired,protocol", [(2, False, "ranges"), (2, True, "ranges"), (3, True, "ranges"), (4, True, "ranges"), (2, True, "root"), (4, False, "root"),
                                                   (2, False, "finalize_ranges"), (3, True, "finalize_ranges"), (4, True, "finalize_ranges")])
def test_multi_rank_merge_equals_single_process(tmp_path, world, paired, protocol):
    spec_args = dict(n_reads=3000, n_loci=300, n_haps=4, paired=paired)
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(world, _free_port(), spec_args, out, protocol), nprocs=world, join=True)
    got = np.load(out)
    spec = synth.SynthSpec(**spec_args)
    t = synth.generate(spec, 0, spec.n_reads)
    exp = orc.ec_from_tuples(t["read_id"], t["locus"], t["hapflag"], spec.n_loci, spec.n_haps)
    for k in ("indptr", "indices", "data", "count"):
        assert np.array_equal(got[k], exp[k]), k
    assert int(got["n_all"]) == exp["n_all"] and int(got["n_valid"]) == exp["n_valid"]
    assert int(got["n_reads"]) == t["n_reads"]


@pytest.mark.parametrize("protocol", ["ranges", "finalize_ranges"])
def test_more_ranks_than_reads(tmp_path, protocol):
    """Shards without a read and key ranges without an EC: 4 ranks, 3 reads."""
    spec_args = dict(n_reads=3, n_loci=50, n_haps=2, paired=False)
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(4, _free_port(), spec_args, out, protocol), nprocs=4, join=True)
    got = np.load(out)
    spec = synth.SynthSpec(**spec_args)
    t = synth.generate(spec, 0, spec.n_reads)
    exp = orc.ec_from_tuples(t["read_id"], t["locus"], t["hapflag"], spec.n_loci, spec.n_haps)
    for k in ("indptr", "indices", "data", "count"):
        assert np.array_equal(got[k], exp[k]), k
    assert int(got["n_reads"]) == t["n_reads"]


def _range_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = np.random.RandomState(7)
    pos = rs.randint(0, 5000, size=(world, 40, 3))
    seen = rs.rand(world, 40, 3) < 0.5
    mn = np.where(seen[rank], pos[rank], np.iinfo(np.int32).max).astype(np.int32)
    mx = np.where(seen[rank], pos[rank] + rs.randint(0, 100, size=(40, 3)), np.iinfo(np.int32).min).astype(np.int32)
    got = ecdist.reduce_ranges(mn, mx)
    np.save(out_path % rank, got)
    dist.barrier()
    dist.destroy_process_group()


def test_range_extremes_reduce_over_ranks(tmp_path):
    """--rangefile with shards: min / max all-reduce, 0 where no rank saw an alignment."""
    world = 2
    out = str(tmp_path / "rng%d.npy")
    mp.spawn(_range_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    rs = np.random.RandomState(7)
    pos = rs.randint(0, 5000, size=(world, 40, 3))
    seen = rs.rand(world, 40, 3) < 0.5
    ext = rs.randint(0, 100, size=(40, 3))           # (rank 0 drew this; rank 1 draws the same numbers from the same seed)
    big = np.iinfo(np.int64).max
    mn = np.where(seen, pos, big).min(axis=0)
    mx = np.where(seen, pos + ext, -1).max(axis=0)
    exp = np.where(seen.any(axis=0), mx - mn + 1, 0)
    for r in range(world):
        assert np.array_equal(np.load(out % r), exp)


def _meta_of(g):
    """cell | file << 22 of global read g (any fixed function: cells straddle the shards)"""
    return ((g * 2654435761) % 37) | ((g % 3) << 22)


def _ms_worker(rank, world, port, spec_args, out_path, per_range=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = synth.SynthSpec(**spec_args)
    R = spec.n_reads
    a, b = rank * R // world, (rank + 1) * R // world
    base = synth.generate(spec, 0, a)["n_reads"] if a else 0
    t = synth.generate(spec, a, b)
    eng = OracleEngine(spec.n_haps)
    eng.build(t)
    eng.meta = [_meta_of(base + i) for i in range(eng.n_reads)]
    fresh = lambda: OracleEngine(spec.n_haps)
    merged = ecdist.exchange_and_merge(eng, fresh, fresh, root=0, finalize_ranges=per_range)
    n_ecs = (len(merged.csr()["indptr"]) - 1) if rank == 0 else None
    n = ecdist.exchange_multisample(eng, merged, n_ecs, root=0)
    if rank == 0:
        assert n == len(merged.triples)
        np.save(out_path, np.array(merged.triples, dtype=np.int64))
    else:
        assert n is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,per_range", [(2, False), (3, False), (4, False), (2, True), (3, True)])
def test_multisample_triples_over_ranks(tmp_path, world, per_range):
    """(EC, cell, file) read counts with the reads sharded over ranks == one process over the whole stream; with the merged tables
    adopted by the root, and (``per_range``) with every rank finalizing its own key range and the root taking the ECs' identities
    off the assembled rows."""
    spec_args = dict(n_reads=3000, n_loci=300, n_haps=4, paired=True)
    out = str(tmp_path / "triples.npy")
    mp.spawn(_ms_worker, args=(world, _free_port(), spec_args, out, per_range), nprocs=world, join=True)
    got = np.load(out)
    spec = synth.SynthSpec(**spec_args)
    one = OracleEngine(spec.n_haps)
    one.build(synth.generate(spec, 0, spec.n_reads))
    rank_of = {k: e for e, (k, _) in enumerate(sorted(one.table.items(), key=lambda kv: kv[1][1]))}
    acc = {}
    for i, ident in enumerate(one.read_ident):
        t = acc.setdefault((rank_of[ident], _meta_of(i)), [0, i])
        t[0] += 1
    exp = np.array(sorted((e, m, c, f) for (e, m), (c, f) in acc.items()), dtype=np.int64)
    assert np.array_equal(got, exp)
