# -*- coding: utf-8 -*-
"""Multi-GPU EC build: contiguous read shards, EC tables merged by key range, one gather.

The reference parallelises over contiguous chunk ranges per process and merges the workers' ordered dicts in
process order (``alntools/bam_utils.py:646-658, 680-724``), which makes EC rank = global first appearance.  Here
every rank builds the table of its own contiguous read shard (``first`` is rebased to the global read numbering on
export), then

1. every rank cuts its table into ``world`` key ranges (``ecb_table_export_parts_device``) and sends range q to
   rank q -- point to point over RCCL/xGMI, every pair of GPUs on its own link, ``world - 1`` messages of
   ``1/world`` of a table each;
2. rank q merges the ``world`` pieces of range q in rank order (``ecb_table_merge_device``): the hashing and key
   comparison of the merge is spread over the GPUs instead of queueing on the root;
3. the merged ranges hold disjoint ECs.  ``finalize_ranges=True`` (single-sample and, since round 4, multisample runs): rank q
   finalizes its own range -- ranks its ECs by first read and emits their CSR rows and counts, 1/world of the work -- and sends
   the rows to the root, which only places the pieces in the global order of first reads (``ecb_assemble_ranges_device``);
   a multisample root takes the ECs' hashes for the second exchange off the finished rows (an EC's hash is a function of its
   key).  ``finalize_ranges=False`` (kept): the merged tables are sent to the root, which loads them without hashing
   (``ecb_table_adopt_device``) and finalizes alone.

The functions only need an *engine* with ``table_sizes/counters/table_export/table_export_parts/table_merge(_many)/
table_adopt(_many)/table_rebase/add_counters/finalize_range/assemble_ranges`` -- :class:`GpuEngine` wraps an :class:`alntools_amd.ecb.EcBuilder`; the CPU ``gloo`` tests
plug in an oracle-backed engine.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

ENTRY_WORDS = 4          # one table entry = 32 bytes = 4 x int64
PAIR_WORDS = 1           # one (locus, mask) pair = 8 bytes = 1 x int64


def ec_keys_words(n_ecs, nnz):
    """int64 words of the packed EC list: hashes | indptr (int32, padded) | indices | data."""
    return n_ecs + (n_ecs + 2) // 2 + 2 * ((nnz + 1) // 2)


def pack_ec_keys(keys, indptr, indices, data):
    """One int64 tensor (one broadcast): [hash per EC | indptr | indices | data], the int32 parts padded to whole words."""
    n_ecs, nnz = keys.numel(), indices.numel()
    out = torch.zeros(ec_keys_words(n_ecs, nnz), dtype=torch.int64, device=keys.device)
    out[:n_ecs] = keys
    w = out[n_ecs:].view(torch.int32)
    a = 2 * ((n_ecs + 2) // 2)
    b = a + 2 * ((nnz + 1) // 2)
    w[:n_ecs + 1] = indptr
    w[a:a + nnz] = indices
    w[b:b + nnz] = data
    return out


def unpack_ec_keys(packed, n_ecs, nnz):
    w = packed[n_ecs:].view(torch.int32)
    a = 2 * ((n_ecs + 2) // 2)
    b = a + 2 * ((nnz + 1) // 2)
    return packed[:n_ecs], w[:n_ecs + 1], w[a:a + nnz], w[b:b + nnz]


def piece_words(n_ecs, nnz):
    """int32 words of one finalized key range: indptr | counts | first reads | indices | data."""
    return 3 * n_ecs + 1 + 2 * nnz


def unpack_piece(packed, n_ecs, nnz):
    """-> views (indptr, counts, firsts, indices, data) of a :func:`piece_words` tensor."""
    a = n_ecs + 1
    return (packed[:a], packed[a:a + n_ecs], packed[a + n_ecs:a + 2 * n_ecs], packed[a + 2 * n_ecs:a + 2 * n_ecs + nnz],
            packed[a + 2 * n_ecs + nnz:a + 2 * n_ecs + 2 * nnz])


class GpuEngine(object):
    """Adapter from :class:`alntools_amd.ecb.EcBuilder` to the merge protocol."""

    def __init__(self, builder, device):
        self.b, self.device = builder, device

    def table_sizes(self):
        return self.b.table_sizes()

    def counters(self):
        return self.b.counters()

    def table_export(self, read_base):
        ne, npairs, _ = self.b.table_sizes()
        ent = torch.empty(max(ne, 1) * ENTRY_WORDS, dtype=torch.int64, device=self.device)
        prs = torch.empty(max(npairs, 1) * PAIR_WORDS, dtype=torch.int64, device=self.device)
        self.b.table_export_device(ent, prs, read_base)
        return ent, prs

    def table_export_parts(self, read_base, n_parts):
        ne, npairs, _ = self.b.table_sizes()
        ent = torch.empty(max(ne, 1) * ENTRY_WORDS, dtype=torch.int64, device=self.device)
        prs = torch.empty(max(npairs, 1) * PAIR_WORDS, dtype=torch.int64, device=self.device)
        eoff, poff = self.b.table_export_parts_device(ent, prs, read_base, n_parts)
        return ent, prs, eoff, poff

    def table_adopt(self, ent, n_entries, prs, n_pairs):
        self.b.table_adopt_device(ent, n_entries, prs, n_pairs)

    def table_rebase(self, ent, n_entries, read_base):
        """First reads of exported entries moved on by ``read_base`` (in place, ahead of a merge on this engine) -> ``ent``."""
        self.b.table_rebase_device(ent, n_entries, read_base)
        return ent

    # multisample across GPUs
    def ec_keys(self, n_ecs):
        """The merged ECs in rank order, as the shards need them to find their own: (hash per EC [int64], CSR A indptr,
        indices, data [int32]) -- one packed int64 tensor plus its layout ``(n_ecs, nnz)``."""
        nnz = self.b.sizes["nnz_a"]
        keys = torch.empty(max(n_ecs, 1), dtype=torch.int64, device=self.device)
        ip = torch.empty(n_ecs + 1, dtype=torch.int32, device=self.device)
        ix = torch.empty(max(nnz, 1), dtype=torch.int32, device=self.device)
        da = torch.empty(max(nnz, 1), dtype=torch.int32, device=self.device)
        self.b.export_ec_keys_device(keys)
        self.b.export_device(ip, ix, da)
        return pack_ec_keys(keys[:n_ecs], ip, ix[:nnz], da[:nnz]), nnz

    def ms_local_triples(self, packed, n_ecs, nnz, read_base):
        keys, ip, ix, da = unpack_ec_keys(packed, n_ecs, nnz)
        n_reads = self.b.table_sizes()[2]
        key = torch.empty(max(n_reads, 1), dtype=torch.int64, device=self.device)
        cnt = torch.empty(max(n_reads, 1), dtype=torch.int32, device=self.device)
        first = torch.empty(max(n_reads, 1), dtype=torch.int32, device=self.device)
        n = self.b.ms_local_triples_device(keys, ip, ix, da, n_ecs, read_base, key, cnt, first)
        return key, cnt, first, n

    def ms_adopt_triples(self, tables):
        return self.b.ms_adopt_triples_device(tables)

    def table_merge_many(self, tables):
        """``tables``: [(entries, n_entries, pairs, n_pairs), ...] merged in that order, one host wait for all."""
        self.b.table_merge_batch_device(tables)

    def table_adopt_many(self, tables):
        self.b.table_adopt_batch_device(tables)

    def table_merge(self, ent, n_entries, prs, n_pairs):
        self.b.table_merge_device(ent, n_entries, prs, n_pairs)

    def add_counters(self, a, v, r):
        self.b.add_counters(a, v, r)

    # finalize per key range
    def finalize_range(self, n_all, n_valid, n_reads):
        """On the handle that merged one key range: rank and emit its ECs against the whole run's read numbering ->
        (packed int32 tensor [:func:`piece_words`], n_ecs, nnz)."""
        if not self.b.table_sizes()[0]:
            return torch.zeros(1, dtype=torch.int32, device=self.device), 0, 0
        self.b.add_counters(n_all, n_valid, n_reads)
        s = self.b.finalize()
        n_ecs, nnz = s["n_ecs"], s["nnz_a"]
        packed = torch.empty(piece_words(n_ecs, nnz), dtype=torch.int32, device=self.device)
        ip, cn, fi, ix, da = unpack_piece(packed, n_ecs, nnz)
        self.b.export_piece_device(ip, ix, da, cn, fi)
        return packed, n_ecs, nnz

    def assemble_ranges(self, pieces, n_all, n_valid, n_reads):
        """``pieces``: [(packed, n_ecs, nnz), ...] -> this (empty) handle holds the finalized result."""
        rows = []
        for packed, n_ecs, nnz in pieces:
            ip, cn, fi, ix, da = unpack_piece(packed, n_ecs, nnz)
            rows.append((ip, ix, da, cn, fi, n_ecs, nnz))
        return self.b.assemble_ranges_device(rows, n_reads, n_all, n_valid)


class HostStagedEngine(object):
    """A :class:`GpuEngine` seen through CPU tensors, for process groups that only move host memory (``gloo``): used to
    run the multi-rank protocol with real libecb handles when RCCL cannot be (two ranks sharing one GPU in the tests and
    in ``bench.py``'s ``ECB_DIST_BACKEND=gloo`` rehearsal).  Every table crosses PCIe twice: never a measurement."""
    device = torch.device("cpu")

    def __init__(self, eng):
        self.e, self.dev, self.b = eng, eng.device, eng.b

    def _up(self, t):
        return t.to(self.dev)

    def table_sizes(self):
        return self.e.table_sizes()

    def counters(self):
        return self.e.counters()

    def add_counters(self, *a):
        self.e.add_counters(*a)

    def table_export(self, read_base):
        ent, prs = self.e.table_export(read_base)
        return ent.cpu(), prs.cpu()

    def table_export_parts(self, read_base, n_parts):
        ent, prs, eo, po = self.e.table_export_parts(read_base, n_parts)
        return ent.cpu(), prs.cpu(), eo, po

    def table_merge(self, ent, n, prs, m):
        self.e.table_merge(self._up(ent), n, self._up(prs), m)

    def table_rebase(self, ent, n, read_base):
        return self.e.table_rebase(self._up(ent), n, read_base) if read_base and n else ent     # (stays on the device: _up is a no-op then)

    def table_merge_many(self, tables):
        self.e.table_merge_many([(self._up(a), n, self._up(b), m) for a, n, b, m in tables])

    def table_adopt_many(self, tables):
        self.e.table_adopt_many([(self._up(a), n, self._up(b), m) for a, n, b, m in tables])

    def finalize_range(self, *a):
        packed, n_ecs, nnz = self.e.finalize_range(*a)
        return packed.cpu(), n_ecs, nnz

    def assemble_ranges(self, pieces, *a):
        return self.e.assemble_ranges([(self._up(p), n, m) for p, n, m in pieces], *a)

    def ec_keys(self, n_ecs):
        packed, nnz = self.e.ec_keys(n_ecs)
        return packed.cpu(), nnz

    def ms_local_triples(self, packed, n_ecs, nnz, read_base):
        k, c, f, n = self.e.ms_local_triples(self._up(packed), n_ecs, nnz, read_base)
        return k.cpu(), c.cpu(), f.cpu(), n

    def ms_adopt_triples(self, tables):
        return self.e.ms_adopt_triples([(self._up(k), self._up(c), self._up(f), n) for k, c, f, n in tables])


SECTIONS = None          # measurement only (tools/dist_sections.py): a dict that collects seconds per section of the exchange


def _mark(name, dev, t0):
    """-> now; with SECTIONS set, waits for the device and adds the time since ``t0`` to ``SECTIONS[name]``."""
    if SECTIONS is None:
        return 0.0
    import time
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t = time.perf_counter()
    if name is not None:
        SECTIONS[name] = SECTIONS.get(name, 0.0) + (t - t0)
    return t


def _p2p(ops):
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def exchange_and_merge(engine, make_part_engine, make_root_engine, group=None, root=0, finalize_ranges=False):
    """All ranks call this after pushing their shard.  Returns the merged engine on ``root`` (ready to finalize; already
    finalized with ``finalize_ranges``, where ``finalize()`` only reports the sizes), ``None`` elsewhere.  Two small
    all-gathers (the cuts and read counts before the first exchange, the sizes of the merged ranges before the second) and two
    rounds of point-to-point messages; see the module text."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = engine.device

    # 1. cut by key range, range q -> rank q.  First reads are numbered from the shard's own 0: where a shard starts in the run
    # is only known after the exchange of sizes, which is the same one that announces the cuts (one all-gather, not two).
    t = _mark(None, dev, 0.0)
    ent, prs, eoff, poff = engine.table_export_parts(0, world)
    n_all, n_valid, nreads = engine.counters()
    t = _mark("counts + cut by key range", dev, t)
    cuts = torch.tensor([eoff[q + 1] - eoff[q] for q in range(world)] + [poff[q + 1] - poff[q] for q in range(world)] +
                        [nreads, n_all, n_valid], dtype=torch.int64, device=dev)
    width = 2 * world + 3
    allcuts = torch.empty(world * width, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allcuts, cuts, group=group)
    allcuts = allcuts.view(world, width).cpu().tolist()
    in_e = [allcuts[r][rank] for r in range(world)]
    in_p = [allcuts[r][world + rank] for r in range(world)]
    read_base = [sum(allcuts[x][2 * world] for x in range(r)) for r in range(world)]
    totals = (sum(c[2 * world + 1] for c in allcuts), sum(c[2 * world + 2] for c in allcuts), sum(c[2 * world] for c in allcuts))
    t = _mark("all-gather of cuts", dev, t)
    piece_e = [None] * world
    piece_p = [None] * world
    ops = []
    for r in range(world):
        if r == rank:
            piece_e[r] = ent[eoff[r] * ENTRY_WORDS:eoff[r + 1] * ENTRY_WORDS]
            piece_p[r] = prs[poff[r] * PAIR_WORDS:poff[r + 1] * PAIR_WORDS]
            continue
        if eoff[r + 1] > eoff[r]:
            ops.append(dist.P2POp(dist.isend, ent[eoff[r] * ENTRY_WORDS:eoff[r + 1] * ENTRY_WORDS], r, group))
            if poff[r + 1] > poff[r]:
                ops.append(dist.P2POp(dist.isend, prs[poff[r] * PAIR_WORDS:poff[r + 1] * PAIR_WORDS], r, group))
        if in_e[r]:
            piece_e[r] = torch.empty(in_e[r] * ENTRY_WORDS, dtype=torch.int64, device=dev)
            piece_p[r] = torch.empty(max(in_p[r], 1) * PAIR_WORDS, dtype=torch.int64, device=dev)
            ops.append(dist.P2POp(dist.irecv, piece_e[r], r, group))
            if in_p[r]:
                ops.append(dist.P2POp(dist.irecv, piece_p[r][:in_p[r] * PAIR_WORDS], r, group))
    _p2p(ops)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()

    t = _mark("exchange of key ranges", dev, t)

    # 2. merge my key range, in rank order (= stream order), first reads moved to the run's numbering
    part = make_part_engine()
    part.table_merge_many([(part.table_rebase(piece_e[r], in_e[r], read_base[r]), in_e[r], piece_p[r], in_p[r]) for r in range(world) if in_e[r]])
    del piece_e, piece_p, ent, prs
    t = _mark("merge of my range (incl. the range handle's reset)", dev, t)

    if finalize_ranges:
        return _finalize_ranges(part, make_root_engine, totals, world, rank, dev, group, root)

    # 3. the merged ranges are disjoint: gather them on the root, which adopts them as they are
    pe_n, pp_n, _ = part.table_sizes()
    pe, pp = part.table_export(0)
    mine2 = torch.tensor([pe_n, pp_n], dtype=torch.int64, device=dev)
    sz2 = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sz2, mine2, group=group)
    sz2 = sz2.view(world, 2).cpu().tolist()
    if rank != root:
        ops = []
        if pe_n:
            ops.append(dist.P2POp(dist.isend, pe[:pe_n * ENTRY_WORDS], root, group))
            if pp_n:
                ops.append(dist.P2POp(dist.isend, pp[:pp_n * PAIR_WORDS], root, group))
        _p2p(ops)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()     # keep pe / pp alive until they are on the wire
        return None
    bufs, ops = {}, []
    for r in range(world):
        e_r, p_r = sz2[r]
        if r == root or not e_r:
            continue
        bufs[r] = (torch.empty(e_r * ENTRY_WORDS, dtype=torch.int64, device=dev),
                   torch.empty(max(p_r, 1) * PAIR_WORDS, dtype=torch.int64, device=dev))
        ops.append(dist.P2POp(dist.irecv, bufs[r][0], r, group))
        if p_r:
            ops.append(dist.P2POp(dist.irecv, bufs[r][1][:p_r * PAIR_WORDS], r, group))
    _p2p(ops)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    merged = make_root_engine()
    merged.table_adopt_many([(pe, sz2[r][0], pp, sz2[r][1]) if r == root else (bufs[r][0], sz2[r][0], bufs[r][1], sz2[r][1])
                             for r in range(world) if sz2[r][0]])
    merged.add_counters(*totals)
    return merged


def _finalize_ranges(part, make_root_engine, totals, world, rank, dev, group, root):
    """Step 3 of :func:`exchange_and_merge` for single-sample runs: every rank finalizes the key range it merged, the root
    places the finished rows.  One message of ``piece_words`` int32 per rank."""
    t = _mark(None, dev, 0.0)
    packed, n_ecs, nnz = part.finalize_range(*totals)
    t = _mark("finalize of my range", dev, t)
    mine = torch.tensor([n_ecs, nnz], dtype=torch.int64, device=dev)
    sz = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sz, mine, group=group)
    sz = sz.view(world, 2).cpu().tolist()
    t = _mark("all-gather of piece sizes", dev, t)
    if rank != root:
        if n_ecs:
            _p2p([dist.P2POp(dist.isend, packed, root, group)])
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
        return None
    bufs, ops = {}, []
    for r in range(world):
        if r == root or not sz[r][0]:
            continue
        bufs[r] = torch.empty(piece_words(*sz[r]), dtype=torch.int32, device=dev)
        ops.append(dist.P2POp(dist.irecv, bufs[r], r, group))
    _p2p(ops)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    t = _mark("pieces to the root", dev, t)
    merged = make_root_engine()
    merged.assemble_ranges([(packed if r == root else bufs[r], sz[r][0], sz[r][1]) for r in range(world) if sz[r][0]], *totals)
    _mark("assembly on the root (incl. its handle's reset)", dev, t)
    return merged


def exchange_multisample(engine, merged, n_ecs, group=None, root=0):
    """Multisample (``bam_utils_multisample.py``) after :func:`exchange_and_merge` and the root's finalize: the per-(EC, cell,
    file) read counts, i.e. the merge of the workers' ``ec[key][cell]`` (``:503-576``).  The root broadcasts the final ECs in
    rank order (hash + CSR A row: 110 MB at config 3); every rank finds its own ECs in that list -- by hash, then key against
    row, so the match is exact -- and reduces its reads to distinct (global EC, cell, file) triples; the triples go to the
    root, which adds up the ones a cell has on two ranks.
    ``merged`` / ``n_ecs``: the finalized root engine and its EC count on the root, ignored elsewhere.  Returns the number
    of distinct triples on the root (then ``export_pairs`` works as on one GPU), None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = engine.device
    nreads = engine.table_sizes()[2]
    keys, nnz = merged.ec_keys(n_ecs) if rank == root else (None, 0)
    info = torch.tensor([nreads, n_ecs if rank == root else 0, nnz], dtype=torch.int64, device=dev)
    allinfo = torch.empty(world * 3, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allinfo, info, group=group)
    allinfo = allinfo.view(world, 3).cpu().tolist()
    read_base = sum(a[0] for a in allinfo[:rank])
    n_ecs, nnz = allinfo[root][1], allinfo[root][2]
    if rank != root:
        keys = torch.empty(ec_keys_words(n_ecs, nnz), dtype=torch.int64, device=dev)
    dist.broadcast(keys, src=root, group=group)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()     # libecb works on its own stream: the keys must have landed
    key, cnt, first, n = engine.ms_local_triples(keys, n_ecs, nnz, read_base)
    mine = torch.tensor([n], dtype=torch.int64, device=dev)
    alln = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(alln, mine, group=group)
    alln = alln.cpu().tolist()
    if rank != root:
        if n:
            _p2p([dist.P2POp(dist.isend, t[:n], root, group) for t in (key, cnt, first)])
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
        return None
    bufs, ops = {}, []
    for r in range(world):
        if r == root or not alln[r]:
            continue
        bufs[r] = (torch.empty(alln[r], dtype=torch.int64, device=dev), torch.empty(alln[r], dtype=torch.int32, device=dev),
                   torch.empty(alln[r], dtype=torch.int32, device=dev))
        ops += [dist.P2POp(dist.irecv, t, r, group) for t in bufs[r]]
    _p2p(ops)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    tables = [((key, cnt, first) if r == root else bufs[r]) + (alln[r],) for r in range(world) if alln[r]]
    return merged.ms_adopt_triples(tables)


def reduce_ranges(range_min, range_max, group=None, device=None):
    """`--rangefile` across ranks (the reference's merge of per-process extremes, ``bam_utils.py:713-721``): all-reduce the
    per-(locus, haplotype) minima and maxima of ``reference_start`` (``EcBuilder.export_range_minmax`` of every rank) and
    return ``max - min + 1``, 0 where no rank saw an alignment (``bam_utils.py:756-763``).  Every rank gets the result."""
    import numpy as np
    dev = device if device is not None else torch.device("cpu")
    mn = torch.from_numpy(np.ascontiguousarray(range_min, dtype=np.int32)).to(dev)
    mx = torch.from_numpy(np.ascontiguousarray(range_max, dtype=np.int32)).to(dev)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    mn, mx = mn.cpu().numpy().astype(np.int64), mx.cpu().numpy().astype(np.int64)
    return np.where(mx >= mn, mx - mn + 1, 0)


def exchange_and_merge_on_root(engine, make_root_engine, group=None, root=0):
    """The simpler protocol (kept for comparison and as a fallback): every rank sends its whole table to the root, which
    merges them one after the other.  The root's merges are serial: at 8 ranks they cost more than a rank's own shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ne, npairs, nreads = engine.table_sizes()
    n_all, n_valid, _ = engine.counters()
    dev = engine.device
    mine = torch.tensor([ne, npairs, nreads, n_all, n_valid], dtype=torch.int64, device=dev)
    sizes = torch.empty(world * 5, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.view(world, 5).cpu().tolist()
    read_base = sum(s[2] for s in sizes[:rank])
    ent, prs = engine.table_export(read_base)
    if rank != root:
        if ne:
            ops = [dist.P2POp(dist.isend, ent[:ne * ENTRY_WORDS], root, group)]
            if npairs:
                ops.append(dist.P2POp(dist.isend, prs[:npairs * PAIR_WORDS], root, group))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return None
    bufs, ops, owner = {}, [], []
    for r in range(world):
        e_r, p_r = sizes[r][0], sizes[r][1]
        if r == root or not e_r:
            continue
        bufs[r] = (torch.empty(e_r * ENTRY_WORDS, dtype=torch.int64, device=dev),
                   torch.empty(max(p_r, 1) * PAIR_WORDS, dtype=torch.int64, device=dev))
        ops.append(dist.P2POp(dist.irecv, bufs[r][0], r, group))
        owner.append(r)
        if p_r:
            ops.append(dist.P2POp(dist.irecv, bufs[r][1][:p_r * PAIR_WORDS], r, group))
            owner.append(r)
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if len(reqs) != len(owner):                  # some backends hand back one request for the whole batch
        for req in reqs:
            req.wait()
        reqs, owner = [], []
    merged = make_root_engine()
    for r in range(world):                      # rank order = stream order; later tables keep arriving while we merge
        e_r, p_r = sizes[r][0], sizes[r][1]
        if not e_r:
            continue
        if r == root:
            merged.table_merge(ent, e_r, prs, p_r)
            continue
        for req, o in zip(reqs, owner):
            if o == r:
                req.wait()
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        merged.table_merge(bufs[r][0], e_r, bufs[r][1], p_r)
    merged.add_counters(sum(s[3] for s in sizes), sum(s[4] for s in sizes), sum(s[2] for s in sizes))
    return merged
