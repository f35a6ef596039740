# -*- coding: utf-8 -*-
"""Multi-GPU EC build: contiguous read shards, one table exchange, ordered merge.

The reference parallelises over contiguous chunk ranges per process and merges the
workers' ordered dicts in process order (``alntools/bam_utils.py:646-658, 680-724``), which
makes EC rank = global first appearance.  Here every rank builds the table of its own
contiguous read shard, the tables (tens of MB) are sent to the root over RCCL/xGMI (point to point),
and the root re-inserts them in rank order with ``first = read_base(rank) + local first``.

The functions only need an *engine* with ``table_sizes/counters/table_export/table_merge/
add_counters`` -- :class:`GpuEngine` wraps an :class:`alntools_amd.ecb.EcBuilder`; the CPU
``gloo`` tests plug in an oracle-backed engine.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

ENTRY_WORDS = 4          # one table entry = 32 bytes = 4 x int64
PAIR_WORDS = 1           # one (locus, mask) pair = 8 bytes = 1 x int64


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

    def table_merge(self, ent, n_entries, prs, n_pairs):
        self.b.table_merge_device(ent, n_entries, prs, n_pairs)

    def add_counters(self, a, v, r):
        self.b.add_counters(a, v, r)


def exchange_and_merge(engine, make_root_engine, group=None, root=0):
    """All ranks call this after pushing their shard.  Returns the merged engine on ``root`` (ready to finalize),
    ``None`` elsewhere.  One small all-gather (sizes), then every rank sends its table straight to the root
    (point-to-point over xGMI: the 7 peers of an 8-GPU node use 7 different links at once; a ring all-gather would
    push everything through every link)."""
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
