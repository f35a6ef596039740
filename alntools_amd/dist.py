# -*- coding: utf-8 -*-
"""Multi-GPU EC build: contiguous read shards, one table exchange, ordered merge.

The reference parallelises over contiguous chunk ranges per process and merges the
workers' ordered dicts in process order (``alntools/bam_utils.py:646-658, 680-724``), which
makes EC rank = global first appearance.  Here every rank builds the table of its own
contiguous read shard, the tables (tens of MB) are all-gathered once over RCCL/xGMI, and
the root re-inserts them in rank order with ``first = read_base(rank) + local first``.

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
    """All ranks call this after pushing their shard.  Returns the merged engine on ``root``
    (ready to finalize), ``None`` elsewhere.  One all-gather of sizes, one of tables."""
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
    max_e = max(max(s[0] for s in sizes), 1) * ENTRY_WORDS
    max_p = max(max(s[1] for s in sizes), 1) * PAIR_WORDS
    send = torch.zeros(max_e + max_p, dtype=torch.int64, device=dev)
    send[:ne * ENTRY_WORDS] = ent[:ne * ENTRY_WORDS]
    send[max_e:max_e + npairs * PAIR_WORDS] = prs[:npairs * PAIR_WORDS]
    recv = torch.empty(world * (max_e + max_p), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    if rank != root:
        return None
    merged = make_root_engine()
    recv = recv.view(world, max_e + max_p)
    for r in range(world):                      # rank order = stream order
        e_r, p_r = sizes[r][0], sizes[r][1]
        if e_r:
            merged.table_merge(recv[r, :max_e].contiguous(), e_r, recv[r, max_e:].contiguous(), p_r)
    merged.add_counters(sum(s[3] for s in sizes), sum(s[4] for s in sizes), sum(s[2] for s in sizes))
    return merged
