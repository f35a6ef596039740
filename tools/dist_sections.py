#!/usr/bin/env python
"""GPU box: where a step of the N-GPU protocol spends its time on ONE rank, section by section (dist.SECTIONS), with the real
RCCL backend at world size 1: the collectives' fixed costs and the host's part show, the transfers do not.
usage: python tools/dist_sections.py [workload] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29513")
import torch
import torch.distributed as dist
import bench
from alntools_amd import ecb, synth, dist as ecdist

w = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
R, T, H, paired, _ = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
spec = synth.SynthSpec(R, T, H, paired=paired)
rid, loc, hf, st = bench.generate_shard(spec, 0, R, dev)
b = ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 24, arena_capacity=1 << 26)
part = ecdist.GpuEngine(ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 25, arena_capacity=1 << 26), dev)
root = ecdist.GpuEngine(ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 20, arena_capacity=1 << 20), dev)
eng = ecdist.GpuEngine(b, dev)


def make_part():
    part.b.reset()
    return part


def make_root():
    root.b.reset()
    return root


def step():
    b.reset()
    b.push_device(rid, loc, hf)
    return ecdist.exchange_and_merge(eng, make_part, make_root, root=0, finalize_ranges=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / steps
ecdist.SECTIONS = {}
push = 0.0
for _ in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.reset(); b.push_device(rid, loc, hf)
    torch.cuda.synchronize(); push += time.perf_counter() - t0
    ecdist.exchange_and_merge(eng, make_part, make_root, root=0, finalize_ranges=True)
print("%s, world 1 over RCCL: step %.2f ms (untimed sections); with a device wait at every section boundary:" % (w, plain * 1e3))
print("  %-52s %7.3f ms" % ("reset + push (k_stream)", push / steps * 1e3))
for k, v in ecdist.SECTIONS.items():
    print("  %-52s %7.3f ms" % (k, v / steps * 1e3))
print("  %-52s %7.3f ms" % ("sum", (push + sum(ecdist.SECTIONS.values())) / steps * 1e3))
dist.destroy_process_group()
