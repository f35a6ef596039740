"""One cold push then one push on the kept table (needs the ECB_KEEP_TABLE hook).  For PMC: 2 k_stream dispatches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from alntools_amd import ecb, synth
R, T, H, paired, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
rid, loc, hf, st = bench.generate_shard(synth.SynthSpec(R, T, H, paired=paired), 0, R, dev)
b = ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 24, arena_capacity=1 << 26)
for keep in (0, 1):
    if keep: os.environ["ECB_KEEP_TABLE"] = "1"
    b.reset(); b.push_device(rid, loc, hf); torch.cuda.synchronize()
