"""GPU box: k_stream time on c3 with a cold table (every EC created) vs. a table that already holds every EC (all hits).
Needs a libecb built with the ECB_KEEP_TABLE debug hook in ecb_reset.  Push only, no finalize."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from alntools_amd import ecb, synth
w = sys.argv[1] if len(sys.argv) > 1 else "c3"
R, T, H, paired, _ = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
rid, loc, hf, st = bench.generate_shard(synth.SynthSpec(R, T, H, paired=paired), 0, R, dev)
b = ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 24, arena_capacity=1 << 26)
for keep in (0, 0, 1, 1, 1):
    if keep: os.environ["ECB_KEEP_TABLE"] = "1"
    b.reset()
    b.profile(True)
    b.push_device(rid, loc, hf)
    torch.cuda.synchronize()
    ms, n, _ = b.profile_read()
    b.profile(False)
    print("keep_table=%d k_stream %.3f ms (%d launches)" % (keep, ms, n), flush=True)
