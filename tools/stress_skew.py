"""GPU box: degenerate streams at size -- every read in one of a handful of ECs (hot slots, hot count ranges)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from alntools_amd import ecb, synth
dev = torch.device("cuda", 0)
for R, T, H, paired in ((20_000_000, 2, 1, False), (20_000_000, 3, 2, True), (20_000_000, 5, 8, False)):
    spec = synth.SynthSpec(R, T, H, paired=paired)
    rid, loc, hf, st = bench.generate_shard(spec, 0, R, dev)
    with ecb.EcBuilder(T, H, device=0) as b:
        for it in range(2):
            b.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            b.push_device(rid, loc, hf); s = b.finalize(); dt = time.perf_counter() - t0
        bad = None
        b.reset(); b.push_device(rid, loc, hf); bad = b.verify_device(rid, loc, hf)
    assert s["n_reads"] == st["reads"] and s["valid_alignments"] == st["valid"], (s, st)
    print("reads %d loci %d haps %d paired %s: %d records, %d ECs, step %.2f ms = %.1f G records/s, verify %s" % (
        R, T, H, paired, st["records"], s["n_ecs"], dt * 1e3, st["records"] / dt / 1e9, bad), flush=True)
