"""debug: random wide streams through a chosen lib; which (max_len, batch) fail, and what differs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import _random_stream
from alntools_amd import ecb
from oracle import c_oracle
print("lib", ecb.LIB_PATH)
for max_len in (60, 100, 200):
    for n_loci in (5000, (1 << 27) - 1):
        t = _random_stream(3, 3000, n_loci, 8, max_len, 0.05, "wide")
        exp = c_oracle.ec_from_tuples(t["read_id"], t["locus"], t["hapflag"], 8, threads=2)
        for batch in (None, 777):
            try:
                with ecb.EcBuilder(n_loci, 8) as b:
                    n = len(t["read_id"]); step = batch or n
                    for a in range(0, n, step):
                        sl = slice(a, a + step)
                        b.push(t["read_id"][sl], t["locus"][sl], t["hapflag"][sl])
                    s = b.finalize(); out = b.export()
                ok = all(np.array_equal(out[a], exp[k]) for a, k in (("indptrA", "indptr"), ("indicesA", "indices"), ("dataA", "data"), ("dataN", "count")))
                print(max_len, n_loci, batch, "E", s["n_ecs"], len(exp["count"]), "OK" if ok else "DIFF")
                if not ok:
                    ne = min(len(out["indptrA"]), len(exp["indptr"])) - 1
                    for e in range(ne):
                        ra = (out["indicesA"][out["indptrA"][e]:out["indptrA"][e + 1]].tolist(), out["dataA"][out["indptrA"][e]:out["indptrA"][e + 1]].tolist())
                        rb = (exp["indices"][exp["indptr"][e]:exp["indptr"][e + 1]].tolist(), exp["data"][exp["indptr"][e]:exp["indptr"][e + 1]].tolist())
                        if ra != rb:
                            print(" first differing EC", e, "len", len(ra[0]), len(rb[0])); print("  got", ra[0][:12], ra[1][:12]); print("  exp", rb[0][:12], rb[1][:12]); break
            except ecb.EcbError as ex:
                print(max_len, n_loci, batch, "ERR", ex)
