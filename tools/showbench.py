"""Print the interesting fields of bench.py JSON lines: python tools/showbench.py <log> ..."""
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l); r = d['roofline']
            print("%s: step %.2f ms  %.1f G/s  k_stream %.3f ms x%.1f  %.0f GB/s (%.1f %%)  ecs=%s exact=%s" % (
                f, d['ms_per_step'], d['value'] / 1e9, r['kernel_ms_per_launch'], r['launches_per_step'], r['achieved'], 100 * r['frac'],
                d['config'].get('ecs'), d['config'].get('exactness_pass')))
