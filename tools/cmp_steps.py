#!/usr/bin/env python
"""Compare two per-step timing dumps of bench.py --dump-steps: per kernel totals
and the steps whose time changed most."""
import json, sys
from collections import defaultdict
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
ka, kb = defaultdict(float), defaultdict(float)
for s in a: ka[s['kernel_name']] += s['ms']
for s in b: kb[s['kernel_name']] += s['ms']
print('total %.2f -> %.2f' % (sum(ka.values()), sum(kb.values())))
for k in sorted(set(ka) | set(kb), key=lambda k: -max(ka[k], kb[k])):
    if max(ka[k], kb[k]) > 0.05: print('  %-52s %7.3f -> %7.3f' % (k, ka[k], kb[k]))
d = sorted(zip(a, b), key=lambda p: -abs(p[0]['ms'] - p[1]['ms']))[: int(sys.argv[3]) if len(sys.argv) > 3 else 12]
for x, y in d:
    print('  #%d %-26s %-44s %.3f -> %.3f (%s)' % (x['step'], x['label'], x['kernel_name'], x['ms'], y['ms'], y['kernel_name'] if y['kernel_name'] != x['kernel_name'] else ''))
