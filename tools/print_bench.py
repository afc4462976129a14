"""Reads a bench.py JSON line on stdin, prints the headline figures on one line."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%.0f %s  %.3f ms/step  roofline frac %.3f  avg launch %.1f us' % (d['value'], d['unit'], d['ms_per_step'], r['frac'], r.get('avg_launch_us', 0)))
for k, v in d.get('configs', {}).items():
    print('   %s: %s %s' % (k, v.get('value'), v.get('unit')))
