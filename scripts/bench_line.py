"""Prints the headline figures of gpurun_out/bench_<workload>.log (one bench.py JSON line)."""
import json, sys
w = sys.argv[1]
d = json.loads(open('gpurun_out/bench_%s.log' % w).read().strip().splitlines()[-1])
lf = d.get('longest_first_dispatch') or d.get('index_order_dispatch') or {}
cb = d.get('cpu_baseline') or {}
print(w, '%.4g' % d['value'], d['unit'], 'ms/step', round(d['ms_per_step'], 4), 'frac', d['roofline'].get('frac'),
      'kernel_ms', round(d['roofline'].get('kernel_ms'), 4), 'cpu', round(cb.get('value', 0)),
      'ratio', round(d.get('speedup_vs_cpu_baseline', 0)), 'structured', round((d.get('cpu_structured') or {}).get('value', 0)),
      'other dispatch (hint %s)' % lf.get('dispatch_hint'), '%.4g' % lf.get('value', 0), round(lf.get('ms_per_step', 0), 4),
      'second pass', '%.4g' % (d.get('second_pass') or {}).get('value', 0))
