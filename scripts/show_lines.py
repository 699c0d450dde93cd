"""Headline figures of the bench.py JSON lines in the files given (value, ms per step, second stage)."""
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'n', d['n_gpus'], 'value %.4g' % d['value'], 'ms %.4f' % d['ms_per_step'], 'second %.4g' % (d.get('second_pass') or {}).get('value',0), 'timing', d['timing'])
