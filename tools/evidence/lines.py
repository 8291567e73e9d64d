"""One line per bench JSON file: samples/s, ms/step, roofline frac, traffic, CPU port, strong proxy, inclusive fraction, walks' ms."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r, u = d.get('roofline') or {}, d.get('update_params_inclusive') or {}
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 4), 'frac', round(r.get('frac') or 0, 4), 'traffic', r.get('traffic'),
              'cpu', (d.get('cpu_baseline') or {}).get('value'), 'proxy', (d.get('strong_proxy') or {}).get('value'),
              'incl', round(u.get('fraction_of_step_rate') or 0, 3), 'walks_ms', (d.get('message_passing') or {}).get('ms_per_step'),
              'allreduce_ms', d.get('allreduce_ms'), 'ref_dims', {k: (v.get('value'), v.get('ms_per_step'), v.get('error')) for k, v in (d.get('ref_dims') or {}).items()})
    except Exception as e:
        print(f, 'FAILED', e)
