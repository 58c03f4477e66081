#!/bin/bash
# tools/sweep.sh "<common bench args>" "<variant args 1>" "<variant args 2>" ... : one bench line per variant
common="$1"; shift
for v in "$@"; do
  timeout -k 10 300 python bench.py $common $v --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('%-28s' % '$v', 'ms %.3f' % d['ms_per_step'], 'Mrays/s %.0f' % d['value'], 'samples %d' % c['samples_per_step'], 'disc %d' % c['discarded_samples_per_step'], 'rays %d' % c['rays_per_step'], 'frac %.4f' % d['roofline']['frac'])" || exit 1
done
