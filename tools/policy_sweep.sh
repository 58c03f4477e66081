#!/bin/bash
# tools/policy_sweep.sh "<bench args>" "VAR=val VAR=val" "VAR=val" ... : one bench line per environment setting of the
# first-pass scheduling knobs (rtmi.hip launch_render: RTMI_KEEP_LANES, RTMI_INLANE_NEED, RTMI_INLANE_MAX, ...)
args="$1"; shift
for v in "$@"; do
  env $v timeout -k 10 300 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('%-60s' % '$v', 'ms %.3f' % d['ms_per_step'], 'alone %.3f' % r['call']['ms_alone'], 'Mrays/s %.0f' % d['value'], 'samples %d' % c['samples_per_step'], 'disc %d' % c['discarded_samples_per_step'], 'frac %.4f alone %.4f' % (r['call']['frac_in_flight'], r['call']['frac_alone']))" || exit 1
done
