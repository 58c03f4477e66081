#!/bin/bash
# A/B of library builds with one-call-alone times: tools/ab2.sh "<bench args>" lib1.so lib2.so ...  (interleaved, 2 rounds)
args="$1"; shift
for round in 1 2; do
for lib in "$@"; do
  RTMI_LIB=$PWD/$lib timeout -k 10 300 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s' % '$lib', 'ms %.3f' % d['ms_per_step'], 'alone %.3f' % r['call']['ms_alone'], 'Mrays/s %.0f' % d['value'], 'frac %.4f alone %.4f' % (r['call']['frac_in_flight'], r['call']['frac_alone']))"
done; done
