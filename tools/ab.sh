#!/bin/bash
# A/B bench of library builds: tools/ab.sh "<bench args>" lib1.so lib2.so ...   (interleaved, 2 rounds)
args="$1"; shift
for round in 1 2; do
for lib in "$@"; do
  RTMI_LIB=$PWD/$lib timeout -k 10 300 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'Mrays/s %.0f' % d['value'], 'ms %.3f' % d['ms_per_step'], 'frac %.4f' % d['roofline']['frac'])"
done; done
