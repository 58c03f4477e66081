#!/bin/bash
# tools/skip_ab.sh "<bench args>" : what a stage costs on a whole frame - the product build against the -DRTMI_SKIP=<bit> builds
# (raytracing_amd/librtmi_skip<bit>.so, made with  make -C raytracing_amd/csrc BUILD=build_skipN OUT=../librtmi_skipN.so EXTRA=-DRTMI_SKIP=N ../librtmi_skipN.so).
# Use a fixed-count workload (e.g. --max-samples 4: every pixel takes exactly 4 samples whatever the image), 2 rounds interleaved.
args="$1"
for round in 1 2; do
for lib in raytracing_amd/librtmi.so raytracing_amd/librtmi_skip*.so; do
  RTMI_LIB=$PWD/$lib timeout -k 10 200 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; k=[v for v in r.values()]
print('%-40s ms/step %.3f  first pass alone %.3f ms' % ('$lib', d['ms_per_step'], k[0]['ms']))"
done; done
