#!/bin/bash
# tools/scenes_bench.sh : one bench line per BASELINE configuration at 3840x2160 (f32), CPU oracle timed beside it
mkdir -p gpurun_out/scenes
for spec in "ch07 1spp" "ch07 shipped" "ch11a 1spp" "ch11a shipped" "ch11b 1spp" "ch11b shipped" "teapot 1spp" "teapot shipped" "soft shipped" "dof shipped" "dice 1spp" "dice shipped"; do
  set -- $spec
  echo "== $1 $2" 
  timeout -k 10 280 python bench.py --scene $1 --sampling $2 --steps $( [ "$1" = dice ] && [ "$2" = shipped ] && echo 3 || echo 8 ) --warmup 2 --cpu-seconds 6 > gpurun_out/scenes/$1_$2.json 2> gpurun_out/scenes/$1_$2.err || { echo "FAILED $1 $2 rc=$?"; tail -3 gpurun_out/scenes/$1_$2.err; exit 1; }
  python - gpurun_out/scenes/$1_$2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']; b=d.get('cpu_baseline',{})
print('ms %.2f  Mrays/s %.0f  samples %d  rays %d  frac %.4f  cpu %.1f Mrays/s' % (d['ms_per_step'], d['value'], c['samples_per_step'], c['rays_per_step'], r['frac'], b.get('value',0)))
PY
done
