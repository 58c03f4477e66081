for v in "RTMI_INLANE_NEED=192" "RTMI_INLANE_NEED=4096 RTMI_INLANE_MAX=4096" "RTMI_INLANE_NEED=4096 RTMI_INLANE_MAX=384" "RTMI_INLANE_NEED=1024 RTMI_INLANE_MAX=4096"; do
  env $v timeout -k 10 300 python bench.py --scene dof --steps 2 --warmup 1 --frames-in-flight 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print('$v', ' | '.join('%s %.1f ms' % (k[:28], v['ms']) for k,v in r.items()), 'disc', d['config']['discarded_samples_per_step'])" || exit 1
done
