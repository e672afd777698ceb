for i in 1 2 3; do for f in eager graph; do python bench.py --config 5 --launch $f --no-cpu-baseline --no-live-traffic --no-extras --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=l['steady_state']
print('$f', 'W5 %.4f ms'%l['ms_per_step'], 'steady', {k:(round(v,4) if v else v) for k,v in s['launch_forms_ms'].items()})"; done; done
