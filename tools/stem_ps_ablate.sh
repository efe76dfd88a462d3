# timing-only ablations of stem_ps_kernel (results are WRONG on purpose): PNVO_STEM_DBG = 16 + bits (pnvo_internal.h StemMXArgs::dbg)
for d in ${DBGS:-9 17 18 20 24 32 48 22 54}; do
  echo "== PNVO_STEM_DBG=$d"
  PNVO_STEM_DBG=$d timeout 120 python bench.py --steps 6 --warmup 2 --no-preheat --no-cpu-baseline --no-secondary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('stem ms', round(j['kernels'][0]['ms_per_step'],4) if 'conv1.0' in j['kernels'][0]['name'] else [k for k in j['kernels'] if 'conv1.0' in k['name']][0]['ms_per_step'])
    elif 'pnvo]' in l: print(l.rstrip())
"
done
