# phase cycles of stem_rs_kernel (option stem_form=resident, PNVO_STEM_DBG=9) at 256 pairs, tensor entry
PNVO_STEM_FORM=${FORM:-resident} PNVO_STEM_DBG=9 timeout 200 python bench.py --steps 6 --warmup 2 --no-preheat --no-cpu-baseline --no-secondary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('stem ms', [k for k in j['kernels'] if 'conv1.0' in k['name']][0]['ms_per_step'], 'value', j['value'])
    elif 'pnvo]' in l: print(l.rstrip())
"
