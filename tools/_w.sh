for w in 0 1 2 3; do echo "STAGGER=$w"; PNVO_STAGGER=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']), round(j['ms_per_step'],3))
        for k in sorted(j['kernels'], key=lambda k: k['name']):
            if ('layer1' in k['name']) and 'down' not in k['name']: print('   ', k['name'][-28:], round(k['ms_per_step'],3), k['tflops'] and round(k['tflops'],1))
"; done
