[ -n "$SKIPTEST" ] || python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']), round(j['ms_per_step'],3), j['pose_rel_err_vs_fp64_oracle'])
        tot = 0
        for k in j['kernels']:
            tot += k['ms_per_step']; print('   ', k['name'][-40:], round(k['ms_per_step'],3), k['tflops'] and round(k['tflops'],1))
        print('sum', round(tot,3))
"
