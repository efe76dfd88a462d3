# A/B the conv kernels: PNVO_CONV=generic vs LDS-staged (nt 1 / 2); prints per-layer ms.
[ -n "$SKIPTEST" ] || python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
for v in "PNVO_CONV=generic" "PNVO_CONV3_NT=1" "PNVO_CONV3_NT=2"; do echo "== $v"; env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']), round(j['ms_per_step'],3), j['pose_rel_err_vs_fp64_oracle'])
        for k in j['kernels']: print('   ', k['name'][-40:], round(k['ms_per_step'],3), k['tflops'] and round(k['tflops'],1))
"; done
