[ -n "$SKIPTEST" ] || python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
for d in ${DBGS:-0 1 2 3}; do echo DBG=$d; PNVO_STEM_DBG=$d python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['pose_rel_err_vs_fp64_oracle'], j['kernels'][0]['name'], j['kernels'][0]['ms_per_step'])
"; done
