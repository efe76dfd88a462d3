for w in 4 2 1; do echo "WPW=$w"; PNVO_CONV_WPW=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(round(j['value']), round(j['ms_per_step'],3), j['pose_rel_err_vs_fp64_oracle'])
        for k in sorted(j['kernels'], key=lambda k: k['name']):
            if 'layer4' in k['name'] or 'convs.0' in k['name'] and ('layer2.0' in k['name'] or 'layer3.0' in k['name']) or 'compression' in k['name'] or 'downsample' in k['name'] or 'fc' in k['name']: print('   ', k['name'][-28:], round(k['ms_per_step'],3), k['tflops'] and round(k['tflops'],1))
"; done
