# Sweep one experiment knob of libpnvo over bench.py and print the step time plus the kernels whose name matches a filter.
# Usage: bash tools/knob_sweep.sh <ENV_VAR> "<value> <value> ..." [kernel-name-substring]
#   e.g. bash tools/knob_sweep.sh PNVO_CONV_TILE "0 11 12 21 22 14" layer4      (generic conv wave tile: 10*MT+NT)
#        bash tools/knob_sweep.sh PNVO_WAVE_WGS "2 3 4" layer2                   (wave-private conv: workgroups per CU)
#        bash tools/knob_sweep.sh PNVO_WAVE_NT "1 2" layer3                      (wave-private conv: n-tiles per item)
#        bash tools/knob_sweep.sh PNVO_CONV3_WGS "1 2 3" layer1                  (workgroup-tile conv: workgroups per CU)
var=$1; vals=$2; filt=${3:-conv}
for v in $vals; do
  echo "$var=$v"
  env $var=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | FILT=$filt python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('  ', round(j['value']), 'pairs/s', round(j['ms_per_step'], 3), 'ms')
        for k in sorted(j['kernels'], key=lambda k: k['name']):
            if os.environ['FILT'] in k['name']: print('     ', k['name'][-40:], round(k['ms_per_step'], 3), k['tflops'] and round(k['tflops'], 1))
"
done
