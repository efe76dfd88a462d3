# phase cycles of conv_rows32_kernel: rebuild conv_rows.o with the ablation / profiling code compiled in, run, restore
set -e
cd pointnav-vo_amd/csrc
cp libpnvo.so /tmp/libpnvo.keep
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPNVO_ROWS_ABL=1 -c conv_rows.hip -o conv_rows.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o libpnvo.so
cd ../..
PNVO_ROWS_PROF=1 python - <<'PY' 2>&1 | grep "conv_rows32" | tail -12
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
obs = bench.make_inputs(256, dev, 0)
with torch.no_grad():
    for _ in range(3): model(obs)
torch.cuda.synchronize()
PY
for d in 0 1 2 4 8 15; do echo -n "PNVO_ROWS_DBG=$d "; PNVO_ROWS_DBG=$d python - <<'PY' 2>&1 | tail -1
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
model, _ = bench.build_model(dev)
obs = bench.make_inputs(256, dev, 0)
with torch.no_grad():
    for _ in range(3): model(obs)
    model.timing(True)
    for _ in range(10): model(obs)
    torch.cuda.synchronize()
    kt = {k["name"]: k["total_ms"] / 10 for k in model.timing_read()}
print({n[-16:]: round(v * 1e3, 1) for n, v in kt.items() if "layer1" in n and "convs.3" in n})
PY
done
cp /tmp/libpnvo.keep pointnav-vo_amd/csrc/libpnvo.so
rm pointnav-vo_amd/csrc/conv_rows.o
