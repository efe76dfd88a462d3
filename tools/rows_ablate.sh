# timing-only ablations of conv_rows32_kernel (WRONG results): which part of a half-step costs what
for d in 0 1 2 4 8 3 7 15 12; do
  echo -n "PNVO_ROWS_DBG=$d  "
  PNVO_ROWS_DBG=$d python - <<'PY'
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
