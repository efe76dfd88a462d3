"""PNVO_STEM_DBG=9 python tools/prof_stem_raw.py [B]: phase cycles of the stem on the sensor-frame entry (pnvo_forward_raw)."""
import os, sys
os.environ.setdefault("PNVO_STEM_DBG", "9")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model, sd = bench.build_model(dev)
obs = bench.make_inputs(B, dev, 0)
rgb_f, dep_f = bench.frames_of(obs)
with torch.no_grad():
    for _ in range(8):
        model.forward_raw(rgb_f, dep_f, obs["top_down_view"])
torch.cuda.synchronize()
del model
