import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_parity import build, load_golden
rec = load_golden(sys.argv[1] if len(sys.argv) > 1 else "model_default_341x192_b2.npz")
model, cfg, sd, obs, tobs, _, _ = build(rec)
with torch.no_grad():
    o1, a = model.tap("stem_conv", tobs)
    os.environ["PNVO_STEM"] = "dense"
    o2, b = model.tap("stem_conv", tobs)
a, b = a.cpu().numpy(), b.cpu().numpy()
d = np.abs(a - b)
print("max", d.max(), "ref max", np.abs(b).max())
bad = np.argwhere(d > 1e-3 * np.abs(b).max())
print("n bad", len(bad), "of", d.size)
if len(bad):
    print("first", bad[:10])
    print("rows", np.unique(bad[:, 1])[:40], "cols", np.unique(bad[:, 2])[:60], "ch", np.unique(bad[:, 3]), "n", np.unique(bad[:, 0]))
