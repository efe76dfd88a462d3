import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pointnav_vo_amd import synth
from pointnav_vo_amd.policy import PointNavResNetPolicy, policy_state_dict_spec
import bench_navloop as bn
dev = torch.device("cuda", 0)
W, H = 341, 192
space = bn.Space({"depth": bn.Box((H, W, 1)), "pointgoal_with_gps_compass": bn.Box((2,))})
pol = PointNavResNetPolicy(observation_space=space, action_space=bn.Act(), hidden_size=512, rnn_type="LSTM",
                           num_recurrent_layers=2, backbone="resnet18", vis_types=["depth"])
psd = synth.make_state_dict(policy_state_dict_spec(width=W, height=H), seed=0)
pol.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in psd.items()})
pol = pol.to(dev).eval()
for E in (8, 16, 32):
    depth = torch.rand(E, H, W, 1, device=dev)
    obs = {"depth": depth, "pointgoal_with_gps_compass": torch.rand(E, 2, device=dev)}
    hid = torch.zeros(pol.num_recurrent_layers, E, 512, device=dev)
    prev_a = torch.zeros(E, 1, dtype=torch.long, device=dev)
    masks = torch.ones(E, 1, device=dev)
    for _ in range(5):
        pol.act(obs, hid, prev_a, masks)
    torch.cuda.synchronize()
    def timeit(fn, n=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        th = time.perf_counter() - t0
        torch.cuda.synchronize(); tt = time.perf_counter() - t0
        return 1e3 * th / n, 1e3 * tt / n
    a = timeit(lambda: pol.act(obs, hid, prev_a, masks))
    b = timeit(lambda: pol._net(obs, hid, prev_a, masks))
    c = timeit(lambda: pol._ensure(dev))
    print(f"E={E}: act host {a[0]:.3f} total {a[1]:.3f} ms | _net host {b[0]:.3f} total {b[1]:.3f} | _ensure host {c[0]:.3f}")
