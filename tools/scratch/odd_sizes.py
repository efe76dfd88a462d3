import subprocess, sys, os, numpy as np
sys.path.insert(0, "/root/repo/tests")
import importlib.util
spec = importlib.util.spec_from_file_location("k", "/root/repo/tests/test_gpu_knobs.py"); k = importlib.util.module_from_spec(spec); spec.loader.exec_module(k)
res = {}
for name, env in (("x3", {"PNVO_CONV": "x3"}), ("fp32", {"PNVO_CONV": "fp32"}), ("auto", {}), ("x3nok", {"PNVO_CONV": "x3", "PNVO_X3_KSPLIT": "off"})):
    r = subprocess.run([sys.executable, "-c", k.SIZES], env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
    res[name] = [[float(x) for x in ln.split()[3:]] for ln in r.stdout.splitlines() if ln.startswith("OUT")]
    print(name, r.returncode, r.stderr[-300:] if r.returncode else "")
for other in ("x3", "auto", "x3nok"):
    for i, (a, b) in enumerate(zip(res[other], res["fp32"])):
        a, b = np.array(a).reshape(-1, 3), np.array(b).reshape(-1, 3)
        err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-2)
        print(other, i, err)
