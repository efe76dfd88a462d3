"""Experiment: the headline batch as ONE forward of 256 pairs vs TWO forwards of 128 pairs on two streams (two handles with the same
weights): do the launch / ramp / tail bubbles of one stream's dependent launches fill with the other stream's work?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

DEV = torch.device("cuda", 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for nsplit in (1, 2, 4):
    models = [bench.build_model(DEV)[0] for _ in range(nsplit)]
    obs = bench.make_inputs(B, DEV, 0)
    parts = [{k: v[i * (B // nsplit):(i + 1) * (B // nsplit)].contiguous() for k, v in obs.items()} for i in range(nsplit)]
    streams = [torch.cuda.Stream(DEV) for _ in range(nsplit)]
    with torch.no_grad():
        ref = models[0](obs).clone()

        def step():
            for m, p, s in zip(models, parts, streams):
                with torch.cuda.stream(s):
                    m(p)
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(DEV))
        ms = timeit(step)
        outs = []
        for m, p, s in zip(models, parts, streams):
            with torch.cuda.stream(s):
                outs.append(m(p).clone())
        torch.cuda.synchronize()
    same = torch.equal(torch.cat(outs), ref)
    print(f"B={B} split into {nsplit} stream(s): {ms:.3f} ms per {B} pairs = {B / ms:.1f} k pairs/s, bit-identical to one forward: {same}", flush=True)
