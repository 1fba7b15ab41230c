import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import cytvdn_amd as tv
shape = (256, 256, 128, 128)
x = bench.synth_host(shape, 0)
mu = np.array([1.0, 1.0, 0.5, 0.5], np.float32)
tv.denoise4D(x, mu, 4, quiet=True)
out = {}
for n in (50, 200):
    for tag, env, kw in (("no_rule", {}, {}), ("rule_plain_order", {"TVDN_PIPELINE": "0"}, {"stopping_relative_change": 1e-30}),
                         ("rule_pipelined_start", {}, {"stopping_relative_change": 1e-30}), ("rule_blocking_r5", {"TVDN_STOP_LAG": "0"}, {"stopping_relative_change": 1e-30})):
        os.environ.update(env)
        best = None
        for _ in range(2):
            t0 = time.perf_counter(); r = tv.denoise4D(x, mu, n, quiet=True, **kw); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        for k in env: del os.environ[k]
        out[f"{tag}_{n}_s"] = round(best, 4)
        out[f"{tag}_{n}_Gvoxel_iters_per_s"] = round(x.size * n / best / 1e9, 2)
        del r
print(json.dumps(out))
