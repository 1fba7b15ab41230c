import gc, json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cytvdn_amd as tv
from cytvdn_amd import synth
def rss_mib():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 20
shape = (64, 32, 256, 256)
x = synth.cube(shape, seed=1, dtype=np.float32) + np.float32(0.25)
mu = np.array([1, 1, .5, .5], np.float32)
for env in ({"TVDN_PIPELINE": "0"}, {}):
    os.environ.update(env)
    for _ in range(3):
        tv.denoise4D(x, mu, 6, quiet=True)
    base = rss_mib()
    for i in range(80):
        tv.denoise4D(x, mu, 6, quiet=True)
        if i % 10 == 9:
            gc.collect()
            print(json.dumps({"env": env, "calls": i + 1, "rss_growth_MiB": round(rss_mib() - base, 1)}), flush=True)
    for k in env: del os.environ[k]
