"""Bandwidth of the two OBF pre-pass kernels on a batch of 4K luma planes (run on the GPU box)."""
import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import __graft_entry__ as g
pkg = g.load_package()
import torch
from bench import gen_textured_gpu
W, H = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
luma = torch.stack([gen_textured_gpu(torch, dev, W, H, seed=7 + f)[0] for f in range(n)])
eng = pkg.CuEngine(W, H, max_chains=1)
best = [1e9, 1e9]
for _ in range(3):
    obf, yc, ms = eng.obf_prepass(luma)
    best = [min(best[0], ms[0]), min(best[1], ms[1])]
px = n * W * H
res = {"frames": n, "hist_ms": best[0], "count_ms": best[1],
       "hist_GBps": px / best[0] / 1e6, "count_GBps": (px + px / 8) / best[1] / 1e6,
       "hist_frac_of_8TBps": px / best[0] / 1e6 / 8000, "count_frac_of_8TBps": (px + px / 8) / best[1] / 1e6 / 8000,
       "obf_nonzero_share": float((obf != 0).float().mean())}
print(json.dumps(res))
