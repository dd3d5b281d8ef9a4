"""Stage profile of the inter residual quadtree (diagnostic build -DFCU_PROFILE -DFCU_PROFILE_RQT as libfcu_prof_rqt.so): a
measurement script, not a test.  Decides picture 0 (intra) and picture 1 (P) of `clips` 4K clips and prints the share of the
P picture's CTU time spent in the stages of inter_tu_trials / est_inter_residual_qt / encode_res_and_calc_rd_inter_cu."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g

pkg = g.load_package()
pkg.engine._lib = None
pkg.engine.lib_path = lambda: os.path.join(os.path.dirname(pkg.engine.__file__), os.environ.get("FCU_LIB", "libfcu_prof_rqt.so"))
import torch
from bench import gen_moving_gpu

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 60
amp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shear = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W, H = 3840, 2160
dev = torch.device("cuda", 0)
dec = pkg.lowdelay.LowDelayPDecider(W, H, 32, n_clips=clips, search_range=64, slice_ctus=120, fast_search=1, amp=bool(amp))
for poc in range(2):
    frames = [gen_moving_gpu(torch, dev, W, H, seed=7 + c, poc=poc, shear=shear) for c in range(clips)]
    dec.decide_picture(frames)
names = ["tu_residual_fwd", "tu_rdoq", "tu_dequant_inverse_sse", "tu_variant_bits", "tu_choice_publish", "tu_whole_syntax", "subtree_syntax",
         "cu_syntax", "rqt_total", "-", "ctu_total",
         "rdoq_lane0_setup_tail", "rdoq_lane0_main_loop", "rdoq_lane0_last_pos", "rdoq_lane0_signs", "rdoq_lane0_sign_hiding"]
acc = np.zeros(17)
n = clips * 17
for c in range(0, n, max(1, n // 64)):
    acc += np.array(dec.eng.debug_counters(c), dtype=float)
tot = acc[10]
for i, nm in enumerate(names):
    print("%-24s %6.2f%%" % (nm, 100 * acc[i] / tot))
