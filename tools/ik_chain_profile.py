"""Diagnostic: cycle shares of the IK kernel's phases on the chain protocol (cold head + warm frames).
Needs the profile build: make -C multiview_motion_capture_amd/csrc prof-ik;
MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_ikprof.so python tools/ik_chain_profile.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth  # noqa: E402
from multiview_motion_capture_amd.pipeline import HotPath  # noqa: E402
from multiview_motion_capture_amd.tracker import run_chains  # noqa: E402

L, B = 16, 128
data = synth.generate(B * L, 5, 4, 20260103, chain_len=L)
hp = HotPath(data["K"], data["Rt"])
out = run_chains(hp, torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda(), L, want_info=True)
torch.cuda.synchronize()
info = out["ik_info"].cpu().numpy()  # (B, L, NP, 8)
for name, sl in (("cold head", info[:, 0]), ("warm", info[:, 1:])):
    inf = sl.reshape(-1, 8)
    inf = inf[~np.isnan(inf[:, 5])]
    tot = inf[:, 5].sum()
    ev, gram, model, kry, chk, trs = (inf[:, k].sum() for k in (0, 1, 2, 4, 6, 7))
    print("%s: solves %d, cycles per solve mean %.0f (p50 %.0f p95 %.0f max %.0f)" % (
        name, len(inf), inf[:, 5].mean(), *np.percentile(inf[:, 5], [50, 95, 100])))
    # the model function (ik1_model_step) = J^T J and g + tridiagonalisation and packing + block checks + the first trial step
    print("   shares: eval %.3f | model function %.3f = [J^T J, g %.3f | tridiag + pack %.3f | checks %.3f | trial (tr solve, Q) %.3f | rest %.3f]"
          " | other (driver, retry / fallback trials) %.3f" %
          (ev / tot, model / tot, gram / tot, kry / tot, chk / tot, trs / tot, (model - gram - kry - chk - trs) / tot, 1 - (ev + model) / tot))
