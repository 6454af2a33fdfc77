"""Diagnostic: where a Householder step of the IK tridiagonalisation spends its cycles (GPU).
Build: hipcc ... -DMVMC_IK_PROFILE -DMVMC_TRI_PROFILE (make -C multiview_motion_capture_amd/csrc prof-tri), then
MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_triprof.so python tools/tri_step_profile.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import run_chains
L, B = 16, 128
data = synth.generate(B * L, 5, 4, 20260103, chain_len=L)
hp = HotPath(data["K"], data["Rt"])
out = run_chains(hp, torch.from_numpy(data["kps25"]).cuda(), torch.from_numpy(data["counts"]).cuda(), L, want_info=True)
torch.cuda.synchronize()
info = out["ik_info"].cpu().numpy()[:, 1:].reshape(-1, 8)
info = info[~np.isnan(info[:, 5])]
# slots (ik1_solve's profile mapping): info[0] = prof[0], [7] = prof[3], [1] = prof[4], [6] = prof[6], [4] = prof[5] (tridiag), [2] = prof[2] (model fn)
a, b, c, d, tri, model, tot = (info[:, k].sum() for k in (0, 7, 1, 6, 4, 2, 5))
print("warm solves %d: tridiag + pack %.3f of the solve, model function %.3f" % (len(info), tri / tot, model / tot))
print("inside the Householder steps: row + reductions + reflector scalars %.3f | v broadcast + mat-vec %.3f | h reduction + w broadcast %.3f | rank-2 update %.3f"
      " (of their sum = %.3f of tridiag + pack)" % (a / (a + b + c + d), b / (a + b + c + d), c / (a + b + c + d), d / (a + b + c + d), (a + b + c + d) / tri))
