"""Diagnostic: cycle shares of the IK kernel's phases (needs the -DMVMC_IK_PROFILE build:
MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_prof.so python tools/ik_phase_profile.py).
Shares only -- the instrumented build's run time is not quoted anywhere."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth  # noqa: E402
from multiview_motion_capture_amd.pipeline import HotPath  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 500
data = synth.generate(F, 5, 4, 20260103)
hp = HotPath(data["K"], data["Rt"])
kps = torch.from_numpy(data["kps25"]).cuda()
cnt = torch.from_numpy(data["counts"]).cuda()
assoc = hp.associate(kps, cnt)
tri = hp.triangulate(assoc)
out = hp.solve_cold(assoc, tri)
torch.cuda.synchronize()
inf = out["info"].reshape(-1, 8).cpu().numpy()
ok = ~np.isnan(inf[:, 1])
inf = inf[ok]
tot = inf[:, 5].sum()
# profile build: info = {eval, gram, tridiag-phase total, model, krylov tridiagonalisation, total, block checks, tr-solve} cycles
print("solves", len(inf))
ev, gram, tri_all, ne, kry, chk, trs = (inf[:, k].sum() for k in (0, 1, 2, 3, 4, 6, 7))
print("cycles per solve (mean): total %.0f" % inf[:, 5].mean())
print("shares: eval %.3f model %.3f [gram %.3f krylov %.3f checks %.3f] tr-solve %.3f other %.3f" %
      (ev / tot, ne / tot, gram / tot, kry / tot, chk / tot, trs / tot, 1 - (ev + ne + tri_all + trs) / tot))
