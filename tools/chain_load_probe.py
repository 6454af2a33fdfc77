"""Chain kernel at different loads: cycles per chain (mean / max) and wall time for B chains (256 CUs, 3 workgroup slots
each), and for several workgroups per chain (PARTS=1,2,4 in the environment)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth  # noqa: E402
from multiview_motion_capture_amd.pipeline import HotPath  # noqa: E402
from multiview_motion_capture_amd.tracker import run_chains_fused  # noqa: E402

L = 16
PARTS = [int(x) for x in os.environ.get("PARTS", "1").split(",")]
for B in [int(a) for a in sys.argv[1:]] or [256, 512, 625, 768]:
    data = synth.generate(B * L, 5, 4, 20260103, chain_len=L)
    hp = HotPath(data["K"], data["Rt"])
    kps = torch.from_numpy(data["kps25"]).cuda()
    cnt = torch.from_numpy(data["counts"]).cuda()
    ref = None
    for parts in PARTS:
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = run_chains_fused(hp, kps, cnt, L, want_info=True, parts=parts)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        pc = out["phase_cycles"].cpu().numpy()
        j = torch.nan_to_num(out["joints"])
        if ref is None:
            ref = j
        tmo = None if out["flags"] is None else int(out["flags"][B])
        print("B=%4d parts=%d  %.2f ms  %.0f frames/s  chain Mcycles mean %.1f p95 %.1f max %.1f  (ALS mean %.1f IK mean %.1f)  "
              "timeout flag %s, same results as parts=%d: %s" %
              (B, parts, dt * 1e3, B * L / dt, pc[:, 6].mean() / 1e6, np.percentile(pc[:, 6], 95) / 1e6, pc[:, 6].max() / 1e6,
               pc[:, 1].mean() / 1e6, pc[:, 3].mean() / 1e6, tmo, PARTS[0], bool(torch.equal(ref, j))), flush=True)
