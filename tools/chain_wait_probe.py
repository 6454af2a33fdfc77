"""Share of a chain's resident workgroup time spent BEFORE its frames begin (tables, pose-pair block, hand-over wait), with two
steps in flight -- needs the diagnostic build:  make EXTRA=-DMVMC_CHAIN_WAITPROF, MVMC_LIB_PATH=<that library>
  python tools/chain_wait_probe.py [frames] [steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import run_chains_fused

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = torch.device("cuda", 0)
data = synth.generate(F, 5, 4, 20260103, chain_len=16)
hp = HotPath(data["K"], data["Rt"], device=d)
kps = torch.from_numpy(data["kps25"]).to(d)
counts = torch.from_numpy(data["counts"]).to(d)
streams = [torch.cuda.Stream() for _ in range(3)]
outs = []
for overlap in (1, 2, 3):
    torch.cuda.synchronize()
    res = []
    for s in range(steps):
        st = streams[s % overlap]
        with torch.cuda.stream(st):
            res.append(run_chains_fused(hp, kps, counts, 16, want_info=True))
    torch.cuda.synchronize()
    pc = np.stack([r["phase_cycles"].cpu().numpy() for r in res[2:]])
    work, wait = pc[..., 6].sum(), pc[..., 7].sum()
    print(f"steps in flight {overlap}: work {pc[..., 6].mean() / 1e6:.1f} M cycles per chain, resident before the frames begin "
          f"{pc[..., 7].mean() / 1e6:.1f} M = {wait / (work + wait):.3f} of the resident time")
