"""Bit-level comparison of two builds of the C-ABI library on the bench workload (GPU):
    python tools/lib_diff.py libA.so libB.so [frames] [views] [people] [occlusion] [spurious]
Runs run_chains_fused with each library in a child process and compares params / joints / meta / info."""
import os, subprocess, sys, tempfile
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multiview_motion_capture_amd import synth
    from multiview_motion_capture_amd.pipeline import HotPath
    from multiview_motion_capture_amd.tracker import run_chains_fused, check_chain_flags
    F, Cv, Pp = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    occ, spu = (float(sys.argv[6]), float(sys.argv[7])) if len(sys.argv) > 7 else (0.0, 0.0)
    data = synth.generate(F, Cv, Pp, 20260103, chain_len=16, frame_seed=20260103, occlusion=occ, spurious=spu)
    hp = HotPath(data["K"], data["Rt"])
    kps = torch.from_numpy(data["kps25"]).cuda(); cnt = torch.from_numpy(data["counts"]).cuda()
    out = run_chains_fused(hp, kps, cnt, 16, want_info=True)
    torch.cuda.synchronize()
    if occ == 0.0:
        check_chain_flags(out)      # (with occlusions a chain may outgrow the layout's tables: both libraries void the same chains)
    np.savez(sys.argv[2], **{k: out[k].cpu().numpy() for k in ("params", "joints", "meta", "n_tracks", "ik_info")})
    sys.exit(0)

a, b = sys.argv[1], sys.argv[2]
F = sys.argv[3] if len(sys.argv) > 3 else "2048"
CP = [sys.argv[4] if len(sys.argv) > 4 else "5", sys.argv[5] if len(sys.argv) > 5 else "4"] + sys.argv[6:8]
res = []
for lib in (a, b):
    f = tempfile.mktemp(suffix=".npz")
    subprocess.run([sys.executable, __file__, "--child", f, F] + CP, env=dict(os.environ, MVMC_LIB_PATH=os.path.abspath(lib)), check=True)
    res.append(np.load(f))
for k in ("n_tracks", "meta", "params", "joints", "ik_info"):
    x, y = res[0][k], res[1][k]
    same = np.array_equal(x, y, equal_nan=True)
    m = np.isfinite(x) & np.isfinite(y)
    print(f"{k:9s} bit-identical {same}" + ("" if same else f"  max abs diff {np.abs(x[m].astype(float) - y[m].astype(float)).max():.3e}  differing {int((x != y)[m].sum())} of {m.sum()}"))
