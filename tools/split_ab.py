"""Split BIG path (two co-resident persistent kernels) against the one-kernel BIG path on config-5 data: identical tables, then timing.
    python tools/split_ab.py [frames] [occlusion]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from multiview_motion_capture_amd import synth
from multiview_motion_capture_amd.pipeline import HotPath
from multiview_motion_capture_amd.tracker import check_chain_flags, run_chains_fused

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
occ = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
L = 16
data = synth.generate(F, 8, 8, 20260104, chain_len=L, occlusion=occ, spurious=0.2 if occ else 0.0)
d = torch.device("cuda:0")
hp = HotPath(data["K"], data["Rt"], device=d)
kps, cnt = torch.from_numpy(data["kps25"]).to(d), torch.from_numpy(data["counts"]).to(d)
a = run_chains_fused(hp, kps, cnt, L, want_info=True, split=False)
torch.cuda.synchronize()
check_chain_flags(a)
print("one kernel ok", flush=True)
b = run_chains_fused(hp, kps, cnt, L, want_info=True, split=True)
torch.cuda.synchronize()
check_chain_flags(b)
print("split ok", flush=True)
n = a["n_tracks"].cpu().numpy()
same = torch.equal(a["n_tracks"], b["n_tracks"]) and torch.equal(a["n_dead"], b["n_dead"]) and torch.equal(a["next_id"], b["next_id"])
for k in ("meta", "params", "joints"):
    x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
    bad = [f for f in range(F) if not np.array_equal(x[f, :n[f]], y[f, :n[f]], equal_nan=True)]
    same = same and not bad
    print(k, "frames differing:", len(bad), bad[:5])
print("als_iters equal:", torch.equal(a["als_iters"], b["als_iters"]), "ik_info equal:", torch.equal(torch.nan_to_num(a["ik_info"]), torch.nan_to_num(b["ik_info"])))
print("IDENTICAL" if same else "DIFFERENT")
for name, split in (("one kernel", False), ("split", True), ("one kernel", False), ("split", True)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = run_chains_fused(hp, kps, cnt, L, split=split)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {F / dt / 1e3:.1f} k frames/s ({dt * 1e3:.1f} ms per {F} frames)", flush=True)
pc = b["phase_cycles"].cpu().numpy()
print("split phase Mcycles per chain {graph, als, assign, ik, commit, out, total}:", (pc[:, :7].mean(0) / 1e6).round(2))
pa = a["phase_cycles"].cpu().numpy()
print("one-k phase Mcycles per chain:", (pa[:, :7].mean(0) / 1e6).round(2))
