"""Where does a device solve leave the noise-free oracle's sequence?  (test infrastructure; GPU box)

    python tools/replay_solve.py tests/golden/two_view_junk_cluster.npz    # poses (v, 17, 3), projs (v, 3, 4); a COLD solve (50 + 50)

Runs trf_np.pose_solver_solve_clean with traces, the device's whole solve (mvmc_ik_solve, cold), then re-makes every trial of the oracle's
sequence on the device (mvmc_debug_ik_model_step from the oracle's x_k, Delta_k, alpha_k) and prints the first trial whose accept / reject
decision or new radius differs, with the ratio's distance from the thresholds 0.25 / 0.75 that decide the radius."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main(path):
    import torch
    import oracle_np as o
    import trf_np as t
    from multiview_motion_capture_amd import device as dev
    z = np.load(path)
    poses, projs = z["poses"], z["projs"]
    v = len(poses)
    bd, ref_side = o.skeleton_constants()
    poses18 = [o.add_mid_spine(p) for p in poses]
    obs = np.array(poses18)[:, o.IK_OBS_IDX, :]
    warm = "init" in z.files       # a WARM solve: init (68) = the previous frame's parameters, budget 5 + 5
    budget = 5 if warm else 50
    if warm:
        x0, side0 = z["init"][:57].copy(), z["init"][57:].copy()
        root = x0[:3]
    else:
        p3d = o.triangulate_groups(projs, poses18, 0.01, True)
        root = 0.5 * (p3d[o.COCO_L_HIP, :3] + p3d[o.COCO_R_HIP, :3])
        x0 = np.concatenate([root, np.zeros(54)])
        side0 = ref_side.copy()
    f1 = lambda x: o.ik_residual(x[:3], x[3:57], side0, obs, projs, bd)
    j1 = lambda x, f: t.ik_jacobian(x[:3], x[3:57], side0, obs, projs, False)
    f2 = lambda x: o.ik_residual(x[:3], x[3:57], x[57:], obs, projs, bd)
    j2 = lambda x, f: t.ik_jacobian(x[:3], x[3:57], x[57:], obs, projs, True)
    tr1, tr2 = [], []
    r1 = t.trf(f1, j1, x0, budget, solver="ne_clean", trace=tr1)
    r2 = t.trf(f2, j2, np.concatenate([r1["x"], side0]), budget, solver="ne_clean", trace=tr2)
    print(f"oracle: stage 1 nfev {r1['nfev']} status {r1['status']} cost {r1['cost']:.6f}; stage 2 nfev {r2['nfev']} status {r2['status']} cost {r2['cost']:.6f}")
    d = torch.device("cuda:0")
    kps = np.zeros((1, v, 1, 17, 3))
    kps[0, :, 0] = poses
    mem = -np.ones((1, max(6, v)), dtype=np.int32)
    mem[0, :v] = np.arange(v)
    kps_t, cams_t, mem_t = torch.from_numpy(kps).to(d), torch.from_numpy(np.ascontiguousarray(projs)).to(d), torch.from_numpy(mem).to(d)
    init_t = torch.from_numpy(np.concatenate([x0, side0])[None]).to(d) if warm else torch.zeros((1, 68), dtype=torch.float64, device=d)
    p, j, info = dev.ik_solve(kps_t, cams_t, mem_t, init_t, torch.full((1,), 0 if warm else 1, dtype=torch.uint8, device=d), 50, 5)
    torch.cuda.synchronize()
    info, p = info.cpu().numpy()[0], p.cpu().numpy()[0]
    print(f"device: stage 1 nfev {int(info[1])} status {int(info[2])} cost {info[0]:.6f}; stage 2 nfev {int(info[4])} status {int(info[5])} cost {info[3]:.6f}; "
          f"models {int(info[6])}, eigensolver fallbacks {int(info[7])}")
    # (a) the starting point: the device's cold root (DLT + post-optimisation in the solve's wave) against the oracle's
    pos_o, _ = o.forward_kinematics(r2["x"][:3], r2["x"][3:57], r2["x"][57:], bd)
    print(f"joints: device against oracle {np.abs(j.cpu().numpy()[0] - pos_o).max():.1e} m")
    if not warm:
      p1, _, _ = dev.ik_solve(kps_t, cams_t, mem_t, torch.zeros((1, 68), dtype=torch.float64, device=d), torch.ones(1, dtype=torch.uint8, device=d), 1, 1)
      print(f"cold root: device {p1.cpu().numpy()[0, :3]} oracle {root}; difference {np.abs(p1.cpu().numpy()[0, :3] - root).max():.1e} m")
    # (b) free-running: stage 1 from the ORACLE's start with budgets 2 .. 50 against the oracle's iterate after as many evaluations
    xs, x = {1: x0.copy()}, x0.copy()
    for e in (e for e in tr1 if "model" not in e):
        if e["accepted"]:
            x = e["x"] + e["step"]
        xs[e["nfev"]] = x.copy()
    init = torch.from_numpy(np.concatenate([x0, side0])[None]).to(d)
    drift = []
    for n in range(2, r1["nfev"] + 1):
        pn, _, inf = dev.ik_solve_stages(init, 1, n, kps_t, cams_t, mem_t)
        drift.append(float(np.abs(pn.cpu().numpy()[0, :57] - xs[n]).max()))
    print("stage 1, free-running device against the oracle after n evaluations (max |dx|):")
    print("   " + " ".join(f"{n + 2}:{v:.0e}" for n, v in enumerate(drift)))
    for st, tr in ((0, tr1), (1, tr2)):
        trials = [e for e in tr if "model" not in e]
        models = [e for e in tr if "model" in e]
        weak = [float((e["lam"] / e["lam"][0])[((e["lam"] / e["lam"][0]) > 1e-13) & ((e["lam"] / e["lam"][0]) < 1e-6)].min()) for e in models
                if (((e["lam"] / e["lam"][0]) > 1e-13) & ((e["lam"] / e["lam"][0]) < 1e-6)).any()]
        print(f"stage {st + 1}: {len(trials)} trials, {len(models)} models, {len(weak)} with a weak eigenvalue (weakest {min(weak) if weak else None})")
        par = np.zeros((len(trials), 68))
        for k, e in enumerate(trials):
            par[k, :len(e["x"])] = e["x"]
            if st == 0:
                par[k, 57:] = side0
        r = dev.ik_model_step(kps_t, cams_t, mem_t.expand(len(trials), mem_t.shape[1]).contiguous(), torch.from_numpy(par).to(d), st,
                              torch.tensor([e["Delta"] for e in trials], dtype=torch.float64, device=d),
                              torch.tensor([e["alpha_in"] for e in trials], dtype=torch.float64, device=d))
        torch.cuda.synchronize()
        r = r.cpu().numpy()
        nn = 57 if st == 0 else 68
        shown = 0
        for k, e in enumerate(trials):
            actual = r[k, 0] - r[k, 5]
            pred, step_norm, Delta = r[k, 3], r[k, 4], e["Delta"]
            ratio = actual / pred if pred > 0 else (1.0 if (pred == 0 and actual == 0) else 0.0)
            Delta_new = 0.25 * step_norm if ratio < 0.25 else (2.0 * Delta if (ratio > 0.75 and step_norm > 0.95 * Delta) else Delta)
            same = ((actual > 0) == bool(e["accepted"])) and abs(Delta_new - e["Delta_new"]) <= 1e-9 * Delta
            ds = np.linalg.norm(r[k, 80:80 + nn] - e["step"]) / Delta
            if not same or ds > 1e-6:
                shown += 1
                if shown <= 6:
                    print(f"  trial {k} (nfev {e['nfev']}): device accept {actual > 0} ratio {ratio:.6f} Delta_new {Delta_new:.6g} | oracle accept "
                          f"{e['accepted']} ratio {e['ratio']:.6f} Delta_new {e['Delta_new']:.6g} | step difference / Delta {ds:.1e}, cost_new rel "
                          f"{abs(r[k, 5] - e['cost_new']) / e['cost_new']:.1e}, alpha {r[k, 2]:.6g} vs {e['alpha']:.6g}")
        print(f"  trials that differ (decision, radius, or step beyond 1e-6 Delta): {shown} of {len(trials)}")


if __name__ == "__main__":
    main(sys.argv[1])
