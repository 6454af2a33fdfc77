import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_np as o, trf_np as t
from multiview_motion_capture_amd import synth, device as dev
d = torch.device("cuda:0")
rng = np.random.default_rng(5)
K, Rt, P = synth.make_cameras(5, rng)
bd, side = o.skeleton_constants()
B = 4
kps = np.zeros((B, 5, 1, 17, 3)); init = np.zeros((B, 68))
for b in range(B):
    root = rng.normal(0, 0.05, 3) + np.array([0, 0, 0.05]); ang = rng.normal(0, 0.25, (18, 3))
    pos, _ = o.forward_kinematics(root, ang, side, bd)
    coco = np.zeros((17, 3))
    for sk_j, co_j in zip(o.REPROJ_SKEL_IDX, o.REPROJ_COCO_IDX): coco[co_j] = pos[sk_j]
    coco[1] = coco[2] = coco[0]
    for c in range(5):
        h = P[c] @ np.concatenate([coco, np.ones((17, 1))], axis=1).T
        kps[b, c, 0, :, :2] = (h[:2] / h[2]).T + rng.normal(0, 1.0, (17, 2)); kps[b, c, 0, :, 2] = rng.uniform(0.5, 1.0, 17)
    init[b] = np.concatenate([root + rng.normal(0, 0.02, 3), (ang * 0.2).ravel(), side])
mem = np.arange(B * 5, dtype=np.int32).reshape(B, 5)
np.set_printoptions(precision=5, suppress=True, linewidth=220)
for nfev in (1, 2):
    p, j, info = dev.ik_solve(torch.from_numpy(kps).to(d), torch.from_numpy(P).to(d), torch.from_numpy(mem).to(d),
                              torch.from_numpy(init).to(d), torch.zeros(B, dtype=torch.uint8, device=d), 50, nfev)
    p, info = p.cpu().numpy(), info.cpu().numpy()
    for b in range(B):
        obs = np.array([o.add_mid_spine(kps[b, c, 0]) for c in range(5)])[:, o.IK_OBS_IDX, :]
        f1 = lambda x: o.ik_residual(x[:3], x[3:].reshape(-1, 3), side, obs, P, bd)
        j1 = lambda x, f: t.ik_jacobian(x[:3], x[3:], side, obs, P, False)
        tr = []
        r1 = t.trf(f1, j1, init[b, :57], nfev, solver="ne", trace=tr)
        print("nfev", nfev, "b", b, "gpu cost1 %.6f cpu %.6f" % (info[b, 0], r1["cost"]), "trace", tr[:1])
        if nfev == 2:
            dx_g = p[b, :57] - init[b, :57]; dx_c = r1["x"] - init[b, :57]
            print("  |dx| gpu %.5f cpu %.5f  max|dx_g-dx_c| %.3e" % (np.linalg.norm(dx_g), np.linalg.norm(dx_c), np.abs(dx_g - dx_c).max()))
            print("  gpu dx", dx_g[:12]); print("  cpu dx", dx_c[:12])
