"""Golden vectors of the reference's Quaternions class (Quaternions.py) by RUNNING THE REFERENCE (build container only; test
infrastructure).   PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_quat.py     Only data is written (tests/golden/quat_cases.npz)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    Q = ref_shim.load_modules().quat.Quaternions
    rng = np.random.default_rng(20260106)
    n = 64
    es = rng.uniform(-np.pi, np.pi, (n, 3))
    es[:4] = [[0, 0, 0], [np.pi / 2, 0, 0], [0, np.pi / 2 - 1e-3, 0], [1e-9, -1e-9, 0]]
    raw = rng.normal(size=(n, 4))
    raw2 = rng.normal(size=(n, 4))
    vs, vs2 = rng.normal(size=(n, 3)), rng.normal(size=(n, 3))
    ws = rng.normal(size=(n, 3)) * rng.uniform(0, 2, (n, 1))
    ws[0] = 0.0
    a = rng.uniform(0, 1, n)
    angles = rng.uniform(-np.pi, np.pi, n)
    q_e = Q.from_euler(es)
    qn, qn2 = Q(raw).normalized(), Q(raw2).normalized()
    qn_in, qn2_in = qn.qs.copy(), qn2.qs.copy()   # Quaternions.slerp negates rows of its SECOND argument in place (:419-421): keep the inputs
    near = Q(qn.qs + 1e-3 * raw2).normalized()          # slerp's linear branch
    out = dict(
        es=es, raw=raw, raw2=raw2, vs=vs, vs2=vs2, ws=ws, a=a, angles=angles,
        from_euler=q_e.qs, from_euler_world=Q.from_euler(es, world=True).qs, from_euler_zyx=Q.from_euler(es, order="zyx").qs,
        from_angle_axis=Q.from_angle_axis(angles, vs).qs,
        mul=(Q(raw) * Q(raw2)).qs, neg=(-Q(raw)).qs, rotate=qn * vs, normalized=qn_in, normalized2=qn2_in, lengths=Q(raw).lengths, abs=abs(Q(raw)).qs,
        transforms=q_e.transforms(), transforms_raw=Q(raw).transforms(), from_transforms=Q.from_transforms(q_e.transforms()).qs,
        log=Q(raw).log(), exp=Q.exp(ws).qs, slerp=Q.slerp(Q(qn_in.copy()), Q(qn2_in.copy()), a).qs, near=near.qs.copy(), slerp_near=Q.slerp(Q(qn_in.copy()), Q(near.qs.copy()), a).qs,
        scale=(Q(qn_in.copy()) * a).qs, euler=q_e.euler(), euler_raw=Q(raw).euler(), between=Q.between(vs, vs2).qs,
        id5=Q.id(5).qs, id23=Q.id((2, 3)).qs)
    ang, ax = Q(raw).angle_axis()
    out["angle_axis_angles"], out["angle_axis_axes"] = ang, ax
    np.savez_compressed(os.path.join(OUT, "quat_cases.npz"), **out)
    print("wrote quat_cases.npz:", ", ".join(sorted(out)))


if __name__ == "__main__":
    main()
