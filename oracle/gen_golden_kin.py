"""Golden vectors of the reference's kinematics.ForwardKinematics.forward (kinematics.py:11-31) by RUNNING THE REFERENCE (build
container only; test infrastructure).   PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_kin.py
Only data is written (tests/golden/kin_cases.npz): skeletons (offsets, parents), rotation vectors, and the positions returned."""
import os
import sys
import types

import numpy as np
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ref_shim.install()
    import kinematics as ref_kin
    rng = np.random.default_rng(20260107)
    out = {}
    # (name, parents): a chain and two trees, with the root's parent -1 (reads the last joint) and 0 (reads itself)
    skels = [("chain_m1", [-1, 0, 1, 2, 3]), ("tree_m1", [-1, 0, 0, 1, 1, 2, 5, 5]), ("tree_0", [0, 0, 0, 1, 2, 2, 4]),
             ("body_m1", [-1, 0, 1, 2, 3, 1, 5, 6, 1, 8, 9, 10, 8, 12, 13, 0, 15])]
    for name, parents in skels:
        J = len(parents)
        off = rng.normal(size=(J, 3)) * 0.3
        off[:, 2] += 1.5                      # keep z away from 0: the result is divided by it
        rv = rng.normal(size=(6, J, 3)) * 0.7
        rv[0] = 0.0
        skel = types.SimpleNamespace(offset=off, topology=np.array(parents), chosen_joints=np.arange(J))
        fk = ref_kin.ForwardKinematics(skel)
        pos = np.stack([fk.forward([Rotation.from_rotvec(r) for r in rv[i]]) for i in range(len(rv))])
        out[name + "_parents"], out[name + "_offset"], out[name + "_rotvec"], out[name + "_pos"] = np.array(parents), off, rv, pos
    np.savez_compressed(os.path.join(OUT, "kin_cases.npz"), **out)
    print("wrote kin_cases.npz:", ", ".join(sorted(out)))


if __name__ == "__main__":
    main()
