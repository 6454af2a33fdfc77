"""The ``Quaternions`` class (SURVEY 8b's last boundary row) and ``kinematics.ForwardKinematics`` against vectors recorded by running
the reference (oracle/gen_golden_quat.py, oracle/gen_golden_kin.py)."""
import types

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import load_golden
from multiview_motion_capture_amd.Quaternions import Quaternions
from multiview_motion_capture_amd.kinematics import ForwardKinematics

TOL = 1e-13


@pytest.fixture(scope="module")
def g():
    return load_golden("quat_cases.npz")


def close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= tol, np.abs(a - b).max()


def test_the_live_path_call(g):
    """Quaternions.from_euler(es).transforms() -- inverse_kinematics.py:178-179."""
    close(Quaternions.from_euler(g["es"]).transforms(), g["transforms"])
    close(Quaternions.from_euler(g["es"], world=True).qs, g["from_euler_world"])
    close(Quaternions.from_euler(g["es"], order="zyx").qs, g["from_euler_zyx"])


def test_operators(g):
    q, r = Quaternions(g["raw"].copy()), Quaternions(g["raw2"].copy())
    close((q * r).qs, g["mul"])
    close((q + r).qs, g["mul"])                      # addition is multiplication (:68-69)
    close((-q).qs, g["neg"])
    close(abs(q).qs, g["abs"])
    close(q.lengths, g["lengths"])
    qn = q.normalized()
    close(qn.qs, g["normalized"])
    close(qn * g["vs"], g["rotate"])                 # vectors come back as a plain array
    close((qn * g["a"]).qs, g["scale"])              # scalars per quaternion: slerp from the identity
    close((q / r).qs, (q * -r).qs, 0.0)
    close((q - r).qs, (q * -r).qs, 0.0)
    assert q.shape == (64,) and len(q) == 64 and q.reals.shape == (64,) and q.imaginaries.shape == (64, 3)
    assert q[3:5].shape == (2,) and np.array_equal(q[7].qs, g["raw"][7:8])   # one quaternion is kept as a (1, 4) array
    assert (q == q).all() and not (q != q).any()
    q2 = q.copy()
    q2[0:2] = r[0:2]
    assert np.array_equal(q2.qs[:2], g["raw2"][:2]) and np.array_equal(q.qs, g["raw"])
    assert q.repeat(2, axis=0).shape == (128,) and q.ravel().shape == (256,)
    with pytest.raises(TypeError):
        q * "x"
    with pytest.raises(TypeError):
        Quaternions([1.0, 0, 0, 0])
    with pytest.raises(TypeError):
        q * Quaternions(g["raw"][:3].copy())         # shapes that do not broadcast


def test_maps_and_constructors(g):
    q = Quaternions(g["raw"].copy())
    close(q.transforms(), g["transforms_raw"])
    close(q.log(), g["log"])
    close(q.euler(), g["euler_raw"])
    ang, ax = q.angle_axis()
    close(ang, g["angle_axis_angles"])
    close(ax, g["angle_axis_axes"])
    close(Quaternions.from_transforms(g["transforms"]).qs, g["from_transforms"])
    close(Quaternions.from_angle_axis(g["angles"], g["vs"]).qs, g["from_angle_axis"])
    close(Quaternions.exp(g["ws"]).qs, g["exp"])
    close(Quaternions.between(g["vs"], g["vs2"]).qs, g["between"])
    close(Quaternions.id(5).qs, g["id5"])
    close(Quaternions.id((2, 3)).qs, g["id23"])
    close(Quaternions.id_like(q).qs, Quaternions.id(64).qs, 0.0)
    qn, qn2 = Quaternions(g["normalized"].copy()), Quaternions(g["normalized2"].copy())
    close(Quaternions.slerp(qn, qn2, g["a"]).qs, g["slerp"])
    close(Quaternions.slerp(qn, Quaternions(g["near"].copy()), g["a"]).qs, g["slerp_near"])
    with pytest.raises(NotImplementedError):
        q.euler(order="zyx")
    with pytest.raises(TypeError):
        Quaternions.id(2.5)


def test_constrained_and_average(g):
    """No recorded vectors (the reference's ``average`` imports a module NumPy 2 removed): properties instead."""
    qn = Quaternions(g["normalized"].copy())
    cz = qn.constrained_z()
    assert np.abs(cz.qs[:, 1:3]).max() < 1e-12 and np.allclose(cz.lengths, 1.0)     # a rotation about z
    # it is the closest rotation about z: no other z-rotation has a larger |dot|
    th = np.linspace(-np.pi, np.pi, 721)
    cand = np.stack([np.cos(th / 2), 0 * th, 0 * th, np.sin(th / 2)], axis=-1)
    assert (np.abs(qn.qs @ cand.T).max(axis=1) <= np.abs(qn.dot(cz)) + 1e-5).all()
    same = Quaternions(np.repeat(g["normalized"][:1], 5, axis=0) * np.array([[1], [-1], [1], [-1], [1]]))
    avg = same.average()
    assert abs(abs(float(avg.qs[0] @ g["normalized"][0])) - 1.0) < 1e-12            # the common rotation, up to sign


def test_forward_kinematics_of_kinematics_py():
    k = load_golden("kin_cases.npz")
    names = sorted({n.rsplit("_", 1)[0] for n in k.files})
    assert len(names) == 4
    for name in names:
        parents, off, rv, pos = (k[name + s] for s in ("_parents", "_offset", "_rotvec", "_pos"))
        fk = ForwardKinematics(types.SimpleNamespace(offset=off, topology=parents, chosen_joints=np.arange(len(parents))))
        assert fk.n_joints == len(parents)
        with np.errstate(invalid="ignore", divide="ignore"):
            got = np.stack([fk.forward([Rotation.from_rotvec(r) for r in rv[i]]) for i in range(len(rv))])
        assert np.array_equal(np.isnan(got), np.isnan(pos))       # parents[0] == 0: the root is (0, 0, 0) / 0
        assert np.nanmax(np.abs(got - pos)) <= 1e-13
        assert np.nanmax(np.abs(got[..., 2] - 1.0)) <= 1e-13      # divided by its own z (:30)
    # parents[0] == -1: the root's transform is local[last] @ local[0] (:25-27), so it moves when only the LAST joint's rotation changes
    parents, off, rv = k["chain_m1_parents"], k["chain_m1_offset"], k["chain_m1_rotvec"][1].copy()
    fk = ForwardKinematics(types.SimpleNamespace(offset=off, topology=parents, chosen_joints=None))
    a = fk.forward([Rotation.from_rotvec(r) for r in rv])
    rv[-1] += 0.3
    b = fk.forward([Rotation.from_rotvec(r) for r in rv])
    assert np.abs(a[1] - b[1]).max() > 1e-3
