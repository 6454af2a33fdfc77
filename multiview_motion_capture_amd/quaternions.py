"""Quaternion helpers with the conventions of the reference's ``Quaternions`` class (Quaternions.py), as plain functions on
(..., 4) arrays in (w, x, y, z) order -- host-side NumPy, as in the reference.  The live path needs only
``from_euler`` + ``transforms`` (inverse_kinematics.py:178-179), which the device FK / IK kernels evaluate themselves; these functions
are the call surface for code that used the class directly.

The reference's numerical quirks are part of the contract and are kept: axes are normalised with ``+ 1e-10`` in the denominator
(:444), ``exp`` and ``angle_axis`` substitute 0.001 for a vanishing length (:396, :328), ``log`` divides by ``length + 1e-10`` (:200),
``slerp`` blends linearly when the quaternions are closer than 1 - cos = 0.01 (:423).
"""
from __future__ import annotations

import numpy as np

_AXES = {"x": np.array([1.0, 0.0, 0.0]), "y": np.array([0.0, 1.0, 0.0]), "z": np.array([0.0, 0.0, 1.0])}


def identity(shape) -> np.ndarray:
    """Quaternions.id / id_like: unit quaternions of the given leading shape (int or tuple)."""
    shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
    q = np.zeros(shape + (4,))
    q[..., 0] = 1.0
    return q


def multiply(q, r) -> np.ndarray:
    """Quaternions.__mul__ for two quaternion arrays (:96-115): the Hamilton product q r, broadcast over leading axes."""
    q, r = np.broadcast_arrays(np.asarray(q, dtype=float), np.asarray(r, dtype=float))
    qw, qx, qy, qz = (q[..., i] for i in range(4))
    rw, rx, ry, rz = (r[..., i] for i in range(4))
    return np.stack([rw * qw - rx * qx - ry * qy - rz * qz,
                     rw * qx + rx * qw - ry * qz + rz * qy,
                     rw * qy + rx * qz + ry * qw - rz * qx,
                     rw * qz - rx * qy + ry * qx + rz * qw], axis=-1)


def conjugate(q) -> np.ndarray:
    """Quaternions.__neg__ (:150-152): the inverse rotation of a unit quaternion."""
    return np.asarray(q, dtype=float) * np.array([1.0, -1.0, -1.0, -1.0])


def rotate(q, v) -> np.ndarray:
    """Quaternions.__mul__ with a (..., 3) vector array (:117-121): q (0, v) q*."""
    v = np.asarray(v, dtype=float)
    vq = np.concatenate([np.zeros(v.shape[:-1] + (1,)), v], axis=-1)
    return multiply(q, multiply(vq, conjugate(q)))[..., 1:]


def lengths(q) -> np.ndarray:
    return np.sqrt(np.sum(np.asarray(q, dtype=float) ** 2, axis=-1))


def normalized(q) -> np.ndarray:
    q = np.asarray(q, dtype=float)
    return q / lengths(q)[..., None]


def single_pole(q) -> np.ndarray:
    """Quaternions.__abs__ (:154-160): normalised, with the sign chosen so that w >= 0."""
    q = normalized(q).copy()
    flip = q[..., 0] < -q[..., 0]
    q[flip] = -q[flip]
    return q


def from_angle_axis(angles, axis) -> np.ndarray:
    """Quaternions.from_angle_axis (:442-447)."""
    angles, axis = np.asarray(angles, dtype=float), np.asarray(axis, dtype=float)
    axis = axis / (np.sqrt(np.sum(axis ** 2, axis=-1)) + 1e-10)[..., None]
    half = angles / 2.0
    c, s = np.cos(half)[..., None], np.sin(half)[..., None]
    return np.concatenate([c, np.broadcast_to(axis * s, c.shape[:-1] + (3,))], axis=-1)


def from_euler(es, order="xyz", world=False) -> np.ndarray:
    """Quaternions.from_euler (:449-462): q0 (q1 q2) for local axes, q2 (q1 q0) for world axes."""
    es = np.asarray(es, dtype=float)
    q0, q1, q2 = (from_angle_axis(es[..., k], _AXES[order[k]]) for k in range(3))
    return multiply(q2, multiply(q1, q0)) if world else multiply(q0, multiply(q1, q2))


def transforms(q) -> np.ndarray:
    """Quaternions.transforms (:335-366): (..., 3, 3) rotation matrices; the quaternion is used as it is (not normalised)."""
    q = np.asarray(q, dtype=float)
    w, x, y, z = (q[..., i] for i in range(4))
    x2, y2, z2 = x + x, y + y, z + z
    xx, yy, zz = x * x2, y * y2, z * z2
    xy, xz, yz = x * y2, x * z2, y * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    m = np.empty(q.shape[:-1] + (3, 3))
    m[..., 0, 0], m[..., 0, 1], m[..., 0, 2] = 1.0 - (yy + zz), xy - wz, xz + wy
    m[..., 1, 0], m[..., 1, 1], m[..., 1, 2] = xy + wz, 1.0 - (xx + zz), yz - wx
    m[..., 2, 0], m[..., 2, 1], m[..., 2, 2] = xz - wy, yz + wx, 1.0 - (xx + yy)
    return m


def from_transforms(ts) -> np.ndarray:
    """Quaternions.from_transforms (:464-506): magnitudes from the diagonal, signs relative to the largest component."""
    ts = np.asarray(ts, dtype=float)
    d0, d1, d2 = ts[..., 0, 0], ts[..., 1, 1], ts[..., 2, 2]
    mag = np.sqrt(np.clip(np.stack([d0 + d1 + d2 + 1.0, d0 - d1 - d2 + 1.0, -d0 + d1 - d2 + 1.0, -d0 - d1 + d2 + 1.0], axis=-1) / 4.0,
                          0.0, None))
    # sign sources: antisymmetric parts pair w with x/y/z, symmetric parts pair the imaginary components with each other
    a = np.stack([ts[..., 2, 1] - ts[..., 1, 2], ts[..., 0, 2] - ts[..., 2, 0], ts[..., 1, 0] - ts[..., 0, 1]], axis=-1)   # w-x, w-y, w-z
    s = np.stack([ts[..., 1, 0] + ts[..., 0, 1], ts[..., 0, 2] + ts[..., 2, 0], ts[..., 2, 1] + ts[..., 1, 2]], axis=-1)   # x-y, x-z, y-z
    q = mag.copy()
    big = [np.all(mag[..., k:k + 1] >= mag, axis=-1) for k in range(4)]
    # the reference applies the four cases one after the other on the running values (ties fall into several cases)
    c = big[0]
    q[c, 1] *= np.sign(a[c, 0]); q[c, 2] *= np.sign(a[c, 1]); q[c, 3] *= np.sign(a[c, 2])
    c = big[1]
    q[c, 0] *= np.sign(a[c, 0]); q[c, 2] *= np.sign(s[c, 0]); q[c, 3] *= np.sign(s[c, 1])
    c = big[2]
    q[c, 0] *= np.sign(a[c, 1]); q[c, 1] *= np.sign(s[c, 0]); q[c, 3] *= np.sign(s[c, 2])
    c = big[3]
    q[c, 0] *= np.sign(a[c, 2]); q[c, 1] *= np.sign(s[c, 1]); q[c, 2] *= np.sign(s[c, 2])
    return q


def log(q) -> np.ndarray:
    """Quaternions.log (:196-201): rotation vector / 2 of the single-pole form."""
    n = single_pole(q)
    im = n[..., 1:]
    ln = np.sqrt(np.sum(im ** 2, axis=-1))
    return im * (np.arctan2(ln, n[..., 0]) / (ln + 1e-10))[..., None]


def exp(ws) -> np.ndarray:
    """Quaternions.exp (:392-405)."""
    ws = np.asarray(ws, dtype=float)
    ts = np.sum(ws ** 2.0, axis=-1) ** 0.5
    ts = np.where(ts == 0, 0.001, ts)
    ls = np.sin(ts) / ts
    return normalized(np.concatenate([np.cos(ts)[..., None], ws * ls[..., None]], axis=-1))


def slerp(q0, q1, a) -> np.ndarray:
    """Quaternions.slerp (:407-434): shortest arc, linear blend when 1 - |<q0, q1>| < 0.01; a broadcasts over the leading axes."""
    q0, q1 = np.broadcast_arrays(np.asarray(q0, dtype=float), np.asarray(q1, dtype=float))
    q1 = q1.copy()
    a = np.broadcast_to(np.asarray(a, dtype=float), q0.shape[:-1]).astype(float)
    d = np.sum(q0 * q1, axis=-1)
    neg = d < 0.0
    d = np.where(neg, -d, d)
    q1[neg] = -q1[neg]
    lin = (1.0 - d) < 0.01
    w0, w1 = np.where(lin, 1.0 - a, 0.0), np.where(lin, a, 0.0)
    om = np.arccos(d[~lin])
    so = np.sin(om)
    w0[~lin] = np.sin((1.0 - a[~lin]) * om) / so
    w1[~lin] = np.sin(a[~lin] * om) / so
    return w0[..., None] * q0 + w1[..., None] * q1


def scale(q, a) -> np.ndarray:
    """Quaternions.__mul__ with scalars (:123-125): slerp from the identity."""
    q = np.asarray(q, dtype=float)
    return slerp(identity(q.shape[:-1]), q, a)


def euler(q, order="xyz") -> np.ndarray:
    """Quaternions.euler (:242-308), 'xyz' only (the reference raises for every other order)."""
    if order != "xyz":
        raise NotImplementedError("Cannot convert from ordering %s" % order)
    n = normalized(q)
    q0, q1, q2, q3 = (n[..., i] for i in range(4))
    es = np.zeros(n.shape[:-1] + (3,))
    es[..., 2] = np.arctan2(2 * (q0 * q3 - q1 * q2), q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3)
    es[..., 1] = np.arcsin((2 * (q1 * q3 + q0 * q2)).clip(-1, 1))
    es[..., 0] = np.arctan2(2 * (q0 * q1 - q2 * q3), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3)
    return es


def angle_axis(q):
    """Quaternions.angle_axis (:324-333) -> (angles, axes)."""
    n = normalized(q)
    s = np.sqrt(1 - n[..., 0] ** 2.0)
    s = np.where(s == 0, 0.001, s)
    return 2.0 * np.arccos(n[..., 0]), n[..., 1:] / s[..., None]


def between(v0, v1) -> np.ndarray:
    """Quaternions.between (:436-440): the rotation taking direction v0 to v1."""
    v0, v1 = np.asarray(v0, dtype=float), np.asarray(v1, dtype=float)
    w = np.sqrt((v0 ** 2).sum(axis=-1) * (v1 ** 2).sum(axis=-1)) + (v0 * v1).sum(axis=-1)
    return normalized(np.concatenate([w[..., None], np.cross(v0, v1)], axis=-1))


def interpolate(q, weights) -> np.ndarray:
    """Quaternions.interpolate (:239-240) as evidently intended: exp of the weighted mean of the logs along axis 0 (the reference's line
    reads the bound method ``log`` without calling it and raises)."""
    return exp(np.average(log(q), axis=0, weights=weights))
