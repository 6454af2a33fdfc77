"""``Quaternions`` -- the reference's quaternion-array class (``Quaternions.py:4-506``) as a drop-in: same class name, constructor,
operators, properties, methods and class methods, over the functions of ``quaternions.py`` (which hold the arithmetic and the
reference's numerical quirks).  SURVEY.md 8b's last boundary row: ``Quaternions.from_euler(es).transforms()`` is how
``inverse_kinematics.foward_kinematics`` (:178-179) builds its local rotations.

Host-side NumPy, as in the reference; the device FK / IK kernels evaluate the same formulas themselves (``mvmc_common.h``).

Where the reference's text cannot run under Python 3 the evident intent is implemented and said so here:
``__div__`` is also bound to ``/`` (the reference defines only the Python-2 name, so ``q / r`` and ``q - r`` raise there);
``id`` accepts any integer (the reference tests ``long``, a NameError for a non-int); ``interpolate`` calls ``log`` (the reference
reads the bound method without calling it); ``average`` uses ``einsum`` for the removed ``numpy.core.umath_tests.matrix_multiply``;
``reshape`` keeps the reference's behaviour of returning ``self`` unchanged (:235-237 discards the reshaped array).
"""
from __future__ import annotations

import numpy as np

from . import quaternions as _q


class Quaternions:
    def __init__(self, qs):
        if isinstance(qs, np.ndarray):
            self.qs = qs[None] if qs.ndim == 1 else qs          # a single quaternion becomes a (1, 4) array (:26)
        elif isinstance(qs, Quaternions):
            self.qs = qs                                        # as the reference (:30-32): the object itself is stored
        else:
            raise TypeError("Quaternions must be constructed from iterable, numpy array, or Quaternions, not %s" % type(qs))

    def __str__(self):
        return "Quaternions(" + str(self.qs) + ")"

    def __repr__(self):
        return "Quaternions(" + repr(self.qs) + ")"

    @classmethod
    def _broadcast(cls, sqs, oqs, scalar=False):
        """:45-66 -- equal rank required; size-1 axes are repeated; a float spreads over the quaternion shape."""
        if isinstance(oqs, float):
            return sqs, oqs * np.ones(sqs.shape[:-1])
        ss = np.array(sqs.shape[:-1] if scalar else sqs.shape)
        os_ = np.array(oqs.shape)
        if len(ss) != len(os_) or not np.all((ss == os_) | (os_ == 1) | (ss == 1)):
            raise TypeError("Quaternions cannot broadcast together shapes %s and %s" % (sqs.shape, oqs.shape))
        if np.all(ss == os_):
            return sqs, oqs
        for ax in np.where(ss == 1)[0]:
            sqs = sqs.repeat(os_[ax], axis=ax)
        for ax in np.where(os_ == 1)[0]:
            oqs = oqs.repeat(ss[ax], axis=ax)
        return sqs, oqs

    # ---- operators (:68-160)
    def __add__(self, other):
        return self * other

    def __sub__(self, other):
        return self / other

    def __mul__(self, other):
        if isinstance(other, Quaternions):                      # Hamilton product
            a, b = Quaternions._broadcast(self.qs, other.qs)
            return Quaternions(_q.multiply(a, b))
        if isinstance(other, np.ndarray) and other.shape[-1] == 3:   # rotate 3-vectors; returns a plain array
            vs = Quaternions(np.concatenate([np.zeros(other.shape[:-1] + (1,)), other], axis=-1))
            return (self * (vs * -self)).imaginaries
        if isinstance(other, (np.ndarray, float)):              # scale: slerp from the identity
            return Quaternions.slerp(Quaternions.id_like(self), self, other)
        raise TypeError("Cannot multiply/add Quaternions with type %s" % str(type(other)))

    def __div__(self, other):
        if isinstance(other, Quaternions):
            return self * (-other)
        if isinstance(other, (np.ndarray, float)):
            return self * (1.0 / other)
        raise TypeError("Cannot divide/subtract Quaternions with type %s" % str(type(other)))

    __truediv__ = __div__

    def __eq__(self, other):
        return self.qs == other.qs

    def __ne__(self, other):
        return self.qs != other.qs

    __hash__ = None

    def __neg__(self):
        return Quaternions(_q.conjugate(self.qs))

    def __abs__(self):
        return Quaternions(_q.single_pole(self.qs))

    def __iter__(self):
        return iter(self.qs)

    def __len__(self):
        return len(self.qs)

    def __getitem__(self, k):
        return Quaternions(self.qs[k])

    def __setitem__(self, k, v):
        self.qs[k] = v.qs

    # ---- views (:174-237)
    @property
    def lengths(self):
        return _q.lengths(self.qs)

    @property
    def reals(self):
        return self.qs[..., 0]

    @property
    def imaginaries(self):
        return self.qs[..., 1:4]

    @property
    def shape(self):
        return self.qs.shape[:-1]

    def repeat(self, n, **kwargs):
        return Quaternions(self.qs.repeat(n, **kwargs))

    def normalized(self):
        return Quaternions(_q.normalized(self.qs))

    def copy(self):
        return Quaternions(np.copy(self.qs))

    def reshape(self, s):
        self.qs.reshape(s)
        return self

    def ravel(self):
        return self.qs.ravel()

    def dot(self, q):
        return np.sum(self.qs * q.qs, axis=-1)

    # ---- maps (:196-366)
    def log(self):
        return _q.log(self.qs)

    def constrained(self, axis):
        """:203-218 -- the rotation about ``axis`` closest to each quaternion (1-D arrays of quaternions)."""
        axis = np.asarray(axis, dtype=float)
        base = -2 * np.arctan2(self.reals, np.sum(axis * self.imaginaries, axis=-1))
        top = Quaternions.exp(axis[None] * ((base + np.pi)[:, None] / 2.0))
        bot = Quaternions.exp(axis[None] * ((base - np.pi)[:, None] / 2.0))
        pick = self.dot(top) > self.dot(bot)
        return Quaternions(np.where(pick[:, None], top.qs, bot.qs))

    def constrained_x(self):
        return self.constrained(np.array([1, 0, 0]))

    def constrained_y(self):
        return self.constrained(np.array([0, 1, 0]))

    def constrained_z(self):
        return self.constrained(np.array([0, 0, 1]))

    def interpolate(self, ws):
        return Quaternions(_q.interpolate(abs(self).qs, ws))

    def euler(self, order="xyz"):
        return _q.euler(self.qs, order)

    def average(self):
        """:310-322 -- the eigenvector of sum q q^T that is closest to all members (1-D arrays only)."""
        if len(self.shape) != 1:
            raise NotImplementedError("Cannot average multi-dimensionsal Quaternions")
        system = np.einsum("ni,nj->ij", self.qs, self.qs)
        _, v = np.linalg.eigh(system)
        proj = self.qs @ v
        return Quaternions(v[:, np.argmin((1.0 - proj ** 2).sum(axis=0))])

    def angle_axis(self):
        return _q.angle_axis(self.qs)

    def transforms(self):
        return _q.transforms(self.qs)

    # ---- constructors (:371-506)
    @classmethod
    def id(cls, n):
        if isinstance(n, tuple) or (isinstance(n, (int, np.integer)) and not isinstance(n, bool)):
            return Quaternions(_q.identity(n))
        raise TypeError("Cannot Construct Quaternion from %s type" % str(type(n)))

    @classmethod
    def id_like(cls, a):
        return Quaternions(_q.identity(tuple(a.shape)))

    @classmethod
    def exp(cls, ws):
        return Quaternions(_q.exp(ws))

    @classmethod
    def slerp(cls, q0s, q1s, a):
        """:408-434; unlike the reference, the rows of ``q1s`` are not negated in place."""
        fst, snd = cls._broadcast(q0s.qs, q1s.qs)
        fst, a = cls._broadcast(fst, a, scalar=True)
        snd, a = cls._broadcast(snd, a, scalar=True)
        return Quaternions(_q.slerp(fst, snd, a))

    @classmethod
    def between(cls, v0s, v1s):
        return Quaternions(_q.between(v0s, v1s))

    @classmethod
    def from_angle_axis(cls, angles, axis):
        return Quaternions(_q.from_angle_axis(angles, axis))

    @classmethod
    def from_euler(cls, es, order="xyz", world=False):
        return Quaternions(_q.from_euler(es, order, world))

    @classmethod
    def from_transforms(cls, ts):
        return Quaternions(_q.from_transforms(ts))
