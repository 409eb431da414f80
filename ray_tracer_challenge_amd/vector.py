"""Host-side 4-tuples (points, vectors) for the scene front-end.

Mirrors the names of the reference's `vector.py` (point/vector/dot/cross/normalize/
magnitude, reference vector.py:81-201) so that scene-building code written against the
reference reads the same here.  These objects only ever live on the host: the scene
compiler reads their `.arr` and the device never sees them.
"""
import numpy as np


class Tuple4(object):
    """fp64 (x, y, z, w); w = 1 for points, 0 for vectors.  `.arr` is the ndarray."""
    __slots__ = ("arr",)

    def __init__(self, arr):
        self.arr = np.asarray(arr, dtype=np.float64)

    @property
    def shape(self):
        return self.arr.shape

    def compare(self, other):
        return bool(np.all(np.isclose(self.arr, _arr(other))))

    def __getitem__(self, i):
        return self.arr[i]

    def __len__(self):
        return len(self.arr)

    def __add__(self, o):
        return Tuple4(self.arr + _arr(o))

    def __sub__(self, o):
        return Tuple4(self.arr - _arr(o))

    def __mul__(self, s):
        return Tuple4(self.arr * s)

    __rmul__ = __mul__

    def __truediv__(self, s):
        return Tuple4(self.arr / s)

    def __neg__(self):
        return Tuple4(-self.arr)

    def __repr__(self):
        return repr(self.arr)


def _arr(x):
    """ndarray view of a Tuple4, a reference ImmutablePointOrVector, or a sequence."""
    a = getattr(x, "arr", x)
    return np.asarray(a, dtype=np.float64)


def point_or_vector(x, y, z, w):
    return Tuple4([x, y, z, w])


def point(x, y, z):
    return Tuple4([x, y, z, 1.0])


def vector(x, y, z):
    return Tuple4([x, y, z, 0.0])


def magnitude(v):
    a = _arr(v)
    if a.shape != (4,) or a[3] != 0.0:
        raise ValueError("Only use this function with vectors.")
    return np.sqrt(np.sum(np.square(a)))


def normalize(v):
    # reciprocal-multiply, as the reference does (vector.py:170)
    return Tuple4(_arr(v) * (1.0 / magnitude(v)))


def dot(a, b):
    return np.vdot(_arr(a), _arr(b))


def cross(a, b):
    a = _arr(a)
    b = _arr(b)
    return vector(a[1] * b[2] - a[2] * b[1],
                  a[2] * b[0] - a[0] * b[2],
                  a[0] * b[1] - a[1] * b[0])
