"""Host-side 4x4 transforms for the scene front-end.

Same public names as the reference's `matrix.py` (matrix/inverse/transpose/
matrix_multiply/translation/scaling/rotation_*/shearing and `Transform.from_yaml`,
reference matrix.py:42-132, 350-525).  Inversion happens here, on the host, once per
scene (np.linalg.inv like reference matrix.py:393); the device only ever sees the
pre-inverted, collapsed 3x4 affine parts the scene compiler emits.
"""
import numpy as np

from .vector import Tuple4, _arr


class Matrix(object):
    """fp64 square matrix; `.arr` is the ndarray (same attribute the reference exposes)."""
    __slots__ = ("arr",)

    def __init__(self, arr):
        self.arr = np.array(arr, dtype=np.float64)

    @property
    def shape(self):
        return self.arr.shape

    def compare(self, other):
        return bool(np.all(np.isclose(self.arr, _arr(other))))

    def __getitem__(self, i):
        return self.arr[i]

    def __mul__(self, b):
        return matrix_multiply(self, b)

    __matmul__ = __mul__

    def __repr__(self):
        return repr(self.arr)


def matrix(*args):
    n = {4: 2, 9: 3, 16: 4}.get(len(args))
    if n is None:
        return None
    return Matrix(np.array(args, dtype=np.float64).reshape(n, n))


_EYE4 = np.eye(4)


def matrix4x4identity():
    m = Matrix.__new__(Matrix)          # every shape gets its own (an OBJ mesh: one per triangle), so skip the conversions
    m.arr = _EYE4.copy()
    return m


def matrix_multiply(a, b):
    out = np.matmul(_arr(a), _arr(b))
    if out.ndim == 1:
        return Tuple4(out)
    return Matrix(out)


def transpose(a):
    return Matrix(_arr(a).T)


def determinant(a):
    return np.linalg.det(_arr(a))


def is_invertible(a):
    return determinant(a) != 0.0


def inverse(a):
    if not is_invertible(a):
        raise ValueError("Matrix supplied is not invertible.")
    return Matrix(np.linalg.inv(_arr(a)))


def translation(x, y, z):
    m = np.eye(4)
    m[0:3, 3] = (x, y, z)
    return Matrix(m)


def scaling(x, y, z):
    return Matrix(np.diag([x, y, z, 1.0]))


def rotation_x(rad):
    c, s = np.cos(rad), np.sin(rad)
    return matrix(1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1)


def rotation_y(rad):
    c, s = np.cos(rad), np.sin(rad)
    return matrix(c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1)


def rotation_z(rad):
    c, s = np.cos(rad), np.sin(rad)
    return matrix(c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1)


def shearing(xy, xz, yx, yz, zx, zy):
    return matrix(1, xy, xz, 0, yx, 1, yz, 0, zx, zy, 1, 0, 0, 0, 0, 1)


class Transform(object):
    """YAML transform lists -> one matrix (reference matrix.py:110-132).

    The list is applied first-to-last to the object, i.e. each entry is
    left-multiplied onto the accumulated matrix.
    """
    _OPS = {
        "translate": (translation, 3),
        "scale": (scaling, 3),
        "rotate-x": (rotation_x, 1),
        "rotate-y": (rotation_y, 1),
        "rotate-z": (rotation_z, 1),
        "shear": (shearing, 6),
    }

    translate = staticmethod(translation)
    scale = staticmethod(scaling)
    rotate_x = staticmethod(rotation_x)
    rotate_y = staticmethod(rotation_y)
    rotate_z = staticmethod(rotation_z)
    shear = staticmethod(shearing)

    @classmethod
    def from_yaml(cls, obj):
        acc = np.eye(4)
        for entry in obj:
            op = cls._OPS.get(entry[0])
            if op is None:
                raise ValueError("Unknown translation type: {}".format(entry[0]))
            fn, nargs = op
            acc = np.matmul(fn(*entry[1:1 + nargs]).arr, acc)
        return Matrix(acc)
