"""Stand-in for the un-vendored PyPI `noise` package so that the reference's `shapes.py`
imports in the build container (reference shapes.py:1496).  Only PerturbedPattern calls it,
no shipped scene uses that pattern, and its parity is recorded as unpinned."""


def pnoise3(*args, **kwargs):
    raise NotImplementedError("PyPI `noise` is not installed in this image")
