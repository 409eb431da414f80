"""Framebuffer and PPM encoding (drop-in for the reference's `canvas.py`).

Layout contract (reference canvas.py:57-64, renderer.py:835): a canvas is an fp64
array indexed `[x, y] -> rgb`, shape `(hsize, vsize, 3)`.  The renderer's device
kernels write exactly this layout, so a frame copies straight into a `Canvas`.

PPM contract (reference canvas.py:116-181): header `P6\\n{w} {h}\\n255\\n`, rows y-outer /
x-inner, per component `>= max -> 255`, `<= 0 -> 0`, else `int(c * 255 / max + 0.5)`;
`max` is 1.0 when `default_clamping` is true, else the largest component on the canvas;
one trailing `\\n`.
"""
import numpy as np


def color(r, g, b):
    return np.array([r, g, b], dtype=np.float64)


class Canvas(object):
    """`(width, height, 3)` fp64 framebuffer, indexable as `canvas[x, y]`."""

    def __init__(self, width, height, data=None):
        self.shape = (width, height)
        if data is None:
            self.np_arr = np.zeros((width, height, 3), dtype=np.float64)
        else:
            data = np.asarray(data, dtype=np.float64)
            if data.shape != (width, height, 3):
                raise ValueError("canvas data must have shape (width, height, 3)")
            self.np_arr = data

    def __getitem__(self, index):
        return self.np_arr[index]

    def __setitem__(self, index, value):
        self.np_arr[index] = value

    def __array__(self, dtype=None, copy=None):
        return self.np_arr if dtype is None else self.np_arr.astype(dtype)


def canvas(w, h):
    return Canvas(w, h)


def pixel_at(canvas, x, y):
    return canvas[x, y]


def write_pixel(canvas, x, y, color):
    if x < 0 or x >= canvas.shape[0] or y < 0 or y >= canvas.shape[1]:
        return
    canvas[x, y] = color


def write_pixels(canvas, block, x_start, y_start, shape):
    canvas[x_start:x_start + shape, y_start:y_start + shape] = block


def _as_array(canvas):
    """(w, h, 3) ndarray view of a Canvas, an ndarray, or a reference Canvas."""
    if isinstance(canvas, np.ndarray):
        return canvas
    if hasattr(canvas, "np_arr") and isinstance(canvas.np_arr, np.ndarray):
        return canvas.np_arr
    if hasattr(canvas, "shared_arr"):  # reference Canvas (canvas.py:61)
        return np.ctypeslib.as_array(canvas.shared_arr)
    return np.asarray(canvas)


def construct_ppm_header(canvas):
    return "P6\n{} {}\n255\n".format(canvas.shape[0], canvas.shape[1])


def quantize(arr, default_clamping=False):
    """(w, h, 3) fp64 -> (h, w, 3) uint8 following reference canvas.py:133-156."""
    arr = _as_array(arr)
    max_component = 1.0
    if not default_clamping:
        max_component = np.max(arr)
    a = np.transpose(arr, (1, 0, 2))
    with np.errstate(all="ignore"):
        mid = (a * 255 / max_component + 0.5)
        mid = np.where(np.isfinite(mid), mid, 0.0).astype(np.int64)
    out = np.where(a >= max_component, 255, np.where(a <= 0, 0, mid))
    return out.astype(np.uint8)


def construct_ppm_body(canvas, default_clamping=False):
    return bytearray(quantize(canvas, default_clamping).tobytes())


def construct_ppm(canvas, default_clamping=True):
    header = construct_ppm_header(canvas)
    body = construct_ppm_body(canvas, default_clamping)
    body.append(10)
    return b"".join([header.encode("utf-8"), body])


class TextureCanvas(object):
    """Image texture; `pixel_at(row, col)` = texel / global max (reference canvas.py:40-55)."""

    def __init__(self, file_path=None, pixels=None):
        self.file_path = file_path
        if pixels is None:
            from PIL import Image
            with Image.open(file_path, "r") as img:
                pixels = np.array(img)
        self.pixels = np.asarray(pixels)
        self.scaling_factor = np.float64(np.max(self.pixels))
        self.width = self.pixels.shape[1]
        self.height = self.pixels.shape[0]

    def pixel_at(self, x, y):
        raw = self.pixels[x, y]
        return color(np.float64(raw[0]) / self.scaling_factor,
                     np.float64(raw[1]) / self.scaling_factor,
                     np.float64(raw[2]) / self.scaling_factor)
