"""MI355X-native renderer behind the `render(camera, world) -> canvas` surface of
gbordelon/ray_tracer_challenge.  See DESIGN.md for the path and its boundary.

Submodules keep the reference's module names (`canvas`, `matrix`, `vector`, `shapes`,
`renderer`, `yaml_parser`, `obj_parser`, `config`); `compile` is the scene compiler and
`_ffi` the ctypes binding of `include/rtc_api.h`.
"""
__version__ = "0.1.0"
