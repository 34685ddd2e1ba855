"""pathtracer_amd -- MI355X-native radiance loop of ethanabrooks/pathtracer.

The product is libpt.so (HIP kernels behind the C ABI of include/pt.h); this
package is the thin host-side mirror of the reference's interface for that
path.  Importing it loads libpt.so and fails loudly if it is missing.
"""
from . import _lib  # noqa: F401  (loads libpt.so or raises)
from ._lib import (PT_FLAG_NO_PRIMARY_REUSE, PT_FLAG_LDS_TILES, PT_FLAG_NO_STEAL, PT_FLAG_NO_BVH, PT_RGB8_MEAN_CLAMP, PT_RGB8_MEAN_GAMMA, PT_RGB8_REFERENCE,
                   PT_ERR_UNSUPPORTED, PT_ERR_LIMIT, PT_ERR_INVALID, PtError)
from .render import Renderer, repa3ToImage, traces
from .scene import (Disk, InfinitePlane, Sphere, Rectangle, Object, Params, disk, infLight, infPlane, infPlane2, infPlane3,
                    light, objects, random_disk_scene, random_rectangle_scene, random_sphere_scene, synthetic_scene)

__all__ = ["Renderer", "traces", "repa3ToImage", "Params", "Object", "Disk", "InfinitePlane", "objects",
           "infPlane", "infPlane2", "infPlane3", "light", "infLight", "disk", "random_disk_scene", "random_sphere_scene", "random_rectangle_scene", "synthetic_scene", "Sphere", "Rectangle", "PtError",
           "PT_RGB8_REFERENCE", "PT_RGB8_MEAN_CLAMP", "PT_RGB8_MEAN_GAMMA", "PT_FLAG_NO_PRIMARY_REUSE", "PT_FLAG_LDS_TILES", "PT_FLAG_NO_STEAL", "PT_FLAG_NO_BVH", "PT_ERR_UNSUPPORTED", "PT_ERR_LIMIT", "PT_ERR_INVALID"]
