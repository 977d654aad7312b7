from . import camera_utils, meshes, shapes, synthetic

__all__ = ["camera_utils", "meshes", "shapes", "synthetic"]
