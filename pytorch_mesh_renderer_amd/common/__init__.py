from . import camera_utils, shapes, synthetic

__all__ = ["camera_utils", "shapes", "synthetic"]
