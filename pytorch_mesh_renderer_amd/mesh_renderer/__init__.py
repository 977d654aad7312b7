"""Same export surface as the reference's src/mesh_renderer/__init__.py:1-5."""
from .render import render, tone_mapper, tone_mapper_uint8, to_uint8
from .rasterize import rasterize
from . import losses
from .graphs import capture_step, CapturedStep

__version__ = '0.0.1'
name = 'mesh_renderer'
