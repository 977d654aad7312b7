"""Same export surface as the reference's src/soft_mesh_renderer/__init__.py:1."""
from .render import render
