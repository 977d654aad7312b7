"""Parity oracle for the rasterize_triangles hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (pytorch_mesh_renderer_amd)
never imports this module; its HIP path fails loudly instead of falling back.

Contents
  mr_oracle.c / libmr_oracle.so  C restatement of the reference C++ kernel
        (src/mesh_renderer/kernels/rasterize_triangles.cpp:131-273,302-419),
        pinned bit-for-bit by tests/golden/* and by oracle/_ref when present.
  _ref/rasterize_triangles_cpp.so  the reference's own kernel compiled from
        /root/reference by oracle/Makefile (git-ignored, travels to the GPU box).
  shading.py  torch-CPU restatement of the eager attribute interpolation and
        Phong shading (src/mesh_renderer/rasterize.py:112-150,
        src/mesh_renderer/render.py:157-228,287-386), pinned by goldens
        generated from the reference's Python.
"""
from .kernel import (forward, backward, max_threads, have_reference_kernel,
                     reference_forward, reference_backward, build)

__all__ = ["forward", "backward", "max_threads", "have_reference_kernel",
           "reference_forward", "reference_backward", "build"]
