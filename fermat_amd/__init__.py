"""fermat_amd — MI355X-native (gfx950) drop-in for NVlabs/fermat's -pt wavefront path tracer.

The product is libfermat_pt_hip.so (hand-written HIP kernels + C-ABI, include/fermat_pt_hip.h).  This package is the thin
Python side used by the tests and bench.py: ctypes declarations of the C-ABI and a `Renderer` that keeps the scene, textures
and frame buffer in torch device tensors (torch = device memory, streams and torch.distributed plumbing) and hands the library
plain device pointers, exactly as Fermat's RenderingContext hands its renderer a RenderingContextView.

There is NO CPU fallback: creating a Renderer without the built extension or without a GPU raises.
"""
from .api import (lib, lib_path, build_extension, FptError, Renderer, default_options, default_bpt_options, default_psf_options, tile_pixel_lists,  # noqa: F401
                  RAY_DTYPE, HIT_DTYPE, VPL_DTYPE)
from . import scene  # noqa: F401

__all__ = ["lib", "lib_path", "build_extension", "FptError", "Renderer", "default_options", "default_bpt_options", "default_psf_options", "tile_pixel_lists", "scene"]
