"""nerf-ca_amd: MI355X-native implementation of NeRF-CA's ray-sampling -> MLP -> compositing path.

Import as ``nerfca_amd`` (see ``nerfca_amd.py`` at the repository root: the directory name has a
hyphen).  ``model/`` and ``train/`` mirror the reference's packages of the same names.
"""
from . import _capi  # noqa: F401
from .fused import render_rays, eval_points, set_precision  # noqa: F401

__all__ = ["render_rays", "eval_points", "set_precision"]
