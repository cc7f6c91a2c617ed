"""Import-name shim: ``from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer`` (Edit_core/tetgs_scene/tetgs_model.py:7) resolves to the MI355X-native package
when the repository root is on ``sys.path``."""
from youreditableavatar_amd.diff_gaussian_rasterization import (  # noqa: F401
    GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians, _C)
