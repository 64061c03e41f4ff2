"""fountain_amd: MI355X-native path-tracing core behind akofke/fountain's render()."""
from . import _abi
from .api import (Backend, DirectLightingIntegrator, Film, FountainError, PathIntegrator, PerspectiveCamera,
                  RandomSampler, SamplerIntegrator, Scene, SceneBuilder, Transform, default_backend, load_ply_ascii,
                  make_rays)

__all__ = ["Backend", "DirectLightingIntegrator", "Film", "FountainError", "PathIntegrator", "PerspectiveCamera",
           "RandomSampler", "SamplerIntegrator", "Scene", "SceneBuilder", "Transform", "default_backend",
           "load_ply_ascii", "make_rays", "_abi"]
