"""fountain_amd: MI355X-native path-tracing core behind akofke/fountain's render()."""
from . import _abi
from .api import (Backend, DirectLightingIntegrator, Film, FountainError, PathIntegrator, PbrtScene, PerspectiveCamera,
                  RandomSampler, SamplerIntegrator, Scene, SceneBuilder, Transform, WhittedIntegrator, default_backend, film_resolve_device, load_ply,
                  load_ply_ascii, make_rays, read_exr, write_exr)

__all__ = ["Backend", "DirectLightingIntegrator", "Film", "FountainError", "PathIntegrator", "PbrtScene", "PerspectiveCamera",
           "RandomSampler", "SamplerIntegrator", "Scene", "SceneBuilder", "Transform", "WhittedIntegrator", "default_backend",
           "film_resolve_device", "load_ply", "load_ply_ascii", "make_rays", "read_exr", "write_exr", "_abi"]
