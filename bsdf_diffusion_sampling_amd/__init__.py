"""MI355X-native neural-BSDF importance sampler (the sample()/pdf() hot path of
fzy28/BSDF_diffusion_sampling), see DESIGN.md."""
from . import weights  # noqa: F401

__all__ = ["weights"]
__version__ = "0.1.0"
