"""Drop-in for the reference's ``spatial_vae.models`` (same import path, same classes):
``import spatial_vae.models as models`` keeps working; the classes live in spatial_vae_amd.models."""
from spatial_vae_amd.models import InferenceNetwork, ResidLinear, SpatialGenerator, VanillaGenerator  # noqa: F401
