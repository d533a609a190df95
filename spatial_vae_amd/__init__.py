"""MI355X-native spatial-VAE decoder hot path.

Host side (Python, PyTorch-ROCm for device memory / streams / autograd plumbing) of the C-ABI
library built from ``csrc/`` (hand-written HIP for gfx950).  The public surface mirrors the
reference's ``spatial_vae.models`` classes and ``eval_minibatch`` functions; see DESIGN.md.
"""
from .build import build, library_path  # noqa: F401
