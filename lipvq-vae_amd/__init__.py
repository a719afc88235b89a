"""lipvq-vae_amd -- MI355X (gfx950) native LipVQ-VAE action tokenizer.

The directory name carries a hyphen (it is the project's name); import it as
``lipvq_vae_amd`` -- the loader module of that name at the repository root maps it here.

Importing this package loads the in-tree HIP library (``_lipvq_hip.so``) and fails loudly
if it is missing: the tokenizer path has no CPU or eager-PyTorch fallback.
"""
from . import _capi, ops  # noqa: F401  (loads the shared object)

__all__ = ["ops"]
