"""deep_sdf.utils.decode_sdf counterpart (deep_sdf/utils.py:64-75)."""
import torch

from ..utils.utils import hip_decoder_for


def decode_sdf(decoder, latent_vector, queries):
    """Hand-head SDF [M,1] of normalised xyz queries through the HIP decoder (the legacy single-output API)."""
    if latent_vector is None:
        raise NotImplementedError("latent-free decoders are not SeparateDecoder-shaped")
    hip = hip_decoder_for(decoder)
    hip.set_sample(latent_vector)
    h, _ = hip.decode_points(queries)
    return h.unsqueeze(1)
