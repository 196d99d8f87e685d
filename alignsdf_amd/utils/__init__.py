from . import mesh, utils  # noqa: F401
from .utils import decode_sdf_multi_output, kinematic_embedding  # noqa: F401
