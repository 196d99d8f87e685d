from . import chamfer  # noqa: F401
