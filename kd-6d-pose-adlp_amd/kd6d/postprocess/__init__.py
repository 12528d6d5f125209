from .postprocess import PostProcessor  # noqa: F401
