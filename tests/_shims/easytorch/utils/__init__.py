from .dist import master_only  # noqa: F401
