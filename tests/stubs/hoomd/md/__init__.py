from . import _md, force  # noqa: F401
