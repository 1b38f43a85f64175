from . import parameterdicts, typeconverter  # noqa: F401
