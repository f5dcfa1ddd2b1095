from .modules import SRLModules  # noqa: F401
