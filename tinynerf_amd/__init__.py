"""tinynerf_amd: MI355X-native ray-marching hot path with the tinynerf core/models API."""
from . import _lib  # noqa: F401

__all__ = ["core", "models"]
