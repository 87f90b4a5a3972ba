"""MI355X-native contrastive hot path (CLIP / FILIP) behind the reference's Python API."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
