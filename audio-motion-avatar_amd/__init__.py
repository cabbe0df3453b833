"""MI355X-native audio-driven Gaussian-avatar rendering hot path."""
from ._lib import AmavError, LIB_PATH  # noqa: F401
