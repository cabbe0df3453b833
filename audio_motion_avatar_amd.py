"""Importable alias of the package directory `audio-motion-avatar_amd/` (a hyphen is not a Python identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("audio-motion-avatar_amd")
