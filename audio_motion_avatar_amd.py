"""Import shim: makes the package directory `audio-motion-avatar_amd/` importable as `audio_motion_avatar_amd`
(a hyphen is not a Python identifier).  The directory is loaded under this module's own name, so every submodule
exists exactly once (`audio_motion_avatar_amd.ops`, `.renderer`, ...)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "audio-motion-avatar_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
