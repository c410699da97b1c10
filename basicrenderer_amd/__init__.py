"""basicrenderer_amd: MI355X-native visibility-buffer path of BasicRenderer (libbrmi.so) + harness."""
from .capi import brmi_lib, scene_lib, LIB_DIR  # noqa: F401
from .scene import Scene, PRESETS  # noqa: F401

__all__ = ["Scene", "PRESETS", "brmi_lib", "scene_lib"]
