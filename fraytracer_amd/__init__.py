"""fraytracer_amd — MI355X-native implementation of FrayTracer's per-pixel SDF ray-marching hot path.

The product is libfraytracer_hip.so (hand-written HIP for gfx950 behind a C ABI, include/fraytracer_hip.h).
This package is the host-side mirror of the reference's F# scene-composition API over that ABI.
Importing it fails if the shared library has not been built; there is no CPU fallback.
"""
from ._lib import FrayTracerError, LIB_PATH, build_info, source_hash
from .api import (FColor, SdfForm, SdfMaterial, SdfObject, SdfLight, SdfScene, Lens, Camera, ImageSize, Image, Ray,
                  Device, DeviceScene, SceneTrace, realise, render_multi, glibc_build_of_this_host)

__all__ = ["FColor", "SdfForm", "SdfMaterial", "SdfObject", "SdfLight", "SdfScene", "Lens", "Camera", "ImageSize",
           "Image", "Ray", "Device", "DeviceScene", "SceneTrace", "realise", "render_multi", "FrayTracerError", "LIB_PATH", "build_info", "source_hash", "glibc_build_of_this_host"]
