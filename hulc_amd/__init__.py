"""hulc_amd — MI355X-native (gfx950) HULC training step behind the reference's LightningModule surface.

Importing the package is cheap; the HIP library (``hulc_amd/csrc/libhulc_hip.so``) is loaded on first use
of ``hulc_amd.Hulc`` / ``hulc_amd.lib`` and a missing library raises — there is no CPU fallback.
"""
from importlib import import_module

__all__ = ["Hulc", "GCBC", "spec"]


def __getattr__(name):
    if name in ("Hulc", "GCBC"):
        return getattr(import_module(".hulc", __name__), name)
    if name in ("spec", "lib", "config", "trainer"):
        return import_module("." + name, __name__)
    raise AttributeError(name)
