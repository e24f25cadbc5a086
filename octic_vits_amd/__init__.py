"""octic_vits_amd — MI355X-native (gfx950) engine for the octic-ViT block of davnords/octic-vits.

Hot ops are hand-written HIP kernels behind a C ABI (include/octic_hip.h, liboctic_hip.so);
this package is the Python host side that mirrors the reference's module API
(octic_vits/d8_layers.py, model.py, deit_models.py) on top of it.
"""
__version__ = "0.1.0"
