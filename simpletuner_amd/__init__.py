"""simpletuner_amd — MI355X-native (gfx950) diffusion train step behind SimpleTuner's trainer / plugin surface.

Compute lives in csrc/ (hand-written HIP, C ABI in include/st355.h); this package is the host side that mirrors the
reference's Python interfaces (ModelFoundation plugin, optimizer registry entry, EMAModel, step loop).
"""
__version__ = "0.1.0"
