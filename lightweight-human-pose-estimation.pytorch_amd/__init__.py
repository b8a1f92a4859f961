"""MI355X-native Lightweight-OpenPose inference path (imported as ``lwpose_amd``).

Host-side mirror of the reference's hot-path API; all compute goes through the C-ABI
library ``liblwpose_hip.so`` (csrc/, include/lwpose.h) — there is no CPU fallback.
"""
__version__ = "0.1.0"
