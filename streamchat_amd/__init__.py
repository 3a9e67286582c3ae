"""streamchat_amd — MI355X-native hot path of StreamChat's streaming video-understanding inference.

Python host code mirrors the reference's call surface (`utiles.py`, `memory_bank`, `longva` model
wrappers, `inference_streaming_longva_v2.py`); the arithmetic runs in hand-written gfx950 HIP kernels
behind the C ABI of include/streamchat_hip.h (libstreamchat_hip.so, bound with ctypes in `_lib`).
Importing the package does not load the library; the first kernel call does, and fails loudly if it
is not built (`python -m streamchat_amd.build`)."""
__version__ = "0.1.0"
