"""MI355X-native engine for the depth-soft Show-Attend-and-Tell training hot path.

Only what the path needs lives here:
  csrc/              hand-written HIP kernels (gfx950) + the C-ABI (include/dic.h)
  _lib.py            ctypes loader for libdic_hip.so (fails loudly when absent)
  Captioning_models  host-side mirror of the reference's Encoder/Decoder/attention call surface
  engine.py          fused train step (encoders -> decoder -> loss -> backward -> AdamW) + data parallel
  synthetic.py       procedural weights / inputs (no dataset or checkpoint ships with the reference)
"""
__version__ = "0.1.0"
