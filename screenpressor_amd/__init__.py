"""MI355X-native ScreenPressor v4 lossless encode/decode path.

The product path is the HIP library behind include/scpr_amd.h
(screenpressor_amd/csrc); this package is the thin Python host mirror used by
bench.py and the tests.  There is no CPU fallback: importing
screenpressor_amd.codec without the built library raises.
"""
__version__ = "0.1.0"
