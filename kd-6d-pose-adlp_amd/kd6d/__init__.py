"""kd6d -- MI355X-native KD training step for 6D pose (hot path of GUOShuxuan/kd-6d-pose-adlp).

Host side of the C ABI in include/kd6d.h.  Importing `kd6d.ops` (or anything that computes)
loads csrc/libkd6d.so and raises if it is missing: there is no CPU fallback.
"""
__version__ = "0.1.0"
