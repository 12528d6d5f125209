"""Flags of the sanitized HOST build of libkd6d (python kd-6d-pose-adlp_amd/build.py --sanitize -> csrc/libkd6d_san.so):
AddressSanitizer + UndefinedBehaviorSanitizer on the host code of every csrc/*.hip; the device code is compiled as usual
(no sanitizer exists for plain gfx950 code objects on this pool).  CPU only: this file, tests/test_sanitize_host.py and
tests/san_driver.cpp are listed in .gpurunignore.

    python tools/build_sanitized.py        # same as build.py --sanitize
"""
import os
import sys

SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g1", "-Wno-option-ignored"]

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "..", "kd-6d-pose-adlp_amd"))
    import build
    print(build.build(force="--force" in sys.argv, sanitize=True))
