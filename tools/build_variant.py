#!/usr/bin/env python3
"""Builds an A/B variant of libsvc_hip.so: the named translation units recompiled with extra -D flags, every other object as built.

  tools/build_variant.py <name> <unit.hip>[,<unit.hip>...] [-DFLAG=VALUE ...]
      -> scalable_video_codec_amd/_ab_<name>_libsvc_hip.so   (git-ignored, travels to the GPU box)

The same-box A/B scripts (tools/ab_*.sh) copy such a file over libsvc_hip.so, run, and restore the built one.
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scalable_video_codec_amd import build as b  # noqa: E402


def main() -> None:
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    name, units, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    b.build_hip()
    objs = []
    for s in b.HIP_SOURCES:
        obj = os.path.join(b.OBJ, s.replace(".hip", ".o"))
        if s in units:
            obj = os.path.join(b.OBJ, f"_ab_{name}_" + s.replace(".hip", ".o"))
            subprocess.check_call([b._hipcc(), *b.HIPCC_FLAGS, *flags, "-c", os.path.join(b.CSRC, s), "-o", obj])
        objs.append(obj)
    missing = [u for u in units if u not in b.HIP_SOURCES]
    if missing:
        sys.exit(f"unknown translation unit(s): {missing}")
    out = os.path.join(b.PKG, f"_ab_{name}_libsvc_hip.so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    print(out)


if __name__ == "__main__":
    main()
