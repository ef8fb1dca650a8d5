#!/usr/bin/env python3
"""Build a variant of libeesen_hip.so with extra -D flags on selected sources, for A/B runs on the GPU box.

  python scripts/build_variant.py NAME "-DSOME_EXPERIMENT=1" [source.hip ...]   (default source: lstm_persistent.hip)
  EESEN_HIP_LIBRARY=eesen_amd/lib/variants/libeesen_hip_NAME.so python bench.py ...

Objects of the other sources are taken from the regular build (python -m eesen_amd.build).
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eesen_amd import build as B

name, defs = sys.argv[1], sys.argv[2].split()
srcs = sys.argv[3:] or ["lstm_persistent.hip"]
B.build()
out_dir = os.path.join(ROOT, "eesen_amd", "lib", "variants")
os.makedirs(out_dir, exist_ok=True)
objs = []
for s in B.SOURCES:
    obj = os.path.join(ROOT, "eesen_amd", "lib", s.replace(".", "_") + ".o")
    if s in srcs:
        obj = os.path.join(out_dir, f"{name}_" + s.replace(".", "_") + ".o")
        subprocess.check_call([B.hipcc()] + defs + B.FLAGS + ["-c", os.path.join(ROOT, "eesen_amd", "csrc", s), "-o", obj])
    objs.append(obj)
lib = os.path.join(out_dir, f"libeesen_hip_{name}.so")
subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs +
                      ["-ldl", "-Wl,-Bsymbolic", "-Wl,--version-script=" + os.path.join(ROOT, "eesen_amd", "csrc", "exports.map")])
print(lib)
