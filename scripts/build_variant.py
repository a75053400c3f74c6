"""Build a variant of libmal_hip.so with extra compiler flags into mal_amd/lib/<name>.so (for MAL_HIP_LIB=...).
usage: build_variant.py name -DMAL_STAGE_TIMERS ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mal_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
hipcc = B._hipcc()
objs = []
procs = []
tmp = os.path.join(B.LIBDIR, "_" + name)
os.makedirs(tmp, exist_ok=True)
for s in B.SOURCES:
    o = os.path.join(tmp, os.path.basename(s).replace(".hip", ".o"))
    objs.append(o)
    procs.append(subprocess.Popen([hipcc] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, s), "-o", o]))
for p in procs:
    assert p.wait() == 0
out = os.path.join(B.LIBDIR, name + ".so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
