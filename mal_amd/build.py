"""Build libmal_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m mal_amd.build            # hipcc --offload-arch=gfx950 -> mal_amd/lib/libmal_hip.so

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box
with the gpurun snapshot.  -ffp-contract=off: fused multiply-adds appear only where the
kernels write them (mal_device.h states every rounding point).
"""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmal_hip.so")
SOURCES = ["mal_api.hip", "mal_pass.hip", "mal_warp.hip", "mal_photo.hip", "mal_photo_march.hip", "mal_dyn.hip", "mal_costvol.hip", "mal_epipolar.hip", "mal_pose.hip", "mal_march.hip", "mal_step.hip", "mal_step_ms.hip", "mal_dr_step.hip"]
HEADERS = ["mal_common.h", "mal_device.h", "mal_march.h", "mal_pose.h", "mal_pairs.h", os.path.join("..", "..", "include", "mal_hip.h")]
# csrc/experiments/: formulations that lost their same-box A/Bs (LABBOOK.md) -- whole sources and the .inc halves the shipped
# sources include under #ifdef MAL_EXPERIMENTS.  Not part of the product: compiled only with MAL_EXPERIMENTS=1 in the environment.
EXPERIMENT_SOURCES = [os.path.join("experiments", "mal_tile2.hip")]
EXPERIMENT_INCLUDES = [os.path.join("experiments", f) for f in ("pass_tiled.inc", "pass_tiled_launch.inc", "march3.inc")]
# -amdgpu-sched-strategy=max-ilp: the kernels run at 2-4 waves per SIMD by register count anyway; scheduling for
# ILP instead of occupancy is worth ~2 % on the marching kernels (measured A/B on MI355X)
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17", "-fno-gpu-rdc",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp",
         "-Wall", "-Wno-unused-function", "-I", CSRC]
# MAL_EXPERIMENTS=1 in the environment builds the formulations that were measured slower and are kept for same-box A/B
# only (LABBOOK.md 6: LDS-tiled passes, three-wave pipeline, exporting gradient pass, classified dispatch order); the
# default library does not contain them and mal_set_option refuses their switches
if os.environ.get("MAL_EXPERIMENTS", "0") not in ("", "0"):
    FLAGS = FLAGS + ["-DMAL_EXPERIMENTS"]
    SOURCES = SOURCES + EXPERIMENT_SOURCES
    HEADERS = HEADERS + EXPERIMENT_INCLUDES


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libmal_hip.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, os.path.basename(s).replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out))
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


# ---- vector-ALU price of the marching kernels' row loop, from the compiler's own output -------------------------
# cycles per wave64 instruction on MI355X at >= 2 waves per SIMD, measured by scripts/valu_probe.hip
# (profiles/r02_valu_probe.txt): plain fp32 2.7, packed fp32 4.2, DPP 4.3, transcendental 8, 32-bit integer multiply 8
VALU_COSTS = {"plain": 2.7, "packed": 4.2, "dpp": 4.3, "transcendental": 8.0, "mul32": 8.0, "mov": 2.7, "cmp/select": 2.7,
              "lane": 2.7}
VALU_JSON = os.path.join(LIBDIR, "valu_cost.json")
# the instantiations bench.py prices: name -> substring of the mangled symbol
VALU_KERNELS = {"teacher": "march_teacher_kernelILb0ELb0EE", "teacher_temporal": "march_teacher_kernelILb1ELb0EE",
                "teacher_generic": "march_kernelILb1ELb1ELb1ELb0ELb0ELb0EE",
                "student": re.compile(r"march_student_kernelILi\d+EE"), "student_generic": "march_kernelILb1ELb0ELb0ELb1ELb0ELb0EE",
                "ensemble": "march_kernelILb0ELb0ELb0ELb0ELb0ELb0EE"}


def _valu_class(op):
    if op.startswith("v_pk_"):
        return "packed"
    if "dpp" in op:
        return "dpp"
    if op.startswith(("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return "transcendental"
    if op.startswith(("v_mul_lo", "v_mul_hi")):
        return "mul32"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "mov"
    if op.startswith(("v_cndmask", "v_cmp")):
        return "cmp/select"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    return "plain"


def valu_cost_of(asm_text, key, nth=0):
    """Instruction classes of the longest backward-branch loop (the row loop; ``nth`` = 1: the second longest disjoint one --
    the gradient-only iterations that follow the row loop in the one-row-halo passes) of the kernel whose mangled name
    contains ``key`` in a ``hipcc -S`` listing: {"loop_instructions", "valu_instructions", "pipe_cycles", "classes":
    {class: {"instr", "cycles"}}} with VALU_COSTS as prices."""
    import collections
    import re
    lines = asm_text.split("\n")
    has = (lambda l: key.search(l) is not None) if hasattr(key, "search") else (lambda l: key in l)  # substring or compiled pattern
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and has(l) and l.rstrip().split(":")[0].endswith("E"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    labels, ins = {}, []
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith((";", ".")):
            continue
        ins.append(t)
    spans = []
    for i, t in enumerate(ins):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", t)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < i:
                spans.append((i - tgt, tgt, i))
    spans.sort(reverse=True)
    picked = []
    for sp in spans:  # the longest loops that do not overlap an already picked one
        if all(sp[2] < q[1] or sp[1] > q[2] for q in picked):
            picked.append(sp)
        if len(picked) > nth:
            break
    if len(picked) <= nth:
        raise StopIteration("kernel %s has no loop number %d" % (key, nth))
    best = picked[nth]
    loop = ins[best[1]:best[2] + 1]
    cnt = collections.Counter()
    for t in loop:
        op = re.split(r"\s+", t)[0]
        if op.startswith("v_"):
            cnt[_valu_class(op)] += 1
    classes = {k: {"instr": n, "cycles": round(n * VALU_COSTS[k], 1)} for k, n in cnt.most_common()}
    return {"loop_instructions": len(loop), "valu_instructions": sum(cnt.values()),
            "pipe_cycles": round(sum(c["cycles"] for c in classes.values()), 1), "classes": classes}


def valu_report(force=False, verbose=False):
    """mal_amd/lib/valu_cost.json: the row loop of the marching kernels priced from the listing of the SAME sources and flags
    the shipped library was built from (stamped with the same digest).  bench.py puts it into its `roofline.valu` block."""
    import json
    dig = _digest()
    if not force and os.path.exists(VALU_JSON):
        try:
            old = json.load(open(VALU_JSON))
            if old.get("digest") == dig:
                return old
        except Exception:
            pass
    os.makedirs(LIBDIR, exist_ok=True)
    asm = os.path.join(LIBDIR, "mal_march.s")
    cmd = [_hipcc()] + FLAGS + ["--cuda-device-only", "-S", os.path.join(CSRC, "mal_march.hip"), "-o", asm]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    text = open(asm).read()
    out = {"digest": dig, "source": "hipcc -S --cuda-device-only mal_amd/csrc/mal_march.hip with mal_amd.build.FLAGS; longest "
                                    "backward-branch loop of each kernel = its row loop",
           "costs_cycles_per_wave64_instruction": VALU_COSTS,
           "costs_source": "scripts/valu_probe.hip on MI355X, >= 2 waves per SIMD (profiles/r02_valu_probe.txt)",
           "kernels": {}}
    for name, key in VALU_KERNELS.items():
        try:
            out["kernels"][name] = valu_cost_of(text, key)
            if not name.startswith(("ensemble", "warp")):  # gradient passes: the two gradient-only iterations behind the row loop (one-row halo)
                out["kernels"][name]["drain"] = valu_cost_of(text, key, nth=1)
        except StopIteration:
            pass
    with open(VALU_JSON, "w") as fh:
        json.dump(out, fh, indent=1)
    os.remove(asm)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--valu" in sys.argv:
        import json
        print(json.dumps(valu_report(force=True, verbose=True)["kernels"], indent=1))
