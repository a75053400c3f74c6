#!/usr/bin/env python
"""Per-stage instruction budget of the marching kernels' row loop: compiles mal_march.hip with
-DMAL_STAGE_MARKS (stage boundaries become labelled comments behind scheduling barriers), cuts the row loop of each
priced kernel at the marks and counts instructions per stage and class.  Writes profiles/r04_valu_budget.json.

    python scripts/valu_budget.py [--out profiles/r04_valu_budget.json]

The marked build is NOT the shipped one (the barriers pin the stage order): its totals are listed next to the shipped
listing's (mal_amd.build.valu_report) so the reader sees how far they are apart."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mal_amd import build  # noqa: E402

STAGES = collections.OrderedDict([  # key = the MAL_MARK that OPENS the stage (7 also takes the loop head in front of mark 0)
    (7, "loop-carried moves (the rolled row state), back edge, row counters, parameter block (scalar loads)"),
    (0, "camera block, depth, ray, projection of both frames (Project3D + unnormalise + border clip)"),
    (10, "bilinear tap cells and byte offsets, next row's operand requests, eight texel gathers, tap weights, clip masks"),
    (1, "(epilogue passes: consistency / distillation terms of the gradient row, placed in the gathers' shadow)"),
    (2, "gather wait, bilinear blend of six colour values and their d/du, d/dv, LDS ring write (24 floats)"),
    (3, "24 statistic planes: products x^2, x*y, y^2 and their horizontal 3-sums"),
    (4, "vertical 3-row sums and SSIM of six values (window sums -> S); epilogue passes: the compiler sinks this into 50"),
    (50, "SSIM partials dS/dsum of six values, L1 term, min over the two candidates, automask / weight, stores, winner-only "
         "coefficients of the 18 partial planes (its 41 static v_mov are zero-fills on the not-a-valid-row path, not executed per row)"),
    (5, "horizontal 3-sums of the 18 partial planes incl. the adjoint's border weights"),
    (60, "gradient row: LDS ring read (24 floats)"),
    (61, "(merged by the compiler into 62)"),
    (62, "gradient row: winner's L1 sign, vertical adjoint sums, d loss / d warped colour, chain rule to the disparity (pose "
         "variant: re-derived point, d u / d disp, d v / d disp), pose partials (12 packed accumulators), stores"),
    (63, "roll of the partial-plane sums (hcA, hcB) incl. the top-border weight"),
    (6, "(forward-only passes: epilogue / depth map out)"),
])
# what lies between mark a (inclusive start) and the next mark is attributed to the stage keyed by a


def loop_of(text, key):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    labels, ins = {}, []
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith("; MAL_STAGE"):
            ins.append(t)
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
    return ins[spans[0][1]:spans[0][2] + 1]


def classify(op):
    if op.startswith("v_"):
        return "valu:" + build._valu_class(op)
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def budget(text, key):
    loop = loop_of(text, key)
    cur = 7  # the loop label sits in front of mark 0: the back edge's tail belongs to the loop overhead
    per = collections.OrderedDict((k, collections.Counter()) for k in STAGES)
    for t in loop:
        if t.startswith("; MAL_STAGE"):
            cur = int(t.split()[2])
            continue
        per.setdefault(cur, collections.Counter())[classify(re.split(r"\s+", t)[0])] += 1
    out = []
    tot = collections.Counter()
    for k, c in per.items():
        valu = {n[5:]: v for n, v in c.items() if n.startswith("valu:")}
        cyc = sum(build.VALU_COSTS[n] * v for n, v in valu.items())
        row = {"stage": k, "what": STAGES.get(k, "?"), "instructions": sum(c.values()), "valu": sum(valu.values()),
               "valu_by_class": valu, "pipe_cycles": round(cyc, 1), "salu": c["salu"], "smem": c["smem"], "lds": c["lds"],
               "vmem": c["vmem"], "waitcnt": c["waitcnt"], "nop": c["nop"]}
        out.append(row)
        tot.update({"instructions": row["instructions"], "valu": row["valu"], "pipe_cycles": row["pipe_cycles"],
                    "salu": row["salu"], "smem": row["smem"], "lds": row["lds"], "vmem": row["vmem"]})
    for row in out:
        row["share_of_pipe_cycles"] = round(row["pipe_cycles"] / max(tot["pipe_cycles"], 1e-9), 4)
    return out, dict(tot)


def main():
    out_path = os.path.join(ROOT, "profiles", "r04_valu_budget.json")
    if "--out" in sys.argv:
        out_path = sys.argv[sys.argv.index("--out") + 1]
    asm = os.path.join(build.LIBDIR, "mal_march_marks.s")
    cmd = [build._hipcc()] + build.FLAGS + ["-DMAL_STAGE_MARKS", "--cuda-device-only", "-S",
                                            os.path.join(build.CSRC, "mal_march.hip"), "-o", asm]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    text = open(asm).read()
    os.remove(asm)
    shipped = build.valu_report()
    res = {"source": "hipcc -S -DMAL_STAGE_MARKS of mal_amd/csrc/mal_march.hip with mal_amd.build.FLAGS (digest of the shipped "
                     "sources %s); row loop = the longest backward-branch loop; stage = what lies between two MAL_MARK comments "
                     "(scheduling barriers keep instructions on their side)" % shipped["digest"][:12],
           "cycle_prices": build.VALU_COSTS, "kernels": {}}
    for name in ("teacher", "student", "ensemble"):
        key = build.VALU_KERNELS[name]
        rows, tot = budget(text, key)
        sh = shipped["kernels"].get(name, {})
        res["kernels"][name] = {"stages": [r for r in rows if r["instructions"]], "total_marked_build": tot,
                                "total_shipped_build": {"instructions": sh.get("loop_instructions"), "valu": sh.get("valu_instructions"),
                                                        "pipe_cycles": sh.get("pipe_cycles")}}
    res["verdict_items"] = {
        "i_horizontal_sums_through_lds": "built and measured in six variants (profiles/r04_hsum_variants_ab.txt): all 42 sums through LDS "
            "= -13 % pipe cycles, -6 % instructions, kernel +2.5 % SLOWER; the 18 partial planes alone -1 % (shipped); the probe "
            "(profiles/r04_dpp_probe.txt) prices a DPP 3-sum at 18.4 and the LDS form at 9.1 cycles of a wave's timeline, but the "
            "statistic planes' sums are consumed at once and expose the LDS round trip",
        "ii_sliding_row_offsets": "not built: ~25-30 scalar instructions of stage 7 / 10 (three reflect-clamp-multiply chains become "
            "carried registers); the A/B above shows that -57 instructions of ANY class did not shorten the kernel, so this cannot",
        "iii_pose_partials_9_sums": "no saving: sum a_i*D, sum a_i*D*y, sum a_i per scalar a_i is mul + add + fma + add = 12 packed "
            "instructions for the three scalars, exactly the 9 fma + 3 add of the 12 accumulators of stage 62 (only the 3 plain "
            "X = depth*ray disappear), and the accumulators stay 18 registers",
        "iv_target_window_sums_once": "not built: -44 instructions here (12 DPP + ~10 packed + ~10 plain + 12 v_mov, ~5 % of the pipe "
            "cycles) against +24..36 B/px written by the identity/packing sweep, the one bandwidth-bound kernel of the step "
            "(+6..10 us there for -2 us per marching pass at best); with the A/B above (instruction cuts do not convert) a net loss",
        "conclusion": "the <= 50 us target is not reached: 55.5 -> 54.6 us replayed (0.319 -> 0.324 of 8 TB/s).  The premise (81 % VALU "
            "busy => remove instructions) does not hold up: the kernel's duration is each wave's serial timeline over its 15 + 2 "
            "iterations (issue ~5 cycles per instruction per wave whatever its class + the exposed part of the gather / LDS / "
            "scalar-load round trips), of which the sibling wave hides a fixed share; removing pipe cycles without shortening that "
            "chain moves nothing, and every attempt to reorder the chain (gathers' shadow) lengthened it"}
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    for name, k in res["kernels"].items():
        print(name, "marked", k["total_marked_build"], "shipped", k["total_shipped_build"])
        for r in k["stages"]:
            print("  %3d %4d instr %4d valu %7.1f cyc (%4.1f %%) salu %3d smem %2d lds %2d vmem %2d  %s  %s"
                  % (r["stage"], r["instructions"], r["valu"], r["pipe_cycles"], 100 * r["share_of_pipe_cycles"], r["salu"],
                     r["smem"], r["lds"], r["vmem"], r["valu_by_class"], r["what"][:60]))


if __name__ == "__main__":
    main()
