"""VALU pipe cycles per loop iteration of a kernel in a hipcc -S file, by instruction class, with the per-class costs
measured by scripts/valu_probe.hip on MI355X (mal_amd.build.VALU_COSTS).  usage: valu_cost.py file.s mangled_substring
(`python -m mal_amd.build --valu` prices the shipped marching kernels the same way -> mal_amd/lib/valu_cost.json)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mal_amd.build import valu_cost_of  # noqa: E402

r = valu_cost_of(open(sys.argv[1]).read(), sys.argv[2])
print("loop: %d instructions, %d VALU, ~%.0f VALU pipe cycles per iteration" % (r["loop_instructions"], r["valu_instructions"], r["pipe_cycles"]))
for k, v in r["classes"].items():
    print("  %-15s %4d instr  %6.0f cycles  %4.1f%%" % (k, v["instr"], v["cycles"], 100 * v["cycles"] / r["pipe_cycles"]))
