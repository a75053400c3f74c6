#!/usr/bin/env python
"""Timeline of ONE replayed step from a rocprofv3 kernel trace (…_kernel_trace.csv): start / end of every dispatch relative to
the step's first kernel, so that what runs beside what is visible (the ensemble + student passes beside the producer chain).
    python scripts/step_timeline.py <kernel_trace.csv> [first-kernel-substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "pack_identity_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
if len(starts) < 12:
    sys.exit("too few steps in the trace")
i0, i1 = starts[-6], starts[-5]  # a step from the steady state
t0 = int(rows[i0]["Start_Timestamp"])
print("step of %d dispatches, %.1f us from its first kernel's start to the next step's" % (i1 - i0, (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("void mal::", "").replace("mal::", "")[:70]
    print("%8.1f %8.1f %7.1f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), name))
