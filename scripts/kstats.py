"""Per-kernel statistics from a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME -> DIR/NAME_results.db).
usage: kstats.py results.db [csv_out]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
td = [t for t in tabs if 'kernel_dispatch' in t][0]
ts = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), "
     f"s.arch_vgpr_count, s.sgpr_count, s.group_segment_size from {td} d join {ts} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc")
rows = list(c.execute(q))
tot = sum(r[5] for r in rows)
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,SGPRs,LDS"]
for r in rows:
    lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f,%d,%d,%d' % (r[0], r[1], r[5], r[2], r[3], r[4], 100.0 * r[5] / tot, r[6], r[7], r[8]))
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
for r in rows[:12]:
    print('%-100s n=%5d avg %8.1f us  min %8.1f  vgpr %d' % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[6]))
