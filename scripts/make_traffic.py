"""profiles/traffic.json from the two PMC passes (FETCH_SIZE, WRITE_SIZE; KB per launch) of scripts/gpu_step_target.py.
usage: make_traffic.py fetch.csv write.csv out.json [bench.json of the same box: stamps its clock and kernel time]"""
import csv, json, sys, collections
def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
B, H, W = 12, 192, 640
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/gpu_step_target.py: three whole loss steps, "
                "B=12 192x640), mean KB per launch. HBM bytes = 2*FETCH_SIZE + WRITE_SIZE: FETCH_SIZE is doubled as "
                "MI355X_MICROARCH.md prescribes for gfx950 (cross-checked on pack_identity_kernel, whose stores are reported "
                "exactly).  The ~60-100 MB working set of a replayed step fits the 256 MiB Infinity Cache, so these are "
                "fabric-side request counts, not DRAM bytes.", "kernels": {}}
for k in sorted(f):
    if "mal::" not in k:
        continue
    fb, wb = f[k] * 1024.0, w.get(k, 0.0) * 1024.0
    out["kernels"][k] = {"fetch_kb_raw": round(f[k], 1), "write_kb_raw": round(w.get(k, 0.0), 1),
                         "hbm_bytes_per_launch": int(2 * fb + wb), "bytes_per_px": round((2 * fb + wb) / (B * H * W), 2)}
teacher = [k for k in out["kernels"] if "march_teacher_kernel<false" in k or
           "march_kernel<true, true, true, false, false, false>" in k or "march_kernel<true, true, true, false>" in k]
temporal = [k for k in out["kernels"] if "march_teacher_kernel<true" in k or "march_kernel<true, true, true, false, false, true>" in k]
if temporal:
    out["pass_kernel_teacher_temporal_bytes_per_launch"] = out["kernels"][temporal[0]]["hbm_bytes_per_launch"]
if teacher:
    out["pass_kernel_teacher_bytes_per_launch"] = out["kernels"][teacher[0]]["hbm_bytes_per_launch"]
out["algorithmic_teacher_bytes_per_launch"] = 96 * B * H * W
if len(sys.argv) > 4:  # the bench line measured on the same box in the same refresh: its sustained clock and kernel time
    try:
        d = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
        out["same_box"] = {"shader_clock_mhz": d["roofline"].get("valu", {}).get("shader_clock_mhz"),
                           "teacher_kernel_ms_events": d["roofline"]["kernel_ms"], "ms_per_step": d["ms_per_step"]}
    except Exception as ex:
        out["same_box"] = {"error": str(ex)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
