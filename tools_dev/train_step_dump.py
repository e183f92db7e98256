#!/usr/bin/env python
"""Time-ordered kernel list of the LAST training step in a rocprofv3 (rocpd sqlite) trace: the step boundary is the fused
AdamW multi-tensor kernel.   usage: train_step_dump.py results.db > step.txt"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
wx = "workgroup_size_x" if "workgroup_size_x" in cols else None
sel = "name, start, end" + (f", {gx}" if gx else "") + (f", {wx}" if wx else "")
rows = list(cur.execute(f"select {sel} from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "FusedOptimizerTensorListMetadata" in r[0] or "multi_tensor_apply_kernel" in r[0] and "Adam" in r[0]]
# group consecutive optimizer kernels; a step = from after the previous group's end to this group's end
groups = []
for i in marks:
    if groups and i - groups[-1][-1] < 40:
        groups[-1].append(i)
    else:
        groups.append([i])
if len(groups) < 2:
    lo, hi = 0, len(rows)
else:
    lo, hi = groups[-2][-1] + 1, groups[-1][-1] + 1
t0 = rows[lo][1]
print(f"# {hi - lo} kernels, {(rows[hi - 1][2] - t0) / 1e6:.3f} ms wall from first start to last end; columns: t_start_us dur_us gap_us grid wg name")
prev_end = t0
busy = 0
for r in rows[lo:hi]:
    name, s, e = r[0], r[1], r[2]
    extra = r[3:] if len(r) > 3 else ()
    busy += e - s
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f} {' '.join(str(x) for x in extra):>16s}  {name[:110]}")
    prev_end = max(prev_end, e)
print(f"# busy {busy / 1e6:.3f} ms")
