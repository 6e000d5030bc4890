"""Developer probe: how much of a latency-bound program is spent BETWEEN kernels?  Reads a rocprofv3 --kernel-trace csv of
`bench.py --batch 1` (one find per step: 69 dependent levels of 2-7 bootstraps after the four wide ones) and reports, for the
kernels of the timed steps, the sum of kernel durations, the sum of the idle gaps between consecutive kernels and the gap
distribution -- the number a hipGraph could at best remove."""
import csv
import glob
import sys

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof_gaps/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the single-find section: levels whose blind rotate runs <= 8 workgroups cannot be told from the trace; take every run of
# kernels whose blind-rotate launches last < 3.6 ms (one workgroup per CU) as "latency-bound"
br = [(s, e) for s, e, n in rows if "blind_rotate" in n]
small = [(s, e) for s, e in br if (e - s) < 3.6e6]
print(f"{len(rows)} kernels, {len(br)} blind-rotate launches, {len(small)} of them under 3.6 ms")
gaps, durs = [], []
for i in range(1, len(rows)):
    s0, e0, n0 = rows[i - 1]
    s1, e1, n1 = rows[i]
    if "blind_rotate" in n1 and (e1 - s1) < 3.6e6 or "blind_rotate" in n0 and (e0 - s0) < 3.6e6:
        g = s1 - e0
        if 0 <= g < 1e6:
            gaps.append(g)
for s, e in small:
    durs.append(e - s)
gaps = np.array(gaps, dtype=np.float64)
print(f"small blind-rotate launches: {np.sum(durs) / 1e6:.1f} ms in {len(durs)} launches, mean {np.mean(durs) / 1e3:.1f} us")
print(f"idle gaps next to them: {np.sum(gaps) / 1e6:.2f} ms in {len(gaps)} gaps, median {np.median(gaps) / 1e3:.1f} us, "
      f"p90 {np.percentile(gaps, 90) / 1e3:.1f} us = {100 * np.sum(gaps) / np.sum(durs):.1f} % of the kernel time")
