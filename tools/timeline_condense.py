"""Condenses a tools/timeline.py listing: runs of the same kernel on the main queue become one line (count, summed duration, summed gap).
    python tools/timeline_condense.py gpurun_out/tl_<workload>/timeline.txt [queue]"""
import re
import sys

rows = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+) us  q\s+(\d+)\s+dur\s+([\d.]+)\s+gap\s+(-?[\d.]+)\s+(.*)', l)
    if m:
        rows.append((float(m[1]), int(m[2]), float(m[3]), float(m[4]), m[5]))
queue = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def short(n):
    n = re.sub(r'void |\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+|at::native::', '', n)
    return n[:44]


out = []
for t, q, d, g, n in rows:
    if q != queue:
        continue
    s = short(n)
    if out and out[-1][0] == s:
        out[-1][1] += 1; out[-1][2] += d; out[-1][3] += g
    else:
        out.append([s, 1, d, g, t])
for s, c, d, g, t in out:
    print(f"{t:9.1f} {s:46s} x{c:3d} dur {d:8.1f} gap {g:7.1f}")
print("queue", queue, "kernels", sum(1 for r in rows if r[1] == queue), "busy", round(sum(r[2] for r in rows if r[1] == queue), 1),
      "gaps", round(sum(r[3] for r in rows if r[1] == queue), 1))
