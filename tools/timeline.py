"""Per-step timeline from a rocprofv3 --kernel-trace CSV: for the last full train step, every kernel in start order with its
stream (queue), duration and the idle gap since the previous kernel END on the same queue; totals of busy / idle per queue.

    python tools/timeline.py gpurun_out/prof_<tag>/stats/**/..._kernel_trace.csv [--first conv1_fwd] [--back N]

--back N: the step N steps before the last full one (bench.py ends with nine untimed steps on ONE stream for `roofline.alone`;
--back 12 shows a step of the timed region, weight gradients on the second stream).
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    opts = dict(zip(sys.argv[2::2], sys.argv[3::2]))
    first = opts.get("--first", "conv_w_prep")
    back = int(opts.get("--back", "0"))
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    # a step starts with the first `first` kernel after a non-`first` kernel
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"] and (i == 0 or first not in rows[i - 1]["Kernel_Name"])]
    starts = [i for j, i in enumerate(starts) if j == 0 or i - starts[j - 1] > 20]
    if len(starts) < 3 + back:
        print("steps not found", len(starts))
        return
    a, b = starts[-3 - back], starts[-2 - back]
    step = rows[a:b]
    t0 = step[0]["s"]
    print(f"step wall (start of first kernel to start of next step): {(rows[b]['s'] - t0) / 1e3:.1f} us, {len(step)} kernels")
    last_end = defaultdict(lambda: None)
    busy = defaultdict(float)
    for r in step:
        q = r.get("Queue_Id", "?")
        gap = (r["s"] - last_end[q]) / 1e3 if last_end[q] is not None else 0.0
        last_end[q] = max(r["e"], last_end[q] or 0)
        dur = (r["e"] - r["s"]) / 1e3
        busy[q] += dur
        name = r["Kernel_Name"]
        name = name[:70]
        print(f"{(r['s'] - t0) / 1e3:9.1f} us  q{q:>3}  dur {dur:8.1f}  gap {gap:7.1f}  {name}")
    for q, v in busy.items():
        print(f"queue {q}: busy {v:.1f} us")
    # union of busy intervals over all queues
    iv = sorted((r["s"], r["e"]) for r in step)
    cur_s, cur_e, tot = iv[0][0], iv[0][1], 0
    for s, e in iv[1:]:
        if s > cur_e:
            tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    tot += cur_e - cur_s
    print(f"GPU busy (any queue): {tot / 1e3:.1f} us; idle inside the step: {(rows[b]['s'] - t0 - tot) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
