#!/usr/bin/env python3
"""Launch sequence of ONE training step from a rocprofv3 --kernel-trace csv: every dispatch in start order with its duration and the idle
gap behind the previous kernel -- shows which small launches (fills, copies, casts) sit between the real kernels and what they cost at
small per-GPU batches, where a 2 us fill + its ~1.3 us boundary is no longer noise.

  on the GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $REPO/bench.py --global-batch 4 \
                       --steps 3 --warmup 2 --no-extras --no-cpu-baseline
                   python tools/launch_sequence.py /tmp/tr > gpurun_out/launch_sequence_b4.txt
The step boundaries are found from the optimizer launch (adam_multi_kernel); the LAST complete step is listed."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for a, b in (("at::native::vectorized_elementwise_kernel<4, at::native::", "aten:"), ("at::native::", "aten:"), ("din_gather::", ""), ("din_wgrad::", "")):
        name = name.replace(a, b)
    cut = name.find("(")
    if cut > 0 and not name.startswith("aten:"):
        name = name[:cut]
    return name[:96]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no *kernel_trace.csv under {root}"
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adam_multi_kernel" in r[2]]
    assert len(ends) >= 2, "need at least two optimizer launches to bracket a step"
    step = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = rows[ends[-2]][1]
    print(f"# last complete step: {len(step)} launches, {(step[-1][1] - t0) / 1e6:.3f} ms from the previous optimizer's end to this one's end")
    busy = sum(e - s for s, e, _ in step)
    print(f"# kernel time {busy / 1e6:.3f} ms, idle between kernels {(step[-1][1] - t0 - busy) / 1e6:.3f} ms")
    agg = defaultdict(lambda: [0, 0, 0])
    prev = t0
    for s, e, n in step:
        a = agg[short(n)]
        a[0] += 1; a[1] += e - s; a[2] += max(0, s - prev)
        prev = max(prev, e)
    print("# per kernel: launches, total us, total idle-gap us in FRONT of its launches")
    for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:60]:
        print(f"#  {c:4d} {d / 1e3:9.1f} {g / 1e3:8.1f}  {n}")
    print("# sequence: t_start_us  dur_us  gap_us  kernel")
    prev = t0
    for s, e, n in step:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:7.1f}  {short(n)}")
        prev = max(prev, e)


if __name__ == "__main__":
    main()
