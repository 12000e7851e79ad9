#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes over bench.py (FETCH_SIZE, WRITE_SIZE; kernel-trace only, separate passes as
MI355X_MICROARCH.md prescribes) into per-kernel HBM bytes per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -o x -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -o x -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py out/fetch/x_counter_collection.csv out/write/x_counter_collection.csv profiles/r01_pmc_traffic.json

Corrections (guide, HBM section): the counters are in KiB-like units of 1 KB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads -> doubled; WRITE_SIZE is taken as is (checked here against conv_small_kernel<4,32,...>, whose two launches
per step write 1.41 GB each: 1373 MB counted)."""
import collections, csv, json, re, sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("din_wgrad::", "").replace("din_gather::", "")
    return name.split("(")[0]


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {}
    for k, v in f.items():
        wv = w.get(k, [])
        res[k] = dict(launches=len(v), fetch_bytes_per_launch=2.0 * 1e3 * sum(v) / len(v),
                      write_bytes_per_launch=1e3 * sum(wv) / len(wv) if wv else None)
    workload = sys.argv[4] if len(sys.argv) > 4 else "inv3_bf16"
    global_batch = int(sys.argv[5]) if len(sys.argv) > 5 else 32
    json.dump(dict(workload=workload, global_batch=global_batch, n_gpus=1, source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1 --no-cpu-baseline`; "
                          "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as counted", kernels=res), open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
