#!/usr/bin/env python3
"""Per-launch averages of the counters collected by tools/pmc_layer.sh for the kernels whose name contains <substr>."""
import collections, csv, glob, sys

out, sub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for path in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x) / len(x) for k, x in agg.items()}
for k in sorted(v):
    print(f"{k:28s} {v[k]:16.4g}   ({len(agg[k])} launches)")
g = v.get
if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
    print(f"MFMA busy / SQ busy cycles          {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES'):.3f}   (per-SIMD busy cycles over per-SE busy cycles: compare between variants only)")
if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
    print(f"MFMA busy cycles per SIMD-cycle     {g('SQ_VALU_MFMA_BUSY_CYCLES') / (g('GRBM_GUI_ACTIVE') * 1024):.3f}   (1024 SIMDs)")
if g("SQ_WAVE_CYCLES"):
    w = g("SQ_WAVE_CYCLES")
    print("wave cycles: ACTIVE_INST_ANY %.2f  WAIT_ANY %.2f  WAIT_INST_ANY %.2f  (WAIT_INST_LDS %.2f)" % (
        g("SQ_ACTIVE_INST_ANY", 0) / w, g("SQ_WAIT_ANY", 0) / w, g("SQ_WAIT_INST_ANY", 0) / w, g("SQ_WAIT_INST_LDS", 0) / w))
if g("SQ_LDS_IDX_ACTIVE"):
    print(f"LDS bank conflict cycles / LDS active cycles   {g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.3f}")
if g("SQ_INSTS_MFMA"):
    m = g("SQ_INSTS_MFMA")
    print("per MFMA: SALU %.2f  VALU(non-MFMA) %.2f  LDS %.2f  VMEM_RD %.3f" % (
        g("SQ_INSTS_SALU", 0) / m, (g("SQ_INSTS_VALU", 0) - m) / m, g("SQ_INSTS_LDS", 0) / m, g("SQ_INSTS_VMEM_RD", 0) / m))
if g("TCC_REQ_sum"):
    print(f"L2 hit rate {g('TCC_HIT_sum', 0) / max(g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0), 1):.3f}  requests {g('TCC_REQ_sum'):.4g} (x128 B = {g('TCC_REQ_sum') * 128 / 1e9:.2f} GB)")
if g("FETCH_SIZE") is not None:
    print(f"HBM read {2e3 * g('FETCH_SIZE') / 1e9:.3f} GB per launch (FETCH_SIZE x2, gfx950 wide-read correction), write {1e3 * g('WRITE_SIZE', 0) / 1e9:.3f} GB")
