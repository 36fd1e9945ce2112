"""Derived SQ-counter table from ONE `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d DIR -- <bench>` pass (no trace flags beside it), per kernel:
cycles/XCD = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs); MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles/XCD);
wait% = SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue-stall% = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; lds-stall% = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES.
    python tools/pmc_sq_table.py DIR [substring of the kernel names to keep]"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name") or row.get("kernel_name")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
keep = sys.argv[2] if len(sys.argv) > 2 else ""
print(f"{'kernel':70s} {'launches':>8s} {'cycles/XCD':>11s} {'MfmaUtil':>9s} {'wait%':>7s} {'issue-stall%':>13s} {'lds-stall%':>11s} {'bank-confl/mfma-cyc':>20s} {'VALU/mfma-cyc':>14s}")
for k in sorted(acc):
    d = acc[k]
    if keep not in k or "GRBM_GUI_ACTIVE" not in d:
        continue
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    mf = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    wc = max(avg.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    name = k.replace("void ", "").replace("(anonymous namespace)::", "")
    name = (name[:name.index(">(") + 1] if ">(" in name else name.split("(")[0])[:70]
    print(f"{name:70s} {len(d['GRBM_GUI_ACTIVE']):8d} {cyc:11.4g} {mf / (1024.0 * cyc):9.3f} {100 * avg.get('SQ_WAIT_ANY', 0) / wc:7.1f} "
          f"{100 * avg.get('SQ_WAIT_INST_ANY', 0) / wc:13.1f} {100 * avg.get('SQ_WAIT_INST_LDS', 0) / wc:11.1f} "
          f"{avg.get('SQ_LDS_BANK_CONFLICT', 0) / max(mf, 1.0):20.4f} {avg.get('SQ_INSTS_VALU', 0) / max(mf, 1.0):14.3f}")
