"""Per-kernel means of SQ counters from rocprofv3 --pmc counter_collection.csv files (one or more passes):
   python3 profiles/summarize_sq.py pass1.csv [pass2.csv ...]
Counter values are summed over the chip's shader engines by rocprofv3; ratios of counters of the same pass are what is
meaningful (e.g. SQ_ACTIVE_INST_LDS / SQ_BUSY_CYCLES, SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS, SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES)."""
import csv, sys, re
from collections import defaultdict

def short(name):
    m = re.search(r"sx::(k_[a-z_0-9]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(44) + "".join(c.replace("SQ_", "")[:18].rjust(20) for c in ctrs))
for k in sorted(acc):
    print(k[:43].ljust(44) + "".join(("%.3e" % (sum(acc[k][c]) / len(acc[k][c]))).rjust(20) if acc[k][c] else "".rjust(20) for c in ctrs))
def ratio(k, a, b):
    if acc[k][a] and acc[k][b] and sum(acc[k][b]) > 0:
        return sum(acc[k][a]) / len(acc[k][a]) / (sum(acc[k][b]) / len(acc[k][b]))
    return None
print()
print("kernel".ljust(44) + "VALU/busy".rjust(12) + "LDS/busy".rjust(12) + "MFMA/busy".rjust(12) + "VMEM/busy".rjust(12) + "conflict/LDS".rjust(14) + "wait/wave".rjust(12))
for k in sorted(acc):
    vals = [ratio(k, "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"), ratio(k, "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"), ratio(k, "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"),
            ratio(k, "SQ_ACTIVE_INST_VMEM", "SQ_BUSY_CYCLES"), ratio(k, "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"), ratio(k, "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")]
    print(k[:43].ljust(44) + "".join(("%.3f" % v).rjust(w) if v is not None else "-".rjust(w) for v, w in zip(vals, (12, 12, 12, 12, 14, 12))))
