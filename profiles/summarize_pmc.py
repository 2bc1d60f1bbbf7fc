"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into per-kernel HBM bytes per launch.

Usage: python profiles/summarize_pmc.py <fetch_csv> <write_csv> <out_json>
Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a coalesced streaming read, so reads are doubled (checked here on k_nan_check, a pure read of a known
byte count: 2 x FETCH_SIZE = 0.955 of the algorithmic bytes); WRITE_SIZE is taken as is."""
import collections
import csv
import json
import sys


def load(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sx::", "")
            k = k.split("<")[0]
            agg[k].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        rd = 2.0 * 1024.0 * sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1)
        wr = 1024.0 * sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
        res[k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "launches_sampled": len(f.get(k, []))}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-22s read %8.1f MB  write %8.1f MB" % (k, v["read_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
