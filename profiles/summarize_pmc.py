"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into per-kernel HBM bytes per launch.

Usage: python profiles/summarize_pmc.py <fetch_csv> <write_csv> <out_json> [<libscythe_hip.so the passes ran with>]
The library's sha256 is stored under "_meta": bench.py reports roofline.traffic only when the library it has loaded is
that very build (a changed kernel makes the numbers stale).
Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a coalesced streaming read, so reads are doubled (checked here on k_nan_check, a pure read of a known
byte count: 2 x FETCH_SIZE = 0.955 of the algorithmic bytes); WRITE_SIZE is taken as is."""
import collections
import csv
import json
import sys


def timer_name(kernel):
    """rocprof kernel name -> the timer name bench.py reports (template variants that bench.py times separately keep
    separate rows: the node-space and ring-wise FFT inverses, the node-space and ring-wise HRBL kernels)."""
    k = kernel.split("(")[0].replace("void ", "").replace("sx::", "")
    base, _, targs = k.partition("<")
    targs = targs.rstrip(">").replace(" ", "").split(",") if targs else []
    if base == "k_rl_inverse_fft":           # <LOGL, COPYOUT, NODE, ST>
        return "k_node_fft" if targs[2] in ("true", "1") else "k_rl_inverse"
    if base == "k_phys_hrbl_mfma":           # ring-wise HRBL kernel: the inner rings when the cell kernel takes the rest
        return "k_phys_hrbl_inner"
    return {"k_fl_forward_fft": "k_fl_forward", "k_colmat": "k_zinv", "k_colmat_mfma": "k_zinv", "k_sbw": "k_sbz", "k_sbw_mfma": "k_sbz",
            "k_phys_hrbl_cell": "k_phys_hrbl"}.get(base, base)


def load(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = timer_name(r["Kernel_Name"])
            agg[k].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    lib = sys.argv[4] if len(sys.argv) > 4 else None
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        rd = 2.0 * 1024.0 * sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1)
        wr = 1024.0 * sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
        res[k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "launches_sampled": len(f.get(k, []))}
    if "k_phys_hrbl" not in res and "k_phys_hrbl_inner" in res:      # ring-wise build: one HRBL launch over all rings
        res["k_phys_hrbl"] = res.pop("k_phys_hrbl_inner")
    if lib:
        import hashlib
        res["_meta"] = {"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "lib": lib}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k.startswith("_"):
            continue
        print("%-22s read %8.1f MB  write %8.1f MB" % (k, v["read_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
