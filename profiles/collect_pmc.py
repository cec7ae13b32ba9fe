"""Turn the per-dispatch CSVs of two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md §HBM prescribes) into profiles/pmc_traffic.json, which bench.py reads to fill
roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_<wl>_FETCH_SIZE -- python3 bench.py --workload <wl> ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_<wl>_WRITE_SIZE -- python3 bench.py --workload <wl> ...
    python profiles/collect_pmc.py gpurun_out c2 t1m

Units / corrections (guide): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes
of wide coalesced reads (checked here on k_step_position: 2 x FETCH_SIZE = 60 B/body, WRITE_SIZE = 28 B/body,
both equal to the kernel's byte count), so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch. For the
4-byte plane loads of the solver rows the factor is uncalibrated (between 1 and 2)."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(path, counter):
    f = glob.glob(os.path.join(path, "*", "*counter_collection.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    rows = rows[int(len(rows) * 0.6):]  # the settled part of the run (timed region + profile pass)
    d = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


def main():
    base = sys.argv[1]
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB per launch averaged over the "
                     "last 40% of dispatches; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction)"}
    for wl in sys.argv[2:]:
        fe = per_kernel(os.path.join(base, f"pmc_{wl}_FETCH_SIZE"), "FETCH_SIZE")
        wr = per_kernel(os.path.join(base, f"pmc_{wl}_WRITE_SIZE"), "WRITE_SIZE")
        out[wl] = {k: {"fetch_kib": round(fe[k][0], 1), "write_kib": round(wr.get(k, (0, 0))[0], 1),
                       "launches_sampled": fe[k][1],
                       "traffic_bytes": int((2 * fe[k][0] + wr.get(k, (0, 0))[0]) * 1024)} for k in fe}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
