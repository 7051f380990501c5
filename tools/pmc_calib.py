"""Factor per access shape: requested bytes / (FETCH_SIZE or WRITE_SIZE x 1024), from the two counter passes of
tools/pmc_calib.sh.  Writes profiles/pmc_calibration.json; tools/pmc_traffic.py applies the factors by kernel."""
import collections
import csv
import json
import re
import sys


def dispatches(d, counter):
    acc = collections.defaultdict(list)
    rows = list(csv.DictReader(open(d + "/run_counter_collection.csv")))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        if r["Counter_Name"] == counter:
            acc[re.sub(r"\(.*", "", r["Kernel_Name"]).strip()].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def main():
    fdir, wdir, req_file, out = sys.argv[1:5]
    req = json.loads(open(req_file).read().strip().splitlines()[-1])
    f, w = dispatches(fdir, "FETCH_SIZE"), dispatches(wdir, "WRITE_SIZE")
    shapes = {}

    def second_rep(vals, per_rep, idx):
        return vals[per_rep + idx]                      # rep 0: launches [0, per_rep), rep 1: the ones measured

    shapes["stream_read_16B_per_lane"] = {"requested": req["k_stream_read"]["read"],
                                          "counter": second_rep(f["k_stream_read"], 1, 0)}
    for i, tag in enumerate(("gather_64B_rows_table_1GiB", "gather_64B_rows_table_64MiB")):
        shapes[tag] = {"requested": req["k_gather64"][i]["read"], "counter": second_rep(f["k_gather64"], 2, i)}
    for i, tag in enumerate(("strided_runs_64B", "strided_runs_128B")):
        shapes[tag] = {"requested": req["k_strided_runs"][i]["read"], "counter": second_rep(f["k_strided_runs"], 2, i)}
    shapes["stream_write_16B_per_lane"] = {"requested": req["k_stream_write"]["write"],
                                           "counter": second_rep(w["k_stream_write"], 1, 0)}
    shapes["scatter_write_4B"] = {"requested": req["k_scatter4"]["write"], "counter": second_rep(w["k_scatter4"], 1, 0)}
    shapes["runs_write_64B"] = {"requested": req["k_runs_write"]["write"], "counter": second_rep(w["k_runs_write"], 1, 0)}
    for v in shapes.values():
        v["factor"] = round(v["requested"] / v["counter"], 4) if v["counter"] else None
    doc = {"what": "bytes requested by tools/pmc_calib.hip / (rocprofv3 FETCH_SIZE or WRITE_SIZE in KB x 1024), per access "
                   "shape, MI355X (gfx950), ROCm 7.2; multiply a kernel's counter by the factor of ITS shape",
           "shapes": shapes}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in shapes.items():
        print("%-32s requested %14d  counter %14.0f  factor %s" % (k, v["requested"], v["counter"], v["factor"]))


if __name__ == "__main__":
    main()
