#!/usr/bin/env python3
"""Turns rocprofv3 output directories (copied back from the GPU box under gpurun_out/) into the small summaries kept
under profiles/.  Usage:

    python profiles/summarize.py --tag r02 --commit <sha> --stats gpurun_out/prof_b/headline_kernel_stats.csv \
        --fetch gpurun_out/pmc_fetch/f_counter_collection.csv --write gpurun_out/pmc_write/w_counter_collection.csv \
        --mfma gpurun_out/pmc_mfma/m_counter_collection.csv --workload n200000_symmetric

The PMC passes are separate runs of the same command (FETCH_SIZE and WRITE_SIZE do not fit one pass), each with
--kernel-trace only.  Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE reports half of the bytes of a wide coalesced streaming read and is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Launches of one kernel are grouped by grid size, because the
symmetric sweep runs with different schedules (column groups per launch, block rows per workgroup)."""
import argparse, collections, csv, json, os, shutil


def per_kernel(path, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] not in counters:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            key = (name, int(row["Grid_Size"]), int(row["Workgroup_Size"]))
            acc[key][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"]),
                                                  int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    # a kernel launched with dozens of different grids (the set-up kernel that generates the tiles block row by block row: 782
    # grids at N=200000) is one line, not 782: its launches are pooled under grid size 0
    grids = collections.Counter(k[0] for k in acc)
    for key in [k for k in acc if grids[k[0]] > 16]:
        pooled = acc[(key[0], 0, key[2])]
        for cname, vals in acc.pop(key).items():
            pooled[cname].extend(vals)
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True); ap.add_argument("--commit", default="")
    ap.add_argument("--stats"); ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--mfma")
    ap.add_argument("--workload", required=True); ap.add_argument("--command", default="")
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    if a.stats:
        shutil.copy(a.stats, os.path.join(here, f"{a.tag}_kernel_stats_{a.workload}.csv"))
    if a.fetch and a.write:
        fe, wr = per_kernel(a.fetch, {"FETCH_SIZE"}), per_kernel(a.write, {"WRITE_SIZE"})
        out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only; KB; FETCH_SIZE doubled "
                       "(gfx950: it reports half of a wide coalesced streaming read), WRITE_SIZE exact; per launch, grouped by grid size",
               "commit": a.commit, "command": a.command, "kernels": []}
        for key in sorted(fe, key=lambda k: -sum(v[1] for v in fe[k]["FETCH_SIZE"])):
            f = [v[1] for v in fe[key]["FETCH_SIZE"]]
            w = [v[1] for v in wr.get(key, {}).get("WRITE_SIZE", [])]
            favg, wavg = sum(f) / len(f), (sum(w) / len(w) if w else 0.0)
            out["kernels"].append({"kernel": key[0], "grid_size": key[1], "workgroup_size": key[2], "launches": len(f),
                                   "fetch_KB_avg": round(favg, 1), "write_KB_avg": round(wavg, 1),
                                   "hbm_bytes_per_launch_corrected": round((2.0 * favg + wavg) * 1024.0)})
        with open(os.path.join(here, f"{a.tag}_pmc_traffic_{a.workload}.json"), "w") as f:
            json.dump(out, f, indent=1)
    if a.mfma:
        mm = per_kernel(a.mfma, {"GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"})
        out = {"note": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace; per launch averages. "
                       "GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock_GHz = GRBM_GUI_ACTIVE / 8 / duration (meaningless for kernels "
                       "shorter than ~50 us); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)", "commit": a.commit, "command": a.command, "kernels": []}
        for key in sorted(mm, key=lambda k: -sum(v[2] for v in mm[k]["GRBM_GUI_ACTIVE"])):
            g = mm[key]["GRBM_GUI_ACTIVE"]; b = mm[key].get("SQ_VALU_MFMA_BUSY_CYCLES", [])
            gavg = sum(v[1] for v in g) / len(g); davg = sum(v[2] for v in g) / len(g)
            bavg = sum(v[1] for v in b) / len(b) if b else 0.0
            out["kernels"].append({"kernel": key[0], "grid_size": key[1], "launches": len(g), "duration_us_avg": round(davg / 1e3, 1),
                                   "clock_GHz": round(gavg / 8.0 / davg, 3) if davg else None,
                                   "mfma_busy": round(bavg / (gavg / 8.0 * 1024.0), 4) if gavg else None})
        with open(os.path.join(here, f"{a.tag}_pmc_mfma_{a.workload}.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
