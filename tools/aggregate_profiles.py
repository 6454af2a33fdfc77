"""Turns the raw rocprofv3 outputs of one profiling session (gpurun_out/<dir>) into the summaries kept under
profiles/:  kernel stats (from --kernel-trace --stats) and HBM traffic per launch (from two --pmc passes,
FETCH_SIZE and WRITE_SIZE, which rocprofv3 reports in KB per dispatch).

  python tools/aggregate_profiles.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <tag> [workload key, default 10000x5x4]

Every record written to profiles/pmc_traffic.json carries the hash of the kernel sources it was measured on (bench.kernel_sources_sha);
bench.py reports roofline.traffic = null for a record whose hash is not the current one.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def pmc(dirname, counter):
    f = max(glob.glob(os.path.join(dirname, "*", "*counter_collection.csv")), key=os.path.getmtime)   # newest run (gpurun merges, never deletes)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    stats_dir, fdir, wdir, tag = sys.argv[1:5]
    wkey = sys.argv[5] if len(sys.argv) > 5 else "10000x5x4"
    sys.path.insert(0, ROOT)
    from bench import kernel_sources_sha
    f = max(glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    out = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent"])
        for r in rows:
            if float(r["Percentage"]) < 0.01:
                continue
            w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                        "%.4f" % (float(r["AverageNs"]) / 1e6), r["Percentage"]])
    fe, wr = pmc(fdir, "FETCH_SIZE"), pmc(wdir, "WRITE_SIZE")
    out2 = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.csv")
    with open(out2, "w") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg"])
        for k in sorted(fe):
            if k.startswith("at::") or k.startswith("__amd"):
                continue
            w.writerow([k, len(fe[k]), "%.3f" % (sum(fe[k]) / len(fe[k])), "%.3f" % (sum(wr[k]) / max(1, len(wr[k])))])
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    rec = json.load(open(tp)) if os.path.exists(tp) else {}
    unit = ("bytes per launch (chain_kernel: one launch = the whole step of 10,000 frames; ik1 / als4: one time step of all 625 chains); rocprofv3 --pmc "
            "FETCH_SIZE and WRITE_SIZE in separate passes, KB x 1024.  fetch_counter_bytes is the counter as read; fetch_bytes = that x "
            "fetch_correction, the factor MEASURED on a calibration kernel with the same access pattern and a known byte count "
            "(tools/fetch_calib.hip: on gfx950 the counter tallies a 128-byte request of a streaming read at 64 bytes); 1.0 for the chain "
            "kernels, whose accesses are 8 / 12-byte pieces of small per-frame tables (ingest_kernel calibrates 1:1 against its known 30 MB input)")
    # calibration factors of this session, if the session ran tools/fetch_calib.hip (prof_r04.sh): known bytes / counter
    calib = {}
    cdir = os.path.join(os.path.dirname(os.path.normpath(fdir)), "calib")
    if os.path.isdir(cdir):
        cal = pmc(cdir, "FETCH_SIZE")
        known = 4000000 * 300.0
        for k, v in cal.items():
            if k.startswith("calib_"):
                calib[k] = known / (sum(v) / len(v) * 1024)
        print("FETCH_SIZE calibration (known bytes / counter):", {k: round(v, 3) for k, v in calib.items()})
    for key, kern in (("ik", "ik1_kernel<6>"), ("als", "als4_kernel<double, 32>"), ("chain", "chain_kernel<false>"), ("chain", "chain_kernel<true>"),
                      ("tri", "dlt_kernel"), ("tri", "ingest_dlt_kernel<float>"), ("tri", "ingest_dlt_kernel<double>"), ("tri", "ingest_dlt3_kernel<5>"),
                      ("tri", "ingest_dlt3_kernel<0>"), ("tri", "ingest_dlt3_kernel<5, false>"), ("tri", "ingest_dlt3_kernel<0, false>"),
                      ("tri", "ingest_dlt3_kernel<5, true>"), ("tri", "ingest_dlt3_kernel<0, true>"), ("assoc", "als4_kernel<float, 24>")):
        if kern not in fe or (key == "assoc" and not wkey.startswith("assoc_dlt:")):
            continue
        if key != "assoc" and wkey.startswith("assoc_dlt:"):
            continue
        kf = sum(fe[kern]) / len(fe[kern]) * 1024
        kw = sum(wr[kern]) / len(wr[kern]) * 1024
        corr = {"fetch_counter_bytes": kf, "fetch_correction": 1.0}
        pattern = "calib_gather_dma" if kern.startswith("ingest_dlt3") else ("calib_copy12" if kern.startswith("ingest_dlt_kernel") else None)
        if pattern:
            if pattern in calib:
                corr = {"fetch_counter_bytes": kf, "fetch_correction": calib[pattern],
                        "why": f"measured in the same session on tools/fetch_calib.hip's {pattern} (same access pattern, known byte count)"}
            else:
                # no calibration in this session: MI355X_MICROARCH.md (HBM / rocprofv3) -- gfx950 FETCH_SIZE reports HALF of the bytes of a
                # wide coalesced streaming read
                corr = {"fetch_counter_bytes": kf, "fetch_correction": 2.0,
                        "why": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a coalesced streaming read (MI355X_MICROARCH.md); not calibrated in this session"}
            kf *= corr["fetch_correction"]
        wk = wkey.split(":", 1)[-1]     # ("assoc_dlt:10000x5x4" selects config 3's record: its dominant kernel also runs at the chain heads of config 4)
        mix = (rec.get(f"{key}:{wk}") or {}).get("inst_mix")   # tools/aggregate_insts.py's record carries its own source hash
        rec[f"{key}:{wk}"] = {"kernel": kern, "fetch_bytes": kf, "write_bytes": kw, "bytes": kf + kw,
                                "source": os.path.basename(out2), "src_sha": kernel_sources_sha(), "unit": unit}
        if mix:
            rec[f"{key}:{wk}"]["inst_mix"] = mix
        rec[f"{key}:{wk}"].update(corr)
    json.dump(rec, open(tp, "w"), indent=1)
    print(open(out).read())
    print(open(out2).read())


if __name__ == "__main__":
    main()
