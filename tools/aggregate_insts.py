"""Instruction mix and pipe occupancy of one kernel from the rocprofv3 --pmc passes of tools/prof_insts.sh (+ the SQ pass of
tools/prof_r02.sh):  python tools/aggregate_insts.py <insts dir> <sq dir> <prefix> <key> [kernel substring]
Writes profiles/<prefix>_inst_mix.txt (the per-launch counter means, as collected) and adds an `inst_mix` record to the entry <key>
of profiles/pmc_traffic.json (bench.py copies it into roofline.fp64 when the kernel sources are the ones profiled).
Units (MI355X_MICROARCH.md): SQ_INSTS_* count wave instructions; SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # noqa: E402


def means(d, sub):
    acc = collections.defaultdict(list)
    for f in [max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)]:   # newest run
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    insts, sq, prefix, key = sys.argv[1:5]
    sub = sys.argv[5] if len(sys.argv) > 5 else "chain_kernel"
    mfma_dir = sys.argv[6] if len(sys.argv) > 6 else None      # optional: the pass with SQ_INSTS_MFMA (config 5: als5's matrix-core products)
    c = {}
    for p in ("a", "b", "c", "d"):
        c.update(means(os.path.join(insts, p), sub))
    c.update(means(sq, sub))
    if mfma_dir:
        c.update(means(mfma_dir, sub))
    valu = c["SQ_INSTS_VALU"]
    f64 = {k: c[f"SQ_INSTS_VALU_{k}_F64"] for k in ("FMA", "ADD", "MUL", "TRANS")}
    flop = 64.0 * (2 * f64["FMA"] + f64["ADD"] + f64["MUL"] + f64["TRANS"])      # all 64 lanes of every wave instruction
    rec = {
        "valu_insts": valu, "f64_insts": f64, "f64_share_of_valu": sum(f64.values()) / valu,
        "int_share_of_valu": (c["SQ_INSTS_VALU_INT32"] + c["SQ_INSTS_VALU_INT64"]) / valu,
        "lanes_active_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / valu,
        "flop_per_launch": flop,
        "valu_busy_simd_cycles": 4.0 * c["SQ_ACTIVE_INST_VALU"], "lds_busy_cu_cycles": c["SQ_LDS_IDX_ACTIVE"],
        "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"], "salu_insts": c["SQ_INSTS_SALU"], "lds_insts": c["SQ_INSTS_LDS"],
        "wave_cycles": 4.0 * c["SQ_WAVE_CYCLES"], "src_sha": kernel_sources_sha(), "source": f"{prefix}_inst_mix.txt",
    }
    if "SQ_INSTS_MFMA" in c:
        # v_mfma_f64_16x16x4_f64: 16 x 16 x 4 multiply-adds = 2,048 flop per wave instruction, on the SAME pipe as the vector FMAs
        # (profiles/r03_mfma_f64_rate.txt), so vector + matrix flop share one peak
        rec["mfma_insts"] = c["SQ_INSTS_MFMA"]
        rec["mfma_flop_per_launch"] = 2048.0 * c["SQ_INSTS_MFMA"]
        rec["flop_per_launch"] = flop + rec["mfma_flop_per_launch"]
        rec["mfma_share_of_flop"] = rec["mfma_flop_per_launch"] / rec["flop_per_launch"]
    with open(os.path.join(ROOT, "profiles", f"{prefix}_inst_mix.txt"), "w") as f:
        f.write(f"# per-launch means of {sub} (rocprofv3 --pmc, separate passes: tools/prof_insts.sh, tools/prof_r02.sh)\n")
        for k in sorted(c):
            f.write(f"{k} {c[k]:.0f}\n")
        f.write("# derived\n")
        for k, v in rec.items():
            f.write(f"{k} {v}\n")
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(tp))
    d.setdefault(key, {})["inst_mix"] = rec
    json.dump(d, open(tp, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
