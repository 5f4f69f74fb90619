"""Copy the rocprofv3 summaries of a tools/pmc_run.sh run (gpurun_out/pmc_<tag>) into profiles/ as <round>_* files: counter
rows of this library's kernels only, the kernel-stats tables, the bench line, the rates the tools printed.
usage: python tools/collect_profiles.py r3a r3"""
import csv, glob, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
src = f"gpurun_out/pmc_{tag}"
dst = "profiles"
def counters(name, out):
    path = os.path.join(src, name)
    if not os.path.exists(path):
        print("missing", path); return
    rows = [r for r in csv.DictReader(open(path)) if "tamp_" in r["Kernel_Name"]]
    if not rows:
        print("no tamp rows in", path); return
    keep = [k for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count",
                        "Counter_Name", "Counter_Value") if k in rows[0]]
    with open(os.path.join(dst, out), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=keep); w.writeheader()
        for r in rows: w.writerow({k: r[k] for k in keep})
def copy(name, out):
    path = os.path.join(src, name)
    if os.path.exists(path): shutil.copy(path, os.path.join(dst, out))
    else: print("missing", path)
copy("bench.json", f"{rnd}_bench.json")
copy("stats_kernel_stats.csv", f"{rnd}_compress_kernel_stats.csv")
for c in ("fetch", "write", "sq", "sq2"):
    counters(f"{c}_counter_collection.csv", f"{rnd}_pmc_{c}_counter_collection.csv")
for c in ("prose", "markup", "python"):
    counters(f"realtext_{c}_sq_counter_collection.csv", f"{rnd}_pmc_realtext_{c}_sq_counter_collection.csv")
counters("c5_sq_counter_collection.csv", f"{rnd}_pmc_configs4_sq_counter_collection.csv")
for c in ("fetch", "write"):
    counters(f"dec2_{c}_counter_collection.csv", f"{rnd}_pmc_dec2_{c}_counter_collection.csv")
    counters(f"dec4_{c}_counter_collection.csv", f"{rnd}_pmc_dec4_{c}_counter_collection.csv")
copy("dec4_stats_kernel_stats.csv", f"{rnd}_decode_config4_1048576_kernel_stats.csv")
copy("dec2_stats_kernel_stats.csv", f"{rnd}_decode_variants_kernel_stats.csv")
copy("dec_split_kernel_stats.csv", f"{rnd}_decode_split_kernel_stats.csv")
counters("dec_split_sq_counter_collection.csv", f"{rnd}_pmc_decode_split_sq_counter_collection.csv")
copy("realtext_stats_kernel_stats.csv", f"{rnd}_realtext_kernel_stats.csv")
copy("longdec_v1_kernel_stats.csv", f"{rnd}_long_decode_v1_kernel_stats.csv")
copy("longdec_ext_kernel_stats.csv", f"{rnd}_long_decode_extended_kernel_stats.csv")
with open(os.path.join(dst, f"{rnd}_decode_and_realtext_rates.txt"), "w") as fh:
    for f in ("longdec_v1.log", "longdec_ext.log", "dec4_stats.log", "dec2_stats.log", "realtext_stats.log", "config5.log", "short_msgs.log", "dec_split_pmc.log",
              "realtext_prose_sq.log", "realtext_markup_sq.log", "realtext_python_sq.log", "dec2_fetch.log"):
        p = os.path.join(src, f)
        if os.path.exists(p):
            for line in open(p):
                if "GB/s" in line or "per stream" in line or line.startswith("decode "):
                    fh.write(f"[{f}] {line}")
# summary for the README
import collections
def summ(name, nstreams, label):
    path = os.path.join(dst, name)
    if not os.path.exists(path): return
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, m in acc.items():
        m = {c: sum(v) / len(v) for c, v in m.items()}
        line = f"{label} {k}: " + ", ".join(f"{c}={v:.4g}" for c, v in sorted(m.items()))
        if "SQ_INSTS_VALU" in m:
            line += f" | per stream VALU {m['SQ_INSTS_VALU']/nstreams:.0f} SALU {m.get('SQ_INSTS_SALU',0)/nstreams:.0f} LDS {m.get('SQ_INSTS_LDS',0)/nstreams:.0f}"
            if "GRBM_GUI_ACTIVE" in m and "SQ_ACTIVE_INST_VALU" in m:
                cyc = m["GRBM_GUI_ACTIVE"] / 8
                line += f" cycles {cyc/1e6:.2f}M VALU busy {100*m['SQ_ACTIVE_INST_VALU']*4/(1024*cyc):.0f}%"
        print(line)
summ(f"{rnd}_pmc_sq_counter_collection.csv", 65536, "configs[1]")
summ(f"{rnd}_pmc_sq2_counter_collection.csv", 65536, "configs[1]")
summ(f"{rnd}_pmc_fetch_counter_collection.csv", 65536, "configs[1]")
summ(f"{rnd}_pmc_write_counter_collection.csv", 65536, "configs[1]")
summ(f"{rnd}_pmc_realtext_prose_sq_counter_collection.csv", 32768, "prose ext")
summ(f"{rnd}_pmc_realtext_markup_sq_counter_collection.csv", 32768, "markup ext")
summ(f"{rnd}_pmc_realtext_python_sq_counter_collection.csv", 32768, "python ext")
summ(f"{rnd}_pmc_configs4_sq_counter_collection.csv", 2097152, "configs[4] share")
summ(f"{rnd}_pmc_dec2_fetch_counter_collection.csv", 65536, "decode 65536x4K")
summ(f"{rnd}_pmc_dec2_write_counter_collection.csv", 65536, "decode 65536x4K")
summ(f"{rnd}_pmc_dec4_fetch_counter_collection.csv", 1048576, "configs[3] decode")
summ(f"{rnd}_pmc_dec4_write_counter_collection.csv", 1048576, "configs[3] decode")
