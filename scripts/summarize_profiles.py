"""Turn gpurun_out/prof_<tag>/ (written by scripts/profile_round.sh on the GPU box) into the committed summaries:
profiles/<tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_summary.csv and profiles/pmc_traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 counts a 128-B read as 64 B in FETCH_SIZE
(MI355X_MICROARCH.md, HBM / rocprofv3 section); the two counters come from SEPARATE passes.
usage: python scripts/summarize_profiles.py <tag>
"""
import collections, csv, glob, json, os, re, shutil, sys
from pathlib import Path

tag = sys.argv[1]
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles"


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0].replace(" ", "")
    # conv_igemm16_kernel<..., SLV>: the sliver instantiation (the 3 launches per step with a 16 n + 4 output width) is the same kernel
    # template as the plain one; the PMC summary and pmc_traffic.json aggregate both under the name without the flag
    name = re.sub(r"^(conv_igemm16_kernel<[0-9,]+),(?:true|false)>$", r"\1>", name)
    # conv_bf16_t256_kernel<NTOT,TW>: one kernel template, instantiated per channel-tile count of the block and patch width; aggregated under the bare name
    return re.sub(r"^conv_bf16_t256_kernel<[0-9,]+(?:,(?:true|false))?>$", "conv_bf16_t256_kernel", name)


def keyed(name: str, grid_threads: int) -> str:
    """conv_bf16_t256_kernel serves the large layers (>= 512 workgroups: the launches bench.py's roofline follows, variant ...7) and, since the
    second half of round 3, narrow-block / small-grid launches (variant ...6): two rows, split by grid size"""
    k = short(name)
    m = re.search(r"conv_bf16_t256_kernel<\s*(\d+),\s*(\d+)(?:,\s*([A-Za-z ]+))?(?:,\s*(?:true|false))?>", name)
    if m is not None:
        base = "conv_bf16_t256_kernel<float>" if (m.group(3) and "float" in m.group(3)) else "conv_bf16_t256_kernel"     # (the fp32 form of the same template)
        large = int(m.group(1)) >= 5 and int(m.group(2)) == 32 and grid_threads >= 512 * 256
        return base if large else base + "[narrow / small]"
    return k


def pmc(dirname):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    f = sorted(glob.glob(str(src / dirname / "*" / "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)   # newest run first
    if not f:
        return out
    for r in csv.DictReader(open(f[0])):
        out[keyed(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def durations(dirname):
    out = collections.defaultdict(list)
    f = sorted(glob.glob(str(src / dirname / "*" / "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            out[keyed(r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


shutil.copy(src / "bench.json", dst / f"{tag}_bench.json")
stats = sorted(glob.glob(str(src / "stats" / "*" / "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)[0]
shutil.copy(stats, dst / f"{tag}_bench_kernel_stats.csv")

sd = durations("stats")
split = {k: {"launches": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 2), "total_ms": round(sum(v) / 1e6, 3)} for k, v in sd.items() if k.startswith("conv_bf16_t256_kernel")}
if split:
    json.dump(split, open(dst / f"{tag}_t256_by_grid.json", "w"), indent=1)
fetch, write, sq, sqdur, lds = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq"), durations("pmc_sq"), pmc("pmc_lds")
traffic = {}
rows = []
code_hash = (src / "code_hash.txt").read_text().strip() if (src / "code_hash.txt").exists() else None
for k in sorted(set(fetch) | set(write) | set(sq) | set(lds)):
    fa = fetch[k].get("FETCH_SIZE", [])
    wa = write[k].get("WRITE_SIZE", [])
    row = {"kernel": k, "launches": len(fa) or len(wa)}
    if fa and wa:
        f_avg, w_avg = sum(fa) / len(fa), sum(wa) / len(wa)
        hbm = (2 * f_avg + w_avg) * 1024
        row.update(FETCH_SIZE_KB_avg=round(f_avg, 1), WRITE_SIZE_KB_avg=round(w_avg, 1), hbm_bytes_per_launch=int(hbm))
        traffic[k] = {"launches_sampled": len(fa), "steps_sampled": 3, "FETCH_SIZE_KB_avg": round(f_avg, 1), "WRITE_SIZE_KB_avg": round(w_avg, 1),
                      "hbm_bytes_per_launch": int(hbm), "source": f"profiles/{tag}_pmc_summary.csv", "code": code_hash,
                      "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B reads as 64 B); WRITE_SIZE as is"}
    if k in sq and sq[k].get("SQ_BUSY_CU_CYCLES"):
        busy = sum(sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(1.0, 4 * sum(sq[k]["SQ_BUSY_CU_CYCLES"]))
        row["mfma_busy_frac"] = round(busy, 4)
        d = sqdur.get(k)
        if d:
            row["eff_clock_ghz"] = round(sum(sq[k]["GRBM_GUI_ACTIVE"]) / 8 / sum(d), 3)
    if k in lds and lds[k].get("SQ_BUSY_CU_CYCLES"):
        act = sum(lds[k]["SQ_LDS_IDX_ACTIVE"])
        row["lds_conflict_frac"] = round(sum(lds[k]["SQ_LDS_BANK_CONFLICT"]) / act, 4) if act else 0.0
        # SQ_WAVE_CYCLES counts in units of 4 cycles: wave-quad-cycles / CU-busy cycles = resident waves per SIMD
        row["waves_per_simd"] = round(sum(lds[k]["SQ_WAVE_CYCLES"]) / sum(lds[k]["SQ_BUSY_CU_CYCLES"]), 2)
    rows.append(row)
cols = ["kernel", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_bytes_per_launch", "mfma_busy_frac", "eff_clock_ghz",
        "lds_conflict_frac", "waves_per_simd"]
with open(dst / f"{tag}_pmc_summary.csv", "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=cols)
    w.writeheader()
    for r in rows:
        w.writerow(r)
# merge: kernels measured by this run replace their entries, other kernels (e.g. the other storage type's run) are kept
tj = dst / "pmc_traffic.json"
merged = json.loads(tj.read_text()) if tj.exists() else {}
merged.update(traffic)
json.dump(merged, open(tj, "w"), indent=1)
print(open(dst / f"{tag}_bench.json").read()[-1200:])
for r in sorted(rows, key=lambda r: -r.get("hbm_bytes_per_launch", 0))[:8]:
    print(r)
