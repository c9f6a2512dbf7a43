"""Copy the rocprofv3 summaries of the last GPU run from gpurun_out/ (scratch) into profiles/ (tracked)."""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(ROOT, "gpurun_out"); p = os.path.join(ROOT, "profiles"); rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
shutil.copy(os.path.join(g, "prof_" + rnd, rnd + "_kernel_stats.csv"), os.path.join(p, rnd + "_kernel_stats.csv"))
shutil.copy(os.path.join(g, "bench_" + rnd + ".json"), os.path.join(p, rnd + "_bench.json"))
# registers, spilled registers and scratch bytes of every kernel of the library the session ran, from its code object (no GPU needed: regenerated here so that the
# file always belongs to the tree the profile was taken on)
import subprocess
with open(os.path.join(p, rnd + "_kernel_metadata.txt"), "w") as fh:
    subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_metadata.sh")], stdout=fh, check=False, cwd=ROOT)
out = {}
for f in ("pmc_fetch/fetch_counter_collection.csv", "pmc_write/write_counter_collection.csv", "pmc_sq/sq_counter_collection.csv", "pmc_mem/mem_counter_collection.csv", "pmc_ic/ic_counter_collection.csv"):
    if not os.path.exists(os.path.join(g, f)): continue
    # a bench step is one solve = one launch, or two (k_solve_lean_cl: the first steps of every query, k_solve_lean: the unsolved ones to the
    # end): counters are summed per step = over the dispatches of both kernels / the dispatches of the kernel that ends a step
    rows = list(csv.DictReader(open(os.path.join(g, f)))); agg = collections.defaultdict(float); per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    names = collections.Counter()
    for r in rows:
        if "k_solve" in r["Kernel_Name"]:
            kn = r["Kernel_Name"].split("(")[0]
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            per_kernel[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
            names[(kn, r["Counter_Name"])] += 1
            out.setdefault("dispatch", {})[kn] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        closing = "k_solve_lean" if ("k_solve_lean", k) in names else max((kn for kn, c in names if c == k), key=lambda kn: names[(kn, k)])
        steps = names[(closing, k)]
        out[k] = {"steps": steps, "mean_per_launch": v / steps, "per_kernel_mean": {kn: sum(c[k]) / len(c[k]) for kn, c in per_kernel.items() if k in c}}
fk, wk = out["FETCH_SIZE"]["mean_per_launch"], out["WRITE_SIZE"]["mean_per_launch"]
kernels_seen = sorted({kn for kn, _ in names})
out.pop("dispatch", None)  # (rocprofv3's per-dispatch VGPR / LDS columns are allocation granules of the launch, not the code object's figures: those are in profiles/rNN_kernel_metadata.txt)
summary = {"command": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python bench.py --timed-only --steps 5 --warmup 1 --in-flight 1 "
                      "(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* ...; one solve at a time, so a dispatch's counters hold no neighbour's traffic)",
           "kernel": " + ".join(kernels_seen) + ": the launch(es) of one solve, counters summed per solve (`mean_per_launch`)",
           "kernel_metadata": "profiles/%s_kernel_metadata.txt (llvm-readelf --notes of the code object: registers, spilled registers, scratch bytes)" % rnd,
           "counters": out,
           "hbm_bytes_per_launch": {"fetch_bytes_raw": fk * 1024, "fetch_bytes_corrected_x2_gfx950": 2 * fk * 1024, "write_bytes": wk * 1024,
                                    "total_corrected": 2 * fk * 1024 + wk * 1024,
                                    "note": "FETCH_SIZE/WRITE_SIZE are KiB; gfx950 FETCH_SIZE tallies 128-B requests as 64 B (MI355X_MICROARCH.md, HBM) -> doubled; "
                                            "WRITE_SIZE uncalibrated.  The query data of a 4096-query solve are 1.4 MB in and out; since round 4 no solve kernel of the "
                                            "bench workload uses scratch memory (k_solve_lean_cl64w4 / k_solve_lean_cl4: 0 spilled registers), so what is measured beyond "
                                            "the query data is the problem block and instruction / constant fetches."}}
json.dump(summary, open(os.path.join(p, rnd + "_pmc_k_solve.json"), "w"), indent=1)
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_sources_hash: the figure is valid for the kernel sources it was measured on, and bench.py checks that)
json.dump({"k_solve_hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024, "source": "profiles/%s_pmc_k_solve.json" % rnd, "kernel_sources_sha256": bench.kernel_sources_hash()},
          open(os.path.join(p, "traffic.json"), "w"), indent=1)
# bench.py reads `roofline.traffic` from profiles/traffic.json as it was BEFORE this session's PMC passes ran; the copy kept under
# profiles/ carries the figure of its own session (same library, same box), with the value the line was printed with beside it
b = json.load(open(os.path.join(p, rnd + "_bench.json")))
if isinstance(b.get("roofline"), dict):
    b["roofline"]["traffic_as_printed"] = b["roofline"].get("traffic")
    b["roofline"]["traffic"] = 2 * fk * 1024 + wk * 1024
    b["roofline"]["traffic_provenance_as_printed"] = b["roofline"].get("traffic_provenance")
    b["roofline"]["traffic_provenance"] = "PMC passes of this same session on the same library (profiles/%s_pmc_k_solve.json), filled in by tools/collect_profiles.py" % rnd
    json.dump(b, open(os.path.join(p, rnd + "_bench.json"), "w"))
for k in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES"):
    print(k, "%.4g" % out[k]["mean_per_launch"])
print("hbm bytes/launch %.4g" % summary["hbm_bytes_per_launch"]["total_corrected"], kernels_seen)

# k_stream_fitness: measured HBM bytes per launch (same unit corrections) next to the algorithmic bytes of the bench line
sf = {}
for f, key in (("pmc_sfetch/sfetch_counter_collection.csv", "FETCH_SIZE"), ("pmc_swrite/swrite_counter_collection.csv", "WRITE_SIZE")):
    path = os.path.join(g, f)
    if os.path.exists(path):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "k_stream_fitness" in r["Kernel_Name"] and r["Counter_Name"] == key]
        if v:
            sf[key] = {"launches": len(v), "mean_per_launch_KiB": sum(v) / len(v)}
if len(sf) == 2:
    b = json.load(open(os.path.join(p, rnd + "_bench.json")))
    alg = b.get("streamed_fitness", {}).get("algorithmic_bytes_per_launch")
    meas = 2 * sf["FETCH_SIZE"]["mean_per_launch_KiB"] * 1024 + sf["WRITE_SIZE"]["mean_per_launch_KiB"] * 1024
    json.dump({"kernel": "k_stream_fitness(StreamArgs)", "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python bench.py --no-cpu-baseline --steps 2 --warmup 1",
               "counters": sf, "measured_hbm_bytes_per_launch": meas, "algorithmic_bytes_per_launch": alg, "measured_over_algorithmic": meas / alg if alg else None,
               "note": "genes [unit][D][pop] f64 streamed once (FETCH_SIZE x 2: gfx950 tallies 128-B requests as 64 B), fitness written once; "
                       "a ratio near 1 = every gene byte crosses the HBM interface once, in full 512-byte wavefront segments"},
              open(os.path.join(p, rnd + "_pmc_k_stream_fitness.json"), "w"), indent=1)
    print("k_stream_fitness measured / algorithmic bytes: %.3f" % (meas / alg if alg else float("nan")))

# The statistics file averages over EVERY launch of the profiled command (one per stream to open the streams, the warm-up, the timed steps);
# the bench line's roofline.kernel_ms is the event-bracketed mean of the timed steps only.  The same mean from the kernel trace, next to the
# figure the profiled process itself printed:
tr = os.path.join(g, "prof_" + rnd, rnd + "_kernel_trace.csv")
pl = os.path.join(g, "prof_bench.log")
if os.path.exists(tr) and os.path.exists(pl):
    line = json.loads([l for l in open(pl) if l.startswith("{")][-1])
    kern, k_timed = line["roofline"].get("kernel_trace_name", line["roofline"]["kernel"]), line["steps"]
    rows = sorted((r for r in csv.DictReader(open(tr)) if r["Kernel_Name"].startswith(kern + "(")), key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]
    rec = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --timed-only", "kernel": kern, "launches": len(dur),
           "mean_ms_all_launches (= %s_kernel_stats.csv AverageNs)" % rnd: sum(dur) / len(dur),
           "timed_steps": k_timed, "mean_ms_of_the_timed_launches (the last %d of the trace)" % k_timed: sum(dur[-k_timed:]) / k_timed,
           "roofline.kernel_ms printed by the profiled process (HIP events around the same launches)": line["roofline"]["kernel_ms"],
           "value printed by the profiled process": line["value"], "batches_in_flight": line["config"]["batches_in_flight"],
           "note": "launches overlap (batches_in_flight of them share the chip), so a launch lasts batches_in_flight x ms_per_step; the launches "
                   "before the timed region start on an emptier chip and are shorter.  Round 5: a solve is this launch PLUS the launch of k_solve_lean_cl4h that takes its "
                   "stragglers over when the chip runs empty -- the events bracket both, the trace figure is this kernel alone"}
    # the stragglers' launch of every solve (round 5): on the same stream behind the dense kernel's, so a solve lasts the sum of the two
    rows2 = sorted((r for r in csv.DictReader(open(tr)) if r["Kernel_Name"].startswith("k_solve_lean_cl4h(")), key=lambda r: int(r["Start_Timestamp"]))
    if rows2:
        dur2 = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows2]
        rec["k_solve_lean_cl4h: launches"] = len(dur2)
        rec["k_solve_lean_cl4h: mean_ms_of_the_timed_launches"] = sum(dur2[-k_timed:]) / min(k_timed, len(dur2))
        rec["sum of the two kernels' timed means (what the events around a solve bracket)"] = rec["mean_ms_of_the_timed_launches (the last %d of the trace)" % k_timed] + rec["k_solve_lean_cl4h: mean_ms_of_the_timed_launches"]
    json.dump(rec, open(os.path.join(p, rnd + "_kernel_stats_timed_region.json"), "w"), indent=1)
    print("kernel trace: all %.2f ms, timed %.2f ms, events %.2f ms" % (sum(dur) / len(dur), sum(dur[-k_timed:]) / k_timed, line["roofline"]["kernel_ms"]))
