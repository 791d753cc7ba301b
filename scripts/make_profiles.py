"""gpurun_out/<dir> (rocprofv3 kernel trace of bench.py + FETCH_SIZE / WRITE_SIZE passes of scripts/x2_prof.py + a plain bench line)
-> profiles/<tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_traffic.json.   The input directory is what
scripts/prof_round.sh leaves (trace/, fetch/, write/, bench.json).
usage: python scripts/make_profiles.py gpurun_out/r2 r2 [output directory, default profiles/]"""
import csv, glob, json, os, shutil, sqlite3, subprocess, sys
src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles")
os.makedirs(P, exist_ok=True)
db = sqlite3.connect(glob.glob(src + "/trace/**/*.db", recursive=True)[0])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(os.path.join(P, tag + "_bench_kernel_stats.csv"), "w", newline="") as fo:
    w = csv.writer(fo, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, a, mn, mx in rows:
        w.writerow([n, c, t, round(a, 3), round(100 * t / tot, 2), mn, mx])
shutil.copy(os.path.join(src, "bench.json"), os.path.join(P, tag + "_bench.json"))
k = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "scripts", "pmc_traffic.py"), src + "/fetch", src + "/write"]))
k = {n: v for n, v in k.items() if not n.startswith(("at::", "void at::", "Cijk", "__amd"))}
main = [n for n in k if n.startswith("score_kernel_reg<64, 28")][0]
exact = ([n for n in k if n.startswith("score_kernel_reg<64, 50")] or [None])[0]   # (round 6: the fallback is score_rescan_k for catalogs up to 131 072 items)
front = [n for n in k if n.startswith("score_front_k<64>")]      # round 5: query split + item split + starting thresholds in one launch
rescan = [n for n in k if n.startswith("score_rescan_k<64>")]
if front and rescan:
    total = sum(k[n]["hbm_bytes_per_launch"] for n in (front[0], main, "score_topk_merge_x<64>", rescan[0]))
elif front and exact is None:     # (round 6, end: a flagged user is re-scored inside the merge launch: three launches a call)
    total = sum(k[n]["hbm_bytes_per_launch"] for n in (front[0], main, "score_topk_merge_x<64>"))
elif front:
    total = sum(k[n]["hbm_bytes_per_launch"] for n in (front[0], main, "score_topk_merge_x<64>", exact, "score_topk_merge"))
else:
    total = 2 * k["score_split_k<64>"]["hbm_bytes_per_launch"] + sum(k[n]["hbm_bytes_per_launch"] for n in ("score_bound_k<64>", main, "score_topk_merge_x<64>", exact, "score_topk_merge"))
out = {"collected": "two rocprofv3 passes (the TCC block cannot hold both counters): rocprofv3 --pmc FETCH_SIZE -- python3 scripts/pmc_step.py ; same with --pmc WRITE_SIZE  "
                    "(6 fused SASRec steps at the bench shapes launched eagerly, 3 re_score_topk calls of 22 363 users x 12 101 items, D = 64, K = 50, iid scores, "
                    "3 gathers of 4 Mi rows from a 16 Mi x 64 table)",
       "units": "KB per launch (average over launches)",
       "correction": "gfx950: FETCH_SIZE tallies the 128-B requests of 16-B-per-lane reads at 64 B (MI355X_MICROARCH.md, HBM): hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE",
       "kernels": k,
       "re_score_topk_call": {
           "launches": "score_front_k (query split + item-table split + starting thresholds + zeroing of the call's words, one launch), score_kernel_reg<64,28,28,split>, "
                       "score_topk_merge_x (a user whose certificate fails is re-scored against the whole catalog by the merge wave that finds it so: "
                       "sx_rescan_user; no fallback launch)",
           "hbm_bytes_per_call": total,
           "algorithmic_lower_bound_bytes": 4 * 64 * (22363 + 12101) + 12 * 22363 * 50,
           "note": "above the lower bound: the partial lists (one 56-entry list per user and segment: 41 MB written by the main kernel, read by the merge), "
                   "the split item table streamed once per user block (3 MB x 175 blocks; what misses the XCDs' L2s is served by the Infinity Cache and "
                   "counted here) and the candidates' rows re-read by the merge.  At 0.8 TB/s over the call none of it is what bounds the kernels "
                   "(vector instruction stream and stage barriers, DESIGN.md section 5a)."}}
# config 5: HBM bytes of one step = the engine's kernels of scripts/pmc_c5.py (eager steps over distinct batches; the table's torch init kernels and
# torch's own small launches are left out), summed over their launches, per step
if os.path.isdir(src + "/c5fetch") and glob.glob(src + "/c5fetch/**/*.db", recursive=True):
    k5 = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "scripts", "pmc_traffic.py"), src + "/c5fetch", src + "/c5write"]))
    k5 = {n: v for n, v in k5.items() if n and not n.startswith(("at::", "void at::", "Cijk", "__amd", "at_cuda", "void at_cuda", "void (anonymous", "(anonymous"))}
    nsteps = int(open(src + "/c5fetch.log").read().split("steps")[-1].split()[0])
    tot5 = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in k5.values()) / nsteps
    out["config5_step"] = {"hbm_bytes_per_step": int(tot5), "steps": nsteps,
                           "kernels": {n: {"launches_per_step": round(v["launches"] / nsteps, 2), "hbm_bytes_per_launch": v["hbm_bytes_per_launch"]} for n, v in k5.items()},
                           "algorithmic_bytes_per_step": "about 3 x 3 600 looked-up rows x (8 + 512 read) + ~10 000 distinct rows x 3 tables x (512 read + 512 written) by the "
                                                         "row-sparse Adam + the d = 128 encoder's tape: ~45 MB",
                           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_c5.py (eager steps, each on a batch never seen before: the "
                                     "uniform negatives' rows come from HBM); 2*FETCH+WRITE"}
json.dump(out, open(os.path.join(P, tag + "_pmc_traffic.json"), "w"), indent=1)
# SQ counters of the step's kernels (two passes of 7)
if os.path.isdir(src + "/pmc1") and glob.glob(src + "/pmc1/**/*.db", recursive=True):
    import collections
    sq = {}
    for d in ("pmc1", "pmc2"):
        acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
        for f in glob.glob(f"{src}/{d}/**/*.db", recursive=True):
            for kn, c, v in sqlite3.connect(f).execute("select kernel_name, counter_name, value from counters_collection"):
                short = kn.split("(")[0].replace("void ", "").strip()
                acc[(short, c)] += float(v); cnt[(short, c)] += 1
        for (kn, c), v in acc.items():
            if kn.startswith(("enc_", "tl4::", "tl8::", "scatter_owner", "sasrec_batch_prep", "sasrec_step_stage", "adam_vec4")):
                sq.setdefault(kn, {})[c] = int(v / cnt[(kn, c)])
    sq = {"what": "rocprofv3 --pmc on scripts/pmc_step.py (SASRec/Beauty B=512 fused step, eager launches; two passes of 7 SQ counters: scripts/prof_round.sh), "
                  "per launch averages; SQ_WAVE_CYCLES / SQ_WAIT_* are quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES are cycles summed over the 1 024 SIMDs",
          **sq}
    json.dump(sq, open(os.path.join(P, tag + "_step_pmc.json"), "w"), indent=1)
for n, c, t, a, mn, mx in rows[:12]:
    print(f"{t/1e3:10.1f} us  calls {c:5d}  avg {a/1e3:8.1f} us  {n[:70]}")
print("hbm bytes per score call:", total)
