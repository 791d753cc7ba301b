"""Extracts the published result rows of the hot-path models from the reference's leaderboard data
(`benchmark/<dataset>/{MF-BPR,LightGCN,SASRec}.json`: per seed, train LOSS and the valid / test / best metric dicts) into
tests/golden/benchmark_rows.json.  These rows are DATA the reference's own tooling consumes (recboard/scripts/build-data.mjs:49-66);
they are the only fixtures in /root/reference for the freerec-side metric definitions: with one held-out target
per user, HR@1 == NDCG@1, NDCG@K <= HR@K, NDCG@K >= HR@K / log2(K + 1), and both are non-decreasing in K.
    python tests/golden/make_benchmark_fixture.py [/root/reference]"""
import json
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = {}
for ds in sorted(os.listdir(os.path.join(ref, "benchmark"))):
    for model in ("MF-BPR", "LightGCN", "SASRec"):
        p = os.path.join(ref, "benchmark", ds, model + ".json")
        if not os.path.exists(p):
            continue
        rows = []
        for entry in json.load(open(p)):
            for run in entry["runs"]:
                rows.append({"seed": run["params"].get("seed"), **{k: run["metrics"][k] for k in ("train", "valid", "test", "best") if k in run["metrics"]}})
        out[f"{ds}/{model}"] = rows
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "benchmark_rows.json")
json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
print(len(out), "model/dataset files,", sum(len(v) for v in out.values()), "runs ->", dst)
