"""gpurun_out/pmc_mfma/ (tools/pmc_mfma.sh, GPU box) -> profiles/gmm_mfma_util.json, profiles/cosine_mfma_util.json: per-dispatch means of the
SQ counters of the scoring kernels' FULL-SIZE launches (the largest grid of each kernel name: the re-scoring launches of the split-precision
paths share names with the main ones), the derived MFMA-pipe / VALU / LDS busy fractions, and the sha256 of the kernel source they
were taken on — bench.py quotes `mfma_busy` only when that hash equals the tree's.
    python tools/store_mfma_pmc.py [round-tag]"""
import collections, csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
S = os.path.join(ROOT, "gpurun_out", "pmc_mfma")
git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()


def collect(prefix, names):
    out = {n: collections.OrderedDict() for n in names}
    i = 1
    while os.path.exists(os.path.join(S, "%s_%d.csv" % (prefix, i))):
        rows = list(csv.DictReader(open(os.path.join(S, "%s_%d.csv" % (prefix, i)))))
        for n in names:
            mine = [r for r in rows if r["Kernel_Name"].startswith(n) or ("ssp::" + n) in r["Kernel_Name"] or (" " + n) in r["Kernel_Name"]]
            if not mine:
                continue
            gmax = max(int(r["Grid_Size"]) for r in mine)
            per = collections.defaultdict(list)
            for r in mine:
                if int(r["Grid_Size"]) == gmax:
                    per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in per.items():
                out[n][k] = {"mean": sum(v) / len(v), "launches": len(v), "grid": gmax}
            out[n]["_kernel_name"] = mine[0]["Kernel_Name"][:160]
        i += 1
    return out


def derive(c):
    g = lambda k: c[k]["mean"] if k in c else None
    if g("GRBM_GUI_ACTIVE") is None or g("SQ_VALU_MFMA_BUSY_CYCLES") is None:
        return None
    cyc = g("GRBM_GUI_ACTIVE") / 8.0          # summed over the 8 XCDs
    simds = 1024.0
    d = {"kernel_cycles": cyc, "mfma_busy_fraction": g("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * simds)}
    if g("SQ_ACTIVE_INST_VALU") is not None:
        d["valu_busy_fraction"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (cyc * simds)
    if g("SQ_ACTIVE_INST_LDS") is not None:
        d["lds_issue_fraction"] = 4.0 * g("SQ_ACTIVE_INST_LDS") / (cyc * simds)
    if g("SQ_WAVE_CYCLES") is not None:
        d["waves_per_simd"] = 4.0 * g("SQ_WAVE_CYCLES") / (cyc * simds)
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    return d


def store(prefix, names, fname, workload, stage, sha_names):
    c = collect(prefix, names)
    if not any(len(v) > 1 for v in c.values()):
        print(fname, ": no counters found")
        return
    doc = {"round": R, "git": git, "kernel_source_sha256": bench.kernel_source_sha256(sha_names), "kernel_source_files": list(sha_names),
           "workload": workload,
           "command": "tools/pmc_mfma.sh  (rocprofv3 --pmc <group> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --stages %s --no-cpu-baseline --no-env; "
                      "one pass per counter group; per-dispatch means over the launches with the kernel's largest grid)" % stage,
           "kernels": {}, "notes": "SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs; SQ_ACTIVE_INST_* and "
                                   "SQ_WAVE_CYCLES count in units of 4 cycles."}
    for n in names:
        if len(c[n]) <= 1:
            continue
        kn = c[n].pop("_kernel_name", n)
        doc["kernels"][n] = {"kernel_name": kn, "grid": next(iter(c[n].values()))["grid"], "launches": next(iter(c[n].values()))["launches"],
                             "raw_per_launch": {k: v["mean"] for k, v in c[n].items()}, "derived": derive(c[n])}
    json.dump(doc, open(os.path.join(ROOT, "profiles", fname), "w"), indent=1)
    print(fname, {n: (doc["kernels"][n]["derived"] or {}).get("mfma_busy_fraction") for n in doc["kernels"]})


store("gmm", ["gmm_loglik_kernel", "gmm_loglik_bf16x3_kernel"], "gmm_mfma_util.json",
      "configs[2] at FULL size: 100000 utterances x 298 frames (2.98e7) x 51 models x 64 mixtures x 39 dims", "mfcc,gmm", bench.GMM_SOURCES)
store("cos", ["cosine_reg_kernel", "cosine_bf16x3_kernel"], "cosine_mfma_util.json",
      "configs[4] at FULL size: 1e6 embeddings x 1251 centroids x 256 dims", "mfcc,cosine --utts 2000 (the cosine stage's own size does not depend on --utts)", bench.COSINE_SOURCES)
