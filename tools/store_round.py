"""gpurun_out/<round>/ (made by tools/profile_round.sh on the GPU box) -> profiles/<round>_*: the bench line, the rocprofv3 kernel
summary of the same command, one per-launch listing per stage (so that every quoted roofline fraction can be recomputed from
profiles/ alone) and the PMC readings, each tied to the sha256 of the kernel source it was taken on.
    python tools/store_round.py r03"""
import collections, csv, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha256)
R = sys.argv[1]
S = os.path.join(ROOT, "gpurun_out", R)
P = os.path.join(ROOT, "profiles")
sha = bench.kernel_source_sha256()
git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()


def have(n):
    return os.path.exists(os.path.join(S, n))


for n in ("bench_line.json", "bench_detail.json"):   # the compact line the driver parses + the full result behind it
    if have(n):
        shutil.copy(os.path.join(S, n), os.path.join(P, "%s_%s" % (R, n)))
if have("kernel_stats.csv"):
    shutil.copy(os.path.join(S, "kernel_stats.csv"), os.path.join(P, "%s_kernel_stats.csv" % R))
    md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stats_md.py"), os.path.join(S, "kernel_stats.csv"),
                         "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-env   (MI355X, %s, tree %s)" % (R, git)],
                        capture_output=True, text=True).stdout
    md = "\n".join(l for l in md.splitlines() if "at::native" not in l) + "\n"
    if have("kernel_trace.csv"):   # the same run, per launch: full-size launches of the kernels the rooflines are quoted on
        by = collections.OrderedDict()
        for r in csv.DictReader(open(os.path.join(S, "kernel_trace.csv"))):
            n = r["Kernel_Name"]
            if any(k in n for k in ("mfcc_stream512_kernel<13, 2, 1, 3, 6, 2, 3, 0, 0>", "gmm_loglik_kernel<10, 2, true>", "cosine_reg_kernel<32, false>")):
                by.setdefault(n, []).append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size_X"])))
        md += "\nPer workload, from the kernel trace of the SAME run (the host-fed stages launch these kernels once per 64-MiB slice and the\n" \
              "split-precision scorers once per re-scoring pass / pilot: the per-kernel averages above mix those in):\n\n| kernel | launches of one workload (grouped by duration) | avg ms | min ms | max ms |\n|---|---|---|---|---|\n"
        for n, v in by.items():   # (the stream kernel's grid is persistent and the list-driven launches are sized for the worst case: duration
            # tells.  One kernel instance serves several workloads — configs[2] and the configs[3] sample, the 100 000-utterance pass and the
            # stages that reuse its plan — so launches are grouped by duration: a new group where the next one is under 0.7 x the group's longest)
            ds = sorted((x[0] for x in v), reverse=True)
            groups = []
            for x in ds:
                if groups and x >= 0.7 * groups[-1][0]:
                    groups[-1].append(x)
                else:
                    groups.append([x])
            for grp in groups:
                if len(grp) >= 2 and sum(grp) / len(grp) >= 0.5:
                    md += "| `%s` | %d of %d | %.3f | %.3f | %.3f |\n" % (n.split("(")[0], len(grp), len(v), sum(grp) / len(grp), min(grp), max(grp))
    md += "\n(torch's own elementwise / reduction kernels that build the synthetic inputs are left out of this table; they are in the csv.\n" \
          "This file mixes every stage's launches of a kernel in one row; the per-stage files %s_stage_*.md list the launches one by one.)\n" % R
    open(os.path.join(P, "%s_kernel_stats.md" % R), "w").write(md)

# ---- per-stage listings
DOM = {"mfcc": ("mfcc_stream", "roofline"), "ref26": ("mfcc_stream", "mfcc_ref26_cmvn"), "inrepo": ("mfcc_stream", "mfcc_inrepo"),
       "librosa": ("mfcc_stream2048_kernel", "mfcc_librosa"), "gmm": ("gmm_loglik", "gmm"), "cosine": ("cosine_", "cosine"), "plp": ("mfcc_stream", "plp")}
for st, (kern, key) in DOM.items():
    tr = "stage_%s_kernel_trace.csv" % st
    if not have(tr):
        continue
    rows = [r for r in csv.DictReader(open(os.path.join(S, tr))) if kern in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    line = json.load(open(os.path.join(S, "stage_%s_bench_detail.json" % st)))   # the full result of the stage's run
    shutil.copy(os.path.join(S, "stage_%s_bench_detail.json" % st), os.path.join(P, "%s_stage_%s_bench_detail.json" % (R, st)))
    shutil.copy(os.path.join(S, "stage_%s_kernel_stats.csv" % st), os.path.join(P, "%s_stage_%s_kernel_stats.csv" % (R, st)))
    out = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --stages %s --no-cpu-baseline   (MI355X, %s, tree %s)" % (st, R, git), "",
           "Every launch of the stage's dominant kernel(s), in launch order (ms, rocprofv3 kernel trace).  The set-up launch and the warm-up launches come first;",
           "the bench's own figure (hipEvents, same run) is quoted below each listing.", ""]
    for name, d in by.items():
        out.append("`%s` — %d launches" % (name, len(d)))
        out.append("")
        out.append("| " + " | ".join("%.3f" % x for x in d) + " |")
        out.append("")
        out.append("mean of all %.3f ms, min %.3f, max %.3f; mean of the last %d: %.3f ms" % (sum(d) / len(d), min(d), max(d), min(5, len(d)), sum(d[-5:]) / min(5, len(d))))
        big = [x for x in d if x >= 0.5 * max(d)]   # (re-scoring passes and precision-auto pilots launch the same kernels on a few rows)
        if len(big) != len(d):
            out.append("full-size launches only (>= half of the longest; the short ones are re-scoring passes / pilots on listed rows): %d launches, mean %.3f ms, mean of the last %d: %.3f ms" % (
                len(big), sum(big) / len(big), min(5, len(big)), sum(big[-5:]) / min(5, len(big))))
        out.append("")
    node = line if key == "roofline" else line.get(key, {})
    def rooflines(n, pre=""):
        if isinstance(n, dict):
            for k, v in n.items():
                if k in ("roofline", "front_roofline") and isinstance(v, dict):
                    yield pre + k, v
                elif isinstance(v, dict):
                    yield from rooflines(v, pre + k + ".")
    if key == "roofline":
        rl = [("roofline", line["roofline"])]
    else:
        rl = list(rooflines(node, key + "."))
    for nm, v in rl:
        out.append("bench line `%s`: kernel_ms %.3f -> achieved %.4g %s of peak %.4g = frac %.4f (%s)" % (
            nm, v.get("kernel_ms", float("nan")), v["achieved"], v["unit"], v["peak"], v["frac"], v.get("kernel", "")))
    if st == "gmm" and "gmm_bf16x3" in line:
        v = line["gmm_bf16x3"]["roofline"]
        out.append("bench line `gmm_bf16x3.roofline`: kernel_ms %.3f -> achieved %.4g %s of peak %.4g = frac %.4f (incl. fp32 re-scoring of %d close calls)" % (
            v["kernel_ms"], v["achieved"], v["unit"], v["peak"], v["frac"], line["gmm_bf16x3"]["utterances_rescored_in_fp32"]))
    if st == "gmm" and "gmm_bf16x3_proven_band" in line:
        v = line["gmm_bf16x3_proven_band"]
        out.append("bench line `gmm_bf16x3_proven_band`: kernel_ms %.3f (precision 1: the derived bound; %d utterances listed, candidates re-scored in fp32; arg-max mismatches vs fp32: %d)" % (
            v["kernel_ms"], v["utterances_rescored_in_fp32"], v["argmax_mismatches_vs_fp32_path"]))
    if st == "cosine" and "cosine_bf16x3" in line:
        v = line["cosine_bf16x3"]
        out.append("bench line `cosine_bf16x3.roofline`: kernel_ms %.3f -> achieved %.4g TFLOP/s algorithmic of peak %.4g = frac %.4f (executed: %.3f); arg-min equal to the fp32 path: %s; rows re-scored in fp32: %d" % (
            v["roofline"]["kernel_ms"], v["roofline"]["achieved"], v["roofline"]["peak"], v["roofline"]["frac"], v["roofline"]["frac_executed"],
            v["argmin_equals_fp32_path"], v["rows_rescored_fp32"]))
    open(os.path.join(P, "%s_stage_%s.md" % (R, st)), "w").write("\n".join(out) + "\n")
    print("stage", st, {k: round(sum(v[-5:]) / min(5, len(v)), 3) for k, v in by.items()})


# ---- PMC
def pmc(prefix, kern, also=""):
    agg = collections.OrderedDict()
    i = 1
    while have("pmc_%s_%d.csv" % (prefix, i)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(S, "pmc_%s_%d.csv" % (prefix, i)))):
            if kern in r["Kernel_Name"] and also in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in per.items():
            agg[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
        i += 1
    return agg


a = pmc("512", "mfcc_stream512", ", 0>(")  # (the first kernel of the launch: <..., WALK = 0>; the second exits on one load)
if a:
    line = json.load(open(os.path.join(S, "stage_mfcc_bench_detail.json"))) if have("stage_mfcc_bench_detail.json") else None
    algo = 23848800000
    g = lambda k: a[k]["mean_per_launch"] if k in a else None
    rd = 2.0 * g("FETCH_SIZE") * 1024 if g("FETCH_SIZE") is not None else None
    wr = g("WRITE_SIZE") * 1024 if g("WRITE_SIZE") is not None else None
    if rd is not None and wr is not None:
        doc = {"kernel": "ssp::mfcc_stream512_kernel<13,2,1,3,6,2,3,0,0> (wave-stream, 3 workgroups per CU)", "round": R, "git": git, "kernel_source_sha256": sha,
               "workload": "configs[1]: 100000 x 3 s @16 kHz, 39-d",
               "command": "tools/profile_round.sh %s pmc512  (rocprofv3 --pmc <counter group> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --stages mfcc --no-cpu-baseline; one pass per counter group; mean over the kernel's launches of the pass)" % R,
               "raw": {k: v["mean_per_launch"] for k, v in a.items() if k.startswith(("FETCH", "WRITE", "TCC"))},
               "corrections": "gfx950: FETCH_SIZE counts 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM): read bytes = 2 * FETCH_SIZE * 1024 = TCC_EA0_RDREQ * 128 B; WRITE_SIZE * 1024 is exact (= WRREQ * 64 B)",
               "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
               "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": (rd + wr) / algo}
        json.dump(doc, open(os.path.join(P, "mfcc_hbm_traffic.json"), "w"), indent=1)
        print("hbm traffic / algorithmic = %.4f" % doc["traffic_over_algorithmic"])
    if "SQ_WAVE_CYCLES" in a:
        quads = 100000 * 75  # 298 frames -> 75 quads of 4 frames per utterance
        wc = g("SQ_WAVE_CYCLES")
        census = None
        cf = os.path.join(S, "isa_census.json")
        if os.path.exists(cf):
            census = json.load(open(cf))
        doc = {"kernel": "ssp::mfcc_stream512_kernel<13,2,1,3,6,2,3,0,0>", "round": R, "git": git, "kernel_source_sha256": sha,
               "workload": "configs[1] at FULL size: 100000 x 3 s @16 kHz per pass (7.5e6 quads of 4 frames), 39-d",
               "command": "tools/profile_round.sh %s pmc512" % R,
               "raw_per_launch": {k: v["mean_per_launch"] for k, v in a.items() if k.startswith(("SQ_", "GRBM"))},
               "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (4 shader cycles) summed over waves; GRBM_GUI_ACTIVE is the sum over the 8 XCDs",
               "derived": {"valu_instructions_per_quad": g("SQ_INSTS_VALU") / quads, "lds_instructions_per_quad": g("SQ_INSTS_LDS") / quads,
                           "salu_instructions_per_quad": g("SQ_INSTS_SALU") / quads,
                           "share_of_wave_time": {"issuing (SQ_ACTIVE_INST_ANY)": g("SQ_ACTIVE_INST_ANY") / wc if g("SQ_ACTIVE_INST_ANY") else None,
                                                  "waiting at s_waitcnt (SQ_WAIT_ANY)": g("SQ_WAIT_ANY") / wc if g("SQ_WAIT_ANY") else None,
                                                  "stalled at issue (SQ_WAIT_INST_ANY)": g("SQ_WAIT_INST_ANY") / wc if g("SQ_WAIT_INST_ANY") else None},
                           "valu_active_cycles_per_quad": 4.0 * g("SQ_ACTIVE_INST_VALU") / quads if g("SQ_ACTIVE_INST_VALU") else None,
                           "wave_cycles_per_quad": 4.0 * wc / quads,
                           "valu_active_share_of_simd_time": 3.0 * g("SQ_ACTIVE_INST_VALU") / wc if g("SQ_ACTIVE_INST_VALU") else None,
                           "lds_conflict_share_of_lds_cycles": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None,
                           "effective_clock_ghz": (g("GRBM_GUI_ACTIVE") / 8.0 / (line["roofline"]["kernel_ms"] * 1e-3) / 1e9) if (line and g("GRBM_GUI_ACTIVE")) else None}}
        if census:
            doc["instruction_census"] = census
            doc["valu_issue_cycles_per_quad"] = census["issue_cycles_per_quad"]
        else:
            doc["valu_issue_cycles_per_quad"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / quads
            doc["valu_issue_cycles_source"] = "SQ_ACTIVE_INST_VALU (measured VALU-active cycles per quad), not an instruction census"
        json.dump(doc, open(os.path.join(P, "mfcc_valu_lds_pmc.json"), "w"), indent=1)
        print("512: VALU/quad %.0f, LDS/quad %.1f, VALU share of SIMD time %.3f" % (doc["derived"]["valu_instructions_per_quad"], doc["derived"]["lds_instructions_per_quad"], doc["derived"]["valu_active_share_of_simd_time"] or -1))
b = pmc("2k", "mfcc_stream2048")
if b and "SQ_WAVE_CYCLES" in b:
    g = lambda k: b[k]["mean_per_launch"] if k in b else None
    frames = 200000 * 47
    wc = g("SQ_WAVE_CYCLES")
    doc = {"kernel": "ssp::mfcc_stream2048_kernel (librosa dialect of MFCC_DTW.MFCC_lib: 2048 / 512, 128 mel, top_db, 13-d), 12-wave workgroups = 3 waves per SIMD",
           "round": R, "git": git, "workload": "bench stage librosa: 200000 x 3 s @ 8 kHz (9.4e6 frames)", "command": "tools/profile_round.sh %s pmc2k" % R,
           "raw_per_launch": {k: v["mean_per_launch"] for k, v in b.items()},
           "derived": {"valu_instructions_per_frame": g("SQ_INSTS_VALU") / frames, "lds_instructions_per_frame": g("SQ_INSTS_LDS") / frames,
                       "wave_cycles_per_frame": 4.0 * wc / frames,
                       "share_of_wave_time": {"issuing": g("SQ_ACTIVE_INST_ANY") / wc if g("SQ_ACTIVE_INST_ANY") else None,
                                              "waiting at s_waitcnt": g("SQ_WAIT_ANY") / wc if g("SQ_WAIT_ANY") else None,
                                              "stalled at issue": g("SQ_WAIT_INST_ANY") / wc if g("SQ_WAIT_INST_ANY") else None,
                                              "of which waiting to issue an LDS instruction": g("SQ_WAIT_INST_LDS") / wc if g("SQ_WAIT_INST_LDS") else None},
                       "valu_active_share_of_simd_time": 3.0 * g("SQ_ACTIVE_INST_VALU") / wc if g("SQ_ACTIVE_INST_VALU") else None,
                       "lds_issue_share_of_wave_time": g("SQ_ACTIVE_INST_LDS") / wc if g("SQ_ACTIVE_INST_LDS") else None,
                       "lds_array_cycles_per_frame": g("SQ_LDS_IDX_ACTIVE") / frames if g("SQ_LDS_IDX_ACTIVE") else None,
                       "lds_conflict_share_of_lds_cycles": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None}}
    json.dump(doc, open(os.path.join(P, "mfcc_stream2048_pmc.json"), "w"), indent=1)
    print("2k: VALU/frame %.0f, LDS/frame %.1f" % (doc["derived"]["valu_instructions_per_frame"], doc["derived"]["lds_instructions_per_frame"]))
