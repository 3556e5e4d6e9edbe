"""Markdown rows of README.md's results table from a bench DETAIL file (bench.py --detail; until round 5 the line itself carried everything):
    python tools/readme_table.py profiles/r06_bench_detail.json"""
import json, sys
d = json.load(open(sys.argv[1]))
g = lambda *k: (lambda v: v)(__import__("functools").reduce(lambda a, b: a[b], k, d))
rows = []
r = d["roofline"]
rows.append(("fused 39-d MFCC, 100k × 3 s utterances resident in HBM (configs[1]; wave-stream kernel, software-pipelined quad loop, DCT / Δ / ΔΔ on the matrix cores)",
             "%.3g frames/s (%.2f ms/pass)" % (d["value"], r["kernel_ms"]),
             "%.2f of 8 TB/s HBM; HBM traffic %s × algorithmic (`profiles/mfcc_hbm_traffic.json`); %s of the 157.3 TFLOP/s packed-FMA vector peak on %s counted flop per frame (`tools/flop_count.py`), %s of the add / multiply-mix-weighted peak; the pass holds %s GHz at %s W of the 1400 W package cap (the line's `env.sustained_mfcc`)" % (
                 r["frac"], ("%.3f" % (r["traffic"] / r["algorithmic_bytes_per_launch"])) if r.get("traffic") else "n/a",
                 ("%.2f" % d["roofline_flop"]["frac"]) if (d.get("roofline_flop") or {}).get("frac") else "n/a",
                 (d.get("roofline_flop") or {}).get("flop_per_frame", "n/a"),
                 ("%.2f" % d["roofline_flop"]["frac_mix_weighted"]) if (d.get("roofline_flop") or {}).get("frac_mix_weighted") else "n/a",
                 ("%.2f" % (d["env"]["sustained_mfcc"]["sclk_mhz"]["mean"] / 1e3)) if "env" in d and "sustained_mfcc" in d["env"] and "sclk_mhz" in d["env"]["sustained_mfcc"] else "n/a",
                 ("%.0f" % d["env"]["sustained_mfcc"]["power_w"]["mean"]) if "env" in d and "sustained_mfcc" in d["env"] and "power_w" in d["env"]["sustained_mfcc"] else "n/a")))
v = d["mfcc_ref26_cmvn"]
rows.append(("the reference's `extract_feature` output: 13 cepstra + Δ, scaled per utterance (26-d), scaling inside the same kernel at three waves per SIMD",
             "%.3g frames/s (%.1f ms)" % (v["value"], v["roofline"]["kernel_ms"]), "%.2f of HBM" % v["roofline"]["frac"]))
if "mfcc_host_fed" in d:
    v = d["mfcc_host_fed"]
    pg = v.get("pageable") or {}
    sh = v.get("extract_feature_shim") or {}
    rows.append(("the same 39-d pass HOST-FED (what the reference-shaped callers do): %d × 3 s in pinned host memory through `ssp_mfcc_run(SSP_HOST)` — copy-in / compute / copy-back pipeline inside the call; float32 / int16 PCM (`ssp_mfcc_run_i16`)" % v["utterances"]
                 + ("; pageable memory" if "wall_ms" in pg else "") + ("; the Python call `GMM_UBM.extract_feature` on %d int16 utterances" % sh["utterances"] if "wall_ms" in sh else ""),
                 "%.2g / %.2g frames/s (%.1f / %.1f ms)" % (v["value"], v["value_i16"], v["wall_ms"], v["wall_ms_i16"])
                 + ("; %.2g" % pg["frames_per_s"] if "wall_ms" in pg else "") + ("; %.2g (%.1f ms)" % (sh["frames_per_s"], sh["wall_ms"]) if "wall_ms" in sh else ""),
                 "%.2f / %.2f of the PCIe bound measured in the same run (%.0f GB/s in, %.0f out)" % (v["frac_of_pcie_bound"], v["frac_of_pcie_bound_i16"], v["h2d_gbs"], v["d2h_gbs"])
                 + ("; pageable %.2f" % pg["frac_of_pcie_bound"] if "wall_ms" in pg else "")))
a, b = d["mfcc_inrepo"]["16k"], d["mfcc_inrepo"]["8k"]
rows.append(("in-repo MFCC (arithmetic pinned to the reference's own outputs), 16 kHz 512/256 13-d / 8 kHz, same kernel",
             "%.2g / %.2g frames/s (%.1f / %.1f ms)" % (a["value"], b["value"], a["roofline"]["kernel_ms"], b["roofline"]["kernel_ms"]),
             "%.2f / %.2f of HBM" % (a["roofline"]["frac"], b["roofline"]["frac"])))
v = d["mfcc_librosa"]
rows.append(("librosa-dialect MFCC (`MFCC_DTW.MFCC_lib`: 2048 / 512, 128 mel, top_db), 200k × 3 s at 8 kHz, 2048-point wave-stream kernel",
             "%.2g frames/s (%.1f ms)" % (v["value"], v["roofline"]["kernel_ms"]), "%.2f of HBM (on the package power cap at 2.1 GHz; `profiles/mfcc_stream2048_pmc.json`)" % v["roofline"]["frac"]))
v = d["plp"]
rows.append(("PLP features (sidekit `plp`), 100k × 3 s: Bark front end on the wave-stream kernel's dense-band instance + RASTA / LPC-cepstrum back end",
             "%.3g frames/s (%.1f + %.1f ms)" % (v["value"], v["front_ms"], v["back_ms"]), "front end %.2f of HBM" % v["front_roofline"]["frac"]))
v = d["gmm"]
rows.append(("GMM-UBM scoring, 51 models × 64 mixtures × 39-d, fp32 MFMA (parity path), one launch, per-utterance means fused; 12-byte decision records gathered",
             "%.3g frame-scores/s (%.1f ms)" % (v["value"], v["roofline"]["kernel_ms"]), "%.2f of the fp32 MFMA peak" % v["roofline"]["frac"]
             + (" (MFMA pipe %.0f %% busy, `profiles/gmm_mfma_util.json`)" % (100 * v["roofline"]["mfma_busy"]) if v["roofline"].get("mfma_busy") else "")))
v = d["gmm_bf16x3"]
rows.append(("same, bf16×3 split-precision MFMA + fp32 re-scoring of the close calls' candidate models, calibrated (heuristic) band (%d of %d listed; arg-max mismatches against fp32: %d)" % (v["utterances_rescored_in_fp32"], v["utterances"], v["argmax_mismatches_vs_fp32_path"]),
             "%.2g frame-scores/s (%.1f ms)" % (v["value"], v["roofline"]["kernel_ms"]), "%.2f of the dense bf16 peak on algorithmic FLOPs" % v["roofline"]["frac"]))
if "gmm_bf16x3_proven_band" in d:
    v = d["gmm_bf16x3_proven_band"]
    rows.append(("same with the PROVEN error bound as the band (`ssp_gmm_score` precision 1: %d of %d listed; arg-max mismatches against fp32: %d)" % (v["utterances_rescored_in_fp32"], v["utterances"], v["argmax_mismatches_vs_fp32_path"]),
                 "%.2g frame-scores/s (%.1f ms)" % (v["value"], v["kernel_ms"]), "%.2f of the dense bf16 peak on algorithmic FLOPs" % v["roofline"]["frac"]))
if "gmm_bf16x3_close_calls" in d:
    pts = d["gmm_bf16x3_close_calls"]["points"]
    rows.append(("same shape OFF the best case: speaker means at %s std from the UBM's (0.3 above) — share of utterances listed for fp32 re-scoring, proven band / heuristic band; arg-max mismatches against fp32: %d" % (
                     " / ".join("%g" % p["speaker_offset_std"] for p in pts), sum(p[k]["argmax_mismatches_vs_fp32_path"] for p in pts for k in ("proven_band", "heuristic_band"))),
                 "proven %s ms (%s listed); heuristic %s ms (%s); fp32 path %.0f ms" % (
                     " / ".join("%.1f" % p["proven_band"]["kernel_ms"] for p in pts), " / ".join("%.0f %%" % (100 * p["proven_band"]["fraction_rescored"]) for p in pts),
                     " / ".join("%.1f" % p["heuristic_band"]["kernel_ms"] for p in pts), " / ".join("%.2g %%" % (100 * p["heuristic_band"]["fraction_rescored"]) for p in pts),
                     pts[0]["fp32_kernel_ms"]), "—"))
if "gmm_auto" in d:
    v = d["gmm_auto"]
    pts = (d.get("gmm_bf16x3_close_calls") or {}).get("points", [])
    rows.append(("same, `precision = auto` (pilot on 2 % of the utterances, then the proven-band path or fp32): here and at the offsets above; time over the better fixed choice",
                 "%.1f ms" % v["kernel_ms"] + ("; " + " / ".join("%.1f" % p["auto"]["kernel_ms"] for p in pts if "auto" in p) + " ms" if pts else ""),
                 "ratio %.2f" % v["ratio_to_best_fixed"] + ("; " + " / ".join("%.2f" % p["auto"]["ratio_to_best_fixed"] for p in pts if "auto" in p) if pts else "")
                 + "; arg-max mismatches against fp32: %d" % v.get("mismatches_vs_fp32", v["argmax_mismatches_vs_fp32_path"])))
if "gmm_host_fed" in d and "wall_ms" in d["gmm_host_fed"]:
    v = d["gmm_host_fed"]
    rows.append(("GMM-UBM scoring HOST-FED (`GMM_UBM.py:181-197` hands host arrays): a quarter of the batch's features in pinned host memory through `ssp_gmm_score(SSP_HOST)`, fp32 path — rows copied in ahead of the kernels that score them",
                 "%.3g frame-scores/s (%.1f ms wall)" % (v["value"], v["wall_ms"]),
                 "copy-in %.1f ms + kernels %.1f ms = %.1f ms if staged whole; the longer of the two is %.2f of the wall time; scores bit-equal to the device path: %s" % (
                     v["copy_in_ms"], v["kernel_ms_device_path"], v["sum_ms"], v["overlap"], v["scores_equal_device_path"])))
c3 = d["gmm_cfg3_shape"]
fs = c3.get("bf16x3_full_share")
rows.append(("configs[3] model shape (K = 512, 1251 speakers + UBM): 12 000 utterances per GPU fp32 / bf16×3" + ("; the FULL per-GPU share (150 000 utterances) on bf16×3, measured" if fs else ""),
             "%.2g / %.2g frame-scores/s (%.2f s / %.2f s)" % (c3["f32"]["value"], c3["bf16x3"]["value"], c3["f32"]["kernel_ms"] / 1e3, c3["bf16x3"]["kernel_ms"] / 1e3)
             + ("; full share %.2f s = %.2g frame-scores/s, %d re-scored, %d arg-max mismatches vs the fp32 sample" % (fs["measured_s"], fs["value"], fs["utterances_rescored_in_fp32"], fs["argmax_mismatches_vs_fp32_sample"]) if fs else ""),
             "%.2f of the fp32 MFMA peak (fp32 sample)" % (c3["f32"]["tflops"] / 157.3)))
v = d["gmm_em"]
rows.append(("GMM EM training (E step + M sums per iteration), 3e6 frames × 64 mix × 39-d", "%.2f ms per iteration" % v["kernel_ms"], "%.2f of the fp32 MFMA peak" % v["roofline"]["frac"]))
v = d["cosine"]
rows.append(("cosine scoring, 1e6 × 1251 × 256, fp32 MFMA (parity path)", "%.2g pair-scores/s (%.2f ms)" % (v["value"], v["roofline"]["kernel_ms"]), "%.2f of the fp32 MFMA peak" % v["roofline"]["frac"]
             + (" (MFMA pipe %.0f %% busy, `profiles/cosine_mfma_util.json`)" % (100 * v["roofline"]["mfma_busy"]) if v["roofline"].get("mfma_busy") else "")))
if "cosine_bf16x3" in d:
    v, c = d["cosine_bf16x3"], d.get("cosine_bf16_cascade")
    rows.append(("same, arg-min only: bf16×3 sweep keeping the two best cosines + fp32 re-scoring inside a proven band from a device-side list (arg-min equal to fp32 on all rows: %s)" % v["argmin_equals_fp32_path"]
                 + ("; the cascade with a bf16 sweep in front (equal: %s; rows to later stages: %d / %d — well-separated synthetic embeddings)" % (c["argmin_equals_fp32_path"], c["rows_to_bf16x3"], c["rows_rescored_fp32"]) if c else ""),
                 "%.2g pair-scores/s (%.2f ms)" % (v["value"], v["roofline"]["kernel_ms"]) + ("; %.2g (%.2f ms)" % (c["value"], c["roofline"]["kernel_ms"]) if c else ""),
                 "%.2f of the dense bf16 peak executed (%.2f algorithmic)" % (v["roofline"]["frac_executed"], v["roofline"]["frac"]) + ("; cascade %.2f algorithmic" % c["roofline"]["frac"] if c else "")))
if "cosine_close_calls" in d:
    pts = d["cosine_close_calls"]["points"]
    rows.append(("same shape OFF the best case: embedding noise %s (0.7 above) — rows the cascade hands to its second / third stage; arg-min equal to fp32 on every row: %s" % (
                     " / ".join("%g" % p["embedding_noise"] for p in pts), all(p[k]["argmin_equals_fp32_path"] for p in pts for k in ("bf16x3", "cascade"))),
                 "cascade %s ms (%s to bf16×3, %s to fp32); bf16×3 alone %s ms; fp32 path %.1f ms" % (
                     " / ".join("%.2f" % p["cascade"]["kernel_ms"] for p in pts), " / ".join("%.1f %%" % (100 * p["cascade"]["fraction_to_bf16x3"]) for p in pts),
                     " / ".join("%.2f %%" % (100.0 * p["cascade"]["rows_rescored_fp32"] / p["rows"]) for p in pts),
                     " / ".join("%.2f" % p["bf16x3"]["kernel_ms"] for p in pts), pts[0]["fp32_kernel_ms"]), "—"))
if "cosine_auto" in d:
    v = d["cosine_auto"]
    pts = (d.get("cosine_close_calls") or {}).get("points", [])
    rows.append(("same, `precision = auto` (the pilot is the first round of the cascade's bf16 sweep; then cascade / bf16×3 / fp32): here and at the noise levels above; time over the best fixed choice",
                 "%.2f ms" % v["kernel_ms"] + ("; " + " / ".join("%.2f" % p["auto"]["kernel_ms"] for p in pts if "auto" in p) + " ms" if pts else ""),
                 "ratio %.2f" % v["ratio_to_best_fixed"] + ("; " + " / ".join("%.2f" % p["auto"]["ratio_to_best_fixed"] for p in pts if "auto" in p) if pts else "")
                 + "; arg-min mismatches against fp32: %d" % v.get("mismatches_vs_fp32", 0)))
if "cosine_host_fed" in d and "wall_ms" in d["cosine_host_fed"]:
    v = d["cosine_host_fed"]
    rows.append(("cosine scoring HOST-FED (`d_vector.py:315-319` hands host arrays): 1e6 × 256 embeddings in pinned host memory through `ssp_cosine_identify(SSP_HOST)`, fp32 path, arg-min + minimum",
                 "%.2g pair-scores/s (%.1f ms wall)" % (v["value"], v["wall_ms"]),
                 "copy-in %.1f ms + sweep %.1f ms = %.1f ms if staged whole; the copy is %.2f of the wall time; arg-min equal to the device path: %s" % (
                     v["copy_in_ms"], v["kernel_ms_device_path"], v["sum_ms"], v["overlap"], v["argmin_equals_device_path"])))
v = d["dvector_dnn"]
rows.append(("d-vector network forward 1274→256×4 (one packed object, hidden layers chained in registers), 5e5 embeddings", "%.2g embeddings/s (%.1f ms)" % (v["value"], v["kernel_ms"]), "%.2f of the fp32 MFMA peak" % v["roofline"]["frac"]))
v = d["dvector_pipeline"]
rows.append(("d-vector recogniser end to end (1 s chunks → MFCC 98×13 → network → cosine vs 1251 → arg-min), 3e5 chunks", "%.2g chunks/s" % v["value"], "—"))
v = d["dtw"]
rows.append(("DTW matcher, 128 × 64 pairs of 1222-element sequences", "%.2g pairs/s" % v["value"], "%.2f of the non-FMA fp32 vector rate" % v["roofline"]["frac"]))
if "cpu_baseline" in d:
    cp = d["cpu_baseline_parallel"]
    rows.append(("CPU (the box's host: %s cores, %s usable under the container's cgroup quota): oracle 1 thread / %s worker processes; the reference's own per-frame MFCC loop; the reference's GMM scoring loop (sklearn)" % (
                     cp.get("host_cores", "?"), cp.get("usable_cores", "?"), cp.get("cores", "?")),
                 "%.2g / %.2g frames/s; %.2g frames/s; %.2g frame-scores/s" % (d["cpu_baseline"]["value"], d["cpu_baseline_parallel"]["value"],
                                                                         d["mfcc_inrepo"]["cpu_baseline_reference_loop"]["value"], d["gmm"]["cpu_baseline_reference_loop"]["value"]), "—"))
print("| stage (config) | throughput | roofline |\n|---|---|---|")
for r_ in rows:
    print("| %s | %s | %s |" % r_)
