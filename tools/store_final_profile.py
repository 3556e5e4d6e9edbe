"""Copy gpurun_out/final/{bench_line.json,kernel_stats.csv} into profiles/%s_final_* and render the markdown summary."""
import csv, json, os, shutil, subprocess, sys
ROUND = os.environ.get("ROUND", "r02")
KERN = os.environ.get("KERN", "mfcc_stream512")
shutil.copy('gpurun_out/final/bench_line.json', 'profiles/%s_final_bench_line.json' % ROUND)
shutil.copy('gpurun_out/final/kernel_stats.csv', 'profiles/%s_final_kernel_stats.csv' % ROUND)
md = subprocess.run([sys.executable, 'tools/stats_md.py', 'gpurun_out/final/kernel_stats.csv',
                     'rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline   (MI355X, %s final)' % ROUND],
                    capture_output=True, text=True).stdout
md = "\n".join(l for l in md.splitlines() if "at::native" not in l) + "\n"
d = json.load(open('profiles/%s_final_bench_line.json' % ROUND))
r = [r for r in csv.DictReader(open('profiles/%s_final_kernel_stats.csv' % ROUND)) if KERN in r['Name']][0]
md += ("\n(torch's own elementwise / reduction kernels that build the synthetic inputs are left out of this table; they are in the csv.)\n"
       "The JSON line of the un-profiled run of the same command is `%s_final_bench_line.json` (fused MFCC kernel: %.3f ms median under hipEvents "
       "over the 10 timed steps vs %.3f ms mean here over %s launches incl. warm-up and extra-stage launches; min %.3f ms).\n"
       % (ROUND, d['roofline']['kernel_ms'], float(r['AverageNs']) / 1e6, r['Calls'], float(r['MinNs']) / 1e6))
open('profiles/%s_final_kernel_stats.md' % ROUND, 'w').write(md)
print("value %.4g frames/s, kernel %.3f ms, frac %.4f" % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))
for k in ('gmm', 'gmm_bf16x3', 'cosine', 'gmm_em', 'dvector_dnn', 'dvector_pipeline', 'dtw', 'plp'):
    v = d.get(k, {})
    print(k, "%.4g" % v.get('value', 0), v.get('roofline', {}).get('frac', ''), v.get('kernel_ms', v.get('ms', '')))
