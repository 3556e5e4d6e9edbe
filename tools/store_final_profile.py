"""Copy gpurun_out/final/{bench_line.json,kernel_stats.csv} into profiles/r01_final_* and render the markdown summary."""
import csv, json, shutil, subprocess, sys
shutil.copy('gpurun_out/final/bench_line.json', 'profiles/r01_final_bench_line.json')
shutil.copy('gpurun_out/final/kernel_stats.csv', 'profiles/r01_final_kernel_stats.csv')
md = subprocess.run([sys.executable, 'tools/stats_md.py', 'gpurun_out/final/kernel_stats.csv',
                     'rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline   (MI355X, round 1 final)'],
                    capture_output=True, text=True).stdout
md = "\n".join(l for l in md.splitlines() if "at::native" not in l) + "\n"
d = json.load(open('profiles/r01_final_bench_line.json'))
r = [r for r in csv.DictReader(open('profiles/r01_final_kernel_stats.csv')) if 'mfcc_fused512' in r['Name']][0]
md += ("\n(torch's own elementwise / reduction kernels that build the synthetic inputs are left out of this table; they are in the csv.)\n"
       "The JSON line of the un-profiled run of the same command is `r01_final_bench_line.json` (fused MFCC kernel: %.3f ms mean under hipEvents "
       "over the 10 timed steps vs %.3f ms mean here over %s launches incl. warm-up and extra-stage launches; min %.3f ms).\n"
       % (d['roofline']['kernel_ms'], float(r['AverageNs']) / 1e6, r['Calls'], float(r['MinNs']) / 1e6))
open('profiles/r01_final_kernel_stats.md', 'w').write(md)
print("value %.4g frames/s, kernel %.3f ms, frac %.4f" % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))
for k in ('gmm', 'gmm_bf16x3', 'cosine', 'gmm_em', 'dvector_dnn', 'dvector_pipeline', 'dtw', 'plp'):
    v = d.get(k, {})
    print(k, "%.4g" % v.get('value', 0), v.get('roofline', {}).get('frac', ''), v.get('kernel_ms', v.get('ms', '')))
