"""gpurun_out/final_mfcc/* -> profiles/%s_final_mfcc_only*: per-launch rocprofv3 durations next to bench.py's hipEvent mean."""
import csv, json, os, shutil
ROUND = os.environ.get("ROUND", "r02")
KERN = os.environ.get("KERN", "mfcc_stream512")
shutil.copy('gpurun_out/final_mfcc/kernel_stats.csv', 'profiles/%s_final_mfcc_only_kernel_stats.csv' % ROUND)
shutil.copy('gpurun_out/final_mfcc/bench_line.json', 'profiles/%s_final_mfcc_only_bench_line.json' % ROUND)
rows = [r for r in csv.DictReader(open('gpurun_out/final_mfcc/kernel_trace.csv')) if KERN in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
j = json.load(open('gpurun_out/final_mfcc/bench_line.json'))
st = [r for r in csv.DictReader(open('profiles/%s_final_mfcc_only_kernel_stats.csv' % ROUND)) if KERN in r['Name']][0]
md = """# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --stages mfcc --no-cpu-baseline   (MI355X, %s final)

The headline kernel alone: 1 set-up + 3 warm-up + 10 timed launches of `%s`.

| launch | %s |
|---|%s
| ms (rocprofv3 kernel trace) | %s |

* `--stats` row: calls %s, average %.3f ms (includes the set-up and warm-up launches), min %.3f, max %.3f (`%s_final_mfcc_only_kernel_stats.csv`).
* mean of the 10 timed launches (rocprofv3): **%.3f ms**; median hipEvent duration measured inside `bench.py` in the same run
  (`roofline.kernel_ms` of `%s_final_mfcc_only_bench_line.json`): **%.3f ms** -> `roofline.achieved` %.0f GB/s, `frac` %.4f.
* boxes of the pool differ by a few percent for one binary.
""" % (ROUND, st['Name'], " | ".join(str(i + 1) for i in range(len(d))), "---|" * len(d), " | ".join("%.3f" % x for x in d),
       st['Calls'], float(st['AverageNs']) / 1e6, float(st['MinNs']) / 1e6, float(st['MaxNs']) / 1e6, ROUND,
       sum(d[-10:]) / 10, ROUND, j['roofline']['kernel_ms'], j['roofline']['achieved'], j['roofline']['frac'])
open('profiles/%s_final_mfcc_only.md' % ROUND, 'w').write(md)
print("rocprof %.3f ms vs hipEvents %.3f ms" % (sum(d[-10:]) / 10, j['roofline']['kernel_ms']))
