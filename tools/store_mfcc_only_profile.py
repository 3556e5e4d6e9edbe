"""gpurun_out/final_mfcc/* -> profiles/r01_final_mfcc_only*: per-launch rocprofv3 durations next to bench.py's hipEvent mean."""
import csv, json, shutil
shutil.copy('gpurun_out/final_mfcc/kernel_stats.csv', 'profiles/r01_final_mfcc_only_kernel_stats.csv')
shutil.copy('gpurun_out/final_mfcc/bench_line.json', 'profiles/r01_final_mfcc_only_bench_line.json')
rows = [r for r in csv.DictReader(open('gpurun_out/final_mfcc/kernel_trace.csv')) if 'mfcc_fused512' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
j = json.load(open('gpurun_out/final_mfcc/bench_line.json'))
st = [r for r in csv.DictReader(open('profiles/r01_final_mfcc_only_kernel_stats.csv')) if 'mfcc_fused512' in r['Name']][0]
md = """# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --stages mfcc --no-cpu-baseline   (MI355X, round 1 final)

The headline kernel alone: 1 set-up + 3 warm-up + 10 timed launches of `%s`.

| launch | %s |
|---|%s
| ms (rocprofv3 kernel trace) | %s |

* `--stats` row: calls %s, average %.3f ms (includes the set-up and warm-up launches), min %.3f, max %.3f (`r01_final_mfcc_only_kernel_stats.csv`).
* mean of the 10 timed launches (rocprofv3): **%.3f ms**; mean hipEvent duration measured inside `bench.py` in the same run
  (`roofline.kernel_ms` of `r01_final_mfcc_only_bench_line.json`): **%.3f ms** -> `roofline.achieved` %.0f GB/s, `frac` %.4f.
* boxes of the pool differ by a few percent (10.4 - 10.9 ms for this binary across the runs of this round).
""" % (st['Name'], " | ".join(str(i + 1) for i in range(len(d))), "---|" * len(d), " | ".join("%.3f" % x for x in d),
       st['Calls'], float(st['AverageNs']) / 1e6, float(st['MinNs']) / 1e6, float(st['MaxNs']) / 1e6,
       sum(d[-10:]) / 10, j['roofline']['kernel_ms'], j['roofline']['achieved'], j['roofline']['frac'])
open('profiles/r01_final_mfcc_only.md', 'w').write(md)
print("rocprof %.3f ms vs hipEvents %.3f ms" % (sum(d[-10:]) / 10, j['roofline']['kernel_ms']))
