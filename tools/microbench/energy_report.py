"""Table from tools/microbench/energy.sh's output: energy per wave-instruction over the s_nop baseline.
   python tools/microbench/energy_report.py gpurun_out/energy.txt > profiles/<round>_energy_microbench.md"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
rows, samp = [], []
for l in txt:
    m = re.match(r"sclk (\d+) watts ([\d.]+)", l.strip())
    if m:
        samp.append((int(m.group(1)), float(m.group(2))))
        continue
    m = re.match(r'RESULT op="(.*)" wps=(\d) seconds=([\d.]+) wave_instr_per_s=([\d.e+]+) memtime_ghz=([\d.]+)', l)
    if m:
        sc = sorted(s[0] for s in samp)[len(samp) // 2] if samp else 0
        pw = sorted(s[1] for s in samp)[len(samp) // 2] if samp else 0
        rows.append((m.group(1), int(m.group(2)), float(m.group(4)), sc, pw))
        samp = []
base = {r[1]: r for r in rows if r[0] == "s_nop 0"}
print("# Socket power by instruction kind (MI355X, `tools/microbench/energy.sh`)\n")
print("`issue_bench power`: every SIMD of the chip runs a loop of 32 independent instructions of ONE kind (8 accumulator chains, slowly changing")
print("operands — real data toggles more bits) at 1 and at 3 waves per SIMD for ~3 s; `rocm-smi` is read three times meanwhile (median).")
print("Energy per wave-instruction (64 lanes) = (socket power − power of the `s_nop` loop at the same occupancy) ÷ the chip's wave-instruction rate.\n")
print("| instruction | waves / SIMD | wave-instr / s (chip) | sclk MHz | socket W | nJ per wave-instruction over `s_nop` |")
print("|---|---|---|---|---|---|")
for r in rows:
    b = base.get(r[1])
    e = (r[4] - b[4]) / r[2] * 1e9 if b and r[2] > 0 else float("nan")
    print("| `%s` | %d | %.3g | %d | %.0f | %s |" % (r[0], r[1], r[2], r[3], r[4], "—" if r[0] == "s_nop 0" else "%.2f" % e))
