#!/bin/bash
# diagnostic (GPU box): what sysfs offers about the card an ordinary user may read -> gpurun_out/env_probe.txt
out=${1:-gpurun_out/env_probe.txt}
{
for c in /sys/class/drm/card*/device; do
  [ -e $c/vendor ] || continue
  echo "== $c vendor $(cat $c/vendor 2>/dev/null) device $(cat $c/device 2>/dev/null) bus $(basename $(readlink -f $c))"
  for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent mem_busy_percent current_link_speed power_dpm_force_performance_level; do
    [ -r $c/$f ] && echo "-- $f: $(cat $c/$f 2>/dev/null | tr '\n' ' ')"
  done
  for h in $c/hwmon/hwmon*; do
    for f in $h/*_input $h/*_average $h/*_label $h/power1_cap; do [ -r $f ] && echo "-- ${f#$c/}: $(cat $f 2>/dev/null)"; done
  done
done
which rocm-smi amd-smi 2>/dev/null
} > $out 2>&1
