#!/bin/bash
# What a GPU box of the pool exposes about its clocks / power / temperature (sysfs, rocm-smi, amd-smi): run once per box next to a bench leg, so that
# bench.py's BoxProbe reads the right files.  usage (gpurun): bash tests/tools/box_probe.sh > gpurun_out/box_probe.log 2>&1
ls /sys/class/drm/
for c in /sys/class/drm/card[0-9]*/device; do
  echo "== $c"; ls $c | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk gpu_busy_percent power_dpm_force_performance_level unique_id vendor device; do echo "-- $f"; cat $c/$f 2>&1 | head -12; done
  for h in $c/hwmon/hwmon*; do echo "== $h"; for f in $h/*; do [ -f $f ] && echo "-- $(basename $f): $(cat $f 2>&1 | head -1)"; done; done
  ls -la $c/gpu_metrics 2>&1; od -A d -t u1 -N 8 $c/gpu_metrics 2>&1
done
python -c "import amdsmi; print('amdsmi ok', amdsmi.__file__)" 2>&1 | tail -1
which rocm-smi amd-smi
( time rocm-smi --showclocks --showpower --showtemp --showperflevel --showmaxpower 2>&1 | head -60 ) 2>&1
( time amd-smi metric --power --clock --temperature --json 2>&1 | head -150 ) 2>&1
nproc; uname -r
