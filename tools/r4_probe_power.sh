#!/bin/bash
# what does this box expose for power / clocks?  (run once; informs tools/power_trace.py)
for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_*; do echo "$f $(cat $f 2>/dev/null)"; done
for f in /sys/class/drm/card*/device/hwmon/hwmon*/freq*_input; do echo "$f $(cat $f 2>/dev/null)"; done
ls /sys/class/drm/ | tr '\n' ' '; echo
rocm-smi --showpower --showclocks 2>&1 | head -30
amd-smi metric --power --clock --json 2>&1 | head -80
