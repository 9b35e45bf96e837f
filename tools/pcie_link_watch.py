#!/usr/bin/env python3
"""Polls the PCIe link state of every AMD GPU the kernel exposes (sysfs current_link_speed / current_link_width, pp_dpm_pcie,
power_dpm_force_performance_level) and prints a line whenever it changes -- the evidence behind DESIGN.md section 9's "half-rate link"
finding: a process whose page-locked copies run at 30 instead of 57 GB/s finds the link at 16 GT/s instead of 32 GT/s.
  python tools/pcie_link_watch.py [seconds] [interval]      (background it next to the workload)"""
import glob
import os
import sys
import time


def read(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return '<%s>' % e.__class__.__name__


def gpus():
    out = []
    for d in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        if read(d + '/vendor') == '0x1002':
            out.append(os.path.realpath(d))
    return sorted(set(out))


def state(d):
    dpm = read(d + '/pp_dpm_pcie').replace('\n', ' | ')
    return {'link': read(d + '/current_link_speed') + ' x' + read(d + '/current_link_width'), 'max': read(d + '/max_link_speed') + ' x' + read(d + '/max_link_width'),
            'pp_dpm_pcie': dpm, 'perf_level': read(d + '/power_dpm_force_performance_level')}


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30
    dt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
    devs = gpus()
    print('gpus:', devs, flush=True)
    last = {}
    t0 = time.time()
    while time.time() - t0 < secs:
        for d in devs:
            s = state(d)
            if last.get(d) != s:
                print('%7.2f s  %s  %s' % (time.time() - t0, os.path.basename(d), s), flush=True)
                last[d] = s
        time.sleep(dt)


if __name__ == '__main__':
    main()
