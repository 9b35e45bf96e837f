#!/usr/bin/env python3
"""What clears the half-rate DMA state some processes start in?  One fresh process: probe the copy rate
(zk_ctx_copy_probe) after each of a list of actions, with the device's DPM / runtime-PM sysfs state next to it.
  python tools/exp_link_state.py idle4,probe4g,tables,idle4"""
import glob
import json
import os
import sys
import time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read(p):
    try:
        return open(p).read().strip()
    except OSError:
        return None


def cur(txt):   # the starred level of a pp_dpm_* file
    if not txt:
        return None
    for l in txt.splitlines():
        if l.rstrip().endswith('*'):
            return l.strip()
    return txt.replace('\n', ' | ')[:60]


def dev_state(d):
    s = {k: cur(read(d + '/pp_dpm_' + k)) for k in ('sclk', 'mclk', 'fclk', 'socclk', 'pcie')}
    s['runtime_status'] = read(d + '/power/runtime_status')
    s['link'] = read(d + '/current_link_speed')
    return {k: v for k, v in s.items() if v is not None}


def main():
    actions = sys.argv[1].split(',') if len(sys.argv) > 1 else []
    tag = sys.argv[2] if len(sys.argv) > 2 else ''
    import zkp_ecdsa_amd as Z
    t0 = time.time()
    eng = Z.Engine(0)
    devs = [os.path.realpath(d) for d in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')) if read(d + '/vendor') == '0x1002']
    rec = {'tag': tag, 'create_s': round(time.time() - t0, 2), 'steps': []}

    import ctypes as C
    bus = C.create_string_buffer(64)
    hip = C.CDLL('libamdhip64.so')
    hip.hipDeviceGetPCIBusId(bus, 64, 0)
    mine = [d for d in devs if d.lower().endswith(bus.value.decode().lower())]
    rec['bus'] = bus.value.decode()
    rec['gpu_numa_node'] = read(mine[0] + '/numa_node') if mine else None
    sched_getcpu = C.CDLL(None).sched_getcpu

    def cpu_node():
        cpu = sched_getcpu()
        for x in os.listdir('/sys/devices/system/cpu/cpu%d' % cpu):
            if x.startswith('node'):
                return cpu, int(x[4:])
        return cpu, -1

    def step(name):
        t = time.time()
        r = {'default': eng.copy_probe(0, 256 << 20), 'node0': eng.copy_probe(0, 256 << 20, 0), 'node1': eng.copy_probe(0, 256 << 20, 1)}
        rec['steps'].append({'after': name, 't': round(t - t0, 2), 'd2h_h2d': r, 'cpu_node': cpu_node(), 'dev': dev_state(mine[0]) if mine else None})
    step('create')
    for a in actions:
        if a.startswith('idle'):
            time.sleep(float(a[4:]))
        elif a == 'probe4g':
            eng.copy_probe(0, 4 << 30)
        elif a == 'tables':
            eng.set_comb_bits(24)
            eng.set_params(*eng.synth_params(1), 80)
        elif a == 'tables16':
            eng.set_comb_bits(16)
            eng.set_params(*eng.synth_params(1), 80)
        step(a)
    print(json.dumps(rec), flush=True)
    os._exit(0)


if __name__ == '__main__':
    main()
