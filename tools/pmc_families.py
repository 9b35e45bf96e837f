#!/usr/bin/env python3
"""Per-family and whole-step view of rocprofv3 --pmc passes over the prover (the families of bench.py's gpu_ms_by_family_per_step) or, with --verify,
over the verifier (the families of verify.gpu_ms_by_family_per_step: only the dispatches from the first k_v_header on are counted, divided by --steps).
    python tools/pmc_families.py [--verify] [--steps N] PROOFS <pmc dir> [<pmc dir> ...]
Every directory holds one pass (counter_collection CSVs, any subset of: SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE) plus kernel_trace CSVs for the durations.  SQ_* cycle counters
are quad-cycles summed over waves (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE counts shader-clock cycles per dispatch and is reported summed over the
eight XCDs (check: sum / 8 / kernel time = 2.0-2.1 GHz in a profiled run), so
    SIMD VALU busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
is the share of SIMD-cycles in which a vector instruction was executing (k_tom_commit: 0.94 at two waves per SIMD).  FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory-side port, raw
(16-byte-per-lane loads are tallied at half their bytes on gfx950)."""
import collections
import csv
import glob
import os
import re
import sys

FAMILIES = [
    ('tom_commit', r'k_tom_commit'),
    ('p256_exp_commit', r'k_exp_commit'),
    ('gk_fold', r'k_gk_(scalars|sort|asub|block|finish|cd_scalars|tile|level)|k_gkm_asub'),
    ('hash', r'k_exp_challenge|k_exph_|k_gk_hash|k_padd_hash'),
    ('respond_write', r'k_write_|k_gk_respond|k_padd_respond|k_status_out'),
    ('tom_normalize', r'k_tom_normalize'),
    ('rng_prepass', r'k_rng_prepass'),
    ('scalars', r'k_lista_scalars|k_padd_i7|k_padd_inv|k_padd_scalars'),
    ('tom_derived', r'k_padd_derived'),
    ('p256_normalize', r'k_p256_normalize'),
    ('p256_front', r'k_front'),
    ('p256_rtab', r'k_rtab'),
    ('scan', r'k_scan|k_items|k_words_to_host'),
    ('p256_t1', r'k_t1\b'),
]
VFAMILIES = [
    ('v_msm_tom', r'k_msm_'),
    ('v_msm_p256', r'k_pm_'),
    ('v_p256_exp_points', r'k_v_exp_points|k_v_exp_status|k_p256_normalize'),
    ('v_hash', r'k_v_challenges|k_v_exph_msg|k_exph_|k_v_sample|k_v_padd_hash'),
    ('v_terms', r'k_v_slot_|k_v_proof_'),
    ('v_parse_validate', r'k_v_header|k_v_validate|k_v_unpack|k_offsets_monotonic'),
    ('v_gk_total', r'k_v_gk_|k_gkm_|k_v_clambda'),
    ('v_p256_front_rtab', r'k_v_front_|k_rtab_'),
    ('v_tom_fixed', r'k_v_t1_scalars|k_tom_commit|k_tom_normalize|k_v_derived'),
    ('v_per_proof_sums', r'k_v_straus|k_v_term_tables|k_v_acc_tree|k_v_p256_'),
    ('v_final', r'k_v_final|k_words_to_host|k_default_vseeds'),
]
SETUP = r'k_tomtab|k_pfix|k_ktab|k_gk_etab|k_gkm_(ring|etab)|k_ring|k_synth|k_build|fillBuffer|copyBuffer|k_bytes_to|k_affine_to'


def family(name):
    if re.search(SETUP, name):
        return None
    for f, pat in FAMILIES:
        if re.search(pat, name):
            return f
    return 'other'


def main():
    global FAMILIES
    argv = sys.argv[1:]
    verify = '--verify' in argv
    if verify:
        argv.remove('--verify')
        FAMILIES = VFAMILIES
    steps = 1
    if '--steps' in argv:
        i = argv.index('--steps')
        steps = int(argv[i + 1])
        del argv[i:i + 2]
    proofs = int(argv[0])
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    seen_dur = False
    for d in argv[1:]:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            rows = list(csv.DictReader(open(f)))
            name_of = lambda r: r.get('Kernel_Name', r.get('Kernel Name', '?'))
            first_v = min([int(r['Dispatch_Id']) for r in rows if name_of(r).startswith('k_v_header')] or [0]) if verify else 0
            for r in rows:
                if verify and int(r['Dispatch_Id']) < first_v:
                    continue
                fam = family(name_of(r))
                if fam:
                    cnt[fam][r['Counter_Name']] += float(r['Counter_Value']) / steps
        if not seen_dur:
            for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
                rows = list(csv.DictReader(open(f)))
                t_v = min([float(r['Start_Timestamp']) for r in rows if r.get('Kernel_Name', '?').startswith('k_v_header')] or [0]) if verify else 0
                for r in rows:
                    if verify and float(r['Start_Timestamp']) < t_v:
                        continue
                    fam = family(r.get('Kernel_Name', '?'))
                    if fam:
                        dur[fam] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6 / steps
                        seen_dur = True
    tot = collections.defaultdict(float)
    print('%-16s %8s %10s %8s %8s %8s %9s %10s' % ('family', 'ms', 'VALU/proof', 'valu/wv', 'wait_mem', 'wait_iss', 'SIMD busy', 'B/proof'))
    for fam in [f for f, _ in FAMILIES] + ['other']:
        c = cnt.get(fam)
        if not c:
            continue
        for k, v in c.items():
            tot[k] += v
        tot['ms'] += dur.get(fam, 0.0)
        wc = c.get('SQ_WAVE_CYCLES', 0.0)
        g = c.get('GRBM_GUI_ACTIVE', 0.0)
        kib = c.get('FETCH_SIZE', 0.0) + c.get('WRITE_SIZE', 0.0)
        print('%-16s %8.2f %10.0f %8.3f %8.3f %8.3f %9s %10.0f' % (
            fam, dur.get(fam, 0.0), c.get('SQ_INSTS_VALU', 0.0) / proofs, c.get('SQ_ACTIVE_INST_VALU', 0.0) / wc if wc else 0, c.get('SQ_WAIT_ANY', 0.0) / wc if wc else 0,
            c.get('SQ_WAIT_INST_ANY', 0.0) / wc if wc else 0, '%.3f' % (32 * c.get('SQ_ACTIVE_INST_VALU', 0.0) / (1024 * g)) if g else '-', kib * 1024 / proofs))
    wc, g = tot.get('SQ_WAVE_CYCLES', 0.0), tot.get('GRBM_GUI_ACTIVE', 0.0)
    print('%-16s %8.2f %10.0f %8.3f %8.3f %8.3f %9s %10.0f' % (
        'WHOLE STEP', tot['ms'], tot.get('SQ_INSTS_VALU', 0.0) / proofs, tot.get('SQ_ACTIVE_INST_VALU', 0.0) / wc if wc else 0, tot.get('SQ_WAIT_ANY', 0.0) / wc if wc else 0,
        tot.get('SQ_WAIT_INST_ANY', 0.0) / wc if wc else 0, '%.3f' % (32 * tot.get('SQ_ACTIVE_INST_VALU', 0.0) / (1024 * g)) if g else '-',
        (tot.get('FETCH_SIZE', 0.0) + tot.get('WRITE_SIZE', 0.0)) * 1024 / proofs))
    if g:
        print('# whole step: sum SQ_ACTIVE_INST_VALU = %.4g quad-cycles, sum GRBM_GUI_ACTIVE = %.4g cycles over 8 XCDs (effective clock %.2f GHz), sum SQ_BUSY_CYCLES = %.4g' % (
            tot.get('SQ_ACTIVE_INST_VALU', 0.0), g, g / 8 / (tot['ms'] * 1e-3) / 1e9 if tot['ms'] else 0, tot.get('SQ_BUSY_CYCLES', 0.0)))


if __name__ == '__main__':
    main()
