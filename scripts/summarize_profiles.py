"""Turns gpurun_out/prof_<tag>/ (scripts/collect_profiles.sh) into the summaries committed under profiles/.

usage: python scripts/summarize_profiles.py <tag>
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join('gpurun_out', f'prof_{tag}')
dst = 'profiles'
IN_LAYER = 'gemm_f32_kernel<4, 1, 2, 4, 16, 2, 1, 3, 1'          # WN in-layer GEMM, fp32, 256 x 128 tiles


def find(sub, pattern):
    hits = sorted(glob.glob(os.path.join(src, sub, '**', pattern), recursive=True))
    return hits[0] if hits else None


def copy_stats(sub, name):
    f = find(sub, '*kernel_stats.csv')
    if f:
        shutil.copy(f, os.path.join(dst, name))
        print('wrote', name)
    return f


copy_stats('stats', f'{tag}_bench_kernel_stats.csv')
copy_stats('stats_f16', f'{tag}_waveglow_f16_kernel_stats.csv')
copy_stats('stats_f16x3', f'{tag}_waveglow_f16x3_kernel_stats.csv')
for sub in sorted(glob.glob(os.path.join(src, 'taco_b*'))):
    name = os.path.basename(sub)
    if os.path.isdir(sub) and not name.startswith('taco_pmc'):
        copy_stats(name, f'{tag}_tacotron2_{name[len("taco_"):]}_kernel_stats.csv')
bj = os.path.join(src, 'bench_under_rocprof.json')
if os.path.exists(bj):
    lines = [l for l in open(bj) if l.startswith('{')]
    if lines:
        open(os.path.join(dst, f'{tag}_bench_under_rocprof.json'), 'w').write(lines[-1])


def counters(sub):
    """kernel name -> counter -> list of per-dispatch values (summed over dimensions/XCDs as rocprofv3 reports them)."""
    f = find(sub, '*counter_collection.csv')
    out = {}
    if not f:
        return out
    per = {}
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'], r['Counter_Name'], r['Dispatch_Id'])
        per[k] = per.get(k, 0.0) + float(r['Counter_Value'])
    for (kn, cn, _), v in per.items():
        out.setdefault(kn, {}).setdefault(cn, []).append(v)
    return out


summary = {'round': tag, 'workload': 'bench.py B=8 T=800 (one step)', 'kernels': {}}
merged = {}
for sub in ('pmc_FETCH_SIZE', 'pmc_WRITE_SIZE', 'pmc_SQ_VALU_MFMA_BUSY_CYCLES'):
    for kn, cs in counters(sub).items():
        for cn, vals in cs.items():
            merged.setdefault(kn, {})[cn] = vals
for kn, cs in merged.items():
    if 'gemm_f32_kernel' not in kn and 'wn_' not in kn and 'wino4' not in kn:
        continue
    e = {'launches': max(len(v) for v in cs.values())}
    for cn, vals in cs.items():
        e[cn + '_mean'] = sum(vals) / len(vals)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs:
        busy = sum(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / len(cs['SQ_VALU_MFMA_BUSY_CYCLES'])
        gui = sum(cs['GRBM_GUI_ACTIVE']) / len(cs['GRBM_GUI_ACTIVE'])
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs; GRBM_GUI_ACTIVE sums over the 8 XCDs
        e['mfma_busy_frac'] = busy / (1024.0 * gui / 8.0)
    if 'SQ_WAIT_ANY' in cs and 'SQ_WAVE_CYCLES' in cs:
        e['wait_any_frac'] = sum(cs['SQ_WAIT_ANY']) / max(1.0, sum(cs['SQ_WAVE_CYCLES']))
    summary['kernels'][kn[:120]] = e
for sub in sorted(glob.glob(os.path.join(src, 'taco_pmc_b*'))):
    if not os.path.isdir(sub):
        continue
    taco = counters(os.path.basename(sub))
    if taco:
        summary['tacotron2_' + os.path.basename(sub)[len('taco_pmc_'):] + '_fetch_KB_mean'] = {
            kn[:90]: {'mean_KB': sum(cs['FETCH_SIZE']) / len(cs['FETCH_SIZE']), 'launches': len(cs['FETCH_SIZE'])}
            for kn, cs in taco.items() if 'FETCH_SIZE' in cs and ('persist' in kn or 'fused' in kn or len(cs['FETCH_SIZE']) >= 32)}
# fp16 / split-fp16 WaveGlow kernels: bytes, MFMA busy fraction and effective clock of every GEMM / end-fold kernel
for prec in ('f16', 'f16x3'):
    m = {}
    for sub in (f'pmc_{prec}_FETCH_SIZE', f'pmc_{prec}_WRITE_SIZE', f'pmc_{prec}_SQ_VALU_MFMA_BUSY_CYCLES'):
        for kn, cs in counters(sub).items():
            for cn, vals in cs.items():
                m.setdefault(kn, {})[cn] = vals
    stats = {}
    f = find(f'stats_{prec}', '*kernel_stats.csv')
    if f:
        for r in csv.DictReader(open(f)):
            stats[r['Name']] = float(r['AverageNs'])
    outp = {}
    for kn, cs in m.items():
        if 'gemm_f32_kernel' not in kn and 'wn_' not in kn and 'wino4' not in kn:
            continue
        e = {'launches': max(len(v) for v in cs.values())}
        for cn, vals in cs.items():
            e[cn + '_mean'] = sum(vals) / len(vals)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs:
            busy = sum(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / len(cs['SQ_VALU_MFMA_BUSY_CYCLES'])
            gui = sum(cs['GRBM_GUI_ACTIVE']) / len(cs['GRBM_GUI_ACTIVE'])
            e['mfma_busy_frac'] = busy / (1024.0 * gui / 8.0)
            if kn in stats:
                e['avg_ns_unprofiled_pass'] = stats[kn]
                e['effective_clock_GHz'] = (gui / 8.0) / stats[kn]
        outp[kn[:140]] = e
    if outp:
        summary[f'waveglow_{prec}_kernels'] = outp
json.dump(summary, open(os.path.join(dst, f'{tag}_pmc_counters.json'), 'w'), indent=1)
print('wrote', f'{tag}_pmc_counters.json')
# the entry bench.py copies into `roofline.traffic`: the dominant kernel of the fp32 step = the in-layer GEMM.  With the
# Winograd form (csrc/wn_wino.hip) that is one template in two tile heights (TAG 4: 256 rows for pairs of phases, 128 rows
# for pairs of frames): launch-weighted means over both; without it, the direct kernel.
WINO = 'gemm_f32_kernel<4, 1, '
# round 4: ONE fused kernel per layer (wino4_fused2_kernel)
cands = [(kn, e) for kn, e in summary['kernels'].items() if 'wino4_fused2_kernel' in kn and 'FETCH_SIZE_mean' in e]
if not cands:
    cands = [(kn, e) for kn, e in summary['kernels'].items() if WINO in kn and ', 4, 0, 1, false, 3, 1>' in kn and 'FETCH_SIZE_mean' in e]
if not cands:
    cands = [(kn, e) for kn, e in summary['kernels'].items() if IN_LAYER in kn and 'FETCH_SIZE_mean' in e][:1]
if cands:
    n = sum(e['launches'] for _, e in cands)

    def wmean(key):
        vals = [(e.get(key), e['launches']) for _, e in cands]
        return None if any(v is None for v, _ in vals) else sum(v * w for v, w in vals) / n
    latest = {'round': tag, 'source': f'profiles/{tag}_pmc_counters.json', 'workload': summary['workload'],
              'kernel': ' + '.join(kn for kn, _ in cands), 'wn_in_layer': {
                  'FETCH_SIZE_KB_mean': wmean('FETCH_SIZE_mean'), 'WRITE_SIZE_KB_mean': wmean('WRITE_SIZE_mean'),
                  'launches': n, 'mfma_busy_frac': wmean('mfma_busy_frac'), 'wait_any_frac': wmean('wait_any_frac'),
                  'lds_bank_conflict_cycles': wmean('SQ_LDS_BANK_CONFLICT_mean')}}
    json.dump(latest, open(os.path.join(dst, 'pmc_hbm_traffic_latest.json'), 'w'), indent=1)
    print('wrote pmc_hbm_traffic_latest.json')
