#!/usr/bin/env python
"""Condenses a rocprofv3 run (gpurun_out/prof_rN/{trace,pmc_fetch,pmc_write}) into the small
summaries committed under profiles/ (kernel stats with shortened names + per-kernel PMC means
with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md §HBM)."""
import json
import os
import sys

import pandas as pd


def short(name: str) -> str:
    name = name.replace('void ', '')
    return name if len(name) < 90 else name[:87] + '...'


def main(src: str, dst_prefix: str, site: str = 'caltech') -> None:
    out = {'site': site}
    ks = pd.read_csv(os.path.join(src, 'trace', 'bench_kernel_stats.csv'))
    ks['Name'] = ks['Name'].map(short)
    ks.to_csv(dst_prefix + '_kernel_stats.csv', index=False)
    out['kernel_stats'] = ks.head(6).to_dict(orient='records')
    pmc = {}
    for which, counter in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        path = os.path.join(src, which, 'bench_counter_collection.csv')
        if not os.path.exists(path):
            continue
        df = pd.read_csv(path)
        df = df[df['Kernel_Name'].str.contains('evc::')]
        for kname, grp in df.groupby('Kernel_Name'):
            e = pmc.setdefault(short(kname), {})
            e[counter + '_KB_mean'] = float(grp['Counter_Value'].mean())
            e[counter + '_KB_max'] = float(grp['Counter_Value'].max())
            e['dispatches_' + counter] = int(len(grp))
            e['VGPR'] = int(grp['VGPR_Count'].iloc[0])
            e['SGPR'] = int(grp['SGPR_Count'].iloc[0])
            e['scratch'] = int(grp['Scratch_Size'].iloc[0])
            e['LDS'] = int(grp['LDS_Block_Size'].iloc[0])
    for kname, e in pmc.items():
        if 'FETCH_SIZE_KB_mean' in e and 'WRITE_SIZE_KB_mean' in e:
            # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B
            # requests at 64 B, i.e. reports half the bytes of a coalesced stream -> double it.
            e['hbm_bytes_per_launch_raw'] = (e['FETCH_SIZE_KB_mean'] + e['WRITE_SIZE_KB_mean']) * 1024
            e['hbm_bytes_per_launch_corrected'] = (2 * e['FETCH_SIZE_KB_mean'] + e['WRITE_SIZE_KB_mean']) * 1024
    out['pmc'] = pmc
    for f in ('bench_plain.json', 'bench_traced.json'):
        p = os.path.join(src, f)
        if os.path.exists(p):
            try:
                out[f[:-5]] = json.loads(open(p).read().strip().splitlines()[-1])
            except Exception:
                pass
    # the code object these figures were measured on (bench.py's roofline.traffic refuses a figure from other kernels)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        from bench import code_object_hash
        out['code_object_sha256'] = code_object_hash()
    except Exception as exc:
        out['code_object_sha256'] = None
        print('code_object_hash failed:', exc)
    json.dump(out, open(dst_prefix + '_summary.json', 'w'), indent=1)
    print(json.dumps(out['pmc'], indent=1))
    # proposed profiles/traffic.json: the compact streaming kernel's entry refreshed from this run.  FETCH_SIZE is scaled
    # by the factor calibrated for this kernel's read shapes (profiles/r2_fetch_calibration.txt: 1.40, against the guide's
    # 2.0 for 16 B/lane streams); the raw and the x2 figures are kept as bounds.
    tpath = os.path.join(root, 'profiles', 'traffic.json')
    cq = [(k, e) for k, e in pmc.items() if 'step_kernel_cquad<true' in k and 'FETCH_SIZE_KB_mean' in e and 'WRITE_SIZE_KB_mean' in e]
    if cq and os.path.exists(tpath):
        k, e = max(cq, key=lambda ke: ke[1].get('dispatches_FETCH_SIZE', 0))
        tj = json.load(open(tpath))
        f, w = e['FETCH_SIZE_KB_mean'], e['WRITE_SIZE_KB_mean']
        # pipelined halves (bench.py --pipeline 2, the default): a step is two launches of half the batch each
        cfg = out.get('bench_plain', {}).get('config', {})
        lps = cfg.get('launches_per_step') or (2 if 'half-batch' in str(cfg.get('pipeline', '')) else 1)
        tj[f'{site}_N65536_project1_compact'] = {
            'hbm_bytes_per_launch': int((1.4 * f + w) * 1024), 'launches_per_step': lps,
            'fetch_kib': round(f, 1), 'write_kib': round(w, 1),
            'fetch_factor': 1.4, 'hbm_bytes_per_launch_raw_counters': int((f + w) * 1024),
            'hbm_bytes_per_launch_guide_x2_rule': int((2 * f + w) * 1024), 'kernel': k,
            'code_object_sha256': out['code_object_sha256'],
            'source': f'profiles/{os.path.basename(dst_prefix)}_summary.json (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) x '
                      'profiles/r2_fetch_calibration.txt (read factor 1.40 for this kernel\'s access shapes instead of the '
                      'guide\'s 2.0 for 16 B/lane streams)'}
        json.dump(tj, open(dst_prefix + '_traffic.json', 'w'), indent=1)
        print('proposed traffic.json ->', dst_prefix + '_traffic.json')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 'caltech')
