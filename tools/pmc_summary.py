#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes into profiles/r03_pmc_summary_cfg3s.json (HBM traffic per launch).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --skip-cpu --only all
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ... (same)
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r03_pmc_summary_cfg3s.json

FETCH_SIZE / WRITE_SIZE are in KiB and cannot share a pass (TCC slots). On gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced (16 B/lane) streaming read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): the summary doubles it for kernels
listed in WIDE_READ and leaves the others as reported (uncalibrated access widths).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

WIDE_READ = {'pan_core_sweep_kernel'}          # 16 B / lane streaming reads


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = name.split('(')[0].replace('void ', '').strip()
    m = re.match(r'filter_kernel<(true|false), (true|false)>', name)
    if m:   # (the block pass is the <new> instantiation too)
        return 'filter_kernel<%s>' % ('new' if m.group(2) == 'true' else 'all')
    return re.sub(r'<.*$', '', name)


def load(directory, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get('Counter_Name') != counter:
                continue
            k = short(row['Kernel_Name'])
            tot[k] += float(row['Counter_Value'])
            n[k] += 1
    return tot, n


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    f_tot, f_n = load(fetch_dir, 'FETCH_SIZE')
    w_tot, w_n = load(write_dir, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(f_tot) | set(w_tot)):
        fn, wn = max(f_n.get(k, 0), 1), max(w_n.get(k, 0), 1)
        fetch = f_tot.get(k, 0.0) * 1024 / fn
        write = w_tot.get(k, 0.0) * 1024 / wn
        corrected = fetch * 2 if k in WIDE_READ else fetch
        out[k] = {'launches': f_n.get(k, 0), 'fetch_bytes_per_launch_raw': fetch,
                  'fetch_bytes_per_launch': corrected, 'fetch_corrected_x2': k in WIDE_READ,
                  'write_bytes_per_launch': write, 'hbm_bytes_per_launch': corrected + write}
    json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0',
               'kernels': out}, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
