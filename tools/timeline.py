"""Critical-path view of a rocprofv3 --kernel-trace CSV of one clustering run: per kernel, the
time it occupies on the main stream, and the idle gaps of that stream (host round trips,
launch latency). Usage: python tools/timeline.py <kernel_trace.csv> [main_queue_id|auto [sweep_to_dump]]"""
import re
import sys

import pandas as pd


def short(name):
    m = re.search(r'(\w+_kernel|DeviceScan\w*|\w+Kernel)\b', name)
    base = m.group(1) if m else name[:40]
    m2 = re.search(r'count_kernel<(\d)', name)
    if m2:
        base = 'count<%s>' % ('table', 'new', 'block')[int(m2.group(1))]
    return base


def main():
    d = pd.read_csv(sys.argv[1])
    d = d[~d.Kernel_Name.str.contains('at::native')]
    d['k'] = d.Kernel_Name.map(short)
    d['dur'] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
    # last clustering call only: from the last encode_gather_kernel on
    t0 = d[d.k == 'encode_gather_kernel'].Start_Timestamp.max()
    d = d[d.Start_Timestamp >= t0].sort_values('Start_Timestamp')
    main_q = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != 'auto' else d[d.k == 'encode_gather_kernel'].Queue_Id.iloc[0]
    m = d[d.Queue_Id == main_q]
    s = d[d.Queue_Id != main_q]
    span = (d.End_Timestamp.max() - t0) / 1e3
    print('span %.1f ms; main stream busy %.1f ms in %d launches; side stream busy %.1f ms in %d launches'
          % (span / 1e3, m.dur.sum() / 1e3, len(m), s.dur.sum() / 1e3, len(s)))
    g = m.groupby('k').dur.agg(['sum', 'count', 'mean', 'max']).sort_values('sum', ascending=False)
    g['sum'] /= 1e3
    print('main stream (sum ms, launches, mean us, max us):')
    print(g.to_string(float_format=lambda x: '%.1f' % x))
    g2 = s.groupby('k').dur.agg(['sum', 'count', 'mean', 'max']).sort_values('sum', ascending=False)
    g2['sum'] /= 1e3
    print('side stream:')
    print(g2.to_string(float_format=lambda x: '%.1f' % x))
    # gaps on the main stream, attributed to the kernel that follows
    prev_end = m.End_Timestamp.shift(1)
    gap = ((m.Start_Timestamp - prev_end) / 1e3).clip(lower=0)
    m = m.assign(gap=gap)
    gg = m.groupby('k').gap.agg(['sum', 'count', 'mean']).sort_values('sum', ascending=False)
    gg['sum'] /= 1e3
    print('idle before (sum ms, n, mean us):  total %.1f ms' % (gap.sum() / 1e3))
    print(gg.head(12).to_string(float_format=lambda x: '%.1f' % x))
    # does a sweep have to wait for its head (index + table pass, side stream, started one sweep earlier)?
    init = m[m.k == 'sweep_init_kernel']
    heads = s[s.k == 'count<table>']
    if len(init) > 2 and len(heads) > 2:
        import numpy as np
        t_init = init.End_Timestamp.values
        nxt = m[m.k == 'count<table>'].Start_Timestamp.values            # catch-up pass = first kernel after the wait
        head_end = heads.End_Timestamp.values
        late, gaps = [], []
        for i, t in enumerate(t_init):
            j = np.searchsorted(nxt, t)
            if j >= len(nxt):
                break
            gaps.append((nxt[j] - t) / 1e3)
            h = np.searchsorted(head_end, nxt[j], side='right') - 1     # last head that ended before the catch-up started
            late.append((head_end[h] - t) / 1e3 if h >= 0 else 0.0)
        gaps, late = np.array(gaps), np.array(late)
        print('sweep start: idle between sweep_init and the catch-up pass: mean %.0f us, median %.0f, p90 %.0f; '
              'sweeps whose head finished after sweep_init: %d of %d (mean lateness of those %.0f us)'
              % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90), (late > 0).sum(), len(late),
                 late[late > 0].mean() if (late > 0).any() else 0.0))
    if len(sys.argv) > 3:   # dump the launches of sweeps [N, N + 1): from the N-th sweep_init_kernel on
        n_sw = int(sys.argv[3])
        starts = d[d.k == 'sweep_init_kernel'].Start_Timestamp.values
        a, b = starts[n_sw], starts[n_sw + 2]
        w = d[(d.Start_Timestamp >= a - 200000) & (d.Start_Timestamp < b)]
        for _, r in w.iterrows():
            print('%9.1f %9.1f %s %-28s %7.1f us  wg %d' % ((r.Start_Timestamp - a) / 1e3, (r.End_Timestamp - a) / 1e3,
                  'M' if r.Queue_Id == main_q else 's', r.k, r.dur, r.Grid_Size_X // r.Workgroup_Size_X))


if __name__ == '__main__':
    main()
