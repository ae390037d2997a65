"""Critical-path view of a rocprofv3 --kernel-trace CSV of one clustering run: per kernel, the
time it occupies on the main stream, and the idle gaps of that stream (host round trips,
launch latency). Usage: python tools/timeline.py <kernel_trace.csv> [main_queue_id|auto [window_to_dump]]"""
import re
import sys

import pandas as pd


def short(name):
    m = re.search(r'filter_kernel<(true|false), (true|false)>', name)
    if m:
        return 'filter<%s>' % ('new' if m.group(2) == 'true' else 'all')
    m = re.search(r'(\w+_kernel|\w+Kernel)\b', name)
    return m.group(1) if m else name[:40]


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
    if len(sys.argv) > 3:   # dump the launches of windows [N, N + 1): from the N-th window_init_kernel on
        n_sw = int(sys.argv[3])
        starts = d[d.k == 'window_init_kernel'].Start_Timestamp.values
        a, b = starts[n_sw], starts[n_sw + 2]
        w = d[(d.Start_Timestamp >= a - 200000) & (d.Start_Timestamp < b)]
        for _, r in w.iterrows():
            print('%9.1f %9.1f %s %-28s %7.1f us  wg %d' % ((r.Start_Timestamp - a) / 1e3, (r.End_Timestamp - a) / 1e3,
                  'M' if r.Queue_Id == main_q else 's', r.k, r.dur, r.Grid_Size_X // r.Workgroup_Size_X))


if __name__ == '__main__':
    main()
