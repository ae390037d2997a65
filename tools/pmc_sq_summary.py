"""Per-kernel sums of rocprofv3 --pmc counter_collection CSVs (any counters), as a small table.
Usage: python tools/pmc_sq_summary.py <dir> [<dir> ...]   (run where the big CSVs are; keep the table)"""
import glob
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(__file__))
from timeline import short  # noqa: E402


def main():
    pd.set_option('display.width', 250)
    pd.set_option('display.max_columns', 40)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            df = pd.read_csv(f, usecols=['Kernel_Name', 'Counter_Name', 'Counter_Value'])
            df = df[~df.Kernel_Name.str.contains('at::native')]
            df['k'] = df.Kernel_Name.map(short)
            t = df.pivot_table(index='k', columns='Counter_Name', values='Counter_Value', aggfunc='sum')
            t['launches'] = df[df.Counter_Name == df.Counter_Name.iloc[0]].groupby('k').size()
            print(t.sort_values(t.columns[0], ascending=False).to_string(float_format=lambda x: '%.4g' % x))
            print()


if __name__ == '__main__':
    main()
