#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) result into the text summary committed under profiles/.
usage: tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    lines = [f'# rocprofv3 --kernel-trace --stats summary of {db}',
             f'# {"kernel":90s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"pct":>6s}']
    for name, calls, tot, avg, pct in rows:
        lines.append(f'{name[:92]:92s} {calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}')
    txt = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main()
