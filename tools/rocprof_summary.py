# -*- coding: utf-8 -*-
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a per-kernel table.

    python tools/rocprof_summary.py <results.db> [top] [--steady N]

``--steady N`` keeps only the dispatches after the (N+1)-th last launch of bk_main, i.e. the last N
steps of bench.py (its timed region): warm-up work -- MIOpen's first-call naive_conv_* fallbacks and
find trials -- is then excluded from the table."""
import sqlite3
import sys


def main(path, top=40, steady=0):
    db = sqlite3.connect(path)
    where = ''
    if steady:
        marks = db.execute("select end from kernels where name like '%bk_main%' order by start").fetchall()
        if len(marks) > steady:
            where = ' where start >= %d' % marks[-(steady + 1)][0]
            print('steady-state window: dispatches after launch #%d of bk_main (last %d steps)\n' % (len(marks) - steady, steady))
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels%s group by name order by 3 desc" % where).fetchall()
    total = sum(r[2] for r in rows)
    print('| kernel | calls | total us | avg us | min us | max us | % |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for name, n, tot, avg, mn, mx in rows[:top]:
        print('| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |' % (name[:90], n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                  100.0 * tot / total))
    ours = [r for r in rows if 'rmnet' in r[0]]
    if ours:
        print('\nhand-written kernels (all of them):\n')
        print('| kernel | calls | total us | avg us | min us | max us | % |')
        print('|---|---:|---:|---:|---:|---:|---:|')
        for name, n, tot, avg, mn, mx in ours:
            print('| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.2f |' % (name[:90], n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                      100.0 * tot / total))
    print('\ntotal kernel time %.1f us over %d dispatches (%d distinct kernels)' %
          (total / 1e3, sum(r[1] for r in rows), len(rows)))


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    steady = int(sys.argv[sys.argv.index('--steady') + 1]) if '--steady' in sys.argv else 0
    if '--steady' in sys.argv:
        args = [a for a in args if a != str(steady)] if len(args) > 1 and args[-1] == str(steady) else args
    main(args[0], int(args[1]) if len(args) > 1 else 40, steady)
