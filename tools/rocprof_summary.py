# -*- coding: utf-8 -*-
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a per-kernel table.
    python tools/rocprof_summary.py gpurun_out/prof_a/r01a_results.db > profiles/<name>.md"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print('| kernel | calls | total us | avg us | min us | max us | % |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for name, n, tot, avg, mn, mx in rows[:top]:
        print('| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |' % (name[:90], n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                  100.0 * tot / total))
    print('\ntotal kernel time %.1f us over %d dispatches (%d distinct kernels)' %
          (total / 1e3, sum(r[1] for r in rows), len(rows)))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
