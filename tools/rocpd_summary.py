#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) as a per-kernel stats table
(the same content as rocprofv3's kernel_stats.csv): calls, total/avg/min/max duration, share."""
import re
import sqlite3
import sys


def main(path, out=None, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = cur.execute(f"""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start),
        min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id
        group by s.kernel_name order by 3 desc""").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f'total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches', '',
             '| kernel | calls | total ms | % | avg us | min us | max us |', '|---|---|---|---|---|---|---|']
    for r in rows[:top]:
        name = re.sub(r'^_ZN4deva12_GLOBAL__N_1\d+', 'deva::', r[0])
        name = re.sub(r'\.kd$', '', name)[:110]
        lines.append(f'| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {100 * r[2] / tot:.1f} | {r[3] / 1e3:.1f} | '
                     f'{r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} |')
    text = '\n'.join(lines) + '\n'
    if out:
        with open(out, 'w') as f:
            f.write(text)
    else:
        sys.stdout.write(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
