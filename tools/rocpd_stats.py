#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / share,
plus per (kernel, grid) rows for the conv kernels.  Usage: rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name "
                  "order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
out = ['# rocprofv3 --kernel-trace summary (%s)' % sys.argv[1].split('/')[-2], '',
       'total kernel time %.2f ms over %d dispatches' % (tot / 1e6, sum(r[1] for r in rows)), '',
       '| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds |', '|---|---|---|---|---|---|---|---|---|---|']
for r in rows[:60]:
    out.append('| %s | %d | %.2f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s |' % (
        r[0][:120], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8]))
host = [r for r in rows if ('at::' in r[0] or 'rocclr' in r[0] or 'Cijk' in r[0] or 'hipcub' in r[0] or 'rocprim' in r[0])]
out += ['', '## host-framework (ATen / runtime) kernels: %.2f ms total, %d dispatches' % (sum(r[2] for r in host) / 1e6, sum(r[1] for r in host)), '',
        '| kernel | calls | total ms | avg us |', '|---|---|---|---|']
for r in host[:25]:
    out.append('| %s | %d | %.2f | %.1f |' % (r[0][:150], r[1], r[2] / 1e6, r[3] / 1e3))
out += ['', '## conv kernels by launch geometry (top 40 by time)', '',
        '| kernel | grid | calls | total ms | avg us |', '|---|---|---|---|---|']
rows2 = db.execute("select name, grid_x, grid_y, grid_z, count(*), sum(duration), avg(duration) from kernels "
                   "where name like '%conv_%' group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 40").fetchall()
for r in rows2:
    out.append('| %s | %dx%dx%d | %d | %.2f | %.1f |' % (r[0][:100], r[1], r[2], r[3], r[4], r[5] / 1e6, r[6] / 1e3))
txt = '\n'.join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt + '\n')
print(txt)
