#!/usr/bin/env python3
"""tools/rocprof_summary.py -- per-kernel statistics from a rocprofv3 results database (rocpd SQLite), the same numbers
`rocprofv3 --kernel-trace --stats` prints, as a text table for profiles/."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("""select s.kernel_name, count(*), sum(d.end - d.start) / 1e6, avg(d.end - d.start) / 1e3, min(d.end - d.start) / 1e3, max(d.end - d.start) / 1e3,
                                max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size)
                         from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows)
    print("# %s" % (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]))
    print("# total kernel time %.3f ms over %d dispatches" % (total, sum(r[1] for r in rows)))
    print("%-72s %6s %11s %11s %11s %11s %6s %5s %5s %7s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "scr_B"))
    for name, calls, total_ms, avg, low, high, vgpr, sgpr, lds, scratch in rows:
        short = name
        for prefix in ("_ZN12_GLOBAL__N_1",):
            if short.startswith(prefix):
                short = short[len(prefix):].lstrip("0123456789")
        short = short.split("EN4agpu")[0].split("EPK")[0].split("Ej")[0].split("Em")[0]
        print("%-72s %6d %11.3f %11.1f %11.1f %11.1f %5.1f%% %5s %5s %7s %7s" % (short[:72], calls, total_ms, avg, low, high, 100 * total_ms / total, vgpr, sgpr, lds, scratch))


if __name__ == "__main__":
    main()
