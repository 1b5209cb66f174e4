#!/usr/bin/env python3
"""tools/trace_timeline.py DB [title] -- where the device is IDLE inside the steady steps of a bench run, from a rocprofv3 kernel trace (rocpd SQLite, `rocprofv3 --kernel-trace`).

A step of a queue of samples is the device's (bench.py: kernel_ms_alone_sum 1.47 s of a 1.81 s step at 10^8 fragments): what is the rest?  The union of the kernel intervals says
how long at least one kernel ran; the gaps say when none did -- a read-back the host waited for, a launch that came late, a copy.  Printed: busy / idle time of the last steps,
concurrency (time with two and more kernels in flight), and the longest idle gaps with the kernel that ended in front of each and the one that started behind it."""
import sqlite3
import sys


def short(name):
    for prefix in ("_ZN12_GLOBAL__N_1",):
        if name.startswith(prefix):
            name = name[len(prefix):].lstrip("0123456789")
    return name.split("EN4agpu")[0].split("EPK")[0].split("Ej")[0].split("Em")[0][:56]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select d.start, d.end, s.kernel_name, d.queue_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    if not rows:
        print("no dispatches")
        return
    print("# %s" % (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]))
    # the steady part: from the start of the third-last launch of fragment_pack_kernel (one per sample) to the start of the last one = two whole steps of the queue
    packs = [start for start, end, name, queue in rows if "fragment_pack_kernel" in name]
    if len(packs) >= 4:
        window = (packs[-4], packs[-2])
        what = "two steps of the queue (between the third-last and the last-but-one fragment_pack_kernel)"
        steps = 2
    else:
        window = (rows[0][0], rows[-1][1])
        what = "the whole trace"
        steps = 1
    inside = [(max(start, window[0]), min(end, window[1]), name, queue) for start, end, name, queue in rows if end > window[0] and start < window[1]]
    events = []
    for start, end, name, queue in inside:
        events.append((start, 1)); events.append((end, -1))
    events.sort()
    busy = multi = 0
    depth, last = 0, window[0]
    for at, change in events:
        if depth >= 1:
            busy += at - last
        if depth >= 2:
            multi += at - last
        depth += change
        last = at
    span = window[1] - window[0]
    kernel_sum = sum(end - start for start, end, _, _ in inside)
    print("# %s: %.1f ms per step; a kernel running %.1f ms (%.1f %%), two or more %.1f ms, NO kernel %.1f ms per step; sum of the kernel times %.1f ms per step; %d dispatches per step"
          % (what, span / steps / 1e6, busy / steps / 1e6, 100.0 * busy / span, multi / steps / 1e6, (span - busy) / steps / 1e6, kernel_sum / steps / 1e6, len(inside) // steps))
    # the idle gaps
    merged = []
    for start, end, name, queue in sorted(inside):
        if merged and start <= merged[-1][1]:
            if end > merged[-1][1]:
                merged[-1] = (merged[-1][0], end, merged[-1][2], name)
        else:
            merged.append((start, end, name, name))
    gaps = []
    for (s0, e0, first0, last0), (s1, e1, first1, last1) in zip(merged, merged[1:]):
        gaps.append((s1 - e0, last0, first1))
    by_pair = {}
    for gap, before, after in gaps:
        key = (short(before), short(after))
        entry = by_pair.setdefault(key, [0, 0])
        entry[0] += gap; entry[1] += 1
    print("%-58s %-58s %10s %7s" % ("# idle behind", "in front of", "ms/step", "gaps"))
    for (before, after), (total, count) in sorted(by_pair.items(), key=lambda item: -item[1][0])[:40]:
        print("%-58s %-58s %10.2f %7d" % (before, after, total / steps / 1e6, count // steps))
    # per queue: how long each hardware queue had a kernel running
    queues = {}
    for start, end, name, queue in inside:
        queues.setdefault(queue, []).append((start, end))
    for queue, intervals in sorted(queues.items()):
        intervals.sort()
        total, current_end = 0, None
        for start, end in intervals:
            if current_end is None or start > current_end:
                total += end - start; current_end = end
            elif end > current_end:
                total += end - current_end; current_end = end
        print("# queue %s: %.1f ms per step busy, %d dispatches per step" % (queue, total / steps / 1e6, len(intervals) // steps))


if __name__ == "__main__":
    main()
