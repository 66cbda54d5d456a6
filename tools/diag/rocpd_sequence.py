"""Per-kernel timeline of the LAST graph replay in a rocprofv3 rocpd database (ROCm 7's default --kernel-trace output):
start offset, duration and the gap to the previous kernel's end, in launch order.
  python tools/diag/rocpd_sequence.py gpurun_out/prof_api/api_results.db [kernels per replay]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = list(cur.execute("select name, start, end from kernels order by start"))
per = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0
def short(n):
    n = re.sub(r"\(.*\)$", "", n)
    n = n.replace("pdm::", "")
    return n[:70]
if not per:   # the repeating unit: distance between the last two occurrences of the last kernel name pattern
    names = [r[0] for r in rows]
    # find the period by matching the tail
    for cand in range(8, 200):
        if names[-cand:] == names[-2 * cand:-cand] and names[-cand:] == names[-3 * cand:-2 * cand]:
            per = cand
            break
print(f"{len(rows)} kernel dispatches, {per} per replay")
if "--timeline" in sys.argv:     # the LAST replay as it ran: start and end offsets (kernels of concurrent branches overlap)
    blk = rows[len(rows) - per:]
    t0 = blk[0][1]
    for n, s, e in blk:
        print(f"{(s - t0) / 1e3:9.2f} -> {(e - t0) / 1e3:9.2f}  ({(e - s) / 1e3:7.2f} us)  {short(n)}")
    print(f"span {(max(e for _, _, e in blk) - t0) / 1e3:.1f} us")
    sys.exit(0)
import collections
R = 20   # average over the last R replays
acc = collections.OrderedDict()
tot_dur = tot_gap = 0.0
span = 0.0
for rep in range(R):
    blk = rows[len(rows) - (rep + 1) * per: len(rows) - rep * per]
    span += (blk[-1][2] - blk[0][1]) / 1e3
    prev_end = None
    for i, (n, s, e) in enumerate(blk):
        d = acc.setdefault(i, [short(n), 0.0, 0.0])
        d[1] += (e - s) / 1e3
        if prev_end is not None:
            d[2] += (s - prev_end) / 1e3
        prev_end = e
print(f"{'#':>3} {'kernel':70s} {'dur us':>8} {'gap before us':>14}")
for i, (n, dur, gap) in acc.items():
    print(f"{i:3d} {n:70s} {dur / R:8.2f} {gap / R:14.2f}")
    tot_dur += dur / R; tot_gap += gap / R
print(f"sum of kernel durations {tot_dur:.1f} us + gaps {tot_gap:.1f} us = {tot_dur + tot_gap:.1f} us; first start -> last end {span / R:.1f} us")
