"""Per-step timeline of a hipGraph bench run from a rocprofv3 --kernel-trace CSV: for the last replay, every kernel
with its start offset, duration and stream/queue, so the overlap of the branches can be read off.
usage: python tools/diag/timeline.py <kernel_trace.csv> [nsteps_back]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = from one copy_many (hand-over) group to the next: use the LAST fps_pruned launch as the anchor
grid = lambda r: int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
fps = [i for i, r in enumerate(rows) if "fps_pruned" in r["Kernel_Name"]]
big = max(grid(rows[i]) for i in fps)   # the pipelined (graph) steps launch the segments of several batches at once
anchors = [i for i in fps if grid(rows[i]) == big]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a0, a1 = anchors[-back - 1], anchors[-back]
t0 = int(rows[a0]["Start_Timestamp"])
print(f"step length {(int(rows[a1]['Start_Timestamp']) - t0) / 1e3:.1f} us")
for r in rows[a0:a1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0][:70]
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3} wg{r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>5} grid{r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {name}")

first = {}
last = {}
for r in rows[a0:a1]:
    q = r.get("Queue_Id", "?")
    first.setdefault(q, int(r["Start_Timestamp"]) - t0)
    last[q] = int(r["End_Timestamp"]) - t0
print("queue first-start / last-end (us):", {q: (round(first[q] / 1e3, 1), round(last[q] / 1e3, 1)) for q in first})
