"""Summarises a rocprofv3 --kernel-trace CSV: per-kernel durations, and for the raster kernel the gap between
the end of one dispatch and the start of the next (any stream) plus how much consecutive dispatches overlap.
usage: trace_timeline.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
def col(r, *names):
    for n in names:
        if n in r: return r[n]
    raise KeyError(names)
ev = []
for r in rows:
    name = col(r, "Kernel_Name", "kernel_name")
    s, e = int(col(r, "Start_Timestamp", "start_timestamp")), int(col(r, "End_Timestamp", "end_timestamp"))
    q = col(r, "Queue_Id", "queue_id")
    ev.append((s, e, "raster" if "raster_kernel" in name else ("geometry" if "geometry_kernel" in name else ("vertex" if "vertex_kernel" in name else name[:30])), q))
ev.sort()
skip = len(ev) // 5          # drop warm-up
ev = ev[skip:]
for k in ("geometry", "raster", "vertex"):
    d = [(e - s) / 1000 for s, e, n, q in ev if n == k]
    if d: print(f"{k:9s} n={len(d):5d} dur us mean {st.mean(d):7.2f} p50 {st.median(d):7.2f} min {min(d):7.2f} max {max(d):7.2f}")
ras = [(s, e, q) for s, e, n, q in ev if n == "raster"]
gaps = [(ras[i + 1][0] - ras[i][1]) / 1000 for i in range(len(ras) - 1)]
period = [(ras[i + 1][0] - ras[i][0]) / 1000 for i in range(len(ras) - 1)]
print(f"raster start-to-start period us mean {st.mean(period):.2f} p50 {st.median(period):.2f}")
print(f"raster end -> next raster start us mean {st.mean(gaps):.2f} p50 {st.median(gaps):.2f} (negative = overlap)")
print("queues:", sorted(set(q for _, _, _, q in ev)))
# busy fraction: union of all kernel intervals / span
iv = sorted((s, e) for s, e, n, q in ev)
span = iv[-1][1] - iv[0][0]; busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"GPU has at least one kernel running {100 * busy / span:.1f} % of the time; span {span / 1000:.1f} us for {len(ras)} frames -> {span / 1000 / max(1, len(ras)):.2f} us/frame")
for i in range(40, 52):
    if i < len(ev): print(f"  t={(ev[i][0] - ev[40][0]) / 1000:8.2f} dur {(ev[i][1] - ev[i][0]) / 1000:6.2f} {ev[i][2]:9s} q{ev[i][3]}")
