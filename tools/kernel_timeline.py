"""Timeline of the last gpv_verify_dev call in a rocprofv3 --kernel-trace database: start / end of every kernel relative to the first, with its queue.
    rocprofv3 --kernel-trace -d out -o t -- python tools/one_size_probe.py 1024 2 ; python tools/kernel_timeline.py out/.../t_results.db [n_last_kernels]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, stream_id, grid_x, grid_y, vgpr_count, accum_vgpr_count from kernels order by start").fetchall()
# the last call = everything from the last k_range_check on
last = max(i for i, r in enumerate(rows) if r[0].startswith("k_range_check"))
first_t = min(r[1] for r in rows[last - 2:last + 1] if True)
sel = [r for r in rows[max(0, last - 3):] ]
t0 = min(r[1] for r in sel)
print("%-34s %9s %9s %8s  %5s %6s %8s %6s %5s" % ("kernel", "start_us", "end_us", "dur_us", "queue", "stream", "grid_x", "grid_y", "vgpr"))
for name, s, e, q, st, gx, gy, vg, ag in sel:
    print("%-34s %9.1f %9.1f %8.1f  %5s %6s %8d %6d %5d" % (name.split("(")[0][:34], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, gx, gy, vg + ag))
