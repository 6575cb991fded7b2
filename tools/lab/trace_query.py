import sqlite3, glob, sys
f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
con = sqlite3.connect(f[0])
print("copies:")
for r in con.execute("select name, count(*), avg(duration)/1e3, avg(size)/1e6 from memory_copies group by name"): print("  ", r)
cols = [c[1] for c in con.execute("pragma table_info(kernels)")]
print(cols)
rows = con.execute("select name, start, end, grid_x, workgroup_x, stream_id from kernels order by start").fetchall() if "grid_x" in cols else con.execute("select name, start, end from kernels order by start").fetchall()
cp = con.execute("select name, start, end, size from memory_copies order by start").fetchall()
ev = [(r[1], r[2], r[0][:60].replace("void lws::(anonymous namespace)::", ""), r[3:]) for r in rows] + [(r[1], r[2], r[0], (r[3],)) for r in cp]
ev.sort()
# last call: the last 1/3 of events
t_last = ev[-1][1]
sel = [e for e in ev if e[0] > t_last - 75e6]
t0 = sel[0][0]
for s, e, n, x in sel: print("%8.2f %8.2f %7.2f  %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n, x))
