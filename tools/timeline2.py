import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = []
for name, a, b, q in db.execute("select name, start, end, queue_id from kernels").fetchall():
    ev.append((a, b, name.split("(")[0].replace("void ", "").replace("liodom_dev::", "")[:30], "q%s" % q))
try:
    for name, a, b in db.execute("select name, start, end from memory_copies").fetchall():
        ev.append((a, b, "COPY " + str(name)[:24], "dma"))
except Exception as ex:
    print("no memory copies:", ex)
ev.sort()
ev = ev[-n:]
t0 = ev[0][0]
for a, b, nm, q in ev:
    print("%-32s %-4s %9.2f -> %9.2f  (%6.2f)" % (nm, q, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))
