import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-90s %5d %12.1f %10.2f %6.2f" % (name[:90], calls, total, avg, pct))
