"""Per-kernel counter values of the last frame in a rocprofv3 --pmc database (development aid): pmc_dump.py db [n_last]"""
import re, sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select dispatch_id, kernel_name, counter_name, value, grid_size from counters_collection order by dispatch_id"))
by = defaultdict(dict)
names = {}
for d, k, cn, v, g in rows:
    by[d][cn] = by[d].get(cn, 0) + v
    m = re.search(r'((trace|shadow|resolve|redo)_kernel<[^>]*>|quantize_kernel)', k)
    names[d] = ((m.group(1) if m else k)[:34], g)
keys = sorted({cn for d in by for cn in by[d]})
print("%-36s %9s " % ("kernel", "grid") + " ".join("%14s" % k[-14:] for k in keys))
sel = [d for d in sorted(by) if '_kernel<' in names[d][0]]
for d in sel[-int(sys.argv[2]) if len(sys.argv) > 2 else 0:]:
    print("%-36s %9d " % names[d] + " ".join("%14.0f" % by[d].get(k, 0) for k in keys))
