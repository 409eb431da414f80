"""Are there scratch loads / stores inside the BVH walks?  For every node fetch of a trace_kernel variant (the inline-asm scalar
load marks it) the three innermost enclosing loops of the device assembly and the scratch accesses inside them.
Usage: spill_sites.py [mangled-variant-substring [unit]]   (after tools/kres.py, which leaves the assembly of the two translation
units in $TMPDIR/rtc_k_rtc_api.s and $TMPDIR/rtc_k_rtc_trace_f64.s; unit = rtc_api | rtc_trace_f64, default the fp64 one)"""
import os, re, sys, tempfile
variant = sys.argv[1] if len(sys.argv) > 1 else "trace_kernelIdLb0ELi0ELb0ELb1ELi4E"
unit = sys.argv[2] if len(sys.argv) > 2 else "rtc_trace_f64"
s = open(os.path.join(tempfile.gettempdir(), "rtc_k_%s.s" % unit)).read().split("\n")
a = next(i for i, l in enumerate(s) if re.match(r"^_ZN.*%s.*:\s" % variant, l))
b = next(i for i in range(a, len(s)) if "s_endpgm" in s[i])
k = s[a:b]
print("%s: %d lines, %d scratch_load and %d scratch_store instructions in all" % (
    variant, len(k), sum("scratch_load" in l for l in k), sum("scratch_store" in l for l in k)))
label = {m.group(1): i for i, l in enumerate(k) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = {}
for i, l in enumerate(k):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in label and label[m.group(1)] < i:
        loops[label[m.group(1)]] = max(loops.get(label[m.group(1)], 0), i)
for site in [i for i, l in enumerate(k) if "s_load_dwordx16" in l and "ASM" in k[i - 1]]:
    around = sorted([lp for lp in loops.items() if lp[0] <= site <= lp[1]], key=lambda x: x[1] - x[0])[:3]
    print("  node fetch at line %5d: " % site + " | ".join("loop of %4d lines: %d loads, %d stores" % (
        e - h + 1, sum("scratch_load" in l for l in k[h:e + 1]), sum("scratch_store" in l for l in k[h:e + 1])) for h, e in around))
