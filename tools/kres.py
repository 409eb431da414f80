"""Register / spill / scratch figures of every trace_kernel variant, from the device asm (development aid)."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "ray_tracer_challenge_amd", "csrc")
# the two translation units of the library, each with the extra flags the Makefile gives it
UNITS = [("rtc_api.hip", []), ("rtc_trace_f64.hip", ["-mllvm", "-disable-machine-licm"])]
for unit, extra in UNITS:
    out = os.path.join(tempfile.gettempdir(), "rtc_k_%s.s" % unit.split(".")[0])
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "--cuda-device-only", "-S", "-o", out,
                    os.path.join(csrc, unit)] + extra + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    md = s[s.index("amdhsa.kernels"):]
    for blk in md.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        n = g("name")
        m = re.search(r"(trace_kernel|shadow_kernel|redo_kernel|resolve_kernel)I(\w+?)EEv", n)
        if m:
            print("%-15s %-10s" % m.groups(), "vgpr", g("vgpr_count"), "vspill", g("vgpr_spill_count"),
                  "sgpr", g("sgpr_count"), "sspill", g("sgpr_spill_count"), "lds", g("group_segment_fixed_size"), "scratch", g("private_segment_fixed_size"))
