"""fp32 gate over the random worlds of tests/test_gpu_fuzz.py: share of pixels more than one 8-bit level from the oracle, world by world.
Usage: fp32_dist.py pytest-log [...]   (logs of `pytest tests/test_gpu_fuzz.py -s -k test_random_world_matches_oracle`)"""
import re, sys
import numpy as np
for path in sys.argv[1:]:
    v = []
    for line in open(path):
        m = re.search(r"seed (\d+): fp64 off (\d+), ids (\d+), fp32 off>1 ([\d.]+), ids32 (\d+) of (\d+)", line)
        if m:
            v.append((int(m[1]), float(m[4]), int(m[5])))
    v = np.array(v)
    f = v[:, 1]
    print("%s: %d worlds | share within 0.001: %.3f, 0.002: %.3f, 0.005: %.3f | worst %s" % (
        path, len(v), (f <= 0.001).mean(), (f <= 0.002).mean(), (f <= 0.005).mean(),
        ", ".join("%d:%.4f" % (a, b) for a, b, _ in v[np.argsort(-f)][:12])))
