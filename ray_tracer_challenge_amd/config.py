"""Scene-level render settings (`add: config` YAML node; reference config.py:1-27).

`thread_count` is the reference's worker-process count; here it is the upper bound on the
number of GPUs `render_multi` shards image tiles over.
"""


class GlobalConfig(object):
    def __init__(self, thread_count=4, timeout=30, divide_threshold=1, clamping=False):
        self.thread_count = thread_count
        self.timeout = timeout
        self.divide_threshold = divide_threshold
        self.clamping = clamping

    @classmethod
    def from_yaml(cls, obj):
        return cls(thread_count=int(obj.get("thread-count", 4)),
                   timeout=int(obj.get("timeout", 30)),
                   divide_threshold=int(obj.get("divide-threshold", 1)),
                   clamping=obj.get("clamping", False))
