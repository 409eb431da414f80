"""Scene-level render settings (`add: config` YAML node; reference config.py:1-27).

`thread_count` is the reference's worker-process count; here it is the upper bound on the
number of GPUs `render_multi` shards image tiles over.
"""

# attribute, YAML key, default, conversion -- the whole schema
_FIELDS = (("thread_count", "thread-count", 4, int),
           ("timeout", "timeout", 30, int),
           ("divide_threshold", "divide-threshold", 1, int),
           ("clamping", "clamping", False, lambda v: v))


class GlobalConfig(object):
    def __init__(self, *args, **kwargs):
        names = [f[0] for f in _FIELDS]
        if len(args) > len(names):
            raise TypeError("GlobalConfig takes at most %d arguments" % len(names))
        given = dict(zip(names, args))
        for k, v in kwargs.items():
            if k not in names or k in given:
                raise TypeError("GlobalConfig: unexpected or repeated argument %r" % k)
            given[k] = v
        for attr, _, default, _ in _FIELDS:
            setattr(self, attr, given.get(attr, default))

    @classmethod
    def from_yaml(cls, obj):
        return cls(**{attr: conv(obj[key]) for attr, key, _, conv in _FIELDS if key in obj})

    def __repr__(self):
        return "GlobalConfig(%s)" % ", ".join("%s=%r" % (f[0], getattr(self, f[0])) for f in _FIELDS)
