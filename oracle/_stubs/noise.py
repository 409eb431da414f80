"""Stand-in for the un-vendored PyPI `noise` package (1.2.2) so that the reference's `shapes.py`
imports in the build container (reference shapes.py:1496).  Only PerturbedPattern calls it
(shapes.py:1916-1920).

`pnoise3` below is a transcription of the package's PUBLISHED algorithm -- Ken Perlin's "improved
noise" (2002) as packaged by Casey Duncan: float32 arithmetic, Perlin's reference permutation, a
16-entry gradient table, octaves summed with persistence 0.5 / lacunarity 2 and divided by the sum
of amplitudes -- written independently of oracle/rtc_oracle.c's C restatement (numpy float32
scalars, tables built differently) so that tests/ can hold the two against each other.  It is NOT
the package: a fixture the reference renders through this file pins the reference's own
PerturbedPattern code (which point is perturbed, the z += 1.0 between the three calls, the
scale factor, nesting) and nothing about the noise function itself, whose parity stays UNPINNED.
TEST INFRASTRUCTURE ONLY.
"""
import numpy as np

_F = np.float32

_PERM_TEXT = """
151 160 137 91 90 15 131 13 201 95 96 53 194 233 7 225 140 36 103 30 69 142 8 99 37 240 21 10 23 190 6 148
247 120 234 75 0 26 197 62 94 252 219 203 117 35 11 32 57 177 33 88 237 149 56 87 174 20 125 136 171 168 68 175
74 165 71 134 139 48 27 166 77 146 158 231 83 111 229 122 60 211 133 230 220 105 92 41 55 46 245 40 244 102 143 54
65 25 63 161 1 216 80 73 209 76 132 187 208 89 18 169 200 196 135 130 116 188 159 86 164 100 109 198 173 186 3 64
52 217 226 250 124 123 5 202 38 147 118 126 255 82 85 212 207 206 59 227 47 16 58 17 182 189 28 42 223 183 170 213
119 248 152 2 44 154 163 70 221 153 101 155 167 43 172 9 129 22 39 253 19 98 108 110 79 113 224 232 178 185 112 104
218 246 97 228 251 34 242 193 238 210 144 12 191 179 162 241 81 51 145 235 249 14 239 107 49 192 214 31 181 199 106 157
184 84 204 176 115 121 50 45 127 4 150 254 138 236 205 93 222 114 67 29 24 72 243 141 128 195 78 66 215 61 156 180
"""
PERM = [int(t) for t in _PERM_TEXT.split()] * 2
assert sorted(PERM[:256]) == list(range(256))

# the twelve edge midpoints of a cube, then four of them again to make sixteen
GRAD3 = [(1, 1, 0), (-1, 1, 0), (1, -1, 0), (-1, -1, 0), (1, 0, 1), (-1, 0, 1), (1, 0, -1), (-1, 0, -1),
         (0, 1, 1), (0, -1, 1), (0, 1, -1), (0, -1, -1), (1, 0, -1), (-1, 0, -1), (0, -1, 1), (0, 1, 1)]


def _grad(h, x, y, z):
    g = GRAD3[h & 15]
    return x * _F(g[0]) + y * _F(g[1]) + z * _F(g[2])


def _mix(t, a, b):
    return a + t * (b - a)


def _fade(t):
    return t * t * t * (t * (t * _F(6) - _F(15)) + _F(10))


def _cell(v, repeat):
    """Lattice coordinate below v and the next one, both wrapped at `repeat` and masked to the table."""
    lo = int(np.floor(np.fmod(v, _F(repeat))))
    hi = int(np.fmod(_F(lo + 1), _F(repeat)))
    return lo & 255, hi & 255


def _noise3(x, y, z, repeat, base=0):
    x, y, z = _F(x), _F(y), _F(z)
    (i, ii), (j, jj), (k, kk) = _cell(x, repeat), _cell(y, repeat), _cell(z, repeat)
    i, ii, j, jj, k, kk = i + base, ii + base, j + base, jj + base, k + base, kk + base
    x, y, z = x - np.floor(x), y - np.floor(y), z - np.floor(z)
    one = _F(1)
    fx, fy, fz = _fade(x), _fade(y), _fade(z)
    a, b = PERM[i], PERM[ii]
    aa, ab, ba, bb = PERM[a + j], PERM[a + jj], PERM[b + j], PERM[b + jj]
    near = _mix(fy, _mix(fx, _grad(PERM[aa + k], x, y, z), _grad(PERM[ba + k], x - one, y, z)),
                _mix(fx, _grad(PERM[ab + k], x, y - one, z), _grad(PERM[bb + k], x - one, y - one, z)))
    far = _mix(fy, _mix(fx, _grad(PERM[aa + kk], x, y, z - one), _grad(PERM[ba + kk], x - one, y, z - one)),
               _mix(fx, _grad(PERM[ab + kk], x, y - one, z - one), _grad(PERM[bb + kk], x - one, y - one, z - one)))
    return _mix(fz, near, far)


def pnoise3(x, y, z, octaves=1, persistence=0.5, lacunarity=2.0, repeatx=1024, repeaty=1024, repeatz=1024, base=0):
    if not (repeatx == repeaty == repeatz):
        raise NotImplementedError("stand-in: one repeat for the three axes (the reference passes none)")
    x, y, z = _F(x), _F(y), _F(z)
    if octaves == 1:
        return float(_noise3(x, y, z, repeatx, base))
    if octaves < 1:
        raise ValueError("Expected octaves value > 0")
    freq, amp, top, total = _F(1), _F(1), _F(0), _F(0)
    for _ in range(int(octaves)):
        total = total + _noise3(x * freq, y * freq, z * freq, int(_F(repeatx) * freq), base) * amp
        top = top + amp
        freq = freq * _F(lacunarity)
        amp = amp * _F(persistence)
    return float(total / top)
