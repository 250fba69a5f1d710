"""Counter-based random numbers shared bit-for-bit by host (numpy) and device (HIP).

Every random tensor on the hot path (dropout masks, DropBlock centres, the
noise view, synthetic images/boxes/weights) is a pure function of
(seed, stream, element index), so a CPU checker and the gfx950 kernels draw the
same numbers without shipping tensors around, and a step is reproducible
regardless of launch geometry.  The device twin lives in csrc/odw_rng.h.

The reference draws these from torch's global generators
(modeling/dropblock/drop_block.py:42 on the CPU generator,
modeling/backbone/vgg16.py:153,178 on the device generator); parity runs inject
this generator into the reference instead (tests/golden/make_golden.py).
"""
import numpy as np

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)


def _mix(x):
    x = x ^ (x >> np.uint32(16))
    x = x * _M1
    x = x ^ (x >> np.uint32(15))
    x = x * _M2
    x = x ^ (x >> np.uint32(16))
    return x


def stream_key(seed, stream):
    """Two 32-bit keys from (seed, stream) -- splitmix64 finaliser, host side only."""
    z = (int(seed) * 0x9E3779B97F4A7C15 + (int(stream) + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 31
    return int(z & 0xFFFFFFFF), int(z >> 32)


def bits(seed, stream, n, offset=0):
    """uint32[n]: hash of element indices offset..offset+n-1 under (seed, stream)."""
    k0, k1 = stream_key(seed, stream)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(offset)).astype(np.uint32)
        h = _mix(idx ^ np.uint32(k0))
        h = _mix(h + np.uint32(k1))
    return h


def uniform(seed, stream, n, offset=0):
    """float32[n] in [0,1) with 24 random bits."""
    return (bits(seed, stream, n, offset) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def normal(seed, stream, n, offset=0):
    """float32[n] ~ N(0,1).  Box-Muller on the uniform pair (u[2k], u[2k+1]) of the stream:
    element 2k = r cos(t), element 2k+1 = r sin(t), r = sqrt(-2 ln(1-u[2k])), t = 2 pi u[2k+1]."""
    first = offset - (offset & 1)
    last = offset + n + ((offset + n) & 1)
    u = uniform(seed, stream, last - first, first)
    u1 = np.float32(1.0) - u[0::2]          # (0,1]
    t = np.float32(6.283185307179586) * u[1::2]
    r = np.sqrt(np.float32(-2.0) * np.log(u1), dtype=np.float32)
    z = np.empty(last - first, np.float32)
    z[0::2] = r * np.cos(t, dtype=np.float32)
    z[1::2] = r * np.sin(t, dtype=np.float32)
    return z[offset - first: offset - first + n]


class Streams(object):
    """Hands out consecutive stream ids in call order (one per logical random draw)."""

    def __init__(self, seed, first=0):
        self.seed = int(seed)
        self.next = int(first)

    def take(self):
        s = self.next
        self.next += 1
        return s
