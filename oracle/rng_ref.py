"""Oracle-side restatement of the counter-based generator (TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's
cpu_baseline may import anything under oracle/).

The definition is the one in od_wscl_amd/csrc/odw_rng.h: element `i` of stream (seed, stream) is

    k0, k1 = low / high 32 bits of splitmix64-finalise(seed * 0x9E3779B97F4A7C15 + (stream + 1) * 0xBF58476D1CE4E5B9)
    h      = mix32(mix32(i ^ k0) + k1)          mix32 = the "lowbias32" finaliser (x ^= x>>16; x *= 0x7FEB352D;
    u      = (h >> 8) * 2^-24                           x ^= x>>15; x *= 0x846CA68B; x ^= x>>16)

written here with 64-bit integers and explicit masks (the product's host twin, od_wscl_amd/utils/rng.py, uses uint32
wrap-around) so that the checker does not share code with what it checks; tests/test_oracle_vs_reference.py pins the two
bit for bit."""
import numpy as np

_MASK32 = np.uint64(0xFFFFFFFF)
_MASK64 = 0xFFFFFFFFFFFFFFFF


def _mix32(x):
    x = x & _MASK32
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x7FEB352D)) & _MASK32
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846CA68B)) & _MASK32
    x = x ^ (x >> np.uint64(16))
    return x


def stream_key(seed, stream):
    z = (int(seed) * 0x9E3779B97F4A7C15 + (int(stream) + 1) * 0xBF58476D1CE4E5B9) & _MASK64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & _MASK64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & _MASK64
    z ^= z >> 31
    return z & 0xFFFFFFFF, z >> 32


def bits(seed, stream, n, offset=0):
    k0, k1 = stream_key(seed, stream)
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(offset)) & _MASK32
    h = _mix32(idx ^ np.uint64(k0))
    h = _mix32((h + np.uint64(k1)) & _MASK32)
    return h.astype(np.uint32)


def uniform(seed, stream, n, offset=0):
    return (bits(seed, stream, n, offset) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def normal(seed, stream, n, offset=0):
    """Box-Muller on the pairs (u[2k], u[2k+1]) of the stream, in fp32 like the device kernel."""
    first = offset - (offset & 1)
    last = offset + n + ((offset + n) & 1)
    u = uniform(seed, stream, last - first, first)
    radius = np.sqrt(np.float32(-2.0) * np.log(np.float32(1.0) - u[0::2]), dtype=np.float32)
    angle = np.float32(6.283185307179586) * u[1::2]
    z = np.empty(last - first, np.float32)
    z[0::2] = radius * np.cos(angle, dtype=np.float32)
    z[1::2] = radius * np.sin(angle, dtype=np.float32)
    return z[offset - first: offset - first + n]


class Streams(object):
    """Consecutive stream ids in call order (one per logical random draw)."""

    def __init__(self, seed, first=0):
        self.seed, self.next = int(seed), int(first)

    def take(self):
        s = self.next
        self.next += 1
        return s
