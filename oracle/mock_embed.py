"""MockEmbedder restated in pure Python -- TEST INFRASTRUCTURE ONLY.

Follows src/embed.rs:124-155 of the reference with std's DefaultHasher taken to
be SipHash-1-3 with zero keys (published algorithm: Aumasson & Bernstein; std
does not promise it). PARITY UNPINNED: the reference's tests
(src/embed.rs:992-1036) pin length, unit norm, determinism and text sensitivity
only, never a value, and std's hasher cannot be run here. Written
independently of the C++ in the product so that the two check each other.
"""
import numpy as np

M = (1 << 64) - 1


def _rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & M


def _round(v):
    v0, v1, v2, v3 = v
    v0 = (v0 + v1) & M; v1 = _rotl(v1, 13); v1 ^= v0; v0 = _rotl(v0, 32)
    v2 = (v2 + v3) & M; v3 = _rotl(v3, 16); v3 ^= v2
    v0 = (v0 + v3) & M; v3 = _rotl(v3, 21); v3 ^= v0
    v2 = (v2 + v1) & M; v1 = _rotl(v1, 17); v1 ^= v2; v2 = _rotl(v2, 32)
    return [v0, v1, v2, v3]


def siphash13(data: bytes, k0=0, k1=0) -> int:
    """One-shot SipHash-1-3 of a byte string."""
    v = [k0 ^ 0x736f6d6570736575, k1 ^ 0x646f72616e646f6d, k0 ^ 0x6c7967656e657261, k1 ^ 0x7465646279746573]
    n = len(data)
    for off in range(0, n - n % 8, 8):
        m = int.from_bytes(data[off:off + 8], "little")
        v[3] ^= m
        v = _round(v)
        v[0] ^= m
    b = ((n & 0xFF) << 56) | int.from_bytes(data[n - n % 8:], "little")
    v[3] ^= b
    v = _round(v)
    v[0] ^= b
    v[2] ^= 0xFF
    for _ in range(3):
        v = _round(v)
    return v[0] ^ v[1] ^ v[2] ^ v[3]


def hash_to_vector(text: str, dim: int, normalize=True) -> np.ndarray:
    """The hasher is never reset: value i is the hash of the whole stream so far."""
    stream = b""
    out = np.empty(dim, dtype=np.float32)
    raw = text.encode("utf-8")
    two64 = np.float32(18446744073709551616.0)
    for i in range(dim):
        stream += raw + b"\xff" + i.to_bytes(8, "little")
        h = siphash13(stream)
        out[i] = (np.float32(h) / two64) * np.float32(2.0) - np.float32(1.0)
    if normalize:
        s = np.float32(0.0)
        for x in out:
            s = np.float32(s + np.float32(x * x))
        norm = np.sqrt(s, dtype=np.float32)
        if norm > 0:
            out = (out / norm).astype(np.float32)
    return out
