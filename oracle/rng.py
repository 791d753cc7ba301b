"""Oracle: the engine's counter-based dropout RNG, restated in numpy (TEST INFRASTRUCTURE).

The reference uses torch's Philox stream for its 5 dropout sites (SASRec/main.py:38,41,80,100);
a fused kernel cannot reproduce that stream (SURVEY.md §7 "Dropout/RNG"), so training-mode
equivalence with the reference is statistical.  What CAN be checked exactly is that the HIP
kernels apply *their own* documented mask consistently in forward and backward: this file
restates that mask generator (recboard_amd/csrc/re_rng.h) so the oracle can be run with the
very same keep-masks.

    keep(seed, stream, idx) = fmix32(idx + stream*0x85EBCA77 + seed) >= floor(p * 2^32)
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def rng_u32(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    idx = np.asarray(idx).astype(np.uint64)
    h = (idx + np.uint64((stream * 0x85EBCA77) & 0xFFFFFFFF) + np.uint64(seed & 0xFFFFFFFF)) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def drop_threshold(p: float) -> int:
    return int(min(float(np.float32(p)) * 4294967296.0, 4294967295.0))  # p is a C float in the ABI


def keep_mask(seed: int, stream: int, shape, p: float) -> np.ndarray:
    """Boolean keep-mask for a tensor of `shape`, element id = C-order flat index."""
    n = int(np.prod(shape))
    if p <= 0.0:
        return np.ones(shape, bool)
    h = rng_u32(seed, stream, np.arange(n, dtype=np.uint64))
    return (h >= np.uint32(drop_threshold(p))).reshape(shape)


# stream ids (must match recboard_amd/csrc/re_rng.h)
STREAM_EMBED = 1


def stream_attn(l: int) -> int:
    return 16 * l + 2


def stream_ffn1(l: int) -> int:
    return 16 * l + 3


def stream_ffn2(l: int) -> int:
    return 16 * l + 4
