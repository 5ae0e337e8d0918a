"""Relative-position bucket tables (host side, integer, bit-exact).

Restates T5's `_relative_position_bucket` (transformers models/t5/modeling_t5.py; called from the reference at
VL-T5/src/modeling_t5_our.py:258 for the encoder and inside the decoder's first block) with numpy float32
arithmetic in the same operation order, so the truncation of the log-spaced buckets lands on the same side.
The tables depend only on (query_len, key_len), never on weights, so they are built once per shape on the
host and handed to `vlt5_relbias_build` / `vlt5_relbias_bwd` as an int32 lookup table.
"""
import math
from functools import lru_cache

import numpy as np


def relative_position_bucket(rel: np.ndarray, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> np.ndarray:
    rel = rel.astype(np.int64)
    out = np.zeros_like(rel)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        out = out + (rel > 0).astype(np.int64) * nb
        n = np.abs(rel)
    else:
        n = -np.minimum(rel, 0)
    max_exact = nb // 2
    small = n < max_exact
    with np.errstate(divide="ignore"):
        ratio = n.astype(np.float32) / np.float32(max_exact)
        val = np.log(ratio).astype(np.float32) / np.float32(math.log(max_distance / max_exact))
        val = (val * np.float32(nb - max_exact)).astype(np.float32)
    val = np.where(np.isfinite(val), val, np.float32(0))
    large = max_exact + val.astype(np.int64)
    large = np.minimum(large, nb - 1)
    return out + np.where(small, n, large)


@lru_cache(maxsize=64)
def bucket_table(qlen: int, klen: int, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> np.ndarray:
    q = np.arange(qlen, dtype=np.int64)[:, None]
    k = np.arange(klen, dtype=np.int64)[None, :]
    return relative_position_bucket(k - q, bidirectional, num_buckets, max_distance).astype(np.int32)
