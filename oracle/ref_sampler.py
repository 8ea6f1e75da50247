"""ORACLE (test infrastructure only — the product never imports this): CPU restatement of the sampler the
reference's runner calls at vllm_omni/worker/gpu_ar_model_runner.py:626-657, i.e. upstream vLLM v0.19
`Sampler` (third-party, absent from /root/reference; pinned `vllm==0.19.0`, docker/Dockerfile.rocm:1):

    penalties  (`apply_penalties`: seen tokens  logit > 0 ? logit / r : logit * r)
    temperature (logits / T;  T == 0 -> greedy argmax)
    `apply_top_k_top_p`: top-k masks logits < k-th largest (ties with it kept); top-p, on the ascending
        sort, masks positions whose cumulative probability is <= 1 - p and always keeps the last one —
        equivalently keeps an element while the mass ranked BEFORE it (descending) is < p
    `random_sample`: probs.div_(q).argmax(), q ~ Exp(1)

parity unpinned: the reference holds no golden vectors for sampling and its RNG stream (torch CUDA
generator) cannot be restated; the noise here is Philox4x32-10 (Salmon et al. 2011, "Parallel random
numbers: as easy as 1, 2, 3") keyed by the request seed, counter = (column, 0, offset_lo, offset_hi),
u = ((x0 >> 8) + 0.5) / 2^24, q = -ln(u).  Philox itself is pinned by the known-answer vectors of the
Random123 distribution (tests/test_oracle_golden.py).  Ties at the top-p cut VALUE are all kept (upstream
breaks them by sort position, i.e. arbitrarily).

Everything is float64 on purpose; `margins` tells a test how close each row came to a different answer.
"""
from __future__ import annotations

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: 4 uint32 (arrays broadcast), key: 2 uint32 -> 4 uint32 outputs (Random123 philox4x32-10)."""
    c = [np.asarray(x, dtype=np.uint64) & M32 for x in counter]
    k0, k1 = [np.asarray(x, dtype=np.uint64) & M32 for x in key]
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M32
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M32
        c = [n0, p1 & M32, n2, p0 & M32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return [x.astype(np.uint32) for x in c]


def uniform_noise(vocab: int, seed: int, offset: int) -> np.ndarray:
    """u_i in (0, 1), 24 bits, for columns 0..vocab-1 of one row."""
    seed, offset = int(seed) & (2**64 - 1), int(offset) & (2**64 - 1)
    x0 = philox4x32_10((np.arange(vocab, dtype=np.uint64), 0, offset & 0xFFFFFFFF, offset >> 32),
                       (seed & 0xFFFFFFFF, seed >> 32))[0]
    return ((x0 >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0


def sample_row(logits: np.ndarray, temperature: float, top_k: int = 0, top_p: float = 1.0, rep_penalty: float = 1.0,
               seen: np.ndarray | None = None, seed: int = 0, offset: int = 0):
    """One row.  logits: fp32 values as the device reads them.  Returns (id, margins) where margins holds
    `topp_gap` (|mass - top_p| of the decision closest to flipping, relative) and `race_ratio` (best / second
    score of the exponential race), so that a test can tell a genuine mismatch from a coin flip in the last ulp."""
    v = np.asarray(logits, dtype=np.float32).astype(np.float64)
    V = v.shape[0]
    if seen is not None and rep_penalty != 1.0:
        r = np.float32(rep_penalty).astype(np.float64)
        v = np.where(seen, np.where(v > 0, v / r, v * r), v)
        v = v.astype(np.float32).astype(np.float64)  # the device holds the penalised value in fp32
    margins = {"topp_gap": np.inf, "race_ratio": np.inf}
    if temperature <= 0:
        return int(np.argmax(v)), margins  # first index of the maximum
    inv_t = np.float64(np.float32(1.0) / np.float32(temperature))
    v = (v.astype(np.float32) * np.float32(inv_t)).astype(np.float64)
    if 0 < top_k < V:
        kth = np.partition(v, V - top_k)[V - top_k]
        v = np.where(v < kth, -np.inf, v)
    e = np.exp(v - v.max())
    p = e / e.sum()
    keep = p > 0
    if top_p < 1.0:
        order = np.argsort(-p, kind="stable")
        ps = p[order]
        before = np.cumsum(ps) - ps                      # mass ranked before each element
        # elements tied in value share the `before` of the first of them (ties are kept together)
        first_of_value = np.searchsorted(-ps, -ps, side="left")
        before = before[first_of_value]
        k_sorted = before < top_p
        keep &= np.zeros(V, dtype=bool) | np.isin(np.arange(V), order[k_sorted])
        margins["topp_gap"] = float(np.min(np.abs(before - top_p)) / max(top_p, 1e-30))
    u = uniform_noise(V, seed, offset)
    score = np.where(keep, p / -np.log(u), -np.inf)
    best = int(np.argmax(score))
    if keep.sum() > 1:
        s2 = np.partition(score, V - 2)[V - 2]
        margins["race_ratio"] = float(score[best] / s2) if s2 > 0 else np.inf
    return best, margins


def sample(logits, temperature, top_k=None, top_p=None, rep_penalty=None, seen_bitmap=None, seen_row=None, seed=None,
           offset=None):
    """Batched wrapper with the device ABI's argument meaning (mi_sample_tokens, include/vllm_omni_amd.h)."""
    logits = np.asarray(logits, dtype=np.float32)
    n, V = logits.shape
    ids, margins = [], []
    for r in range(n):
        seen = None
        if seen_bitmap is not None:
            words = np.asarray(seen_bitmap[(seen_row[r] if seen_row is not None else r)]).astype(np.int64) & 0xFFFFFFFF
            seen = ((words[np.arange(V) >> 5] >> (np.arange(V) & 31)) & 1).astype(bool)
        i, m = sample_row(logits[r], float(temperature[r]), int(top_k[r]) if top_k is not None else 0,
                          float(top_p[r]) if top_p is not None else 1.0,
                          float(rep_penalty[r]) if rep_penalty is not None else 1.0, seen,
                          int(seed[r]) if seed is not None else 0, int(offset[r]) if offset is not None else 0)
        ids.append(i)
        margins.append(m)
    return np.asarray(ids, dtype=np.int64), margins
