"""Details of the rows where tests/test_gpu_sampler.py::test_sampler_random_sweep disagrees with the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-omni_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref_sampler  # noqa: E402
from test_gpu_sampler import _run  # noqa: E402

dev = torch.device("cuda:0")
for seed in [int(x) for x in (sys.argv[1:] or ["101", "103", "105", "108", "122", "137"])]:
    rng = np.random.default_rng(4000 + seed)
    vocab = int(rng.choice([33, 64, 65, 255, 1000, 2048, 3072, 4097, 8448, 15744]))
    n = 48
    scale = rng.choice([0.05, 1.0, 3.0, 12.0], size=(n, 1))
    x = rng.standard_normal((n, vocab)) * scale
    grid = rng.random(n) < 0.3
    x[grid] = np.round(x[grid] * 2) / 2
    logits = torch.from_numpy(x.astype(np.float32))
    bf = rng.random() < 0.5
    if bf:
        logits = logits.to(torch.bfloat16)
    lf = logits.float().numpy()
    temperature = rng.choice([0.0, 0.3, 0.9, 1.0, 2.0], size=n).astype(np.float32)
    top_k = rng.choice([0, 1, 2, 50, 64, 65, 300, vocab - 1, vocab, vocab + 7], size=n).astype(np.int32)
    top_p = rng.choice([1.0, 0.999, 0.8, 0.5, 0.05], size=n).astype(np.float32)
    seed_v = rng.integers(0, 2**63, size=n, dtype=np.int64)
    offset = rng.integers(0, 2**40, size=n, dtype=np.int64)
    got, _ = _run(dev, logits, temperature, top_k, top_p, None, None, None, seed_v, offset)
    ref, margins = ref_sampler.sample(lf, temperature, top_k, top_p, None, None, None, seed_v, offset)
    for r in range(n):
        if got[r] == ref[r]:
            continue
        v = lf[r].astype(np.float64)
        order = np.argsort(-v, kind="stable")
        rank = {int(t): i for i, t in enumerate(order)}
        # same row alone, and with neutral top-k / top-p, to see which stage disagrees
        alone, _ = _run(dev, logits[r:r + 1], temperature[r:r + 1], top_k[r:r + 1], top_p[r:r + 1], None, None, None, seed_v[r:r + 1], offset[r:r + 1])
        nok, _ = _run(dev, logits[r:r + 1], temperature[r:r + 1], None, top_p[r:r + 1], None, None, None, seed_v[r:r + 1], offset[r:r + 1])
        nop, _ = _run(dev, logits[r:r + 1], temperature[r:r + 1], top_k[r:r + 1], None, None, None, None, seed_v[r:r + 1], offset[r:r + 1])
        rk, _ = ref_sampler.sample(lf[r:r + 1], temperature[r:r + 1], None, top_p[r:r + 1], None, None, None, seed_v[r:r + 1], offset[r:r + 1])
        rp, _ = ref_sampler.sample(lf[r:r + 1], temperature[r:r + 1], top_k[r:r + 1], None, None, None, None, seed_v[r:r + 1], offset[r:r + 1])
        print(f"seed {seed} V={vocab} bf16={bf} row {r}: T={temperature[r]} k={top_k[r]} p={top_p[r]} scale={scale[r, 0]} grid={grid[r]} | "
              f"got {got[r]} (rank {rank[int(got[r])]}) ref {ref[r]} (rank {rank[int(ref[r])]}) alone {alone[0]} | "
              f"no-top-k: dev {nok[0]} ref {rk[0]} | no-top-p: dev {nop[0]} ref {rp[0]} | {margins[r]}")
