"""Deterministic, torch-free synthetic tensor generator.

Counter-based (splitmix64 -> Box-Muller), so this container, the GPU box, the
golden-vector generator and bench.py all regenerate bit-identical inputs from
(seed, stream) without depending on torch / numpy RNG stream compatibility
(SURVEY.md section 8(d) "Synthetic inputs").

All functions return numpy arrays; callers wrap them in torch tensors.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 finaliser applied to (seed-offset) counters; uint64 in/out."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in (0, 1), never exactly 0 or 1."""
    idx = np.arange(n, dtype=np.uint64) + np.uint64(stream) * np.uint64(1 << 40)
    bits = _splitmix64(idx, seed) >> np.uint64(11)  # 53 bits
    return (bits.astype(np.float64) + 0.5) * (1.0 / (1 << 53))


def normal(shape, seed: int, stream: int = 0, std: float = 1.0, dtype=np.float32) -> np.ndarray:
    """N(0, std^2) via Box-Muller on two uniform streams."""
    n = int(np.prod(shape))
    u1 = uniform01(n, seed, 2 * stream)
    u2 = uniform01(n, seed, 2 * stream + 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (z * std).astype(dtype).reshape(shape)


def student_t4(shape, seed: int, stream: int = 0, std: float = 1.0, dtype=np.float32) -> np.ndarray:
    """Heavy-tailed weights: Student-t (df=4) rescaled to the requested std
    (var of t4 is 2), mimicking LLM outlier weights."""
    n = int(np.prod(shape))
    z = normal((n,), seed, 4 * stream + 100, 1.0, np.float64)
    chi = sum(normal((n,), seed, 4 * stream + 101 + i, 1.0, np.float64) ** 2 for i in range(4))
    t = z / np.sqrt(chi / 4.0)
    return (t * (std / np.sqrt(2.0))).astype(dtype).reshape(shape)


def llm_weight(n_out: int, n_in: int, seed: int, stream: int = 0, std: float = 0.02,
               heavy_tail: bool = False) -> np.ndarray:
    """fp32 weight matrix whose values are exactly fp16-representable (hub
    checkpoints are fp16, reference gptq_pb/run.py:24,29 dtype="auto")."""
    gen = student_t4 if heavy_tail else normal
    w = gen((n_out, n_in), seed, stream, std, np.float32)
    return w.astype(np.float16).astype(np.float32)


def activations(shape, seed: int, stream: int = 7, dtype=np.float16) -> np.ndarray:
    return normal(shape, seed, stream, 1.0, np.float32).astype(dtype)


def calib_inputs(nsamples: int, seqlen: int, hidden: int, seed: int, hot_frac: float = 0.01,
                 hot_scale: float = 20.0) -> np.ndarray:
    """Calibration activations with a few high-variance channels, giving the
    column-concentrated Hessian saliency real models show."""
    x = normal((nsamples, seqlen, hidden), seed, 11, 1.0, np.float32)
    n_hot = max(1, int(hidden * hot_frac))
    hot = (np.argsort(uniform01(hidden, seed, 13))[:n_hot]).astype(np.int64)
    x[..., hot] *= hot_scale
    return x
