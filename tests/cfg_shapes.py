"""Shared builders for the BASELINE.json configurations that need model-shaped layers (test infrastructure; uses the
oracle, never imported by the product path).

configs[2] ("config 3"): llama-7b full model, low_frac 0.95, *hessian* salients, seq 2048 -- its seven decoder linears.
configs[4] ("config 5"): the same layers row-sharded over 2/4/8 GPUs.
Reference anchors: gptq_pb/run_all.sh:1-5 (the configurations), gptq_pb/gptq.py:93-99 (hessian saliency),
gptq_pb/run.py:148-156 (Hessians accumulated from calibration activations).
"""
import numpy as np

from oracle import pb_oracle as O
from pb_llm_amd import synth

# llama-7b decoder layer: name -> (out_features, in_features)   (hidden 4096, inter 11008)
LLAMA7B = {"q_proj": (4096, 4096), "k_proj": (4096, 4096), "v_proj": (4096, 4096), "o_proj": (4096, 4096),
           "gate_proj": (11008, 4096), "up_proj": (11008, 4096), "down_proj": (4096, 11008)}
LLAMA7B_DISTINCT = {"q_proj": ("q_proj", "k_proj", "v_proj", "o_proj"), "gate_proj": ("gate_proj", "up_proj"),
                    "down_proj": ("down_proj",)}


def hinv_diag_synthetic(K: int, seed: int, exact: bool, nsamples: int = 4, seqlen: int = 256) -> np.ndarray:
    """diag(U), U = upper Cholesky factor of H^-1, for column-concentrated calibration activations
    (synth.calib_inputs: 1 % of the channels 20x hotter).  exact: the oracle's full add_batch / damping / Cholesky chain
    (gptq.py:35-51,67-81); else the same quantity for the DIAGONAL of H only (1/sqrt(H_jj + damp)), which keeps the
    K = 11008 case out of a 3 x O(K^3) host factorisation -- the saliency it induces is the same kind: hot input
    channels become salient for every row."""
    X = synth.calib_inputs(nsamples, seqlen, K, seed)
    if exact:
        U, _ = O.hinv_cholesky_upper(O.hessian_from_inputs(X))
        return np.diag(U).astype(np.float32).copy()
    hd = np.zeros(K, np.float64)
    for s in range(nsamples):          # add_batch's running mean of 2 x x^T, diagonal only
        hd = hd * (s / (s + 1)) + (2.0 / (s + 1)) * (X[s].reshape(-1, K).astype(np.float64) ** 2).sum(0)
    hd[hd == 0] = 1
    hd = hd + 0.01 * hd.mean()
    return (1.0 / np.sqrt(hd)).astype(np.float32)


def hessian_layer(N: int, K: int, low_frac: float, seed: int, exact_hessian: bool | None = None):
    """(W, low_mask, rtn result dict) of one layer with hessian-metric salients (RTN values: disable_gptq)."""
    W = synth.llm_weight(N, K, seed=seed)
    d = hinv_diag_synthetic(K, seed, K <= 4096 if exact_hessian is None else exact_hessian)
    mask = O.ptq_low_mask(W, low_frac, "hessian", d, -1)
    return W, mask, O.ptq_rtn(W, mask, 8, -1)
