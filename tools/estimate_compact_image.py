#!/usr/bin/env python3
"""Planning aid (CPU only): how many bytes a GEMM image of BASELINE configs[3]'s layer would take under slot encodings that were
NOT built -- to size the "compact image" item of DESIGN section 9 against the shipped one.
  shipped    4-byte entries {LDS offset : fp16}, dealt round-robin to 64 lanes, slots of whole KiB (16 bytes per lane)
  row-owned  2-byte entries {7-bit column : 8-bit code}, the row implied by the lane (4 lanes per row), slot = plane dword + the
             fullest lane's entries, in units of 16 / 8 / 4 bytes per lane
Usage: python tools/estimate_compact_image.py [N K low_frac]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth

N, K, lf = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (13824, 5120, 0.8)
W = synth.llm_weight(N, K, seed=N % 97)
mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)                       # True = binarized
sal = ~mask
NRB, NH = N // 16, K // 128
cnt = sal.reshape(NRB, 16, NH, 128).sum(axis=3)                            # [record, row, half slab]
n_slot = cnt.sum(axis=1)                                                   # entries per slot
# shipped: words per lane = 1 + ceil(n / 64), vectors of 4 words
nv = np.ceil((1 + np.ceil(n_slot / 64)) / 4)
shipped = nv.sum() * 1024
# row-owned 2-byte entries: lane capacity = ceil(max row count / 4) halfwords + the 4-byte plane word
cap = np.ceil(cnt.max(axis=1) / 4)
out = {"layer": f"{N}x{K} low_frac {lf}", "salient_fraction": round(float(sal.mean()), 4), "blob_MB_approx": round((N * K / 8 + 2 * sal.sum()) / 1e6, 1),
       "entries_per_slot_mean": round(float(n_slot.mean()), 1), "entries_per_slot_max": int(n_slot.max()),
       "fullest_row_per_slot_mean": round(float(cnt.max(axis=1).mean()), 1), "shipped_image_MB": round(shipped / 1e6, 1)}
for gran in (16, 8, 4):
    per_lane = np.ceil((4 + 2 * cap) / gran) * gran
    out[f"row_owned_2B_gran{gran}_MB"] = round(float(per_lane.sum() * 64) / 1e6, 1)
# ideal: no padding at all
out["ideal_4B_entries_MB"] = round((N * K / 8 + 4 * sal.sum()) / 1e6, 1)
out["ideal_2B_entries_MB"] = round((N * K / 8 + 2 * sal.sum()) / 1e6, 1)
import json
print(json.dumps(out))
