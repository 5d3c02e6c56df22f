#!/usr/bin/env python3
"""Time single PB linears of the BASELINE configs (not the headline bench line): per-call
device time over a rotation of layer copies larger than the Infinity Cache."""
import argparse, json, sys, os, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth
from pb_llm_amd.packing import pack_dense
from pb_llm_amd.quant import PBLinear

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="13824x5120,5120x13824,4096x4096,11008x4096,4096x11008")
    ap.add_argument("--M", default="1,4,32")
    ap.add_argument("--low-frac", type=float, default=0.8)
    ap.add_argument("--copies-gb", type=float, default=0.6)
    a = ap.parse_args()
    dev = "cuda:0"
    for shp in a.shapes.split(","):
        N, K = map(int, shp.split("x"))
        W = synth.llm_weight(N, K, seed=N % 97)
        mask = O.ptq_low_mask(W, a.low_frac, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        p = pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0], r["hscale"], r["hzero"],
                       (~mask).astype(np.uint8))
        ncopy = max(2, int(a.copies_gb * 1e9 / p.nbytes))
        layers = [PBLinear(p.to(dev), None) for _ in range(ncopy)]
        for M in map(int, a.M.split(",")):
            x = torch.from_numpy(synth.activations((M, K), 3, 21)).to(dev)
            for l in layers: l(x)
            torch.cuda.synchronize()
            reps = max(3, 400 // ncopy)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                for l in layers: l(x)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * ncopy)
            balg = p.algorithmic_bytes(M)
            print(json.dumps(dict(shape=shp, low_frac=a.low_frac, M=M, us_per_call=round(us, 2), tokens_per_s=round(M / us * 1e6),
                                  alg_GBps=round(balg / us / 1e3, 1), gflops=round(2.0 * N * K * M / us / 1e3, 1),
                                  packed_MB=round(p.nbytes / 1e6, 2), copies=ncopy)), flush=True)

if __name__ == "__main__":
    main()
