#!/usr/bin/env python3
"""Feasibility of a side-stream weight prefetch in sequential decode: llama-7b decoder-layer linears at M = 1 (fused qkv, o,
fused gate+up, down) with a few tiny kernels between them (standing in for norms / rotary / attention), captured in a
hipGraph; with and without a side stream that touches the NEXT group's packed records (one 4-byte read per 128-byte line)
while the tiny kernels run.  Layer copies rotate over more bytes than the Infinity Cache holds."""
import argparse, json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q
from pb_llm_amd.runtime import FusedGemv


def make(N, K, seed, lf):
    W = synth.llm_weight(N, K, seed=seed)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    return Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).packed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--gap", type=int, default=4, help="tiny kernels between two linears")
    a = ap.parse_args()
    dev = "cuda:0"
    H, I = 4096, 11008
    pk = dict(q=make(H, H, 1, 0.9), o=make(H, H, 2, 0.9), gate=make(I, H, 3, 0.9), down=make(H, I, 4, 0.9))
    groups = []      # flat list of (FusedGemv, x) in execution order
    x = torch.from_numpy(synth.activations((1, H), 1, 21)).to(dev)
    xi = torch.from_numpy(synth.activations((1, I), 2, 21)).to(dev)
    for c in range(a.copies):
        groups.append((FusedGemv([pk["q"].to(dev) for _ in range(3)], None, dev), x))
        groups.append((FusedGemv([pk["o"].to(dev)], None, dev), x))
        groups.append((FusedGemv([pk["gate"].to(dev) for _ in range(2)], None, dev), x))
        groups.append((FusedGemv([pk["down"].to(dev)], None, dev), xi))
    lines = [[p.blob.view(torch.int32)[::32] for p in g.packed] for g, _ in groups]
    small = torch.zeros(4096, device=dev)
    side = torch.cuda.Stream()

    def run(prefetch):
        main_s = torch.cuda.current_stream()
        for i, (g, xx) in enumerate(groups):
            g(xx)
            if prefetch:
                side.wait_stream(main_s)                 # start when this group's launch is done
                with torch.cuda.stream(side):
                    for t in lines[(i + 1) % len(groups)]:
                        t.sum()
            for _ in range(a.gap):
                small.add_(1.0)
        if prefetch:
            main_s.wait_stream(side)

    out = {}
    for name, pf in (("no_prefetch", False), ("prefetch", True), ("no_prefetch_again", False)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run(pf)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run(pf)
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        e0.record()
        for _ in range(n): g.replay()
        e1.record(); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) * 1e3 / (n * a.copies), 2)
    # the gap kernels alone
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4 * a.copies * a.gap): small.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): g.replay()
    e1.record(); torch.cuda.synchronize()
    out["gap_kernels_only"] = round(e0.elapsed_time(e1) * 1e3 / (100 * a.copies), 2)
    out["unit"] = "us per decoder layer (4 linear launches + 4 x gap tiny kernels)"
    out["gap"] = a.gap
    print(json.dumps(out))


if __name__ == "__main__":
    main()
