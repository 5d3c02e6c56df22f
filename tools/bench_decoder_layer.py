#!/usr/bin/env python3
"""llama-7b decoder-layer linears (q,k,v,o 4096x4096; gate,up 11008x4096; down 4096x11008) as PB
layers: decode (M=1: fused qkv, o, fused gate+up, down = 4 launches, captured in a hipGraph) and
prefill (M=2048, GEMM regime).  Linears only (no attention / norms); data rotates over enough
layer copies to defeat the Infinity Cache."""
import argparse, json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q
from pb_llm_amd.runtime import GroupedGemv

def make(N, K, seed, lf):
    W = synth.llm_weight(N, K, seed=seed)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    return Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"])

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--low-frac", type=float, default=0.9); ap.add_argument("--copies", type=int, default=10)
    a = ap.parse_args(); dev = "cuda:0"
    H, I = 4096, 11008
    base = dict(q=make(H, H, 1, a.low_frac), o=make(H, H, 2, a.low_frac), gate=make(I, H, 3, a.low_frac), down=make(H, I, 4, a.low_frac))
    pk = {k: v.packed for k, v in base.items()}
    layer_bytes = 4 * pk["q"].nbytes + 2 * pk["gate"].nbytes + pk["down"].nbytes
    alg = 4 * pk["q"].algorithmic_bytes(1) + 2 * pk["gate"].algorithmic_bytes(1) + pk["down"].algorithmic_bytes(1)
    copies = []
    for c in range(a.copies):   # distinct HBM copies of one decoder layer
        qkv = GroupedGemv([pk["q"].to(dev) for _ in range(3)], None, 1, dev, shared_x=True)
        o = Q.PBLinear(pk["o"].to(dev), None)
        gu = GroupedGemv([pk["gate"].to(dev) for _ in range(2)], None, 1, dev, shared_x=True)
        dn = Q.PBLinear(pk["down"].to(dev), None)
        copies.append((qkv, o, gu, dn))
    x = torch.from_numpy(synth.activations((1, H), 1, 21)).to(dev)
    xi = torch.from_numpy(synth.activations((1, I), 2, 21)).to(dev)
    for qkv, o, gu, dn in copies:
        qkv.x[0].copy_(x); gu.x[0].copy_(x)
    def decode_all():
        for qkv, o, gu, dn in copies:
            qkv.launch(); o(x); gu.launch(); dn(xi)
    decode_all(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): decode_all()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): decode_all()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    us_layer = e0.elapsed_time(e1) * 1e3 / (n * a.copies)
    print(json.dumps(dict(mode="decode M=1, 4 launches per decoder layer (hipGraph)", us_per_decoder_layer=round(us_layer, 2),
                          packed_MB_per_layer=round(layer_bytes / 1e6, 1), alg_GBps=round(alg / us_layer / 1e3),
                          tokens_per_s_32_layers_linears_only=round(1e6 / (32 * us_layer)))), flush=True)
    # batched decode (M = 16 rows per forward): seven separate small-batch calls per decoder layer vs q|k|v and gate|up as ONE merged
    # layer each (packing.concat_rows, round 5) -- both captured in a hipGraph, linears only
    from pb_llm_amd.packing import concat_rows
    Mb = int(os.environ.get("PBL_BENCH_MB", 16))
    xb = torch.from_numpy(synth.activations((Mb, H), 5, 21)).to(dev); xbi = torch.from_numpy(synth.activations((Mb, I), 6, 21)).to(dev)
    sep, mer = [], []
    for c in range(a.copies):
        q3 = [Q.PBLinear(pk["q"].to(dev), None) for _ in range(3)]
        o_ = Q.PBLinear(pk["o"].to(dev), None)
        gu = [Q.PBLinear(pk["gate"].to(dev), None) for _ in range(2)]
        dn = Q.PBLinear(pk["down"].to(dev), None)
        sep.append((q3, o_, gu, dn))
        mer.append((Q.PBLinear(concat_rows([m.packed for m in q3]), None), o_, Q.PBLinear(concat_rows([m.packed for m in gu]), None), dn))
    def run_sep():
        for q3, o_, gu, dn in sep:
            for m in q3: m(xb)
            o_(xb)
            for m in gu: m(xb)
            dn(xbi)
    def run_mer():
        for qm, o_, gm, dn in mer:
            qm(xb); o_(xb); gm(xb); dn(xbi)
    res = {}
    for name, fn in (("separate", run_sep), ("merged", run_mer)):
        with torch.no_grad():
            fn(); torch.cuda.synchronize()
            s2 = torch.cuda.Stream()
            with torch.cuda.stream(s2): fn()
            torch.cuda.synchronize()
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb): fn()
            for _ in range(10): gb.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(100): gb.replay()
            e1.record(); torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) * 1e3 / (100 * a.copies), 2)
    print(json.dumps(dict(mode=f"batched decode M={Mb}, linears of a decoder layer (hipGraph): 7 separate calls vs q|k|v and gate|up merged",
                          us_per_decoder_layer=res)), flush=True)
    # prefill
    M = 2048
    xp = torch.from_numpy(synth.activations((M, H), 3, 21)).to(dev); xpi = torch.from_numpy(synth.activations((M, I), 4, 21)).to(dev)
    mods = [(Q.PBLinear(pk["q"].to(dev), None), xp)] * 4 + [(Q.PBLinear(pk["gate"].to(dev), None), xp)] * 2 + [(Q.PBLinear(pk["down"].to(dev), None), xpi)]
    def prefill():
        for m, xx in mods: m(xx)
    prefill(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): prefill()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = 2.0 * M * (4 * H * H + 3 * H * I)
    print(json.dumps(dict(mode="prefill M=2048 (device unpack + library GEMM)", ms_per_decoder_layer=round(ms, 3),
                          tflops=round(flops / ms / 1e9, 1), tokens_per_s_32_layers_linears_only=round(M / (32 * ms * 1e-3)))), flush=True)

if __name__ == "__main__":
    main()
