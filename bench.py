#!/usr/bin/env python3
"""bench.py -- PB-linear GEMV stream benchmark (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: L
(default 224 = the number of linears in llama-7b) independent llama-7b q_proj-shaped
partially-binarized linears (4096x4096, xnor low_frac=0.9 + 10 % int8 salient,
groupsize -1, RTN structure from the oracle's restatement of gptq_pb) each applied to
its own bs=1 fp16 activation vector, executed as ONE grouped HIP launch.  All inputs
are resident in HBM before the timed region; the L layer blobs live at distinct HBM
addresses (> 1.2 GB, beyond the 256 MiB Infinity Cache), so the stream is served by
HBM, not by cache.

value = layer-tokens/s = n_gpus * L * M * steps / t  (whole job).
roofline.achieved = ALGORITHMIC bytes per launch / average launch duration, where
B_alg(N,K,nnz,M) is SURVEY.md 8(d)'s formula (1 bit/weight sign plane + uint8 code
and 8-bit index per salient + row params + row pointers + fp16 x and y).

Multi-GPU (--gpus N, launched by torch.distributed.run): each rank streams its own
L layers and tokens (independent requests; no data-path collective): weak scaling.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_base_layers(n_distinct, N, K, low_frac, seed0):
    """Synthetic PB layers via the oracle's restatement of the reference PTQ RTN path
    (structure only -- the oracle is not timed here and not on the product path)."""
    from oracle import pb_oracle as O
    from pb_llm_amd import synth
    from pb_llm_amd.packing import pack_dense
    out = []
    for i in range(n_distinct):
        W = synth.llm_weight(N, K, seed=seed0 + i)
        mask = O.ptq_low_mask(W, low_frac, "magnitude", None, -1)
        r = O.ptq_rtn(W, mask, 8, -1)
        p = pack_dense(r["W_fq"], r["scale"][0] + r["mean"][0], -r["scale"][0] + r["mean"][0],
                       r["hscale"], r["hzero"], (~mask).astype(np.uint8))
        out.append((p, r["W_fq"]))
    return out


def cpu_baseline(dense_layers, K, M, budget_s=12.0):
    """The reference's CPU path on this box's host cores, via the oracle's statement of it
    (oracle/pb_oracle.py: torch_dense_linear = F.linear over the dense fake-quant fp32 weight, what
    gptq_pb/run.py / eval_after_qat.py execute per layer), on a bounded
    sample.  Also the QAT layer as written (weight re-simulated on every call, item 2)."""
    from oracle import pb_oracle as O
    from pb_llm_amd import synth
    Ws = [torch.from_numpy(w) for w in dense_layers]
    xs = [torch.from_numpy(synth.activations((M, K), 900 + i, 21)).float() for i in range(len(Ws))]
    for w, x in zip(Ws, xs):
        O.torch_dense_linear(x, w)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for w, x in zip(Ws, xs):
            O.torch_dense_linear(x, w)
        n += len(Ws)
    dt = time.perf_counter() - t0
    # as-written QAT forward: a few calls are enough (tens of ms each)
    w = Ws[0]
    mask = w.abs() > w.abs().flatten().kthvalue(int(0.9 * w.numel()))[0]
    scale = w[~mask].abs().mean().view(1, 1)
    O.torch_qat_forward_as_written(xs[0], w, mask, scale)
    t1 = time.perf_counter()
    nq = 5
    for _ in range(nq):
        O.torch_qat_forward_as_written(xs[0], w, mask, scale)
    qat_ms = 1e3 * (time.perf_counter() - t1) / nq
    return dict(value=n * M / dt, unit="layer-tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} fp32 F.linear calls (oracle.torch_dense_linear) over {len(Ws)} distinct dense "
                       f"{Ws[0].shape[0]}x{K} fake-quant weights (cache-cold rotation), {dt:.1f} s",
                ms_per_layer=1e3 * dt / n, qat_forward_as_written_ms_per_layer=qat_ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--layers", type=int, default=224)
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic weight sets (replicated to L blobs)")
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--M", type=int, default=1)
    ap.add_argument("--low-frac", type=float, default=0.9)
    ap.add_argument("--mode", choices=["grouped", "graph", "eager"], default="grouped")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallel", choices=["dp", "tp"], default="dp",
                    help="dp: independent layer streams per rank (weak scaling, no collective); "
                         "tp: every layer K-split across ranks + one RCCL all-reduce of the stacked fp32 partial "
                         "outputs per step (strong scaling)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (any world size, incl. 1)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    import __graft_entry__ as ge
    ge.build()
    from pb_llm_amd import synth
    from pb_llm_amd.quant import PBLinear
    from pb_llm_amd.runtime import GroupedGemv

    tp = a.parallel == "tp" and use_dist
    base = build_base_layers(a.distinct, a.N, a.K, a.low_frac, seed0=1000 + (0 if tp else 17 * rank))
    Kloc, c0 = a.K, 0
    if tp:  # K-split every layer: this rank keeps columns [c0, c1)
        from pb_llm_amd.packing import infer_levels, pack_dense
        from pb_llm_amd.parallel import split_points, COL_ALIGN
        c0, c1 = split_points(a.K, world, COL_ALIGN)[rank:rank + 2]
        Kloc = c1 - c0
        shards = []
        for p_full, Wd in base:
            hi, lo = infer_levels(Wd)
            full = p_full.unpack().numpy()
            assert np.array_equal(full, Wd)
            # per-row levels / code grid are those of the FULL row, identical on every rank
            from pb_llm_amd.packing import infer_code_grid
            ss, sz = infer_code_grid(Wd, hi, lo)
            shards.append((pack_dense(Wd[:, c0:c1], hi, lo, ss, sz), Wd))
        base = shards
    packed = [base[i % a.distinct][0].to(dev) for i in range(a.layers)]
    # .to(dev) of a host blob allocates a fresh device buffer per call: L distinct HBM regions
    assert len({p.blob.data_ptr() for p in packed}) == a.layers
    grp = GroupedGemv(packed, None, M=a.M, device=dev, out_f32=tp)
    for i, t in enumerate(grp.x):
        xi = synth.activations((a.M, a.K), 5000 + i + (0 if tp else 1000 * rank), 21)
        t.copy_(torch.from_numpy(np.ascontiguousarray(xi[:, c0:c0 + Kloc])))
    torch.cuda.synchronize()

    singles = [PBLinear(p, None) for p in packed] if a.mode != "grouped" else None

    def step():
        if a.mode == "grouped":
            grp.launch()
            if tp:
                dist.all_reduce(grp.y_all)
        else:
            for m, x in zip(singles, grp.x):
                m(x)

    graph = None
    if a.mode == "graph":
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        run = graph.replay
    else:
        run = step

    for _ in range(a.warmup):
        run()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_s = e0.elapsed_time(e1) * 1e-3
    if use_dist:
        tt = torch.tensor([wall, dev_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_s = tt.tolist()

    traffic = None
    try:  # HBM bytes per launch from the committed PMC profile of this exact workload (never measured here)
        tj = json.load(open(os.path.join(REPO, "profiles", "traffic_r01.json")))
        key = f"N{a.N}_K{a.K}_L{a.layers}_M{a.M}_lf{a.low_frac}_{a.mode}"
        if tj.get("workload_key") == key and not tp:
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    if rank == 0:
        b_alg = grp.algorithmic_bytes()
        launches = a.steps * (1 if a.mode == "grouped" else a.layers)
        per_launch_s = dev_s / launches
        b_launch = b_alg / (1 if a.mode == "grouped" else a.layers)
        achieved = b_launch / per_launch_s / 1e9
        out = {
            "metric": "PB-linear GEMV tokens/sec + achieved HBM GB/s, llama-7b 4096x4096 low_frac=0.9",
            "value": (1 if tp else world) * a.layers * a.M * a.steps / wall,
            "unit": "layer-tokens/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * wall / a.steps,
            "higher_is_better": True, "scaling": "strong" if tp else "weak", "vs_baseline": None,
            "dtype": "f16 (x, y) / 1-bit + u8 weights, f32 accumulate", "data": "synthetic",
            "config": {"workload": f"llama-7b q_proj {a.N}x{a.K} xnor low_frac={a.low_frac} + int8 salient, "
                                   f"bs={a.M} GEMV, stream of {a.layers} layer blobs at distinct HBM addresses "
                                   f"({a.distinct} distinct weight sets), mode={a.mode}",
                       "layers_per_step": a.layers, "tokens": a.M,
                       "parallelism": (f"tp{world} (every layer K-split, one RCCL all-reduce of [L,M,N] fp32 per step)" if tp
                                       else f"dp{world} (independent layer streams, no collective)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "pbl_gemv_kernel<1,4>" if a.mode == "grouped" else "pbl_gemv_kernel<1,1>",
                         "algorithmic_bytes_per_launch": b_launch,
                         "packed_bytes_per_launch": grp.packed_bytes() / (1 if a.mode == "grouped" else a.layers),
                         "us_per_launch": 1e6 * per_launch_s,
                         "us_per_layer": 1e6 * dev_s / (a.steps * a.layers)},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline([b[1] for b in base], a.K, a.M)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
