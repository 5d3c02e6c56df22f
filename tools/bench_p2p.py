#!/usr/bin/env python3
"""Latency of the K-split decode all-reduce ([1, 4096] fp32 = 16 KB) between TWO PROCESSES on one GPU box: libpbl's one-shot
peer-to-peer all-reduce (eager launches, and 64 dependent all-reduces replayed from ONE hipGraph, as a decoder's 64 K-split
layers would run them) against torch.distributed with the gloo backend (the only collective that accepts two ranks on one
device: RCCL refuses).  A proxy: hipIpc mapping, flag protocol and rank-ordered sum are what runs over xGMI; the link is not.
usage: python tools/bench_p2p.py   (spawns the two ranks itself)"""
import json, os, socket, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from pb_llm_amd.parallel import P2PAllReduce
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4096
    comm = P2PAllReduce(n, dev)
    x = torch.full((n,), float(rank + 1), device=dev)
    res = {}

    def timed(fn, reps):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    comm.all_reduce_(x.clone()); torch.cuda.synchronize()
    res["p2p_eager_us"] = round(timed(lambda: comm.all_reduce_(x), 2000), 2)
    # 64 dependent all-reduces in one graph (the buffer keeps the call number, so a replay advances it)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): comm.all_reduce_(x)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(64): comm.all_reduce_(x)
    res["p2p_graph64_us_per_allreduce"] = round(timed(g.replay, 100) / 64, 2)
    y = torch.ones(n, device=dev)
    res["gloo_us"] = round(timed(lambda: dist.all_reduce(y), 200), 2)
    comm.check(); comm.close()
    out[rank] = res
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mp.Manager().dict()
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    print(json.dumps({"message": "16 KB fp32 all-reduce, 2 processes sharing one MI355X", "rank0": out[0], "rank1": out[1]}))
