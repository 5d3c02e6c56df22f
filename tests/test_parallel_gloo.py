"""Multi-process CPU tests (gloo, world_size 2) of the tensor-parallel PB linear
(pb_llm_amd/parallel.py): shard construction, N-split all-gather, K-split all-reduce.

There is no GPU here, so the per-rank HIP GEMV is replaced INSIDE THE TEST by the oracle
(dense float64 matmul over the shard's unpacked weight); what is under test is the
sharding math and the collective plumbing, which is identical on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pb_oracle as O
from pb_llm_amd import parallel as PP
from pb_llm_amd import synth


def test_split_points():
    assert PP.split_points(4096, 8, 16) == [512 * i for i in range(9)]
    pts = PP.split_points(11008, 8, 128)          # llama-7b down_proj K: 86 blocks of 128
    w = [pts[i + 1] - pts[i] for i in range(8)]
    assert sum(w) == 11008 and all(x % 128 == 0 for x in w) and max(w) - min(w) == 128
    pts = PP.split_points(13824, 8, 128)          # llama-13b: 108 blocks
    assert pts[-1] == 13824 and all(p % 128 == 0 for p in pts)
    pts = PP.split_points(100, 3, 16)             # ragged tail
    assert pts[0] == 0 and pts[-1] == 100 and pts == sorted(pts)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(N=96, K=1024, seed=21):
    W = synth.llm_weight(N, K, seed=seed, heavy_tail=True)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    b = synth.normal((N,), seed, 3, 0.1)
    x = synth.activations((3, K), seed, 21)
    return r, mask, b, x


def _oracle_forward(packed, bias, x, out_f32=False):
    """test-only stand-in for the HIP kernel on CPU ranks"""
    y = O.dense_linear(x.numpy(), packed.unpack().numpy(), None if bias is None else bias.numpy())
    return torch.from_numpy(y.astype(np.float32 if out_f32 else np.float16))


def _worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pb_llm_amd.parallel as PPw
        import pb_llm_amd.quant as Qw
        PPw.pb_linear_forward = _oracle_forward
        Qw.PBLinear.forward = lambda self, x: _oracle_forward(self.packed, self.pbl_bias, x)
        r, mask, b, x = _case()
        Wt, bt, mt, xt = torch.from_numpy(r["W_fq"]), torch.from_numpy(b), torch.from_numpy(mask), torch.from_numpy(x)
        ref = O.dense_linear(x, r["W_fq"], b)
        out = {}
        # N-split: no reduction, all-gather of the outputs
        shard, rows = PPw.shard_linear(Wt, bt, mt, "n", rank, world, -1, r["hscale"], r["hzero"])
        np.testing.assert_array_equal(shard.weight.float().cpu().numpy(), r["W_fq"][rows[0]:rows[1]].astype(np.float32))
        y = PPw.PBLinearNSplit(shard, rows, Wt.shape[0])(xt)
        out["n"] = O.parity_errors(y.numpy(), ref)[0]
        # K-split with PTQ side information
        shard, cols = PPw.shard_linear(Wt, bt, mt, "k", rank, world, -1, r["hscale"], r["hzero"])
        np.testing.assert_array_equal(shard.weight.float().cpu().numpy(), r["W_fq"][:, cols[0]:cols[1]])
        assert (shard.bias is not None) == (rank == 0)
        y = PPw.PBLinearKSplit(shard, cols)(xt)
        out["k"] = O.parity_errors(y.numpy(), ref)[0]
        # K-split from a flattened checkpoint (levels inferred on full rows, shared by all shards)
        shard, cols = PPw.shard_linear(Wt, None, None, "k", rank, world)
        np.testing.assert_array_equal(shard.weight.float().cpu().numpy(), r["W_fq"][:, cols[0]:cols[1]])
        y = PPw.PBLinearKSplit(shard, cols)(xt)
        out["k_flat"] = O.parity_errors(y.numpy(), O.dense_linear(x, r["W_fq"]))[0]
        # pre-sharded input (output of an N-split layer feeding a K-split layer, Megatron style)
        y2 = PPw.PBLinearKSplit(shard, cols, input_is_sharded=True)(xt[..., cols[0]:cols[1]])
        assert torch.equal(y, y2)
        # fused-push eligibility (round 5): how many tokens one push pass takes follows the rank's OWN shard; the
        # ranks must agree on the minimum or one waits on counters while its peer waits on flags
        assert PPw.agree_min(4 - rank) == 4 - (world - 1)
        import ctypes as C
        from pb_llm_amd import _lib
        ks = PPw.PBLinearKSplit(shard, cols, collective="p2p")
        own = _lib.lib().pbl_linear_push_max_tokens(C.byref(shard.packed.layer_struct(None)))
        owns = [None] * world
        dist.all_gather_object(owns, own)
        assert 1 <= min(owns) <= 4 and ks.push_max_tokens == min(owns)
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tensor_parallel_world2_gloo():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
        assert len(results) == world
        for rank in range(world):
            for k, v in results[rank].items():
                assert v < 1e-3, (rank, k, v)


def test_push_limit_depends_on_the_shards_own_data():
    """pbl_linear_push_max_tokens (= one GEMV pass within the LDS budget) is a function of the shard's width AND of its fullest
    record: two shards of one layer can answer differently -- which is why PBLinearKSplit agrees on the minimum over the ranks"""
    import ctypes as C
    from pb_llm_amd import _lib
    L = _lib.lib()

    def lim(K, max_nch, NRB=256, G=1):
        lay = _lib.PblLayer(None, None, NRB * 16, K, (K + 511) // 512, G, NRB, 0xC, max_nch, 0)
        return L.pbl_linear_push_max_tokens(C.byref(lay))

    assert lim(4096, 420) == 4 and lim(512, 60) == 4
    a, b = lim(5504, 600), lim(5504, 700)                 # tp2 shards of an 11008-wide layer at ~20 % salients, one a little fuller
    assert 1 <= b < a <= 4, (a, b)
    assert lim(4096, 420, G=32) == 0                      # column groups: never fused
    assert [lim(5504, n) for n in (100, 400, 800, 1600, 3200)] == sorted((lim(5504, n) for n in (100, 400, 800, 1600, 3200)), reverse=True)
