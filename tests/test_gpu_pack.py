"""-m gpu: the device-side packer (csrc/pbl_pack.hip, packing.pack_dense_dev) produces the host packer's blob BYTE FOR BYTE,
and the module paths that now pack on the GPU (QAT eval, BinaryLinear / XnorBinaryLinear, LowHighGPTQ.to_pb) agree with the
host route.  The reference has no packed format (gptq_pb/gptq.py:180-184 writes dense weights back)."""
import numpy as np
import pytest
import torch

from oracle import pb_format_ref as F
from oracle import pb_oracle as O
from pb_llm_amd import _lib, synth
from pb_llm_amd import quant as Q
from pb_llm_amd.packing import infer_levels, pack_dense, pack_dense_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def both(Wd, hi, lo, ss, sz, sal, f16):
    host = pack_dense(Wd, hi, lo, ss, sz, sal, sal_f16=f16)
    dev = pack_dense_dev(T(Wd), T(hi), T(lo), T(ss), T(sz), T(sal), sal_f16=f16)
    torch.cuda.synchronize()
    assert (dev.N, dev.K, dev.G, dev.NRB, dev.flags, dev.max_nch, dev.max_nexc, dev.nnz, dev.nexc) == \
           (host.N, host.K, host.G, host.NRB, host.flags, host.max_nch, host.max_nexc, host.nnz, host.nexc)
    hb, db = host.blob.numpy(), dev.blob.cpu().numpy()
    assert hb.size == db.size
    if not np.array_equal(hb, db):
        bad = np.nonzero(hb != db)[0]
        raise AssertionError(f"{bad.size} differing bytes, first at {bad[:8]}")
    return host, dev


@pytest.mark.parametrize("N,K,lf,f16,gs,exc", [(256, 1024, 0.9, False, -1, 0), (100, 1536, 0.9, True, -1, 3), (33, 520, 0.8, True, -1, 2),
                                               (64, 2048, 0.995, False, -1, 0), (48, 1024, 0.5, True, -1, 5), (64, 1024, 0.9, True, 128, 2),
                                               (4096, 4096, 0.9, True, -1, 4), (16, 777, 0.9, False, -1, 1)])
def test_device_packer_is_byte_identical(N, K, lf, f16, gs, exc):
    W = synth.llm_weight(N, K, seed=N + K, heavy_tail=True)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    Wd = r["W_fq"].copy()
    if f16:
        Wd = Wd.astype(np.float16).astype(np.float32)
    for e in range(exc):                                         # values on neither level nor the code grid
        Wd[(7 * e) % N, (131 * e + 5) % K] = np.float32(np.float16(0.123 + e)) if f16 else np.float32(0.123 + e)
    Wd[N - 1, K - 1] = np.nan if exc else Wd[N - 1, K - 1]       # a non-finite value is an exception, never a code
    hi, lo = infer_levels(Wd, gs, mask)
    sal = (~mask).astype(np.uint8)
    host, dev = both(Wd, hi, lo, np.asarray(r["hscale"], np.float32).reshape(-1), np.asarray(r["hzero"], np.float32).reshape(-1), sal, f16)
    if not exc:
        np.testing.assert_array_equal(F.decode(dev.blob.cpu().numpy()), Wd)          # and the independent decoder reads it back
    # without a salient mask and without a code grid (fully binarized layers): sign(0) entries become code entries
    if N <= 256:
        Wb = np.sign(synth.llm_weight(N, K, seed=3)).astype(np.float32)
        Wb[0, 5] = 0.0
        one = np.ones(N, np.float32)
        both(Wb, one, -one, one, np.zeros(N, np.float32), None, False)


def test_modules_pack_on_the_gpu_like_on_the_host():
    """QAT layer / BinaryLinear / XnorBinaryLinear with GPU weights: _pack() runs on the device and yields the blob the host
    route yields for the same module on the CPU; LowHighGPTQ.to_pb() packs from quantizer state without leaving the GPU."""
    W = synth.llm_weight(96, 1024, seed=5, heavy_tail=True)
    for dt in (torch.float32, torch.float16):
        cpu = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W).to(dt), None, 0.1).eval()
        cpu.gen_outlier_mask()
        gpu = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W).to(dt), None, 0.1).to(DEV).eval()
        gpu.gen_outlier_mask()
        pc, pg = cpu._pack(), gpu._pack()
        assert pg.blob.is_cuda and np.array_equal(pc.blob.numpy(), pg.blob.cpu().numpy())
    for cls in (Q.BinaryLinear, Q.XnorBinaryLinear):
        Wz = W.copy(); Wz[3, 7] = 0.0
        pc = cls(torch.from_numpy(Wz), None)._pack()
        pg = cls(torch.from_numpy(Wz), None).to(DEV)._pack()
        assert pg.blob.is_cuda
        if cls is Q.BinaryLinear:                     # (Xnor: row means / scales are reduced in another order on the GPU)
            assert np.array_equal(pc.blob.numpy(), pg.blob.cpu().numpy())
        else:
            np.testing.assert_allclose(pg.unpack().numpy(), pc.unpack().numpy(), rtol=2e-6, atol=1e-9)
    from pb_llm_amd.ptq import LowHighGPTQ
    lin = torch.nn.Linear(1024, 96, bias=True).half().to(DEV)
    lin.weight.data = torch.from_numpy(W).half().to(DEV)
    g = LowHighGPTQ(lin, "magnitude", -1, 8, disable_gptq=True)
    g.add_batch(torch.from_numpy(synth.calib_inputs(1, 64, 1024, 3)).to(DEV))
    g.fasterquant(0.9)
    pb = g.to_pb()
    assert pb.pbl_blob.is_cuda and pb.packed.flags & _lib.PBL_FLAG_SAL_F16
    assert torch.equal(pb.weight.cpu(), lin.weight.data.cpu())          # the blob IS the written-back fp16 weight
    assert pb.packed.nexc <= 4                                          # levels and code grid from the quantizer state: nothing off-grid
    x = torch.from_numpy(synth.activations((2, 1024), 4, 21)).to(DEV)
    ref = torch.nn.functional.linear(x, lin.weight.data, lin.bias.data)
    rel, ratio = O.parity_errors(pb(x).float().cpu().numpy(), ref.float().cpu().numpy().astype(np.float64))
    assert rel < 2e-3


@pytest.mark.parametrize("N,K,lf,f16,gs", [(256, 1024, 0.9, True, -1), (100, 1536, 0.9, False, -1), (64, 1024, 0.9, True, 128), (4096, 4096, 0.95, True, -1)])
def test_from_dense_on_the_gpu_is_byte_identical_to_the_host_route(N, K, lf, f16, gs):
    """PBLinear.from_dense with a GPU weight, the PTQ low mask and the HighQuantizer state (what gptq_pb/run.py has in hand
    after fasterquant, gptq.py:155,180-184): levels found on the device, blob from the device packer -- the bytes the host
    route (numpy level inference + host packer) produces for the same tensors; a sign(0) value that is rarer than both levels
    becomes an exception on both routes; a row where it is NOT rarer sends the whole layer through the host route."""
    W = synth.llm_weight(N, K, seed=N + K + 1, heavy_tail=True)
    W[1, 9] = 0.0
    mask = O.ptq_low_mask(W, lf, "magnitude", None, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    Wd = torch.from_numpy(r["W_fq"])
    Wd = Wd.half() if f16 else Wd
    lm = torch.from_numpy(mask)
    host = Q.PBLinear.from_dense(Wd, None, lm, gs, r["hscale"], r["hzero"])
    dev = Q.PBLinear.from_dense(Wd.to(DEV), None, lm.to(DEV), gs, r["hscale"], r["hzero"])
    assert dev.pbl_blob.is_cuda and not host.pbl_blob.is_cuda
    assert np.array_equal(host.pbl_blob.numpy(), dev.pbl_blob.cpu().numpy())
    if N == 64:       # a row-group holding ONE value at all binarized positions but two: three values, the middle one not rarer
        Wb = r["W_fq"].copy()
        g0 = np.nonzero(mask[5, :128])[0]
        Wb[5, g0] = Wb[5, g0[0]]
        Wb[5, g0[1]] += 0.5; Wb[5, g0[2]] += 1.0
        Wt = torch.from_numpy(Wb).half() if f16 else torch.from_numpy(Wb)
        assert Q._from_dense_dev(Wt.to(DEV), lm.to(DEV), gs, r["hscale"], r["hzero"]) is None
        h2 = Q.PBLinear.from_dense(Wt, None, lm, gs, r["hscale"], r["hzero"])
        d2 = Q.PBLinear.from_dense(Wt.to(DEV), None, lm.to(DEV), gs, r["hscale"], r["hzero"])
        assert np.array_equal(h2.pbl_blob.numpy(), d2.pbl_blob.cpu().numpy())


@pytest.mark.parametrize("gs,f16", [(-1, True), (128, True), (-1, False)])
def test_from_quantizers_on_the_gpu_is_byte_identical_to_the_host_route(gs, f16):
    """PBLinear.from_quantizers (the PTQ API INTEGRATION.md shows) with GPU tensors: the composition q_high * ~mask + q_low * mask
    (gptq_pb/gptq.py:119-127) and the packing both run on the device (round 3: .cpu(), a Python loop over the groups, the host
    packer) and give the host route's blob byte for byte"""
    N, K = 192, 1024
    W = synth.llm_weight(N, K, seed=77, heavy_tail=True)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, gs)
    r = O.ptq_rtn(W, mask, 8, gs)
    G = 1 if gs == -1 else K // gs
    mean, scale = r["mean"].reshape(G, N, 1), r["scale"].reshape(G, N, 1)
    dt = torch.float16 if f16 else torch.float32
    host = Q.PBLinear.from_quantizers(torch.from_numpy(W), torch.from_numpy(mask), mean, scale, r["hscale"], r["hzero"], groupsize=gs, dtype=dt)
    dev = Q.PBLinear.from_quantizers(T(W), T(mask), T(mean), T(scale), T(r["hscale"]), T(r["hzero"]), groupsize=gs, dtype=dt)
    assert dev.pbl_blob.is_cuda and not host.pbl_blob.is_cuda
    assert np.array_equal(host.pbl_blob.numpy(), dev.pbl_blob.cpu().numpy())
    ref = r["W_fq"].astype(np.float16).astype(np.float32) if f16 else r["W_fq"]
    assert np.array_equal(dev.weight.float().cpu().numpy(), ref)
