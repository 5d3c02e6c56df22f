"""-m gpu: salient selection and the 8-bit row quantizer on the device (pbl_kth_pair / pbl_outlier_mask /
pbl_quant8_rows, pb_llm_amd/prep.py).  Integer / order-statistic work: BIT-EXACT against numpy, against the oracle
(oracle.weight_quant_8bit, pinned by golden G3) and against the reference's gen_outlier_mask outputs (golden G4)."""
import numpy as np
import pytest
import torch

from oracle import pb_oracle as O
from pb_llm_amd import _lib, prep, synth
from pb_llm_amd import quant as Q
from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()


def T(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n", [1, 7, 1000, 65537, 1 << 20])
def test_kth_pair_exact(dt, n):
    w = T(synth.llm_weight(1, n, seed=n, heavy_tail=True).reshape(-1)).to(dt)
    if n >= 1000:
        w[::7] = w[3]                      # many duplicates
        w[1::50] = 0.0
        w[2::50] = -0.0
    flat = np.sort(w.float().cpu().numpy())
    for k_lo, k_hi in {(1, n), (max(1, n // 20), max(1, n - n // 20)), ((n + 1) // 2, (n + 1) // 2)}:
        got = prep.kth_pair(w, k_lo, k_hi).cpu().numpy()
        assert got[0] == flat[k_lo - 1] and got[1] == flat[k_hi - 1], (k_lo, k_hi)
    with pytest.raises(IndexError):
        prep.kth_pair(w, 0, n)             # torch.kthvalue(k=0) raises in the reference as well
    with pytest.raises(IndexError):
        prep.kth_pair(w, 1, n + 1)


def test_kth_pair_llama_layer_matches_torch_kthvalue():
    W = T(synth.llm_weight(4096, 4096, seed=2, heavy_tail=True))
    n = W.numel()
    k_lo, k_hi = int(n * 0.05), int(n * 0.95)
    got = prep.kth_pair(W, k_lo, k_hi)
    ref = torch.stack([torch.kthvalue(W.view(-1).cpu(), k_lo)[0], torch.kthvalue(W.view(-1).cpu(), k_hi)[0]])
    assert torch.equal(got.cpu(), ref)
    mask = prep.outlier_mask(W, got)
    assert mask.dtype == torch.bool and torch.equal(mask, (W < ref[0].to(DEV)) | (W > ref[1].to(DEV)))


@pytest.mark.parametrize("dt", [np.float32, np.float16])
@pytest.mark.parametrize("shape", [(8, 64), (33, 777), (16, 13824)])
def test_quant8_rows_bit_exact_vs_oracle(dt, shape):
    N, K = shape
    W = synth.llm_weight(N, K, seed=N + K, heavy_tail=True).astype(dt)
    W[0, :] = 0.25                          # constant row: range 0 -> nan codes -> 0
    W[1, :] = np.abs(W[1, :]) + dt(0.7)     # min rounds to +1: the "rounded zero point" quirk with a nonzero zero point
    ref = O.weight_quant_8bit(W)
    codes = O.weight_quant_8bit(W, simulated=False)
    Wd = T(W)
    sc, zp = prep.quant8_rows_(Wd)
    got = Wd.cpu().numpy()
    assert got.dtype == W.dtype and np.array_equal(got.view(np.uint16 if dt == np.float16 else np.uint32),
                                                   ref.view(np.uint16 if dt == np.float16 else np.uint32))
    # the packer's affine form reproduces every weight from an integer code
    rec = codes.astype(np.float32) * sc.cpu().numpy()[:, None] + zp.cpu().numpy()[:, None]
    assert np.array_equal(rec.astype(dt)[2:], ref[2:])


def test_quant8_rows_matches_golden_g3():
    g = golden("g3_weight_quant_8bit")
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16)):
        Wd = T(g["W"]).to(dt)
        prep.quant8_rows_(Wd)
        assert np.array_equal(Wd.float().cpu().numpy(), g[f"sim_{tag}"])


@pytest.mark.parametrize("tag,dt", [("f32", torch.float32), ("f16", torch.float16)])
def test_gen_outlier_mask_on_gpu_matches_reference_golden_g4(tag, dt):
    """the module's gen_outlier_mask with the weight on the GPU == the reference module on the CPU, bit for bit"""
    g = golden("g4_pb_qat_linear")
    W = synth.llm_weight(768, 768, seed=4, heavy_tail=True)
    W[7, 9] = 0.0
    m = Q.BinaryXnorExceptOutliersLinear(torch.from_numpy(W).to(dt), None, 0.1).to(DEV)
    m.eval()
    m.gen_outlier_mask()
    assert m.weight.is_cuda and m.outlier_mask.is_cuda
    assert np.array_equal(np.packbits(m.outlier_mask.cpu().numpy()), g[f"mask_{tag}"])
    assert np.array_equal(m.weight.data.float().cpu().numpy(), g[f"w_hat_{tag}"].astype(np.float32))
    assert m.binary_scale.dtype == dt and tuple(m.binary_scale.shape) == (1, 1)
    ref_s = float(g[f"binary_scale_{tag}"].reshape(()))
    assert abs(float(m.binary_scale) - ref_s) <= (1.2e-7 if dt == torch.float32 else 1e-3) * ref_s
    assert abs(m.outlier_nbits - float(g[f"outlier_nbits_{tag}"])) < 1e-9
    x = synth.normal((3, 768), 4, 5, 1.0)
    with torch.no_grad():
        y = m(T(x).to(dt))
    mask = m.outlier_mask.cpu().numpy()
    w_hat = m.weight.data.cpu().numpy()
    ref = O.pb_qat_forward(x.astype(w_hat.dtype), w_hat, mask, m.binary_scale.cpu().numpy(), None, 1.0)
    rel, ratio = O.parity_errors(y.float().cpu().numpy(), ref)
    assert rel < 1e-3 and ratio < 1.0
