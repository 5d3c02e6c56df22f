#!/usr/bin/env python3
"""BASELINE configs[2] ("config 3"): llama-7b-shaped full model (HF LlamaForCausalLM, random init, synthetic tokens),
low_frac 0.95 with HESSIAN salients, on one MI355X.
  prefill  seq 2048, batch 1: dense fp16 fake-quant weights (what the reference evaluates, gptq_pb/eval_ppl_utils.py:55-64)
           vs PBLinear with the library backend (pbl_unpack_dev + library GEMM per layer) vs the fused kernel (pbl_gemm_f16)
  decode   one token per forward (no KV cache: every linear runs at M = 1): PBLinear eager, fused q/k/v + gate/up
           (harness.fuse_decode_), fused + hipGraph (harness.GraphedForward); dense fp16 for reference
The 7 linears of ONE decoder layer are quantised on the GPU (LowHighGPTQ, hessian metric, RTN values, Hessians from
column-concentrated calibration activations) and every one of the 32 layers gets its OWN device copy of the 7 blobs, so
decode streams 32 x 50 MB from distinct HBM addresses."""
import copy, json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from transformers import LlamaConfig, LlamaForCausalLM
from pb_llm_amd import harness as H, ptq, synth
from pb_llm_amd.quant import PBLinear

SEQ = int(os.environ.get("SEQ", 2048))
LAYERS = int(os.environ.get("LAYERS", 32))
MODES = os.environ.get("MODES", "prefill,decode").split(",")
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=LAYERS, num_attention_heads=32,
                  num_key_value_heads=32, vocab_size=32000, max_position_embeddings=4096)
torch.manual_seed(0)
with torch.device("cuda"):
    model = LlamaForCausalLM(cfg).half().eval()
model.config.use_cache = False
ids = torch.randint(0, 32000, (1, SEQ), device="cuda")


def timeit(fn, n=5):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.time() - t0) / n


# quantise layer 0 with the hessian metric (fake-quant weights written back like gptq_pb does), tie the other layers' DENSE
# weights to it (the dense baseline is compute bound at seq 2048 and only a reference at M = 1)
l0 = model.model.layers[0]
host_blobs, meta = {}, {}
for name, lin in H.find_layers(l0).items():
    q = ptq.LowHighGPTQ(lin, "hessian", -1, 8, disable_gptq=True)
    X = torch.from_numpy(synth.calib_inputs(2, 512, lin.in_features, seed=len(host_blobs) + 1)).cuda()
    for s_ in range(X.shape[0]):
        q.add_batch(X[s_:s_ + 1])
    q.fasterquant(0.95)
    pb = q.to_pb()
    host_blobs[name] = pb.packed.blob.cpu()
    meta[name] = pb
    q.free()
for layer in model.model.layers[1:]:
    for name, lin in H.find_layers(layer).items():
        lin.weight = H.find_layers(l0)[name].weight
out = dict(model=f"llama-7b shape ({LAYERS} layers), random init", low_frac=0.95, metric="hessian",
           salient_fraction={n: round(m.packed.nnz / (m.packed.N * m.packed.K), 4) for n, m in meta.items()},
           packed_MB_per_decoder_layer=round(sum(b.numel() for b in host_blobs.values()) / 1e6, 1),
           dense_MB_per_decoder_layer=round(sum(m.in_features * m.out_features * 2 for m in meta.values()) / 1e6, 1))
tok1 = ids[:, :1]
if "prefill" in MODES:
    out["prefill_dense_fp16_ms"] = round(timeit(lambda: model(ids)) * 1e3, 2)
if "decode" in MODES:
    out["decode_dense_fp16_ms_per_token"] = round(timeit(lambda: model(tok1), 20) * 1e3, 3)
with torch.no_grad():
    ref = model(ids[:, :64]).logits[0, -4:].float()
from pb_llm_amd.packing import PackedWeight
for layer in model.model.layers:
    for name in host_blobs:
        parent = layer
        *path, leaf = name.split(".")
        for p_ in path: parent = getattr(parent, p_)
        m0 = meta[name].packed
        pk = PackedWeight(host_blobs[name].cuda(), m0.N, m0.K, m0.P, m0.G, m0.NRB, m0.flags, m0.max_nch, m0.max_nexc, m0.nnz, m0.nexc)
        setattr(parent, leaf, PBLinear(pk, None))
with torch.no_grad():
    got = model(ids[:, :64]).logits[0, -4:].float()
out["logits_rel_max_diff_vs_dense"] = float((got - ref).abs().max() / ref.abs().max())
if "prefill" in MODES:
    from pb_llm_amd import quant as Qm
    for backend in ("library", "tuned", "auto"):
        Qm.GEMM_BACKEND = backend
        t = timeit(lambda: model(ids))
        out[f"prefill_pb_{backend}_ms"] = round(t * 1e3, 2)
        out[f"prefill_tokens_per_s_pb_{backend}"] = round(SEQ / t)
    imgs = [getattr(m_.packed, "_gemm_image", (None, None))[1] for m_ in model.modules() if isinstance(m_, PBLinear)]
    out["layers_with_gemm_image"] = sum(1 for i_ in imgs if i_ is not None)
    out["gemm_image_GB"] = round(sum(i_.data.numel() for i_ in imgs if i_ is not None) / 1e9, 2)
    Qm.GEMM_BACKEND = "auto"
if "decode" in MODES:
    out["decode_pb_eager_ms_per_token"] = round(timeit(lambda: model(tok1), 20) * 1e3, 3)
    n = H.fuse_decode_(model)
    out["fused_groups"] = n
    out["decode_pb_fused_eager_ms_per_token"] = round(timeit(lambda: model(tok1), 20) * 1e3, 3)
    g = H.GraphedForward(model, tok1)
    t = timeit(lambda: g.replay(tok1), 50)
    out["decode_pb_fused_graph_ms_per_token"] = round(t * 1e3, 3)
    out["decode_tokens_per_s_pb_fused_graph"] = round(1.0 / t, 1)
if os.environ.get("BF16", "1") == "1":
    # the same model in bf16 (how HF LLaMA checkpoints ship; qat/run_qat.py:120): everything around the packed linears is cast,
    # the linears take bf16 activations through the same kernels (round 5: prepare + GEMM epilogue at prefill, inside the GEMV /
    # the fused launches at decode)
    from pb_llm_amd import quant as Qm
    for m_ in model.modules():
        if not isinstance(m_, PBLinear):
            for n_, p_ in list(m_._parameters.items()):
                if p_ is not None:
                    p_.data = p_.data.bfloat16()
    Qm.GEMM_BACKEND = "auto"
    if "prefill" in MODES:
        t = timeit(lambda: model(ids))
        out["prefill_pb_auto_bf16_ms"] = round(t * 1e3, 2)
    if "decode" in MODES:
        out["decode_pb_fused_eager_bf16_ms_per_token"] = round(timeit(lambda: model(tok1), 20) * 1e3, 3)
        gb = H.GraphedForward(model, tok1)
        t = timeit(lambda: gb.replay(tok1), 50)
        out["decode_pb_fused_graph_bf16_ms_per_token"] = round(t * 1e3, 3)
print(json.dumps(out))
