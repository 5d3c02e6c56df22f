#!/usr/bin/env python3
"""BASELINE config 2: llama-7b-shaped full model forward (HF LlamaForCausalLM, random init, synthetic tokens), seq 2048,
batch 1, on one MI355X -- dense fp16 weights (what the reference evaluates after gptq_pb writes its fake-quant weights back)
vs every decoder Linear swapped for a PBLinear (packed 1-bit + 8-bit salient weights; device unpack + library GEMM at this M).
To keep the setup short the 7 linears of ONE decoder layer are quantised (RTN branch, low_frac 0.95, magnitude metric, GPU
pipeline) and packed, and the 32 layers share them; prefill is GEMM-bound, so weight reuse in the caches does not matter."""
import json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from transformers import LlamaConfig, LlamaForCausalLM
from pb_llm_amd import harness as H, ptq

SEQ = int(os.environ.get("SEQ", 2048))
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                  num_key_value_heads=32, vocab_size=32000, max_position_embeddings=4096)
torch.manual_seed(0)
with torch.device("cuda"):
    model = LlamaForCausalLM(cfg).half().eval()
model.config.use_cache = False
ids = torch.randint(0, 32000, (1, SEQ), device="cuda")


def timeit(m, n=5):
    with torch.no_grad():
        m(ids); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n): m(ids)
        torch.cuda.synchronize()
    return (time.time() - t0) / n


# quantise layer 0 (fake-quant weights written back like gptq_pb does), tie the other layers to it
l0 = model.model.layers[0]
packed = {}
for name, lin in H.find_layers(l0).items():
    q = ptq.LowHighGPTQ(lin, "magnitude", -1, 8, disable_gptq=True)
    q.add_batch(torch.zeros(1, 8, lin.in_features, device="cuda"))      # RTN + magnitude: the Hessian is not used
    q.H += torch.eye(lin.in_features, device="cuda")
    q.fasterquant(0.95)
    packed[name] = q.to_pb().to("cuda")
for layer in model.model.layers[1:]:
    for name, lin in H.find_layers(layer).items():
        lin.weight = H.find_layers(l0)[name].weight                        # share the dense fake-quant weights too
t_dense = timeit(model)
with torch.no_grad():
    ref = model(ids).logits[0, -4:].float()
for layer in model.model.layers:
    for name in packed:
        parent = layer
        *path, leaf = name.split(".")
        for p_ in path: parent = getattr(parent, p_)
        setattr(parent, leaf, packed[name])
t_pb = timeit(model)
with torch.no_grad():
    out = model(ids).logits[0, -4:].float()
rel = float((out - ref).abs().max() / ref.abs().max())
print(json.dumps(dict(model="llama-7b shape, random init", seq=SEQ, batch=1,
                      dense_fp16_forward_ms=round(t_dense * 1e3, 1), pb_packed_forward_ms=round(t_pb * 1e3, 1),
                      tokens_per_s_pb=round(SEQ / t_pb), tokens_per_s_dense=round(SEQ / t_dense),
                      logits_rel_max_diff=rel,
                      packed_MB_per_decoder_layer=round(sum(p.packed.nbytes for p in packed.values()) / 1e6, 1),
                      dense_MB_per_decoder_layer=round(sum(p.in_features * p.out_features * 2 for p in packed.values()) / 1e6, 1))))
