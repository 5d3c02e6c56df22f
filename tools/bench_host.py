#!/usr/bin/env python3
"""Host cost of one eager PBLinear call (M = 1, 4096 x 4096, low_frac 0.9): the native dispatcher (torch.ops.pbllm_native.linear,
csrc/pbl_torch.cpp) against the ctypes path (PBL_NATIVE=0), a dense nn.Linear for scale.  The kernel itself takes ~5 us
back to back, so a stream of calls is host bound on every path: us per call = host time per call.  Call sites this is about:
every `module(x)` of qat/run_qat.py:45-66, utils.py:103-123, gptq_pb/eval_ppl_utils.py:55-64 at batch 1."""
import json, os, subprocess, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)


def measure():
    from oracle import pb_oracle as O
    from pb_llm_amd import synth, quant as Q, _lib
    N = K = 4096
    W = synth.llm_weight(N, K, seed=3)
    mask = O.ptq_low_mask(W, 0.9, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to("cuda:0")
    dense = torch.nn.Linear(K, N, bias=False).half().to("cuda:0")
    res = {"native_dispatcher": _lib.native_linear() is not None}
    for M in (1, 4, 16):
        x = torch.from_numpy(synth.activations((M, K), 5, 21)).to("cuda:0")
        xb = x.bfloat16()
        xf = x.float() * 1.0009765625

        def torch_split():                      # fp32 x the way the route did it before round 5's split / join kernels: torch ops around ONE kernel call
            hi = xf.half()
            yy = Q.pb_linear_forward(layer.packed, None, torch.cat([hi, (xf - hi.float()).half()], 0), out_f32=True)
            return yy[:M] + yy[M:]
        for name, fn in (("pb", lambda: layer(x)), ("pb_bf16", lambda: layer(xb)), ("pb_f32", lambda: layer(xf)), ("pb_f32_torch_split", torch_split),
                         ("dense", lambda: dense(x))):
            with torch.no_grad():
                for _ in range(200): fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 3000
                for _ in range(n): fn()
                torch.cuda.synchronize()
                res[f"{name}_M{M}_us_per_call"] = round((time.perf_counter() - t0) / n * 1e6, 2)
    return res


if __name__ == "__main__":
    if os.environ.get("PBL_HOST_CHILD"):
        print(json.dumps(measure()))
    else:
        out = {}
        for tag, env in (("native", {}), ("ctypes", {"PBL_NATIVE": "0"})):
            e = dict(os.environ, PBL_HOST_CHILD="1", **env)
            out[tag] = json.loads(subprocess.check_output([sys.executable, os.path.abspath(__file__)], env=e).decode().strip().splitlines()[-1])
        print(json.dumps(out))
