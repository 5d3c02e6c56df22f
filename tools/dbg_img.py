import sys, os, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from cfg_shapes import hessian_layer, LLAMA7B
from pb_llm_amd import quant as Q, synth
N, K = LLAMA7B["down_proj"]
for seed in (302, 301):
    W, mask, r = hessian_layer(N, K, 0.95, seed=seed)
    layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to("cuda:0")
    p = layer.packed
    x = torch.from_numpy(synth.activations((2048, K), 77, 21)).cuda()
    img = Q.gemm_image(p)
    print("seed", seed, "colmax max", img.max_entries, "nv hist", np.bincount(np.digitize(list(img.colmax), [193, 449, 705, 961]), minlength=5).tolist())
    y_img = Q.fused_gemm_forward(p, None, x, image=img)
    y_old = Q.fused_gemm_forward(p, None, x)
    bad = (y_img != y_old)
    print("mismatching elements", int(bad.sum()), "of", bad.numel())
    if bad.any():
        rows = bad.any(0).nonzero().flatten().cpu().numpy(); toks = bad.any(1).nonzero().flatten().cpu().numpy()
        print("rows", len(rows), rows[:40], "records", np.unique(rows // 16)[:40], "row tiles", np.unique(rows // 128))
        print("tokens", len(toks), toks[:10], toks[-5:])
        d = (y_img.float() - y_old.float()).abs()
        print("max abs diff", float(d.max()), "ref max", float(y_old.float().abs().max()))
        # per column-of-K effect cannot be seen directly; per row count of salients in hot columns
        c = (~mask).reshape(N // 16, 16, K // 128, 128).sum(axis=(1, 3))
        recs = np.unique(rows // 16)
        print("entries per slot for a bad record (max over h):", c[recs[0]].max(), "argmax h", c[recs[0]].argmax(), " overall max", c.max())
