#!/usr/bin/env python3
"""Timeline of the GEMM-regime kernel from inside (build/libpbl_trace.so = tools/build_variant.sh trace -DPBL_TRACE=1).

Every wave stamps s_memrealtime (100 MHz) and s_memtime (shader clock) at entry, at the start of its main loop, at its end and
at exit, and adds up the shader cycles it is parked at the workgroup barrier / at counted vmcnt waits.  Printed per role
(consumer = MFMA waves 0..7, producer = expand waves 8..11): dispatch skew, start-up, loop, epilogue, the real shader clock,
and who waits for whom.  Usage: PBL_LIB=build/libpbl_trace.so python tools/trace_gemm.py [NxK:low_frac ...]"""
import ctypes as C, json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
os.environ.setdefault("PBL_LIB", os.path.join(REPO, "build", "libpbl_trace.so"))
from oracle import pb_oracle as O
from pb_llm_amd import synth, quant as Q, _lib

M = int(os.environ.get("PBL_BENCH_M", 2048))
NC = int(os.environ.get("PBL_TRACE_NCONS", 8))          # consumer waves of the build (then the producers)
NP = int(os.environ.get("PBL_TRACE_NPROD", 4))
SHAPES = sys.argv[1:] or ["4096x4096:0.95"]
IMG = os.environ.get("PBL_TRACE_IMG", "0") == "1"         # the round-4 kernel over the GEMM image (4 MFMA + 4 expanding waves)
if IMG:
    NC, NP = 4, 4
L = _lib.lib()
set_trace = L.pbl_debug_trace_gemm_img if IMG else L.pbl_debug_trace_gemm
set_trace.restype = None
set_trace.argtypes = [C.c_void_p]


def stats(a):
    a = np.asarray(a, dtype=np.float64)
    return dict(min=round(float(a.min()), 2), p50=round(float(np.median(a)), 2), p95=round(float(np.percentile(a, 95)), 2), max=round(float(a.max()), 2))


for t in SHAPES:
    shp, lf = t.split(":"); lf = float(lf)
    N, K = map(int, shp.split("x"))
    W = synth.llm_weight(N, K, seed=N % 97)
    mask = O.ptq_low_mask(W, lf, "magnitude", None, -1)
    r = O.ptq_rtn(W, mask, 8, -1)
    layer = Q.PBLinear.from_dense(torch.from_numpy(r["W_fq"]).half(), None, torch.from_numpy(mask), -1, r["hscale"], r["hzero"]).to("cuda:0")
    x = torch.from_numpy(synth.activations((M, K), 3, 21)).cuda()
    lst = Q.gemm_list(layer.packed)
    nwg = ((N // 16 + 7) // 8) * ((M + 255) // 256)
    trace = torch.zeros(nwg * 16 * 16, dtype=torch.int64, device="cuda:0")
    run = lambda: Q.fused_gemm_forward(layer.packed, None, x, prepared=lst)
    if IMG:
        img = Q.gemm_image(layer.packed)
        run = lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img)
        if os.environ.get("PBL_TRACE_XF", "0") == "1":            # round 6: the kernel that reads x from a fragment-major copy
            xfr = Q.x_fragments(x)
            run = lambda: Q.fused_gemm_forward(layer.packed, None, x, image=img, x_frag=xfr)
    # warm: sustained launches without the probe buffer (the stamps are skipped), then ONE traced launch in the same stream
    set_trace(None)
    import time
    t0 = time.time()
    while time.time() - t0 < float(os.environ.get("PBL_BENCH_PREHEAT_S", 1.0)):
        for _ in range(20): run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us_plain = e0.elapsed_time(e1) * 1e3 / 20
    set_trace(trace.data_ptr())
    for _ in range(3): run()                      # the last launch's stamps survive
    torch.cuda.synchronize()
    set_trace(None)
    tr = trace.cpu().numpy().reshape(nwg, 16, 16).astype(np.int64)
    tr = tr[:, :NC + NP]
    rt = tr[..., 0:8:2].astype(np.float64) * 0.01          # us (100 MHz)
    ct = tr[..., 1:8:2].astype(np.float64)
    t_first = rt[..., 0].min()
    out = dict(lib=os.path.basename(os.environ["PBL_LIB"]) + ("/img" if IMG else ""), shape=shp, low_frac=lf, M=M, us_per_call_events=round(us_plain, 1), workgroups=nwg)
    span = rt[..., 3].max() - t_first
    out["kernel_span_us"] = round(float(span), 2)
    dt = rt[..., 3] - rt[..., 0]; dc = ct[..., 3] - ct[..., 0]
    out["sclk_mhz"] = round(float(np.median(dc[dt > 0] / dt[dt > 0])), 1)
    clk = out["sclk_mhz"]
    for name, sl in (("consumer", slice(0, NC)), ("producer", slice(NC, NC + NP))):
        out[name] = dict(
            entry_after_first_us=stats(rt[:, sl, 0] - t_first),
            startup_us=stats(rt[:, sl, 1] - rt[:, sl, 0]),
            loop_us=stats(rt[:, sl, 2] - rt[:, sl, 1]),
            tail_us=stats(rt[:, sl, 3] - rt[:, sl, 2]),
            exit_after_first_us=stats(rt[:, sl, 3] - t_first),
            barrier_wait_us=stats(tr[:, sl, 8] / clk),
            vmcnt_wait_us=stats(tr[:, sl, 9] / clk))
    out["barrier_wait_by_wave_us"] = [round(float(np.median(tr[:, w, 8] / clk)), 1) for w in range(NC + NP)]
    out["vmcnt_wait_by_wave_us"] = [round(float(np.median(tr[:, w, 9] / clk)), 1) for w in range(NC + NP)]
    out["simd_by_wave"] = [int(np.bincount((tr[:, w, 10] >> 4) & 3, minlength=4).argmax()) for w in range(NC + NP)]
    # placement: XCC id and CU id spread
    xcc = tr[:, 0, 11] & 0xF
    out["wg_per_xcc"] = np.bincount(xcc, minlength=8).tolist()
    print(json.dumps(out), flush=True)
