#!/usr/bin/env python3
"""EXPERIMENT (not product): the headline GEMV with its sign-plane phase on the MX matrix cores, end to end.

Takes the bench workload (224 packed 4096x4096 layers, low_frac 0.9, one token), re-lays the 1 KiB sign-plane tiles of every
record out ROW-MAJOR (what a format version 3 would store), prepares x as four block-scaled E4M3 terms, and runs
`pbl_proto_gemv_rowmajor` from a library built with -DPBL_PROTO_ROWMAJOR (tools/build_variant.sh proto -DPBL_PROTO_ROWMAJOR=1).
Checks the result against the shipped kernel on the same layers and times the 224-layer launch like bench.py does.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (workload construction only)
from oracle import pb_format_ref as F  # noqa: E402
from pb_llm_amd import _lib, synth  # noqa: E402
from pb_llm_amd.packing import PackedWeight  # noqa: E402
from pb_llm_amd.runtime import GroupedGemv  # noqa: E402


def k4(kb, e):
    """column (within a 128-column step) that element e of lane group kb of the 4-bit MFMA operand meets (tools/ubench_fp4_sign.hip)"""
    return 16 * (4 * (kb & 1) + 2 * (e // 16) + (kb >> 1)) + e % 16


# bit (4 n + c) of dword j of lane (kb, r)  <->  column 128 j + k4(kb, 8 c + n) of the panel; class c of that column
COL_OF = np.zeros((4, 4, 32), np.int64)          # [kb][j][bit]
CLASS_OF_COL = np.zeros(512, np.int64)
for kb_ in range(4):
    for j_ in range(4):
        for bit_ in range(32):
            n_, c_ = bit_ // 4, bit_ % 4
            col_ = 128 * j_ + k4(kb_, 8 * c_ + n_)
            COL_OF[kb_, j_, bit_] = col_
            CLASS_OF_COL[col_] = c_
A_FAC = np.array([4.0, 2.0, 1.0, -1.0])          # s = A v - B for the nibble classes {1, 1.5}, {0, 1}, {0, 2}, {1, -1}
B_FAC = np.array([5.0, 1.0, 1.0, 0.0])


def to_rowmajor(blob: np.ndarray) -> np.ndarray:
    """PBL1 (version 2) blob -> the same blob with every sign-plane tile re-laid out row-major (G == 1 layers)."""
    out = blob.copy()
    h = F.read_header(blob)
    P, NRB = h["P"], h["NRB"]
    assert h["G"] == 1
    rb_info = blob[h["rb_off_pos"]: h["rb_off_pos"] + 16 * (NRB + 1)].view(np.uint32).reshape(NRB + 1, 4)
    tiles_off = 512
    lane = np.arange(64)[:, None, None]
    di = np.arange(4)[None, :, None]
    el = np.arange(2)[None, None, :]
    col_in_panel = (128 * di + 2 * lane + el).reshape(-1)          # [64 * 4 * 2]
    weights = (np.uint64(1) << np.arange(32, dtype=np.uint64))
    for b in range(NRB):
        off = int(rb_info[b, 0]) * 16 + tiles_off
        tiles = blob[off: off + P * 1024].view(np.uint32).reshape(P, 64, 4)
        bits = np.zeros((16, P, 512), np.uint8)
        for rho in range(16):
            pos = rho + 8 if rho < 8 else rho - 8
            bb = np.stack([(tiles >> np.uint32(pos)) & 1, (tiles >> np.uint32(16 + pos)) & 1], -1).reshape(P, -1)   # [P, 512] in (l, i, e) order
            bits[rho][:, col_in_panel] = bb
        g = bits[:, :, COL_OF]                                    # [16 r, P, 4 kb, 4 j, 32 bit]
        words = (g.astype(np.uint64) * weights).sum(-1).astype(np.uint32)     # [16, P, 4, 4]
        new = np.transpose(words, (1, 2, 0, 3)).reshape(P, 64, 4)  # lane = 16 kb + r
        out[off: off + P * 1024] = np.ascontiguousarray(new).view(np.uint8).reshape(-1)
    return out


def e4m3_encode(v: np.ndarray) -> np.ndarray:
    """round to nearest even to OCP E4M3 (saturating at +-448), returns the byte codes"""
    v = np.asarray(v, np.float64)
    sgn = (np.signbit(v)).astype(np.uint8) << 7
    a = np.minimum(np.abs(v), 448.0)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -30))).astype(np.int64)
    e = np.clip(e, -6, 8)
    q = np.round(a / 2.0 ** (e - 3)).astype(np.int64)              # significand in units of 2^(e-3): 8..15 normal, 0..7 subnormal
    bump = q == 16
    e = np.where(bump, e + 1, e)
    q = np.where(bump, 8, q)
    normal = q >= 8
    code = np.where(normal, ((e + 7) << 3) | (q - 8), q)          # subnormals: exponent field 0
    code = np.where((e + 7 > 15) | ((e + 7 == 15) & (q - 8 > 6)), 0x7E, code)   # saturate at 448 (0x7E)
    return (code.astype(np.uint8) | sgn)


def e4m3_decode(c: np.ndarray) -> np.ndarray:
    c = c.astype(np.int64)
    s, e, m = c >> 7, (c >> 3) & 15, c & 7
    v = np.where(e > 0, (1 + m / 8.0) * 2.0 ** (e - 7), (m / 8.0) * 2.0 ** -6)
    return np.where(s == 1, -v, v)


def x_terms(x: np.ndarray, Kp: int) -> np.ndarray:
    """fp16 x [K] -> the kernel's per-layer buffer: terms[4][Kp] u8, scales[4][Kp / 32] u8, float X, float Xb (+ padding)"""
    K = x.shape[0]
    xf = np.zeros(Kp, np.float64)
    xf[:K] = x.astype(np.float64)
    cls = CLASS_OF_COL[np.arange(Kp) % 512]
    xp = xf * A_FAC[cls]
    Xb = float((xf * B_FAC[cls]).sum())
    X = float(xf.sum())
    terms = np.zeros((4, Kp), np.uint8)
    scales = np.full((4, Kp // 32), 127, np.uint8)
    # MX scale blocks follow the 4-bit operand's K order (tools/ubench_fp4_sign.hip): within a 128-column step the scale that
    # lane group kb supplies covers the columns that pair with kb's elements: 16-column pieces t = 4 (kb & 1) + 2 h + (kb >> 1)
    kb_of_piece = np.array([0, 2, 0, 2, 1, 3, 1, 3])
    blk = (np.arange(Kp) // 128) * 4 + kb_of_piece[(np.arange(Kp) % 128) // 16]          # scale block of every column
    order = np.argsort(blk, kind="stable")                                               # columns grouped by block (32 each)
    rem = xp[order].reshape(-1, 32).copy()
    for t in range(4):
        m = np.abs(rem).max(1)
        ex = np.where(m > 0, np.ceil(np.log2(np.maximum(m, 1e-300) / 448.0)), 0).astype(np.int64)
        ex = np.clip(ex, -127, 127)
        sc = 2.0 ** ex
        codes = e4m3_encode(rem / sc[:, None])
        terms[t][order] = codes.reshape(-1)
        scales[t] = (ex + 127).astype(np.uint8)
        rem = rem - e4m3_decode(codes) * sc[:, None]
    resid = float(np.abs(rem).max())
    buf = np.concatenate([terms.reshape(-1), scales.reshape(-1), np.array([X, Xb, 0, 0], np.float32).view(np.uint8),
                          np.zeros(16, np.uint8)])
    return buf, resid


def main():
    dev = "cuda:0"
    Lc, N, K, lf, distinct = 224, 4096, 4096, 0.9, 4
    lib = C.CDLL(os.path.join(REPO, "build", os.environ.get("PBL_PROTO_LIB", "libpbl_proto.so")))
    vp, u32 = C.c_void_p, C.c_uint32
    lib.pbl_proto_gemv_rowmajor.restype = C.c_int
    lib.pbl_proto_gemv_rowmajor.argtypes = [vp, vp, vp, vp, C.c_int, u32, u32, u32, C.c_int, C.c_int, vp]
    base = bench.build_base_layers(distinct, N, K, lf, seed0=1000)
    full = [bench.pack_slice(b) for b in base]
    t0 = time.time()
    rm = [PackedWeight.from_blob(torch.from_numpy(to_rowmajor(p.blob.numpy()))) for p in full]
    t_conv = time.time() - t0
    xs_host = [synth.activations((1, K), 5000 + i, 21) for i in range(Lc)]
    # shipped kernel on the version-2 blobs
    g2 = GroupedGemv([full[i % distinct].to(dev) for i in range(Lc)], None, M=1, device=dev)
    for t, xh in zip(g2.x, xs_host):
        t.copy_(torch.from_numpy(xh))
    y2 = [y.clone() for y in g2.launch()]
    torch.cuda.synchronize()
    # prototype on the row-major blobs
    rmd = [rm[i % distinct].to(dev) for i in range(Lc)]
    Kp = rmd[0].P * 512
    resid = 0.0
    xt = []
    for xh in xs_host:
        buf, r = x_terms(xh[0], Kp)
        resid = max(resid, r)
        xt.append(torch.from_numpy(buf).to(dev))
    structs = (_lib.PblLayer * Lc)(*[p.layer_struct(None) for p in rmd])
    layers_dev = torch.from_numpy(np.frombuffer(bytes(structs), dtype=np.uint8).copy()).to(dev)
    y = torch.empty(Lc, N, dtype=torch.float16, device=dev)
    x_ptrs = torch.tensor([t.data_ptr() for t in g2.x], dtype=torch.int64, device=dev)
    xt_ptrs = torch.tensor([t.data_ptr() for t in xt], dtype=torch.int64, device=dev)
    y_ptrs = torch.tensor([y[i].data_ptr() for i in range(Lc)], dtype=torch.int64, device=dev)
    max_nch = max(p.max_nch for p in rmd)
    st = torch.cuda.current_stream().cuda_stream
    out = {"convert_s_per_layer": round(t_conv / distinct, 2), "x_terms_max_residual": resid}
    for wpb in (4, 8):
        def launch():
            rc = lib.pbl_proto_gemv_rowmajor(layers_dev.data_ptr(), x_ptrs.data_ptr(), xt_ptrs.data_ptr(), y_ptrs.data_ptr(), Lc,
                                             rmd[0].NRB, K, max_nch, wpb, 0, st)
            assert rc == 0, rc
        y.zero_()
        launch()
        torch.cuda.synchronize()
        err = max(float((y[i].float() - y2[i][0].float()).abs().max()) for i in range(Lc))
        ref = max(float(y2[i][0].float().abs().max()) for i in range(Lc))
        # pre-heat like bench.py, then 20 timed launches (the driver's count) and 200
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 2.0:
            for _ in range(32):
                launch()
            torch.cuda.synchronize()
        res = {"max_abs_diff_vs_shipped": err, "max_abs_y": ref}
        for steps in (20, 200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                launch()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / steps
            b_alg = sum(p.algorithmic_bytes(1) for p in rmd)
            res[f"us_per_launch_{steps}"] = round(us, 1)
            res[f"roofline_frac_{steps}"] = round(b_alg / us / 1e3 / 8000.0, 4)
        out[f"wpb{wpb}"] = res
    # the shipped kernel, same process, same timing loop
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 2.0:
        for _ in range(32):
            g2.launch()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        g2.launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    out["shipped"] = {"us_per_launch_200": round(us, 1), "roofline_frac_200": round(g2.algorithmic_bytes() / us / 1e3 / 8000.0, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
