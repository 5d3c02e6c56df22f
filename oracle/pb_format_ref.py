"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Independent numpy decoder of the PBL1 blob.

Written from the format description in include/pbl.h, NOT from the C++ packer, so
that tests can check  decode(pack(W)) == W  without trusting either side, and
can evaluate the packed-form GEMV on the CPU:  y = decode(blob) @ x.
Never imported by the product path (pb_llm_amd/).
"""
from __future__ import annotations

import struct

import numpy as np

HDR = struct.Struct("<10I3Q4I")  # pbl_blob_header, 80 bytes


def _a16(x):
    return (x + 15) & ~15


def _a128(x):
    return (x + 127) & ~127


def read_header(blob: np.ndarray) -> dict:
    f = HDR.unpack(blob[:80].tobytes())
    keys = ["magic", "version", "N", "K", "P", "G", "NRB", "flags", "max_nch", "max_nexc",
            "nnz", "nexc", "blob_bytes", "rb_off_pos", "r0", "r1", "r2"]
    h = dict(zip(keys, f))
    assert h["magic"] == 0x314C4250 and h["version"] == 2, "bad PBL1 (version 2) blob"
    assert h["blob_bytes"] == blob.size
    return h


def decode(blob: np.ndarray) -> np.ndarray:
    """PBL1 blob (uint8 array) -> dense fp32 [N, K]."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    h = read_header(blob)
    N, K, P, G, NRB = h["N"], h["K"], h["P"], h["G"], h["NRB"]
    gs = K // G
    rb_info = blob[h["rb_off_pos"]: h["rb_off_pos"] + 16 * (NRB + 1)].view(np.uint32).reshape(NRB + 1, 4)
    rb_off = rb_info[:, 0].astype(np.int64) * 16
    W = np.zeros((NRB * 16, P * 512), np.float32)

    # column owned by (lane l, dword i, element e) of panel p:  512p + 128i + 2l + e
    lane = np.arange(64)[:, None, None]
    di = np.arange(4)[None, :, None]
    el = np.arange(2)[None, None, :]
    col_in_panel = 128 * di + 2 * lane + el                     # [64, 4, 2]

    for b in range(NRB):
        rec = blob[rb_off[b]: rb_off[b + 1]]
        nfull, ntail, nexc, off_sal = struct.unpack("<4I", rec[:16].tobytes())
        assert (nfull, ntail, nexc) == tuple(int(v) for v in rb_info[b, 1:]), "rb_info disagrees with record header"
        rowinfo = rec[16:144].view(np.dtype([("start", "<u2"), ("nfull", "<u2"), ("tailidx", "<u2"),
                                             ("ntail", "u1"), ("pad", "u1")]))
        params = rec[144:400].view(np.float32).reshape(16, 4)
        ghl = rec[400:400 + 16 * G * 8].view(np.float32).reshape(16, G, 2) if G > 1 else None
        tiles_off = off_sal - P * 1024
        assert tiles_off == _a128(400 + (128 * G if G > 1 else 0)) and rb_off[b] % 128 == 0
        tiles = rec[tiles_off:off_sal].view(np.uint32).reshape(P, 64, 4)
        for rho in range(16):
            pos = rho + 8 if rho < 8 else rho - 8
            # bit (e*16 + pos) of each dword
            bits = np.stack([(tiles >> np.uint32(pos)) & 1, (tiles >> np.uint32(16 + pos)) & 1], -1)  # [P,64,4,2]
            cols = (np.arange(P)[:, None, None, None] * 512 + col_in_panel[None]).reshape(-1)
            row_bits = np.zeros(P * 512, np.uint8)
            row_bits[cols] = bits.reshape(-1)
            if G > 1:
                hi_c = np.repeat(ghl[rho, :, 0], gs)
                lo_c = np.repeat(ghl[rho, :, 1], gs)
                hi_c = np.pad(hi_c, (0, P * 512 - K))
                lo_c = np.pad(lo_c, (0, P * 512 - K))
            else:
                hi_c = np.full(P * 512, params[rho, 0], np.float32)
                lo_c = np.full(P * 512, params[rho, 1], np.float32)
            W[b * 16 + rho] = np.where(row_bits == 1, hi_c, lo_c)
        nch = nfull + ntail
        s = off_sal
        col0 = rec[s: s + 2 * nch].view(np.uint16).astype(np.int64)
        s += _a128(2 * nch)
        delta = rec[s: s + 16 * nch].reshape(nch, 16).astype(np.int64)
        s += _a128(16 * nch)
        code = rec[s: s + 16 * nch].reshape(nch, 16).astype(np.float32)
        s += _a128(16 * nch)
        tailcnt = rec[s: s + ntail].astype(np.int64)
        s += _a16(ntail)
        if h["flags"] & 0x3:   # HAS_GROUPS | SAL_F16: per-chunk row ids
            crow = rec[s: s + nch].astype(np.int64)
            s += _a16(nch)
            for rho in range(16):   # crow must agree with rowinfo
                ri = rowinfo[rho]
                assert (crow[int(ri["start"]): int(ri["start"]) + int(ri["nfull"])] == rho).all()
                assert (crow[nfull + int(ri["tailidx"]): nfull + int(ri["tailidx"]) + int(ri["ntail"])] == rho).all()
        exc = rec[s: s + 8 * nexc].view(np.dtype([("col", "<u2"), ("row", "<u2"), ("value", "<f4")]))
        s = _a16(s + 8 * nexc)
        NS = (K + 255) // 256
        assert h["flags"] & 0x8, "version 2 blobs carry the slab index (PBL_FLAG_SLABS)"
        slab = rec[s: s + 16 * NS * 4].view(np.uint32).reshape(16, NS).astype(np.int64)
        assert _a128(s + 16 * NS * 4) == rec.size, "record size disagrees with the layout"
        assert not (delta & 1).any(), "deltas are stored doubled"
        cols = col0[:, None] + np.cumsum(delta // 2, axis=1)
        sal16 = bool(h["flags"] & 0x2)   # PBL_FLAG_SAL_F16: values went through an fp16 round trip
        # slab index, re-derived from the chunk lists: per (row, 256-column slab) the cumulative count of chunks that
        # START left of the slab's right edge, and whether a chunk from the left reaches into the slab
        for rho in range(16):
            ri = rowinfo[rho]
            fidx = np.arange(int(ri["start"]), int(ri["start"]) + int(ri["nfull"]))
            tidx = nfull + np.arange(int(ri["tailidx"]), int(ri["tailidx"]) + int(ri["ntail"]))
            for q in range(NS):
                lo_c, hi_c = 256 * q, 256 * (q + 1)
                e = int(slab[rho, q])
                assert (e & 0xFFFF) == int((col0[fidx] < hi_c).sum()) and ((e >> 16) & 0xFF) == int((col0[tidx] < hi_c).sum())
                assert ((e >> 24) & 1) == int(((col0[fidx] < lo_c) & (cols[fidx, -1] >= lo_c)).any())
                assert ((e >> 25) & 1) == int(((col0[tidx] < lo_c) & (cols[tidx, -1] >= lo_c)).any())
                assert (e >> 26) == 0

        def deq(ss, code_vals, sz):
            v = (ss * (code_vals - sz)).astype(np.float32)
            return v.astype(np.float16).astype(np.float32) if sal16 else v

        for rho in range(16):
            ri = rowinfo[rho]
            ss, sz = np.float32(params[rho, 2]), np.float32(params[rho, 3])
            for ch in range(int(ri["start"]), int(ri["start"]) + int(ri["nfull"])):
                W[b * 16 + rho, cols[ch]] = deq(ss, code[ch], sz)
            for t in range(int(ri["tailidx"]), int(ri["tailidx"]) + int(ri["ntail"])):
                ch, n = nfull + t, tailcnt[t]
                W[b * 16 + rho, cols[ch, :n]] = deq(ss, code[ch, :n], sz)
                if h["flags"] & 0x4:   # PBL_FLAG_TAIL_REPEAT: padding repeats the last entry with step 0
                    assert 1 <= n < 16 and (delta[ch, n:] == 0).all() and (code[ch, n:] == code[ch, n - 1]).all()
        for e in exc:
            W[b * 16 + int(e["row"]), int(e["col"])] = e["value"]
    return W[:N, :K].copy()


def stats(blob: np.ndarray) -> dict:
    h = read_header(np.ascontiguousarray(blob, dtype=np.uint8))
    return h
