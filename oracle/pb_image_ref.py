"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Independent numpy decoder of the GEMM image "PBI3".

Written from the layout description in pb_llm_amd/csrc/pbl_gemm_img.hip / include/pbl.h (image = [header 64 B][rbase: NRB u32]
[rtab: NRB x 128 u32][slots][levels: NRB x G x 16 u32]), NOT from the build kernel, so that a test can check
decode(build(blob)) == the fp16 weights a dense copy of the layer holds  without trusting the kernels that multiply from the image.
Never imported by the product path (pb_llm_amd/).
"""
from __future__ import annotations

import struct

import numpy as np

HDR = struct.Struct("<8I4Q")       # magic, NH, NRB, G, K, N, flags, nvmax; rtab_off, slots_off, levels_off, total
MAGIC = 0x33494250                  # "PBI3"


def read_header(img: np.ndarray) -> dict:
    f = HDR.unpack(img[:64].tobytes())
    h = dict(zip(["magic", "NH", "NRB", "G", "K", "N", "flags", "nvmax", "rtab_off", "slots_off", "levels_off", "total"], f))
    assert h["magic"] == MAGIC, "not a PBI3 image"
    return h


def decode(img: np.ndarray) -> tuple[np.ndarray, dict]:
    """image bytes (uint8) -> (W [N, K] float16: what the kernels' LDS tiles hold, statistics).

    Slot of (record r, half slab h): rtab[r][h] = offset (256-byte units from the record's start rbase[r]) | nv << 16; nv vectors of
    [64 lanes][4] u32, word w of lane l at u32 index (w >> 2) * 256 + 4 l + (w & 3).  Word 0: the lane's sign-plane dword (bit
    16 e + pos: column 2 l + e of the half slab, row pos + 8 if pos < 8 else pos - 8; 1 = level hi).  Words 1 ..: entries
    {offset << 16 | fp16 bits}, offset = row * 256 + ((2 * column in the half slab) ^ (row << 4)); applied after the plane, in any
    order (padding repeats an entry of the slot, or rewrites (row 0, column 0) with the plane's value).  Level row of (record, group):
    16 words {(hi - lo) mod 2^16 << 16 | lo} as fp16 bit patterns."""
    h = read_header(img)
    NH, NRB, G, K, N = h["NH"], h["NRB"], h["G"], h["K"], h["N"]
    assert NH == (K + 127) // 128 and NRB == (N + 15) // 16 and h["total"] <= img.size
    u32 = img[: (img.size // 4) * 4].view(np.uint32)
    rbase = u32[16:16 + NRB].astype(np.int64)
    rtab = u32[h["rtab_off"] // 4: h["rtab_off"] // 4 + NRB * 128].reshape(NRB, 128)
    levels = u32[h["levels_off"] // 4: h["levels_off"] // 4 + NRB * G * 16].reshape(NRB, G, 16)
    gs = K // G
    W = np.zeros((NRB * 16, NH * 128), np.uint16)
    lanes = np.arange(64)
    nslots = np.zeros(6, np.int64)
    run = 0
    for r in range(NRB):
        assert rbase[r] == run, "records' slots are laid out back to back"
        off_expect = 0
        for hs in range(NH):
            t = int(rtab[r, hs]); off, nv = t & 0xFFFF, t >> 16
            assert 1 <= nv <= 5 and off == off_expect, (r, hs, t)
            off_expect += 4 * nv
            nslots[nv] += 1
            base = (h["slots_off"] + (int(rbase[r]) + off) * 256) // 4
            slot = u32[base: base + nv * 256].reshape(nv, 64, 4)               # [vector][lane][component]
            words = slot.transpose(1, 0, 2).reshape(64, 4 * nv)                 # [lane][word]
            lev = levels[r, (hs * 128) // gs]
            lo = (lev & 0xFFFF).astype(np.uint16)
            hi = ((lev & 0xFFFF) + (lev >> 16)).astype(np.uint16)               # (mod 2^16)
            tile = np.zeros((16, 128), np.uint16)
            d = words[:, 0]
            for pos in range(16):
                rho = pos + 8 if pos < 8 else pos - 8
                for e in range(2):
                    bit = (d >> (16 * e + pos)) & 1
                    tile[rho, 2 * lanes + e] = np.where(bit == 1, hi[rho], lo[rho])
            ent = words[:, 1:].reshape(-1)
            o = (ent >> 16).astype(np.int64)
            row = o >> 8
            col = ((o & 0xFF) ^ ((row << 4) & 0xFF)) >> 1
            assert row.max(initial=0) < 16
            vals = (ent & 0xFFFF).astype(np.uint16)
            # every position is written with ONE value, however often it occurs (padding is idempotent)
            key = row * 128 + col
            order = np.argsort(key, kind="stable")
            ks, vs = key[order], vals[order]
            same = ks[1:] == ks[:-1]
            assert np.all(vs[1:][same] == vs[:-1][same]), "two different values for one position of a slot"
            tile[row, col] = vals
            W[16 * r: 16 * r + 16, 128 * hs: 128 * hs + 128] = tile
        run += off_expect
    assert run * 256 == h["levels_off"] - h["slots_off"]
    stats = dict(slots_by_kib=nslots[1:].tolist(), bytes=int(h["total"]))
    return W[:N, :K].view(np.float16), stats
