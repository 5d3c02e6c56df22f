// pbl_host.cpp -- host side of libpbl.so: PBL1 packer / unpacker / blob validation.
//
// The reference has no packed format (1-bit weights live as dense fp16,
// gptq_pb/gptq.py:180-184; its intended storage is only *accounted* in
// quant/outlier_quantizer.py:116-122).  This file defines ours; the layout is
// documented in include/pbl.h and DESIGN.md.
#include "../../include/pbl.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline size_t align16(size_t x) { return (x + 15) & ~size_t(15); }
inline size_t align128(size_t x) { return (x + 127) & ~size_t(127); }

// Kernel-form dequant used for the representability check and by the unpacker:
// the value HighQuantizer produces, scale * (q - zero) (gptq_pb/high_quant.py:6-8).
inline float dequant(float sscale, float szero, int q) { return sscale * (float(q) - szero); }
// the same value after the checkpoint's fp16 round trip (round-to-nearest-even)
inline float dequant_f16(float sscale, float szero, int q) { return float(_Float16(dequant(sscale, szero, q))); }
// fp16-checkpoint entries only: the matrix-core kernels keep MINUS the weight, fl16((-scale) * (q - zero)), in their fp16 tile
// and recognise an entry by "any bit set" (a weight of value 0 is -0 there).  An entry whose negated product would be +0 --
// zero or negative scale, or a product that underflows from the positive side -- is therefore not coded (it becomes an exception).
inline bool sal16_storable(float sscale, float szero, int q) {
    const _Float16 u = _Float16((-sscale) * (float(q) - szero));
    uint16_t bits;
    std::memcpy(&bits, &u, 2);
    return bits != 0;
}

struct Entry { uint16_t col; uint8_t code; };

struct RecBuild {
    std::vector<uint16_t> col0_full, col0_tail;
    std::vector<uint8_t> delta_full, code_full, delta_tail, code_tail, tailcnt, crow_full, crow_tail;
    std::vector<pbl_exception> exc;
    std::vector<uint32_t> slab;     // [16][NS]
    pbl_rowinfo ri[16];
    void clear() {
        slab.clear();
        col0_full.clear(); col0_tail.clear(); delta_full.clear(); code_full.clear();
        delta_tail.clear(); code_tail.clear(); tailcnt.clear(); exc.clear(); crow_full.clear(); crow_tail.clear();
        std::memset(ri, 0, sizeof(ri));
    }
};

// Bit index inside dword i of a lane for (row-in-block rho, element e).
inline int bit_index(int rho, int e) {
    int pos = rho < 8 ? rho + 8 : rho - 8;
    return e * 16 + pos;
}

size_t record_fixed_bytes(uint32_t P, uint32_t G) { return size_t(PBL_TILES_OFF(G)) + size_t(P) * 1024; }

size_t record_sal_bytes(size_t nch, size_t ntail, size_t nexc, bool has_crow, uint32_t K) {
    return PBL_SAL_BYTES(uint32_t(nch), uint32_t(ntail), uint32_t(nexc), has_crow, K);
}

}  // namespace

extern "C" {

const char* pbl_status_string(int s) {
    switch (s) {
        case PBL_OK: return "ok";
        case PBL_ERR_INVALID_ARG: return "invalid argument";
        case PBL_ERR_BAD_BLOB: return "not a valid PBL1 blob";
        case PBL_ERR_UNSUPPORTED: return "unsupported shape or option";
        case PBL_ERR_MISALIGNED: return "pointer not 16-byte aligned";
        case PBL_ERR_CAPACITY: return "output buffer too small";
        case PBL_ERR_LAUNCH: return "kernel launch failed";
        case PBL_ERR_NOT_REPRESENTABLE: return "value not representable in the packed format";
        default: return "unknown status";
    }
}

int pbl_version(void) { return PBL_VERSION; }

}  // extern "C" (reopened below)

namespace {

// Everything one 16-row record contributes to the blob, built independently of every other record.
struct RecImage {
    std::vector<uint8_t> bytes;     // the record as it lies in the blob (empty in a size query)
    size_t rec_bytes = 0;
    uint32_t nfull = 0, ntail = 0, nexc = 0;
    uint64_t nnz = 0;
    int status = PBL_OK;
};

struct PackArgs {
    const float *W, *hi, *lo, *sscale, *szero;
    const uint8_t* sal_mask;
    uint32_t N, K, G, gs, P;
    bool sal16, want_bytes;
    size_t fixed, tiles_off;
};

void build_record(const PackArgs& a, uint32_t b, RecImage& out) {
    const uint32_t N = a.N, K = a.K, G = a.G, P = a.P;
    RecBuild rb;
    rb.clear();
    std::vector<Entry> ents;
    std::vector<uint32_t> tile(size_t(P) * 256, 0u);
    pbl_rowparams params[16];
    std::memset(params, 0, sizeof(params));
    std::vector<float> ghl;
    if (G > 1) ghl.assign(size_t(16) * G * 2, 0.f);
    uint64_t nnz = 0;

    for (int rho = 0; rho < 16; ++rho) {
        const uint32_t r = b * 16 + rho;
        rb.ri[rho].start = uint16_t(rb.col0_full.size());
        rb.ri[rho].tailidx = uint16_t(rb.col0_tail.size());
        if (r >= N) continue;
        const float* w = a.W + size_t(r) * K;
        const float ss = a.sscale ? a.sscale[r] : 0.f, sz = a.szero ? a.szero[r] : 0.f;
        params[rho] = {a.hi[size_t(r) * G], a.lo[size_t(r) * G], ss, sz};
        if (G > 1)
            for (uint32_t g = 0; g < G; ++g) {
                ghl[(size_t(rho) * G + g) * 2 + 0] = a.hi[size_t(r) * G + g];
                ghl[(size_t(rho) * G + g) * 2 + 1] = a.lo[size_t(r) * G + g];
            }
        ents.clear();
        for (uint32_t c = 0; c < K; ++c) {
            const float v = w[c];
            const uint32_t g = c / a.gs;
            const float h = a.hi[size_t(r) * G + g], l = a.lo[size_t(r) * G + g];
            const bool forced = a.sal_mask && a.sal_mask[size_t(r) * K + c];
            int bit;
            if (!forced && v == h) bit = 1;
            else if (!forced && v == l) bit = 0;
            else {
                bit = 1;  // sparse entries correct against `hi`
                bool coded = false;
                if (a.sscale && ss != 0.f && std::isfinite(v)) {
                    float qf = std::nearbyint(v / ss + sz);
                    for (int dq = 0; dq <= 2 && !coded; ++dq) {
                        int q = int(qf) + (dq == 0 ? 0 : (dq == 1 ? 1 : -1));
                        if (q >= 0 && q <= 255 && (a.sal16 ? dequant_f16(ss, sz, q) == v && sal16_storable(ss, sz, q) : dequant(ss, sz, q) == v)) {
                            ents.push_back({uint16_t(c), uint8_t(q)});
                            coded = true;
                        }
                    }
                }
                if (!coded) rb.exc.push_back({uint16_t(c), uint16_t(rho), v});
            }
            if (bit) {
                const uint32_t p = c / 512, cc = c % 512, i = cc / 128, l2 = (cc % 128) / 2, e = cc & 1;
                tile[(size_t(p) * 64 + l2) * 4 + i] |= 1u << bit_index(rho, int(e));
            }
        }
        nnz += ents.size();
        // greedy chunking: close at 16 entries or when the next column step exceeds PBL_MAX_GAP
        size_t i = 0;
        uint16_t nfull = 0, ntail = 0;
        const uint32_t NS = PBL_NSLABS(K);
        if (rb.slab.empty()) rb.slab.assign(size_t(16) * NS, 0u);
        std::vector<uint32_t> fe(NS, 0u), te(NS, 0u), fb(NS, 0u), tb(NS, 0u);
        while (i < ents.size()) {
            size_t j = i + 1;
            while (j < ents.size() && j - i < 16 && ents[j].col - ents[j - 1].col <= PBL_MAX_GAP) ++j;
            const size_t cnt = j - i;
            {   // slab index: the chunk covers columns [first, last]
                const uint32_t s0 = ents[i].col / PBL_SLAB_COLS, s1 = ents[j - 1].col / PBL_SLAB_COLS;
                const bool full = cnt == 16;
                for (uint32_t s = s0; s < NS; ++s) (full ? fe : te)[s] += 1;          // first column < 256 (s+1)
                for (uint32_t s = s0 + 1; s <= s1; ++s) (full ? fb : tb)[s] = 1;     // reaches in from the left
            }
            uint8_t d[16] = {0}, q[16] = {0};
            for (size_t k = 0; k < cnt; ++k) {
                d[k] = k ? uint8_t(2 * (ents[i + k].col - ents[i + k - 1].col)) : 0;  // byte step in the fp16 x tile
                q[k] = ents[i + k].code;
            }
            for (size_t k = cnt; k < 16; ++k) q[k] = q[cnt - 1];   // PBL_FLAG_TAIL_REPEAT: padding repeats the last entry
            if (cnt == 16) {
                rb.col0_full.push_back(ents[i].col);
                rb.delta_full.insert(rb.delta_full.end(), d, d + 16);
                rb.code_full.insert(rb.code_full.end(), q, q + 16);
                rb.crow_full.push_back(uint8_t(rho));
                ++nfull;
            } else {
                rb.col0_tail.push_back(ents[i].col);
                rb.delta_tail.insert(rb.delta_tail.end(), d, d + 16);
                rb.code_tail.insert(rb.code_tail.end(), q, q + 16);
                rb.tailcnt.push_back(uint8_t(cnt));
                rb.crow_tail.push_back(uint8_t(rho));
                ++ntail;
            }
            i = j;
        }
        if (ntail > 255) { out.status = PBL_ERR_UNSUPPORTED; return; }
        rb.ri[rho].nfull = nfull;
        rb.ri[rho].ntail = uint8_t(ntail);
        for (uint32_t s = 0; s < NS; ++s) rb.slab[size_t(rho) * NS + s] = fe[s] | (te[s] << 16) | (fb[s] << 24) | (tb[s] << 25);
    }
    if (rb.slab.empty()) rb.slab.assign(size_t(16) * PBL_NSLABS(K), 0u);

    const size_t nfull = rb.col0_full.size(), ntail = rb.col0_tail.size(), nch = nfull + ntail;
    if (nch > 65535) { out.status = PBL_ERR_UNSUPPORTED; return; }
    const bool has_crow = G > 1 || a.sal16;
    out.rec_bytes = a.fixed + record_sal_bytes(nch, ntail, rb.exc.size(), has_crow, K);
    out.nfull = uint32_t(nfull); out.ntail = uint32_t(ntail); out.nexc = uint32_t(rb.exc.size()); out.nnz = nnz;
    if (!a.want_bytes) return;
    out.bytes.assign(out.rec_bytes, 0);
    uint8_t* rec = out.bytes.data();
    pbl_rec_header h = {uint32_t(nfull), uint32_t(ntail), uint32_t(rb.exc.size()), uint32_t(a.fixed)};
    std::memcpy(rec, &h, sizeof(h));
    std::memcpy(rec + 16, rb.ri, sizeof(rb.ri));
    std::memcpy(rec + 16 + 128, params, sizeof(params));
    if (G > 1) std::memcpy(rec + PBL_REC_GHL_OFF, ghl.data(), ghl.size() * 4);
    std::memcpy(rec + a.tiles_off, tile.data(), tile.size() * 4);
    uint8_t* s = rec + a.fixed;
    const uint32_t nch32 = uint32_t(nch), ntail32 = uint32_t(ntail);
    uint16_t* col0 = reinterpret_cast<uint16_t*>(s);
    for (size_t k = 0; k < nfull; ++k) col0[k] = rb.col0_full[k];
    for (size_t k = 0; k < ntail; ++k) col0[nfull + k] = rb.col0_tail[k];
    uint8_t* dl = s + PBL_SAL_DELTA_OFF(nch32);
    if (nfull) std::memcpy(dl, rb.delta_full.data(), nfull * 16);
    if (ntail) std::memcpy(dl + nfull * 16, rb.delta_tail.data(), ntail * 16);
    uint8_t* cd = s + PBL_SAL_CODE_OFF(nch32);
    if (nfull) std::memcpy(cd, rb.code_full.data(), nfull * 16);
    if (ntail) std::memcpy(cd + nfull * 16, rb.code_tail.data(), ntail * 16);
    if (ntail) std::memcpy(s + PBL_SAL_TAILCNT_OFF(nch32), rb.tailcnt.data(), ntail);
    if (has_crow) {
        uint8_t* cr = s + PBL_SAL_CROW_OFF(nch32, ntail32);
        if (nfull) std::memcpy(cr, rb.crow_full.data(), nfull);
        if (ntail) std::memcpy(cr + nfull, rb.crow_tail.data(), ntail);
    }
    if (!rb.exc.empty())
        std::memcpy(s + PBL_SAL_EXC_OFF(nch32, ntail32, has_crow), rb.exc.data(), rb.exc.size() * sizeof(pbl_exception));
    std::memcpy(s + PBL_SAL_SLAB_OFF(nch32, ntail32, uint32_t(rb.exc.size()), has_crow), rb.slab.data(), rb.slab.size() * 4);
}

}  // namespace

extern "C" {

int pbl_pack_dense_f32(const float* W, uint32_t N, uint32_t K, uint32_t G,
                       const float* hi, const float* lo, const float* sscale, const float* szero,
                       const uint8_t* sal_mask, uint32_t flags, void* out, size_t cap, size_t* out_bytes) {
    if (flags & ~PBL_FLAG_SAL_F16) return PBL_ERR_INVALID_ARG;
    const bool sal16 = flags & PBL_FLAG_SAL_F16;
    if (!W || !hi || !lo || !out_bytes || N == 0 || K == 0 || G == 0) return PBL_ERR_INVALID_ARG;
    if (K > 32767 || N > (1u << 24)) return PBL_ERR_UNSUPPORTED;
    if (G > 1 && (K % G != 0 || (K / G) % 128 != 0)) return PBL_ERR_UNSUPPORTED;
    const uint32_t P = (K + PBL_PANEL_COLS - 1) / PBL_PANEL_COLS;
    const uint32_t NRB = (N + 15) / 16;
    const size_t rboff_pos = sizeof(pbl_blob_header);
    const size_t rec0 = align128(rboff_pos + size_t(NRB + 1) * sizeof(pbl_rec_info));
    uint8_t* blob = static_cast<uint8_t*>(out);

    PackArgs a;
    a.W = W; a.hi = hi; a.lo = lo; a.sscale = sscale; a.szero = szero; a.sal_mask = sal_mask;
    a.N = N; a.K = K; a.G = G; a.gs = K / G; a.P = P; a.sal16 = sal16; a.want_bytes = blob != nullptr;
    a.fixed = record_fixed_bytes(P, G);
    a.tiles_off = a.fixed - size_t(P) * 1024;

    // records are independent: build their images on all host cores, then lay them out in order.  The blob is
    // byte-identical for any thread count.
    std::vector<RecImage> imgs(NRB);
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char* env = std::getenv("PBL_PACK_THREADS")) nthreads = unsigned(std::max(1, std::atoi(env)));
    nthreads = std::max(1u, std::min(std::min(nthreads, 64u), NRB / 8u + 1u));
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
        for (uint32_t b = next.fetch_add(1); b < NRB; b = next.fetch_add(1)) build_record(a, b, imgs[b]);
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(work);
    } catch (...) {
        // thread limit reached: the threads that did start (and this one) share the work
    }
    work();
    for (auto& t : pool) t.join();

    std::vector<pbl_rec_info> rb_info(NRB + 1);
    size_t cur = rec0;
    uint64_t nnz = 0, nexc_total = 0;
    uint32_t max_nch = 0, max_nexc = 0;
    for (uint32_t b = 0; b < NRB; ++b) {
        const RecImage& im = imgs[b];
        if (im.status != PBL_OK) return im.status;
        rb_info[b] = {uint32_t(cur / 16), im.nfull, im.ntail, im.nexc};
        max_nch = std::max(max_nch, im.nfull + im.ntail);
        max_nexc = std::max(max_nexc, im.nexc);
        nexc_total += im.nexc;
        nnz += im.nnz;
        if (blob) {
            if (cur + im.rec_bytes > cap) return PBL_ERR_CAPACITY;
            std::memcpy(blob + cur, im.bytes.data(), im.rec_bytes);
        }
        cur += im.rec_bytes;
    }
    rb_info[NRB] = {uint32_t(cur / 16), 0, 0, 0};
    *out_bytes = cur;
    if (blob) {
        if (cur > cap) return PBL_ERR_CAPACITY;
        pbl_blob_header h;
        std::memset(&h, 0, sizeof(h));
        h.magic = PBL_MAGIC; h.version = PBL_VERSION; h.N = N; h.K = K; h.P = P; h.G = G; h.NRB = NRB;
        h.flags = (G > 1 ? PBL_FLAG_HAS_GROUPS : 0) | (sal16 ? PBL_FLAG_SAL_F16 : 0) | PBL_FLAG_TAIL_REPEAT | PBL_FLAG_SLABS;
        h.max_nch = max_nch; h.max_nexc = max_nexc; h.nnz = nnz; h.nexc = nexc_total;
        h.blob_bytes = cur; h.rb_off_pos = uint32_t(rboff_pos);
        std::memcpy(blob, &h, sizeof(h));
        std::memset(blob + rboff_pos, 0, rec0 - rboff_pos);
        std::memcpy(blob + rboff_pos, rb_info.data(), rb_info.size() * sizeof(pbl_rec_info));
    }
    return PBL_OK;
}

int pbl_blob_describe(const void* host_blob, size_t bytes, pbl_layer* out) {
    if (!host_blob || !out || bytes < sizeof(pbl_blob_header)) return PBL_ERR_INVALID_ARG;
    pbl_blob_header h;
    std::memcpy(&h, host_blob, sizeof(h));
    if (h.magic != PBL_MAGIC || h.version != PBL_VERSION || h.blob_bytes != bytes) return PBL_ERR_BAD_BLOB;
    if (h.N == 0 || h.K == 0 || h.K > 32767 || h.N > (1u << 24)) return PBL_ERR_BAD_BLOB;
    if (h.P != (h.K + 511) / 512 || h.NRB != (h.N + 15) / 16 || h.G == 0) return PBL_ERR_BAD_BLOB;
    if (h.G > 1 && (h.K % h.G != 0 || (h.K / h.G) % 128 != 0)) return PBL_ERR_BAD_BLOB;
    if ((h.flags & ~PBL_FLAG_KNOWN) || !(h.flags & PBL_FLAG_SLABS) || !(h.flags & PBL_FLAG_TAIL_REPEAT)) return PBL_ERR_BAD_BLOB;
    if (bool(h.flags & PBL_FLAG_HAS_GROUPS) != (h.G > 1) || h.rb_off_pos != sizeof(pbl_blob_header)) return PBL_ERR_BAD_BLOB;
    {   // walk every record: nothing below trusts a count or an offset it has not bounded first
        const uint8_t* blob = static_cast<const uint8_t*>(host_blob);
        const size_t rec0 = align128(sizeof(pbl_blob_header) + size_t(h.NRB + 1) * sizeof(pbl_rec_info));
        if (rec0 > bytes) return PBL_ERR_BAD_BLOB;
        const pbl_rec_info* info = reinterpret_cast<const pbl_rec_info*>(blob + h.rb_off_pos);
        const size_t fixed = record_fixed_bytes(h.P, h.G);
        const bool has_crow = h.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16);
        const uint32_t NS = PBL_NSLABS(h.K);
        uint64_t nnz = 0, nexc = 0;
        size_t expect = rec0;
        for (uint32_t b = 0; b < h.NRB; ++b) {
            const pbl_rec_info ri = info[b];
            const size_t off = size_t(ri.off16) * 16;
            if (off != expect) return PBL_ERR_BAD_BLOB;                       // contiguous, hence monotone and 128-aligned
            const uint64_t nch = uint64_t(ri.nfull) + ri.ntail;
            if (nch > 65535 || nch > h.max_nch || ri.nexc > h.max_nexc) return PBL_ERR_BAD_BLOB;
            const size_t rbytes = fixed + record_sal_bytes(nch, ri.ntail, ri.nexc, has_crow, h.K);
            if (off + rbytes > bytes || off + rbytes < off) return PBL_ERR_BAD_BLOB;
            expect = off + rbytes;
            const uint8_t* rec = blob + off;
            pbl_rec_header rh;
            std::memcpy(&rh, rec, sizeof(rh));
            if (rh.nfull != ri.nfull || rh.ntail != ri.ntail || rh.nexc != ri.nexc || rh.off_sal != fixed) return PBL_ERR_BAD_BLOB;
            const pbl_rowinfo* row = reinterpret_cast<const pbl_rowinfo*>(rec + PBL_REC_ROWINFO_OFF);
            uint32_t sf = 0, st = 0;
            for (int r = 0; r < 16; ++r) {                                    // rows own consecutive ranges of both lists
                if (row[r].start != sf || row[r].tailidx != st) return PBL_ERR_BAD_BLOB;
                sf += row[r].nfull; st += row[r].ntail;
            }
            if (sf != ri.nfull || st != ri.ntail) return PBL_ERR_BAD_BLOB;
            const uint8_t* s = rec + fixed;
            const uint32_t nch32 = uint32_t(nch);
            const uint16_t* col0 = reinterpret_cast<const uint16_t*>(s);
            const uint8_t* delta = s + PBL_SAL_DELTA_OFF(nch32);
            const uint8_t* tailcnt = s + PBL_SAL_TAILCNT_OFF(nch32);
            const uint8_t* crow = s + PBL_SAL_CROW_OFF(nch32, ri.ntail);
            for (uint32_t c = 0; c < nch32; ++c) {
                const uint32_t cnt = c < ri.nfull ? 16u : tailcnt[c - ri.nfull];
                if (cnt < 1 || cnt > 16 || (c >= ri.nfull && cnt == 16) || delta[size_t(c) * 16] != 0) return PBL_ERR_BAD_BLOB;
                uint32_t col = col0[c];
                for (uint32_t k = 1; k < 16; ++k) {
                    const uint32_t d = delta[size_t(c) * 16 + k];
                    if ((d & 1u) || (k < cnt ? d == 0 : d != 0)) return PBL_ERR_BAD_BLOB;   // doubled steps, strictly rising, padding steps 0
                    col += d / 2;
                }
                if (col >= h.K) return PBL_ERR_BAD_BLOB;
                if (has_crow && crow[c] > 15) return PBL_ERR_BAD_BLOB;
                if (h.flags & PBL_FLAG_SAL_F16) {      // every coded value must be recognisable in the kernels' fp16 tile (sal16_storable)
                    const pbl_rowparams* prm = reinterpret_cast<const pbl_rowparams*>(rec + PBL_REC_PARAMS_OFF);
                    const uint8_t* code = s + PBL_SAL_CODE_OFF(nch32);
                    for (uint32_t k = 0; k < cnt; ++k)
                        if (!sal16_storable(prm[crow[c]].sscale, prm[crow[c]].szero, code[size_t(c) * 16 + k])) return PBL_ERR_BAD_BLOB;
                }
                nnz += cnt;
            }
            const pbl_exception* exc = reinterpret_cast<const pbl_exception*>(s + PBL_SAL_EXC_OFF(nch32, ri.ntail, has_crow));
            for (uint32_t k = 0; k < ri.nexc; ++k)
                if (exc[k].col >= h.K || exc[k].row > 15) return PBL_ERR_BAD_BLOB;
            nexc += ri.nexc;
            const uint32_t* slab = reinterpret_cast<const uint32_t*>(s + PBL_SAL_SLAB_OFF(nch32, ri.ntail, ri.nexc, has_crow));
            for (int r = 0; r < 16; ++r) {
                uint32_t pf = 0, pt = 0;
                for (uint32_t q = 0; q < NS; ++q) {
                    const uint32_t e = slab[size_t(r) * NS + q];
                    const uint32_t f = PBL_SLAB_FE(e), t = PBL_SLAB_TE(e);
                    if ((e >> 26) || f < pf || t < pt || f > row[r].nfull || t > row[r].ntail) return PBL_ERR_BAD_BLOB;
                    if (PBL_SLAB_FBACK(e) > pf || PBL_SLAB_TBACK(e) > pt) return PBL_ERR_BAD_BLOB;
                    pf = f; pt = t;
                }
                if (pf != row[r].nfull || pt != row[r].ntail) return PBL_ERR_BAD_BLOB;   // the last slab has seen every chunk
            }
        }
        if (expect != bytes || size_t(info[h.NRB].off16) * 16 != bytes) return PBL_ERR_BAD_BLOB;
        if (nnz != h.nnz || nexc != h.nexc) return PBL_ERR_BAD_BLOB;
    }
    out->blob = nullptr; out->bias = nullptr;
    out->N = h.N; out->K = h.K; out->P = h.P; out->G = h.G; out->NRB = h.NRB; out->flags = h.flags;
    out->max_nch = h.max_nch; out->max_nexc = h.max_nexc;
    return PBL_OK;
}

int pbl_unpack_dense_f32(const void* host_blob, size_t bytes, float* Wout) {
    pbl_layer L;
    int st = pbl_blob_describe(host_blob, bytes, &L);
    if (st != PBL_OK) return st;
    if (!Wout) return PBL_ERR_INVALID_ARG;
    const uint8_t* blob = static_cast<const uint8_t*>(host_blob);
    pbl_blob_header h;
    std::memcpy(&h, blob, sizeof(h));
    const pbl_rec_info* rb_info = reinterpret_cast<const pbl_rec_info*>(blob + h.rb_off_pos);
    const uint32_t N = L.N, K = L.K, G = L.G, P = L.P, gs = K / G;
    const size_t fixed = record_fixed_bytes(P, G), tiles_off = fixed - size_t(P) * 1024;
    for (uint32_t b = 0; b < L.NRB; ++b) {
        const uint8_t* rec = blob + size_t(rb_info[b].off16) * 16;
        pbl_rec_header rh;
        std::memcpy(&rh, rec, sizeof(rh));
        const pbl_rowinfo* ri = reinterpret_cast<const pbl_rowinfo*>(rec + 16);
        const pbl_rowparams* pr = reinterpret_cast<const pbl_rowparams*>(rec + 144);
        const float* ghl = reinterpret_cast<const float*>(rec + PBL_REC_GHL_OFF);
        const uint32_t* tile = reinterpret_cast<const uint32_t*>(rec + tiles_off);
        const uint32_t nch = rh.nfull + rh.ntail;
        const uint8_t* s = rec + rh.off_sal;
        const bool has_crow = L.flags & (PBL_FLAG_HAS_GROUPS | PBL_FLAG_SAL_F16);
        const uint16_t* col0 = reinterpret_cast<const uint16_t*>(s);
        const uint8_t* delta = s + PBL_SAL_DELTA_OFF(nch);
        const uint8_t* code = s + PBL_SAL_CODE_OFF(nch);
        const uint8_t* tailcnt = s + PBL_SAL_TAILCNT_OFF(nch);
        const pbl_exception* exc = reinterpret_cast<const pbl_exception*>(s + PBL_SAL_EXC_OFF(nch, rh.ntail, has_crow));
        for (int rho = 0; rho < 16; ++rho) {
            const uint32_t r = b * 16 + rho;
            if (r >= N) continue;
            float* w = Wout + size_t(r) * K;
            for (uint32_t c = 0; c < K; ++c) {
                const uint32_t p = c / 512, cc = c % 512, i = cc / 128, l2 = (cc % 128) / 2, e = cc & 1;
                const uint32_t word = tile[(size_t(p) * 64 + l2) * 4 + i];
                const int bit = (word >> bit_index(rho, int(e))) & 1;
                const uint32_t g = c / gs;
                const float hv = G > 1 ? ghl[(size_t(rho) * G + g) * 2] : pr[rho].hi;
                const float lv = G > 1 ? ghl[(size_t(rho) * G + g) * 2 + 1] : pr[rho].lo;
                w[c] = bit ? hv : lv;
            }
            auto apply = [&](size_t ch, int cnt) {
                uint32_t col = col0[ch];
                for (int k = 0; k < cnt; ++k) {
                    col += delta[ch * 16 + k] / 2u;
                    if (col < K)
                        w[col] = (L.flags & PBL_FLAG_SAL_F16) ? dequant_f16(pr[rho].sscale, pr[rho].szero, code[ch * 16 + k])
                                                              : dequant(pr[rho].sscale, pr[rho].szero, code[ch * 16 + k]);
                }
            };
            for (uint32_t k = 0; k < ri[rho].nfull; ++k) apply(size_t(ri[rho].start) + k, 16);
            for (uint32_t k = 0; k < ri[rho].ntail; ++k)
                apply(size_t(rh.nfull) + ri[rho].tailidx + k, tailcnt[ri[rho].tailidx + k]);
        }
        for (uint32_t k = 0; k < rh.nexc; ++k) {
            const uint32_t r = b * 16 + exc[k].row;
            if (r < N && exc[k].col < K) Wout[size_t(r) * K + exc[k].col] = exc[k].value;
        }
    }
    return PBL_OK;
}

}  // extern "C"
