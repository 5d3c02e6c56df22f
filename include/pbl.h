/*
 * pbl.h -- C ABI of libpbl.so: MI355X (gfx950) partially-binarized linear layer.
 *
 * This is the drop-in boundary for ONE hot path of hahnyuan/PB-LLM: the forward of
 *   quant.BinaryLinear                     (quant/quantizer.py:75-86)
 *   quant.XnorBinaryLinear                 (quant/quantizer.py:172-193)
 *   quant.BinaryXnorExceptOutliersLinear   (quant/outlier_quantizer.py:33-123)
 *   nn.Linear holding GPTQ-PB fake-quant weights (gptq_pb/gptq.py:155,180-184)
 * all of which the reference evaluates as  F.linear(x, w_sim, bias)  over a dense
 * simulated weight.  The reference is pure Python with no FFI; the Python module
 * layer in pb_llm_amd/quant.py mirrors its classes and calls these entry points
 * through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions: plain pointers and sizes, no torch types.  Device pointers are
 * BORROWED (the caller owns all memory); nothing here allocates, frees or
 * synchronises device memory; kernels are asynchronous on the given hipStream_t
 * and are hipGraph-capturable.  Every function returns a pbl_status (0 = ok).
 * No exceptions cross the ABI.  Reentrant; no global mutable state.
 *
 * ---------------------------------------------------------------------------
 * Unified math:  for output row r
 *   y_r = sum_j v_rj * x_j + b_r,
 *   v_rj = hi_{r,g(j)} if bit_rj else lo_{r,g(j)}          (dense 1-bit plane)
 *          except at "salient" positions, where v_rj = sscale_r * (q_rj - szero_r)
 *          (uint8 code q, per-row affine = HighQuantizer's scale*(q-zero)), and at rare "exception" positions,
 *          where v_rj is an explicit fp32 value.
 * hi/lo are the two values the binarized weights of a row(-group) take:
 *   BinaryLinear: (+1,-1);  Xnor: (+a_r,-a_r);  QAT PB layer: (+a,-a) per tensor;
 *   GPTQ-PB: (mu+alpha, mu-alpha) per row and column group.
 *
 * ---------------------------------------------------------------------------
 * Packed format "PBL1" (one blob per layer, identical on host and device):
 *
 *   [pbl_blob_header 80 B][rb_info: pbl_rec_info[NRB+1], 16 B each][pad to 128 B]
 *   [record 0][record 1]...[record NRB-1]        NRB = ceil(N/16); every record starts on a 128-B line
 *
 * rb_info[b] = {record offset in units of 16 B, nfull, ntail, nexc}: ONE 16-byte scalar load
 * tells a wavefront where its record is and how long its salient lists are, so every
 * load of the record can be issued at once (rb_info[NRB].off16 = end of blob).
 * A record holds one ROW-BLOCK of 16 output rows and is the unit of work of one
 * wavefront.  P = ceil(K/512) column panels.  Record layout (PBL_* macros below; tiles, col0,
 * delta and code start on 128-B lines):
 *   +0    pbl_rec_header (16 B): nfull, ntail, nexc, off_sal (bytes from record start)
 *   +16   rowinfo[16]  (8 B each): u16 start, u16 nfull, u16 tailidx, u8 ntail, u8 0
 *   +144  params[16]   (16 B each): f32 hi, lo, sscale, szero     (group 0 / G==1)
 *   +400  [G>1 only]   ghl[16][G] (8 B each): f32 hi, lo per column group
 *   +T    tiles[P]     1 KiB each: the sign plane of 16 rows x 512 columns (T = PBL_TILES_OFF(G) = 512 for G == 1)
 *   +off_sal: col0[nch] (u16, nch = nfull+ntail), delta[nch][16] (u8),
 *             code[nch][16] (u8), tailcnt[ntail_pad16] (u8), [flags & (HAS_GROUPS|SAL_F16)] crow[nch_pad16]
 *             (u8, row-in-block of each chunk), exc[nexc] (pbl_exception, 8 B),
 *             slab[16][NS] (u32, version 2): the column-slab index of the salient lists, NS = ceil(K/256)
 *
 * Sign-plane tile p: lane l (0..63) owns 4 dwords at byte ((p*64+l)*4+i)*4, i=0..3.
 *   dword i covers columns c = 512p + 128i + 2l + e, e in {0,1}.
 *   bit b of the dword: e = b>>4; pos = b&15; row-in-block rho = pos>=8 ? pos-8 : pos+8.
 *   (Rows 8..15 are read after one `<< 8`, so every row sits on one of the fp16
 *    bit positions 8..15 and is unpacked to a two-valued fp16 constant by ONE
 *    v_and(_or)_b32 -- no per-weight shift; see DESIGN.md "bit classes".)
 *   bit = 1 <=> the weight takes value hi (salient/exception positions store 1).
 *
 * Salient entries of a row are sorted by column and cut greedily into chunks of up
 *   to 16: a chunk closes after 16 entries or when the next column step exceeds 127.
 *   Chunk c: col0[c] = column of entry 0; delta[c][k] = 2 * (column step from entry
 *   k-1 to k), i.e. the BYTE step in an fp16 x vector (delta[c][0] = 0); code[c][k] = q.  Chunks with exactly 16 entries are
 *   "full"; the others are "tail" chunks and carry their count in tailcnt[].  With PBL_FLAG_TAIL_REPEAT the
 *   unused entries of a tail chunk have step 0 and REPEAT the last entry's code, so a reader that writes
 *   positions (rather than accumulates) may process all 16 entries of any chunk without looking at tailcnt.
 *   Chunk order within a record: full chunks of row 0, row 1, ... row 15 (indices
 *   0..nfull-1), then the tail chunks of row 0, row 1, ... (indices nfull..nch-1).
 *   rowinfo[r]: full chunks [start, start+nfull), tail chunks
 *   [hdr.nfull + tailidx, hdr.nfull + tailidx + ntail).
 *
 * Slab index (version 2, PBL_FLAG_SLABS): the GEMV walks a row's chunks from left to right; the matrix-core kernels
 *   walk the COLUMNS in slabs of 256 and need, per (row, slab), the row's chunks that overlap the slab.  The chunks of
 *   a row cover disjoint, ordered column intervals, so that is one contiguous range of its full chunks and one of its
 *   tail chunks, and at most one chunk straddles a slab boundary.  slab[rho][s] packs
 *     bits  0..15  fe: number of the row's full chunks whose first column is < 256 (s+1)
 *     bits 16..23  te: the same for its tail chunks
 *     bit  24      fback: a full chunk that starts left of column 256 s reaches into the slab
 *     bit  25      tback: the same for a tail chunk
 *   Full chunks overlapping slab s: [fe(s-1) - fback(s), fe(s)) relative to rowinfo.start, fe(-1) = 0; tails likewise
 *   relative to rowinfo.tailidx.  (Computed by the kernel in round 1 with a per-launch counting sort.)
 */
#ifndef PBL_H_
#define PBL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBL_MAGIC 0x314C4250u /* "PBL1" */
#define PBL_VERSION 2
#define PBL_SLAB_COLS 256
#define PBL_ROWS_PER_BLOCK 16
#define PBL_PANEL_COLS 512
#define PBL_CHUNK 16
#define PBL_MAX_GAP 127 /* largest column step inside a chunk (stored doubled in a u8) */
#define PBL_MAX_TOKENS_PER_LAUNCH 4 /* M handled per weight pass by the GEMV kernel */

/* Record layout arithmetic (shared by the packer, the decoder and the kernels).  Records and the
 * big arrays inside them start on 128-byte lines: the weight stream is read with non-temporal
 * 1 KiB wave-loads, and a load that straddles a line fetches the shared line twice (measured:
 * +8 % FETCH_SIZE with 16-byte alignment). */
#define PBL_ALIGN16(x) (((x) + 15u) & ~15u)
#define PBL_ALIGN128(x) (((x) + 127u) & ~127u)
#define PBL_REC_ROWINFO_OFF 16u
#define PBL_REC_PARAMS_OFF 144u
#define PBL_REC_GHL_OFF 400u
#define PBL_TILES_OFF(G) PBL_ALIGN128(400u + ((G) > 1u ? 128u * (G) : 0u))
#define PBL_SAL_DELTA_OFF(nch) PBL_ALIGN128((nch) * 2u)                                   /* from off_sal */
#define PBL_SAL_CODE_OFF(nch) (PBL_SAL_DELTA_OFF(nch) + PBL_ALIGN128((nch) * 16u))
#define PBL_SAL_TAILCNT_OFF(nch) (PBL_SAL_CODE_OFF(nch) + PBL_ALIGN128((nch) * 16u))
#define PBL_SAL_CROW_OFF(nch, ntail) (PBL_SAL_TAILCNT_OFF(nch) + PBL_ALIGN16(ntail))
#define PBL_SAL_EXC_OFF(nch, ntail, has_crow) (PBL_SAL_CROW_OFF(nch, ntail) + ((has_crow) ? PBL_ALIGN16(nch) : 0u))
#define PBL_NSLABS(K) (((K) + PBL_SLAB_COLS - 1u) / PBL_SLAB_COLS)
#define PBL_SAL_SLAB_OFF(nch, ntail, nexc, has_crow) PBL_ALIGN16(PBL_SAL_EXC_OFF(nch, ntail, has_crow) + (nexc) * 8u)
#define PBL_SAL_BYTES(nch, ntail, nexc, has_crow, K) \
    PBL_ALIGN128(PBL_SAL_SLAB_OFF(nch, ntail, nexc, has_crow) + 16u * PBL_NSLABS(K) * 4u)
#define PBL_SLAB_FE(e) ((e) & 0xFFFFu)
#define PBL_SLAB_TE(e) (((e) >> 16) & 0xFFu)
#define PBL_SLAB_FBACK(e) (((e) >> 24) & 1u)
#define PBL_SLAB_TBACK(e) (((e) >> 25) & 1u)

typedef enum {
    PBL_OK = 0,
    PBL_ERR_INVALID_ARG = -1,
    PBL_ERR_BAD_BLOB = -2,
    PBL_ERR_UNSUPPORTED = -3,
    PBL_ERR_MISALIGNED = -4,
    PBL_ERR_CAPACITY = -5,
    PBL_ERR_LAUNCH = -6,
    PBL_ERR_NOT_REPRESENTABLE = -7
} pbl_status;

/* flags in pbl_blob_header.flags */
#define PBL_FLAG_HAS_GROUPS 0x1u /* G > 1: per-(row,group) hi/lo in ghl */
#define PBL_FLAG_SAL_F16 0x2u    /* salient values are fl16(sscale*(q-szero)): the layer came from an fp16
                                    checkpoint (gptq_pb/gptq.py:182 `.to(fp16)`).  Unpack and the GEMV both
                                    apply the fp16 rounding (v_cvt_pk_f16_f32), so the layer is reproduced
                                    bit-exactly. */
#define PBL_FLAG_TAIL_REPEAT 0x4u /* tail-chunk padding repeats the last entry (step 0, same code); set by this packer */
#define PBL_FLAG_SLABS 0x8u       /* the records carry the column-slab index (format version 2) */
#define PBL_FLAG_KNOWN 0xFu

/* Limits (every entry point answers PBL_ERR_UNSUPPORTED / PBL_ERR_BAD_BLOB beyond them; pb_llm_amd/packing.py names the limit):
 *   in_features  K <= 32767          16-bit column indices with pre-doubled byte steps; shard wider layers along K
 *   out_features N <= 2^24
 *   column groups: K % G == 0 and K / G a multiple of 128; the kernels additionally want K / G a power of two
 *   tokens per GEMV pass <= PBL_MAX_TOKENS_PER_LAUNCH (pbl_linear_f16 loops / routes to the matrix-core kernel beyond)
 *   matrix-core kernels: K % 8 == 0, x 16-byte aligned, <= 32 tokens per pass; GEMM-regime kernel: K % 8 == 0, groups of a multiple of 128 columns
 *   pbl_quant8_rows: K <= 16384 (prep.py falls back to the torch path above)
 */
typedef struct {
    uint32_t magic, version;
    uint32_t N, K;          /* out_features, in_features */
    uint32_t P;             /* column panels = ceil(K/512) */
    uint32_t G;             /* column groups (1 = whole row); groupsize = K/G, multiple of 128 */
    uint32_t NRB;           /* row blocks = ceil(N/16) */
    uint32_t flags;
    uint32_t max_nch;       /* max chunks in any record (LDS sizing) */
    uint32_t max_nexc;      /* max exceptions in any record */
    uint64_t nnz;           /* salient code entries, total */
    uint64_t nexc;          /* exception entries, total */
    uint64_t blob_bytes;    /* total size of the blob */
    uint32_t rb_off_pos;    /* byte offset of rb_info[] from blob start (= sizeof header = 80) */
    uint32_t reserved[3];
} pbl_blob_header;

typedef struct { uint32_t nfull, ntail, nexc, off_sal; } pbl_rec_header;
typedef struct { uint32_t off16, nfull, ntail, nexc; } pbl_rec_info;
typedef struct { uint16_t start, nfull, tailidx; uint8_t ntail, pad; } pbl_rowinfo;
typedef struct { float hi, lo, sscale, szero; } pbl_rowparams;
typedef struct { uint16_t col; uint16_t row; float value; } pbl_exception;

/* A layer as the kernels see it.  `blob` is a DEVICE pointer to a PBL1 blob
 * (16-B aligned); the scalar fields replicate its header so no device read is
 * needed on the host.  `bias` (device, fp32 [N]) may be NULL. */
typedef struct {
    const void*  blob;
    const float* bias;
    uint32_t N, K, P, G, NRB, flags, max_nch, max_nexc;
} pbl_layer;

const char* pbl_status_string(int status);
int pbl_version(void);

/* ---------------- host-side packer (CPU; replaces nothing in the reference: the
 * reference stores 1-bit weights as dense fp16, gptq_pb/gptq.py:180-184) ---------- */

/* Pack a dense simulated weight W [N,K] (fp32, row-major, torch nn.Linear layout)
 * given the two binarized values of every (row, group): hi, lo [N*G]; and the
 * per-row affine of the salient codes: sscale, szero [N] (may be NULL when the layer
 * has no salient weights).  An element equal to hi (lo) becomes bit 1 (0); any
 * other element becomes a uint8 code entry if  fl32(sscale*(q-szero)) == W  for an
 * integer q in [0,255], else an fp32 exception entry.  So unpack(pack(W)) == W
 * bit-exactly for ANY input; off-grid values only cost bytes (8 B each).  If out == NULL only *out_bytes is computed.
 * sal_mask (u8 [N*K], may be NULL): when given, positions with sal_mask != 0 are
 * forced into the code/exception list even if they equal hi or lo. */
int pbl_pack_dense_f32(const float* W, uint32_t N, uint32_t K, uint32_t G,
                       const float* hi, const float* lo,
                       const float* sscale, const float* szero,
                       const uint8_t* sal_mask, uint32_t flags /* PBL_FLAG_SAL_F16 or 0 */,
                       void* out, size_t out_capacity, size_t* out_bytes);

/* Device-side packer: the same blob, BYTE FOR BYTE, from device tensors (csrc/pbl_pack.hip; one wavefront per record).
 * Two steps because the blob's size depends on the data:
 *   1. pbl_pack_dev_count -> counts_out [NRB][PBL_PACK_COUNT_WORDS] u32 (device): word 0 record bytes, 1 nfull, 2 ntail,
 *      3 nexc, 4 coded entries, 5 status (non-zero: the layer exceeds the format's limits), 8.. per-row counts;
 *   2. the caller prefix-sums word 0 into rec_off [NRB+1] (u64, device; rec_off[0] = PBL_ALIGN128(80 + 16 (NRB+1)), the
 *      last entry = blob size), reduces max(nfull+ntail), max(nexc), sum(word 4), sum(nexc), allocates the blob and calls
 *      pbl_pack_dev_write, which writes header, rb_info and every record (padding zeroed).
 * All pointers are device pointers (W fp32 [N,K]; hi, lo [N,G]; sscale, szero [N] or NULL; sal_mask u8 [N,K] or NULL). */
#define PBL_PACK_COUNT_WORDS 56
int pbl_pack_dev_count(const float* W, uint32_t N, uint32_t K, uint32_t G, const float* hi, const float* lo,
                       const float* sscale, const float* szero, const uint8_t* sal_mask, uint32_t flags,
                       uint32_t* counts_out, void* stream);
int pbl_pack_dev_write(const float* W, uint32_t N, uint32_t K, uint32_t G, const float* hi, const float* lo,
                       const float* sscale, const float* szero, const uint8_t* sal_mask, uint32_t flags,
                       const uint32_t* counts, const uint64_t* rec_off, uint64_t blob_bytes, uint32_t max_nch, uint32_t max_nexc,
                       uint64_t nnz, uint64_t nexc, void* blob_out, void* stream);

/* Validate a host blob and fill a pbl_layer (blob/bias pointers are left NULL).  The WHOLE structure is checked, not
 * only the header: record offsets monotone, 128-byte aligned and inside the blob; every record's header equal to its
 * rb_info entry and its size equal to the layout macros; rowinfo ranges consistent with the chunk counts; every chunk's
 * columns, every exception's row / column and every slab entry in range; header maxima >= the per-record values.  A
 * blob that passes cannot make pbl_unpack_dense_f32 or a kernel read outside it.  PBL_ERR_BAD_BLOB otherwise. */
int pbl_blob_describe(const void* host_blob, size_t bytes, pbl_layer* out);

/* Reconstruct the dense simulated weight (fp32 [N,K]) from a host blob
 * (to_regular_linear, quant/outlier_quantizer.py:108-114). */
int pbl_unpack_dense_f32(const void* host_blob, size_t bytes, float* W_out);

/* ---------------- device entry points ------------------------------------------- */

/* Bytes of dynamic LDS the GEMV kernel needs for this layer at m tokens per pass
 * (informational; the launchers compute it themselves). */
size_t pbl_gemv_lds_bytes(const pbl_layer* layer, int m);

/* y[M,N] = x[M,K] @ W_sim^T + bias.  x, y: device fp16, row-major, contiguous
 * (replaces F.linear(x, w_sim, bias): quant/outlier_quantizer.py:105,
 * quant/quantizer.py:86,193).  M >= 1: up to PBL_MAX_TOKENS_PER_LAUNCH tokens run on the bit-unpacking GEMV,
 * more on the matrix-core kernel (pbl_gemm_mfma_f16, weights streamed once per 32 tokens) when the layer
 * qualifies, else in GEMV passes of 4.  y_f32 != 0: y is fp32 instead of fp16.
 * stream: hipStream_t (as void*). */
int pbl_linear_f16(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* stream);

/* Reconstruct the dense simulated weight ON THE DEVICE: W_out [N,K] row-major, fp16
 * (out_f32 == 0) or fp32.  This is the GEMM-regime path (M >= ~16, prefill): unpack into a
 * transient workspace, then a plain library GEMM (rocBLAS / hipBLASLt through torch) --
 * exactly the arithmetic the reference runs on its dense fake-quant weight
 * (gptq_pb/gptq.py:180-184 + nn.Linear).  For fp16 output every value must be
 * fp16-representable to be exact (true for layers packed from an fp16 checkpoint). */
int pbl_unpack_dev(const pbl_layer* layer, void* W_out, int out_f32, void* stream);

/* Matrix-core kernel for 1 <= M <= 32 tokens, any layer with K % 8 == 0, PBL_FLAG_TAIL_REPEAT | PBL_FLAG_SLABS, column
 * groups (if any) of a power-of-two size >= 128, and x 16-B aligned (else PBL_ERR_UNSUPPORTED): ONE pass over the packed
 * weights for all tokens.  A workgroup of 4
 * waves owns 4 records and walks the columns in 256-column slabs; the slab of x is staged in LDS once for the 4 waves;
 * per slab a wave builds class-coded sign-plane A fragments in registers, scatters its record's salient entries of the
 * slab (found through the slab index) into an fp16 tile, derives the salient mask from that tile, and contracts with
 * v_mfma_f32_16x16x32_f16; fp32 decode identical to the GEMV.  y fp16 (y_f32 == 0) or fp32.  Column-group layers fold the
 * accumulators into per-row totals at every group boundary.  Up to 16 tokens a workgroup keeps one x tile (three workgroups
 * per CU), above that two.  pbl_linear_f16_ws routes here by itself (see pbl_linear_workspace_bytes). */
int pbl_gemm_mfma_f16(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* stream);

/* The same kernel with a K split: a layer with few records (N = 4096: 256) cannot fill 1024 SIMDs with one wave per
 * record, so the column slabs are divided over up to KS workgroups per record group; each writes an fp32 partial
 * y to `workspace` ([KS][M][N] floats, device, 16-B aligned) and a second small kernel adds them in a FIXED order
 * (deterministic, no atomics).  pbl_mfma_workspace_bytes() is what this layer needs for M tokens (0: no split is used);
 * workspace == NULL or too small: runs unsplit.  pbl_linear_f16_ws is pbl_linear_f16 with that workspace handed through. */
size_t pbl_mfma_workspace_bytes(const pbl_layer* layer, int M);
/* What pbl_linear_f16_ws wants as workspace for M tokens of this layer: pbl_mfma_workspace_bytes() when it will route the
 * call to the matrix-core kernel (more tokens than one GEMV pass takes at two workgroups per CU), 0 when the GEMV serves it.
 * Without the workspace a call of up to 8 tokens falls back to GEMV passes; larger ones run the matrix-core kernel unsplit. */
size_t pbl_linear_workspace_bytes(const pbl_layer* layer, int M);
int pbl_gemm_mfma_f16_ws(const pbl_layer* layer, const void* x, void* y, int M, int y_f32,
                         void* workspace, size_t workspace_bytes, void* stream);
int pbl_linear_f16_ws(const pbl_layer* layer, const void* x, void* y, int M, int y_f32,
                      void* workspace, size_t workspace_bytes, void* stream);

/* L independent layers in ONE launch (decode-time fused QKV / gate+up, and the
 * stream benchmark of SURVEY.md 8(d)).  layers_dev: DEVICE array of L pbl_layer;
 * x_dev / y_dev: DEVICE arrays of L pointers (fp16 [M,K_l] / fp16 [M,N_l]);
 * max_NRB, max_K, max_nch, max_nexc: maxima over the group (the host knows them).  M <= 4.
 * group_flags: bit 0 = some layer has column groups (the launch then runs the column-group kernel: every group size a power
 * of two, every K a multiple of 128; group-free layers may ride along), bit 1 = some layer has PBL_FLAG_SAL_F16.
 * y_f32 != 0: every y_l is fp32 (tensor-parallel partial sums). */
int pbl_gemv_f16_grouped(const pbl_layer* layers_dev, const void* const* x_dev, void* const* y_dev,
                         int L, int M, uint32_t max_NRB, uint32_t max_K, uint32_t max_nch,
                         uint32_t max_nexc, int group_flags, int y_f32, void* stream);

/* GEMM regime (M > 32 tokens: prefill, large batches) straight from the packed format: y[M,N] = x[M,K] (fp16) @ W^T + bias
 * with fp32 accumulation, y fp16 (y_f32 == 0) or fp32.  Every weight enters the contraction as the fp16 number
 * pbl_unpack_dev(.., out_f32 = 0) would produce for it -- the row(-group)'s level, the fp16-rounded salient value, the
 * exception value -- so the result is the one of a library fp16 GEMM on the unpacked layer up to summation order; exact
 * weights for layers packed from an fp16 checkpoint (PBL_FLAG_SAL_F16).  Any layer with K % 8 == 0, PBL_FLAG_SLABS |
 * PBL_FLAG_TAIL_REPEAT and column groups (if any) of a multiple of 128 columns; x and y 16-B aligned; otherwise
 * PBL_ERR_UNSUPPORTED and the caller falls back to pbl_unpack_dev + a library GEMM.  A workgroup of 8 waves owns 8 records
 * x 256 tokens: 4 producer waves rebuild the fp16 weight tile of the next 128 columns in LDS (sign plane through
 * v_pk_mad_u16, salients through the slab index) while 4 consumer waves multiply the previous one with
 * v_mfma_f32_32x32x16_f16 against x, which each consumer stages for its own 64 tokens by LDS-DMA
 * (csrc/pbl_gemm_big.hip); the dense weight never exists in HBM.  Replaces nn.Linear over the dense fake-quant checkpoint
 * at seq 2048 (gptq_pb/eval_ppl_utils.py:55-64, evaluate.py:126-145). */
int pbl_gemm_f16_ex(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* stream);
/* The same with a transient workspace (device, 16-B aligned, pbl_gemm_workspace_bytes(layer, M) bytes; 0 = none wanted: at most
 * one 256-token tile, or K > 16256).  A workgroup of the GEMM kernel decodes a record's salient chunks once per token tile and
 * half slab, in a single in-order producer wave -- the critical path at 5-10 % salients.  With the workspace a small kernel
 * ahead of the GEMM decodes every chunk ONCE per call into 4-byte words {position in the stage image : fp16 value}, grouped
 * per (record, 128-column half slab), and the GEMM's producers copy them: 4 B per salient entry of scratch (the dense weight
 * would be 2 B per WEIGHT), identical results bit for bit.  workspace == NULL or too small: pbl_gemm_f16_ex. */
size_t pbl_gemm_workspace_bytes(const pbl_layer* layer, int M);
int pbl_gemm_f16_ws(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, void* workspace, size_t workspace_bytes,
                    void* stream);
int pbl_gemm_f16(const pbl_layer* layer, const void* x, void* y, int M, void* stream);   /* = pbl_gemm_f16_ex(.., 0, ..) */
/* The two halves of pbl_gemm_f16_ws, for callers that build a layer's salient list once and use it many times, or build the
 * NEXT layer's list on a second stream while this layer's GEMM runs (the perplexity loops call the same linears batch after
 * batch: gptq_pb/eval_ppl_utils.py:55-64).  The list depends on the blob only -- never on x or M -- and stays valid until the
 * blob changes.  pbl_gemm_list_bytes: its size (0: K > 16256, no list; use pbl_gemm_f16_ex).  pbl_gemm_prepare: one small
 * kernel on `stream`.  pbl_gemm_f16_prepared: the GEMM over a prepared list, any M >= 1, results bit-identical to
 * pbl_gemm_f16_ex / _ws.  The library keeps no reference to the workspace between calls. */
size_t pbl_gemm_list_bytes(const pbl_layer* layer);
int pbl_gemm_prepare(const pbl_layer* layer, void* workspace, size_t workspace_bytes, void* stream);
int pbl_gemm_f16_prepared(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, const void* workspace,
                          size_t workspace_bytes, void* stream);

/* GEMM regime over a per-layer GEMM IMAGE (round 4, csrc/pbl_gemm_img.hip) -- the form the prefill path uses by default.
 * Callers that run the same linears batch after batch at more than 32 rows (the reference's perplexity loops,
 * gptq_pb/eval_ppl_utils.py:55-64, evaluate.py:126-145, call every nn.Linear with 2048 rows) re-lay the packed layer ONCE
 * into slots -- one per (16-row record, 128-column half slab): per lane the sign-plane dword and the salient entries as
 * ready-to-store {LDS offset : fp16 value} words, padded with idempotent repeats -- plus one row of level pairs per (record,
 * column group).  The GEMM kernel over an image has four matrix-core waves that only read fragments and multiply, and four
 * waves that expand one slot per 64-column step with ~105 instructions and stage x by LDS-DMA; results are bit-identical to
 * pbl_gemm_f16_ex / _ws / _prepared.  Every slot is sized for its own entry count: 1 KiB (up to 192 entries), 2 KiB (448),
 * 3 KiB (704), 4 KiB (960) or 5 KiB (1216); a table of 128 words per record says where its slots are.  8.6 MB for a
 * 4096 x 4096 layer at 5 % magnitude salients (packed blob 5.1 MB, dense fp16 33.5 MB).
 *   pbl_gemm_image_stats_bytes  size of the statistics buffer (device) the next call fills; 0 = no image for this layer (K % 8,
 *                          more than 127 half slabs, a group size that is not a multiple of 128): use pbl_gemm_f16_ws
 *   pbl_gemm_image_stats   two small kernels: entry counts, slot sizes and record starts into stats_dev (16-byte aligned, any
 *                          content).  Its first TWO uint32 words are the geometry: geom[0] = all slots in 256-byte units,
 *                          geom[1] = the largest slot in KiB (0xFFFFFFFF: a slot with more than 1216 entries: no image); the
 *                          caller reads them back ONCE per layer and passes the HOST copy to the other calls
 *   pbl_gemm_image_bytes   size of the image for that geometry; 0 = no image for this layer
 *   pbl_gemm_image_build   one small kernel on `stream` into caller-owned memory (16-byte aligned) from the layer and stats_dev
 *                          (which may be released behind it); the image is valid until the blob changes
 *   pbl_gemm_f16_image     y[M, N] = x[M, K] . W^T (+ bias), fp16 x, fp16 or fp32 y, any M >= 1
 * The library keeps no reference to the image, the statistics buffer or geom between calls. */
size_t pbl_gemm_image_stats_bytes(const pbl_layer* layer);
int pbl_gemm_image_stats(const pbl_layer* layer, void* stats_dev, void* stream);
size_t pbl_gemm_image_bytes(const pbl_layer* layer, const uint32_t* geom);
int pbl_gemm_image_build(const pbl_layer* layer, const uint32_t* geom, const void* stats_dev, void* image, size_t image_bytes, void* stream);
/* x as a FRAGMENT-MAJOR copy (round 6).  Taking pbl_gemm_f16_image's loop apart showed a quarter of a round going into staging x through
 * LDS (2 MB per CU and 128 x 256 tile by LDS-DMA, ~50 B/clk/CU whatever the source; profiles/r06_gemm.md).  With this copy a matrix-core B
 * fragment is 1 KiB of contiguous memory and the kernel's MFMA waves load it straight into registers: no x tile passes through LDS.
 *   pbl_x_fragment_bytes   size of the copy of x [M, K] fp16: tokens up to a whole 256-token tile x (K up to a 64-column step, + 64) x 2
 *   pbl_x_to_fragments     x [M, K] fp16 (device, rows ldx elements apart) -> the copy (16-byte aligned): for every block of 32 tokens and
 *                          every 16-column k-step 1 KiB -- lane l of 64: token l & 31, columns 16 ks + 8 (l >> 5) .. + 7; zeros beyond M / K.
 *                          ONE small streaming kernel per DISTINCT x: q / k / v of a decoder layer share a copy, gate / up another.
 *   pbl_gemm_f16_image_xf  pbl_gemm_f16_image_ws with x given as that copy (same M, K, plans, workspace): bit-identical results. */
size_t pbl_x_fragment_bytes(int M, uint32_t K);
int pbl_x_to_fragments(const void* x_f16, int M, uint32_t K, size_t ldx, void* x_fragments, void* stream);
int pbl_gemm_f16_image_xf(const pbl_layer* layer, const void* x_fragments, void* y, int M, int out_dtype, const float* tok_scale,
                          const void* image, size_t image_bytes, const uint32_t* geom, void* workspace, size_t workspace_bytes, void* stream);

/* The RESIDUAL image of an fp32-grid layer (round 6): same buffer size, geometry words and statistics buffer as pbl_gemm_image_build,
 * but every value v of the layer (row levels, salient values fl32(sscale (q - szero)), exceptions) is stored as fp16(4096 (v - fp16(v)))
 * where the ordinary image stores fp16(v).  The reference's fp32-only module classes (quant/quantizer.py:75-86,172-193: weights forced
 * to fp32) and QAT's fp32 master weights (utils.py:34-36) multiply with fp32 values an fp16 tile cannot hold; with both images
 *     W = W_hi + 2^-12 W_lo   (up to 2^-22 |W|),    y = x W_hi^T + 2^-12 x W_lo^T
 * so the GEMM regime of those layers runs on pbl_gemm_f16_image_ws / pbl_gemm_small_image_ws twice (fp32 results) + pbl_act_f32_join3
 * instead of pbl_unpack_dev + an fp32 library GEMM.  PBL_ERR_UNSUPPORTED for PBL_FLAG_SAL_F16 layers (their values are fp16). */
int pbl_gemm_image_build_residual(const pbl_layer* layer, const uint32_t* geom, const void* stats_dev, void* image, size_t image_bytes, void* stream);
int pbl_gemm_f16_image(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, const void* image, size_t image_bytes,
                       const uint32_t* geom, void* stream);
/* Shapes whose tiles are not a whole number of rounds of the chip (5120 x 5120 at 2048 rows: 320 tiles of 128 x 256 on 256 CUs; a
 * 300-token prompt: 64 tiles) -- pbl_gemm_f16_image_ws cuts the launch into a full part (whole rounds, straight to y) and a tail
 * whose tiles are split along K so that tail x splits fills the chip once; the tail's fp32 partial tiles go through `workspace`
 * (pbl_gemm_image_workspace_bytes(layer, M) bytes, 16-byte aligned, any content; 0: the plan is one launch) and a small kernel adds
 * them in split order (deterministic), adds the bias, applies tok_scale and casts.  Without a workspace (NULL / too small, and in
 * pbl_gemm_f16_image / _ex) everything is ONE launch, bit-identical to pbl_gemm_f16_ws / _prepared; with it the tail's tiles differ
 * by fp32 summation order.  pbl_gemm_image_plan reports the plan (mode 0 one launch / 1 token tail / 2 row tail, cut, splits, half
 * slabs per split, region tokens x columns).  Caller: the reference's perplexity loops at 2048 rows on llama-13b's 5120-row layers,
 * and every prompt shorter than a chip's worth of tiles (gptq_pb/eval_ppl_utils.py:55-64, evaluate.py:126-145). */
size_t pbl_gemm_image_workspace_bytes(const pbl_layer* layer, int M);
int pbl_gemm_image_plan(const pbl_layer* layer, int M, uint64_t* out6);
int pbl_gemm_f16_image_ws(const pbl_layer* layer, const void* x, void* y, int M, int out_dtype, const float* tok_scale,
                          const void* image, size_t image_bytes, const uint32_t* geom, void* workspace, size_t workspace_bytes,
                          void* stream);
/* The same with the result's type named (PBL_DTYPE_F16 / PBL_DTYPE_F32: exactly the call above) and, for bf16 activations
 * (qat/run_qat.py:120 `bf16=True`; F.linear(x_bf16, w, b), quant/outlier_quantizer.py:101-106 under bf16 autocast),
 * PBL_DTYPE_BF16 with tok_scale [M] (device, fp32): x is the fp16 copy pbl_act_bf16_prepare made of the bf16 activations and
 * y[t, r] = bf16(acc[t, r] * tok_scale[t] + bias[r]) is scaled and cast in the kernel's epilogue.  tok_scale must be NULL for
 * the other two types. */
int pbl_gemm_f16_image_ex(const pbl_layer* layer, const void* x, void* y, int M, int out_dtype, const float* tok_scale,
                          const void* image, size_t image_bytes, const uint32_t* geom, void* stream);
/* The same product for 1 <= M <= 64 rows of x over the same image (HBM-bound: the image is read once, also for 33 - 64 rows; every
 * wave owns 32 rows of W and a range of 128-column half slabs, K is split over the grid).  workspace: pbl_gemm_small_image_workspace_bytes(layer, M) bytes,
 * 16-byte aligned, any content (the splits' fp32 partial outputs, added in split order by a second small kernel: deterministic);
 * NULL / too small: one split.
 * Replaces F.linear at a small serving batch (quant/outlier_quantizer.py:101-106 at <= 64 rows; BASELINE.json configs[3]). */
size_t pbl_gemm_small_image_workspace_bytes(const pbl_layer* layer, int M);
int pbl_gemm_small_image_ws(const pbl_layer* layer, const void* x, void* y, int M, int y_f32, const void* image, size_t image_bytes,
                            const uint32_t* geom, void* workspace, size_t workspace_bytes, void* stream);
/* The same for bf16 activations (see pbl_act_bf16_prepare below): x_f16 / tok_scale [M] are the prepare step's outputs, y [M, N] is
 * out_dtype PBL_DTYPE_BF16 or PBL_DTYPE_F32 = cast(acc * tok_scale[token] + bias[row]) -- pbl_act_finish folded into the kernel
 * that adds the K splits, i.e. two launches behind the prepare step instead of three, the same bits.  PBL_ERR_UNSUPPORTED, nothing
 * launched, when the layer runs as ONE split (no workspace, or a layer that fills the device without a split): run
 * pbl_gemm_small_image_ws with an fp32 result and pbl_act_finish instead. */
int pbl_gemm_small_image_act(const pbl_layer* layer, const void* x_f16, void* y, int M, int out_dtype, const float* tok_scale,
                             const void* image, size_t image_bytes, const uint32_t* geom, void* workspace, size_t workspace_bytes, void* stream);

/* bf16 activations (round 5; csrc/pbl_act.hip).  The reference's QAT and evaluation run under bf16 (qat/run_qat.py:120; HF LLaMA
 * checkpoints), i.e. F.linear(x_bf16, w, b).  The packed kernels multiply fp16 tiles with fp32 accumulation; bf16 -> fp16 is exact
 * inside fp16's range and a token's row may be scaled by a power of two without changing a product's significand:
 *   pbl_act_bf16_prepare  x [M, K] bf16 (rows ldx elements apart) -> x_f16 [M, K] fp16 contiguous and tok_scale [M] fp32 with
 *                         x_f16[t, :] = x[t, :] / tok_scale[t], tok_scale[t] = 2^max(0, exponent(amax_t) - 14): exact for every
 *                         finite bf16 input (what falls under fp16's subnormals is > 2^-38 below the token's maximum).  A token
 *                         that holds inf / NaN becomes the indicator row (finite -> 0, +-inf -> +-1, NaN -> NaN) with
 *                         tok_scale[t] = +inf: y = (W . indicator) * inf gives +-inf by the sign of the weight an infinity
 *                         meets, NaN where that weight is 0 and NaN rows for NaN inputs, as F.linear does (deviation: several
 *                         infinities in ONE token whose products disagree in sign give +-inf by the weights' sum, not NaN).
 *   pbl_act_finish        y_out [M, N] (PBL_DTYPE_F32 / _F16 / _BF16) = cast(y_f32[t, r] * tok_scale[t] + bias[r]); tok_scale and
 *                         bias may be NULL.  Behind the kernels that serve <= 64 rows (they write fp32); the GEMM-regime kernel
 *                         scales and casts in its own epilogue (pbl_gemm_f16_image_ex).
 * One small streaming kernel each: no host synchronisation, the same behaviour eagerly and under hipGraph capture. */
int pbl_act_bf16_prepare(const void* x_bf16, int M, uint32_t K, size_t ldx, void* x_f16, float* tok_scale, void* stream);
int pbl_act_finish(const float* y_f32, const float* tok_scale, const float* bias, int M, uint32_t N, void* y_out, int out_dtype,
                   void* stream);
/* fp32 activations (the reference's fp32-only module classes, quant/quantizer.py:78-80,175-177; QAT's fp32 master weights,
 * utils.py:34-36): F.linear(x_f32, w, b).  The packed kernels are linear in x, so x is split into two fp16 terms, both run through
 * the kernel in ONE call with an fp32 result, and the halves are added:
 *   pbl_act_f32_split  x [M, K] fp32 (rows ldx elements apart) -> x_f16 [2 M, K] fp16 contiguous + tok_scale [M] fp32: per token
 *                      s = 2^max(0, exponent(amax) - 14) (pbl_act_bf16_prepare's rule), rows [0, M) = fp16(x / s), rows [M, 2 M) =
 *                      fp16(x / s - fp16(x / s)), tok_scale[t] = s; a token holding inf / NaN becomes its indicator row (finite -> 0,
 *                      +-inf -> +-1, NaN -> NaN; low term 0) with tok_scale[t] = +inf, so the result has F.linear's non-finite
 *                      pattern.  tok_scale == NULL: the unscaled form (s = 1; overflows to a NaN row for |x| >= 65520);
 *   pbl_act_f32_join   y_out [M, N] (out_dtype) = cast((y_f32[t, r] + y_f32[M + t, r]) * tok_scale[t] + bias[r]) for y_f32 [2 M, N]
 *                      (16-byte aligned, like bias and y_out); tok_scale and bias may be NULL.
 * One small kernel each, no host synchronisation. */
int pbl_act_f32_split(const float* x, int M, uint32_t K, size_t ldx, void* x_f16, float* tok_scale, void* stream);
int pbl_act_f32_join(const float* y_f32, const float* tok_scale, const float* bias, int M, uint32_t N, void* y_out, int out_dtype, void* stream);
/* y_out [M, N] (out_dtype) = cast((y_f32[t, r] (+ y_f32[M + t, r] if two_terms) + lo_scale * y_lo[t, r]) * tok_scale[t] + bias[r]): the terms
 * of an fp32-grid layer multiplied from its two images (above; lo_scale = 2^-12).  y_f32 is [2 M, N] for fp32 activations (the two fp16
 * terms of pbl_act_f32_split through the ordinary image) or [M, N] (two_terms = 0: fp16 / scaled bf16 activations); y_lo [M, N] is the HIGH
 * activation term through the residual image.  tok_scale / bias may be NULL; pointers 16-byte aligned.  One small kernel. */
int pbl_act_f32_join3(const float* y_f32, int two_terms, const float* y_lo, float lo_scale, const float* tok_scale, const float* bias, int M,
                      uint32_t N, void* y_out, int out_dtype, void* stream);

/* bf16 activations in ONE launch (decode: one GEMV pass, M <= 4 rows of a group-free layer): x [M, K] bf16 -> y [M, N] bf16 (fp32 with
 * y_f32).  The kernel's staging phase does what pbl_act_bf16_prepare does and its epilogue what pbl_act_finish does -- the same bits
 * as the three-launch form.  PBL_ERR_UNSUPPORTED: more rows than one pass takes, or column groups -- run pbl_act_bf16_prepare +
 * pbl_linear_f16_ws (fp32 out) + pbl_act_finish instead.  pbl_gemv_bf16_fused_host: pbl_gemv_f16_fused_host with bf16 x and a bf16
 * (fp32) result.  Replaces F.linear under `bf16=True` (qat/run_qat.py:120) at decode time. */
int pbl_linear_bf16(const pbl_layer* layer, const void* x_bf16, void* y, int M, int y_f32, void* stream);
int pbl_gemv_bf16_fused_host(const pbl_layer* layers_host, const uint64_t* y_off_host, const void* x_bf16, void* y, int L, int M,
                             uint32_t ldy, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, int y_f32, void* stream);

/* Decode-time FUSED projections (q/k/v, gate/up: layers that read the same activation): L layers, ONE x [M, K] (fp16,
 * 16-B aligned rows are not required), ONE output matrix y [M, ldy] in which layer l owns the columns
 * [y_off[l], y_off[l] + N_l).  Unlike pbl_gemv_f16_grouped no pointer table refers to x or y, so the caller may pass
 * freshly allocated tensors on every call and capture the launch in a hipGraph.  M <= 4; biases through the pbl_layer
 * entries as usual.  The callers this replaces: three / two separate nn.Linear calls per decoder layer
 * (gptq_pb/eval_ppl_utils.py:55-64 via the HF attention / MLP modules). */
int pbl_gemv_f16_fused(const pbl_layer* layers_dev, const uint64_t* y_off_dev, const void* x, void* y, int L, int M,
                       uint32_t ldy, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, int y_f32, void* stream);
/* The same launch for at most PBL_FUSED_INLINE_MAX layers with HOST arrays: the descriptors and offsets are copied into the
 * kernel arguments, so no device table is read (one dependent memory round trip less per workgroup; q/k/v and gate/up groups
 * are 3 and 2 layers).  Graph-capturable like the device-table form: the arguments are captured by value. */
#define PBL_FUSED_INLINE_MAX 4
int pbl_gemv_f16_fused_host(const pbl_layer* layers_host, const uint64_t* y_off_host, const void* x, void* y, int L, int M,
                            uint32_t ldy, uint32_t max_NRB, uint32_t K, uint32_t max_nch, int group_flags, int y_f32, void* stream);

/* ---------------- QAT step, weight side ------------------------------------------ */
/* The elementwise work of one training step of BinaryXnorExceptOutliersLinear
 * (quant/outlier_quantizer.py:83-99; straight-through estimator quant/quantizer.py:18-25), fused into
 * streaming kernels; the step's GEMMs stay library GEMMs.  W, mask (1 byte per element, nonzero = salient,
 * the layout of a torch.bool tensor), out, g: device pointers, 16-byte aligned, n = N*K elements.
 * dtype codes PBL_DTYPE_*.  scale: DEVICE float (written by pbl_qat_scale, read by the others -- no host sync). */
#define PBL_DTYPE_F32 0
#define PBL_DTYPE_F16 1
#define PBL_DTYPE_BF16 2
#define PBL_QAT_PARTIALS 1024
/* bytes of device scratch pbl_qat_scale needs */
size_t pbl_qat_workspace_bytes(void);
/* *scale_out = mean |W_i| over mask_i == 0 (binary_scale, outlier_quantizer.py:90-93); nan if none.
 * Deterministic (fixed grid and tree). */
int pbl_qat_scale(const void* W, int w_dtype, const uint8_t* mask, size_t n, void* workspace, float* scale_out, void* stream);
/* out_i = mask_i ? W_i * outlier_scale : sign(W_i) * scale   (outlier_quantizer.py:94-98), sign(0) = 0.
 * Products are rounded to W's dtype like the reference, then converted to out_dtype (f32 -> f16/bf16 is the
 * autocast cast in front of F.linear).  Supported: f32->f32/f16/bf16, f16->f16, bf16->bf16. */
int pbl_qat_wsim(const void* W, int w_dtype, const uint8_t* mask, const float* scale, float outlier_scale,
                 void* out, int out_dtype, size_t n, void* stream);
/* in place: g_i (= dL/dw_sim_i) *= mask_i ? (train_outlier ? outlier_scale : 0) : scale   -> dL/dW_i. */
int pbl_qat_wgrad(void* g, int g_dtype, const uint8_t* mask, const float* scale, float outlier_scale, int train_outlier,
                  size_t n, void* stream);

/* ---------------- producer side: salient selection and the 8-bit row quantizer -------------------- */
/* BinaryXnorExceptOutliersLinear.gen_outlier_mask (quant/outlier_quantizer.py:54-81) on the device, bit-identical
 * to the reference's CPU arithmetic.  W: device, n = N*K elements, dtype PBL_DTYPE_*; everything stays on `stream`. */
size_t pbl_prep_workspace_bytes(void);
/* out2[0] = k_lo-th smallest, out2[1] = k_hi-th smallest element of W (1-based ranks, torch.kthvalue :57-66),
 * as DEVICE floats.  Exact: 3-pass radix select over order-preserving keys.  k outside [1, n] -> INVALID_ARG
 * (torch raises).  workspace: pbl_prep_workspace_bytes() device bytes, 16-B aligned. */
int pbl_kth_pair(const void* W, int w_dtype, size_t n, uint64_t k_lo, uint64_t k_hi, void* workspace, float* out2, void* stream);
/* mask_out[i] = (W[i] < thr2[0]) | (W[i] > thr2[1])   (strict, :69); one byte per element (torch.bool layout). */
int pbl_outlier_mask(const void* W, int w_dtype, size_t n, const float* thr2, uint8_t* mask_out, void* stream);
/* weight_quant_8bit(W, simulated=True) IN PLACE, per row of W[N,K] (quant/outlier_quantizer.py:10-29: integer-rounded
 * zero point, wrapping uint8 cast, arithmetic in the reference's dtypes and order).  code_scale[r] = range_r / 255 and
 * code_zp[r] = zero point: W_hat[r,j] = code * code_scale[r] + code_zp[r] with an integer code 0..255.  K <= 16384. */
int pbl_quant8_rows(void* W, int w_dtype, uint32_t N, uint32_t K, float* code_scale, float* code_zp, void* stream);

/* HighQuantizer.calibrate(weight=True) as gptq_pb/run.py:132-137 configures it (per channel, asymmetric, no mse
 * search; gptq_pb/high_quant.py:29-67,95-102): scale[r] = (max(row,0) - min(row,0)) / maxq, zero[r] = round(-min/scale),
 * correctly rounded (bit-identical to the reference on the host).  W [N,K] fp32 device. */
int pbl_high_calibrate(const float* W, uint32_t N, uint32_t K, float maxq, float* scale, float* zero, void* stream);

/* One 128-column block of LowHighGPT.fasterquant's column loop (gptq_pb/gptq.py:129-168), all rows in one launch.
 * W [N,K] fp32 (columns c0 .. c0+ncols are read, then overwritten with their quantised values); U [K,K] fp32 = upper
 * Cholesky factor of H^-1 (:76-81); low_mask [N,K] bytes, nonzero = binarized (:83-99); hscale/hzero [N], maxq: the
 * HighQuantizer (high_quant.py:6-8); mean/scale [N]: the LowQuantizer row parameters of the block's group
 * (low_quant.py:25-32,75-82); err_out [N,128] receives Err1 (zero padded) for the caller's trailing update
 * W[:, c0+ncols:] -= Err1 @ U[c0:c0+ncols, c0+ncols:] (a library GEMM); losses [N] += sum_i (w-q)^2/d^2 / 2.
 * feedback == 0: round-to-nearest branch (:119-127), no error propagation.  ncols <= 128. */
int pbl_gptq_block(float* W, uint32_t N, uint32_t K, uint32_t c0, uint32_t ncols, const float* U, const uint8_t* low_mask,
                   const float* hscale, const float* hzero, float maxq, const float* mean, const float* scale,
                   float* err_out, float* losses, int feedback, void* stream);

/* ---------------- multi-GPU: one-shot all-reduce of K-split partial outputs -----------------------
 * The reference has no multi-GPU path (evaluate.py:56-62).  One process per GPU; every rank creates one communication
 * buffer with pbl_comm_alloc (THE exception to "nothing here allocates": a peer-mapped buffer must be its own
 * allocation), exports it (pbl_ipc_export -> 64-byte handle, exchanged by the host layer, e.g. torch.distributed
 * all_gather), opens the other ranks' handles (pbl_ipc_open) and then calls pbl_p2p_allreduce_f32 with the table of
 * mapped pointers: x[0:n] (fp32, device) <- sum over ranks, in rank order (bit-identical on every rank), as ONE kernel:
 * push to the peers' slots over xGMI, flag, bounded wait, local sum (csrc/pbl_comm.hip).  max_elems: the capacity the
 * buffers were sized for (pbl_p2p_buffer_bytes_world(max_elems, world); pbl_p2p_buffer_bytes sizes for 16 ranks).
 * The call number that tags the flags is either an argument (pbl_p2p_allreduce_f32: seq = 1, 2, 3, ... the same on every
 * rank) or kept in the buffer itself (pbl_p2p_allreduce_f32_dev): the launch then has no per-call argument, so it can be
 * captured in a hipGraph together with the K-split GEMV in front of it and replayed; y_f16 (optional, device, n halves)
 * additionally receives the sum rounded to fp16 -- the K-split layer's output dtype -- which saves the cast launch.  Use ONE of
 * the two forms per buffer.  A wait that times out (3 s: a peer died or never launched) sets the buffer's status word
 * (pbl_p2p_check) and writes NaN over the slice: a wrong sum is never silent. */
#define PBL_P2P_MAX_WORLD 16
#define PBL_P2P_MAX_BLOCKS 64
#define PBL_IPC_HANDLE_BYTES 64
size_t pbl_p2p_buffer_bytes(size_t max_elems);
size_t pbl_p2p_buffer_bytes_world(size_t max_elems, int world);
int pbl_comm_alloc(size_t bytes, void** dev_ptr_out);          /* zero-filled device memory, uncached where supported */
int pbl_comm_free(void* dev_ptr);
int pbl_ipc_export(void* dev_ptr, void* handle64_out);
int pbl_ipc_open(const void* handle64, void** dev_ptr_out);
int pbl_ipc_close(void* dev_ptr);
int pbl_p2p_allreduce_f32(void* const* peer_bufs, int rank, int world, float* x, size_t n, size_t max_elems, uint32_t seq,
                          void* stream);
int pbl_p2p_allreduce_f32_dev(void* const* peer_bufs, int rank, int world, float* x, void* y_f16, size_t n, size_t max_elems,
                              void* stream);
/* Fused K-split layer (round 4): pbl_linear_f16_push runs this rank's column shard as ONE GEMV pass (M <= 4, group-free layer)
 * whose row owners write their fp32 partials straight into slot [call & 1][rank] of EVERY rank's communication buffer (7 xGMI
 * stores + 1 local) and count each finished 16-row record at the peers; pbl_p2p_reduce_f32_dev waits until every rank has
 * pushed `expect_records` (= layer->NRB) records, sums the slots in rank order into y_f32 and / or y_f16 (either may be NULL) and
 * publishes the call number (device counted: both launches are hipGraph-capturable and replayable).  Against pbl_linear_f16 +
 * pbl_p2p_allreduce_f32_dev the partial never makes the round trip through local HBM and the push has no launch of its own.
 * PBL_ERR_UNSUPPORTED from the push (column groups, more tokens than one pass takes): run the unfused pair.  The bias is added by
 * rank 0's partial only. */
int pbl_linear_f16_push(const pbl_layer* layer, const void* x, int M, void* const* peer_bufs, int rank, int world, size_t max_elems,
                        void* stream);
/* The largest M pbl_linear_f16_push takes for THIS shard (0: never -- column groups).  It depends on the shard's own data
 * (its fullest record sizes the LDS): the ranks of a K-split layer must agree on ONE limit (the minimum over the ranks) before
 * they choose between the fused pair and GEMV + all-reduce, or one rank waits on counters while its peer waits on flags. */
int pbl_linear_push_max_tokens(const pbl_layer* layer);
int pbl_p2p_reduce_f32_dev(void* const* peer_bufs, int rank, int world, float* y_f32, void* y_f16, size_t n, size_t max_elems,
                           uint32_t expect_records, void* stream);

int pbl_p2p_check(const void* own_buf);                        /* SYNCHRONOUS: 1 if a wait timed out since the last check (the word is then cleared) */

#ifdef __cplusplus
}
#endif
#endif /* PBL_H_ */
