/*
 * gcn_spmm.h — C-ABI of the MI355X (gfx950) GraphConvolution hot path.
 *
 * This is the drop-in boundary for the sparse x dense products of LinChen-65/pygcn's
 * GraphConvolution layer.  The reference has no FFI of its own for this path: it calls PyTorch,
 *
 *     output = torch.spmm(adj, support)              pygcn/layers.py:34     (forward,  A · B)
 *     grad_support = adj.t() @ grad_output           torch autograd `mm` mat2 formula, triggered
 *                                                    by loss.backward() at pygcn/train.py:157
 *     output + self.bias                             pygcn/layers.py:35-36  (fused epilogue)
 *
 * so the entry points below are what a binding for those three lines has to call.  Plain
 * pointers and sizes only — no torch types.  Every `const void*`/pointer marked DEVICE must be
 * hipMalloc'ed memory of the current device; the launchers never allocate, never synchronise
 * and enqueue on the stream they are given (so they are hipGraph-capturable).  The planner
 * entry points (`*_host`) are pure CPU code on HOST arrays.
 *
 * Return value of every function: 0 on success, a hipError_t (> 0) for HIP failures,
 * a negative GCN_E_* for argument errors.  gcn_last_error() gives a message.
 */
#ifndef GCN_SPMM_H
#define GCN_SPMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCN_E_BADARG   (-1)   /* null pointer / negative size / inconsistent plan            */
#define GCN_E_ALIGN    (-2)   /* (reserved) alignment the selected kernel cannot handle       */
#define GCN_E_WORKSPACE (-3)  /* workspace smaller than gcn_spmm_workspace_bytes()            */
#define GCN_E_CAPACITY (-4)   /* planner output arrays too small                              */

#define GCN_DTYPE_F32  0      /* B, C fp32; fp32 accumulate  (configs C1-C4)                  */
#define GCN_DTYPE_BF16 1      /* B, C bf16 storage; fp32 values and accumulate (config C5)    */

#define GCN_ABI_VERSION 25

#define GCN_DEFAULT_ITEM_COST   64     /* work units (stored entries + rows) per row-batch item */
#define GCN_DEFAULT_LONG_THRESH 256    /* rows with more stored entries are chunked             */

/*
 * A CSR adjacency (or its transpose) plus the static launch schedule built for it once.
 * All pointers are DEVICE pointers.  The schedule splits the rows of the matrix into
 *   - row-batch items : runs of <= 64 consecutive "short" rows whose stored entries are
 *                       contiguous; one wavefront streams one item;
 *   - long rows       : rows with more than `long_thresh` stored entries, cut into chunks of
 *                       `long_thresh` entries that are summed by separate wavefronts into a
 *                       partial slab and then added in chunk order (bitwise reproducible;
 *                       no float atomics).
 * Replaces: the torch sparse tensor `adj` the reference builds at pygcn/utils.py:407-414 and
 * passes to GraphConvolution.forward (pygcn/layers.py:32).
 */
typedef struct gcn_csr_plan {
    int64_t n_rows;
    int64_t n_cols;
    int64_t nnz;
    const void *rowptr;        /* [n_rows+1] int32 or int64                                  */
    int32_t rowptr_is64;
    int32_t long_thresh;       /* chunk length L for long rows                               */
    const int32_t *col;        /* [nnz] column index of every stored entry                   */
    const float *val;          /* [nnz] fp32 value of every stored entry                     */
    int64_t n_items;
    const int32_t *items;      /* [2*n_items] (first_row, end_row) per row-batch item        */
    int64_t n_chunks;
    const int32_t *chunk_row;  /* [n_chunks] row every chunk belongs to                      */
    const int64_t *chunk_e0;   /* [n_chunks] first stored entry of the chunk                 */
    int64_t n_long;
    const int32_t *long_row;   /* [n_long] row index of every long row                       */
    const int32_t *long_chunk0;/* [n_long+1] first chunk of every long row                   */
} gcn_csr_plan;

/* ABI history: 21 = round 2's surface.  25 (round 4, late): new entry point gcn_gemm_atg256_f32_b3_colsum (the
 * weight gradient with the bias gradient Σ G[rows] as a side result); gcn_gemm_atg256_workspace_bytes grew by
 * 1 KiB per workgroup; struct gcn_gemm_epilogue gained keep_bits_out / mask_bits at its END (zero them);
 * nothing else changed.  24 (round 4): new entry points gcn_gemm_xw256_f32_b3 /
 * gcn_gemm_atg256_f32_b3 (the fp32-EQUIVALENT three-part bf16 form of the 256-wide GEMMs with the
 * full option set of the _h2 entry points: row lists, forward epilogue, backward mask, max|Y|) and
 * gcn_gemm_xw256_b3_workspace_bytes; nothing existing changed.  23 (round 3, late): at p = 1/2 the dropout keep function
 * draws 128 one-bit fields per Philox call (other p unchanged) — masks at p = 1/2 differ from ABI
 * 22's; no signature or struct changed.  22 (round 3): the dropout keep function draws eight 16-bit
 * fields per Philox call instead of four 32-bit words and takes a row base (drop_row_base in both
 * epilogue structs) and the product can report max|result| (gcn_epilogue.c_absmax); new entry
 * points gcn_nll_log_softmax_backward_colsum, gcn_gemm_atg_bf16, gcn_sddmm_csr, gcn_rows_pack_count /
 * gcn_rows_pack_values / gcn_rows_unpack / gcn_bits_row_counts; gcn_gemm_xw_bf16
 * takes a backward mask.  The three options that leave
 * rows of an output unwritten (c_skip_zero_rows, c_row_select, skip_zero_rows of the backward
 * sweeps) are EXPERIMENTAL: the product uses them only inside single autograd nodes that own both
 * the producer and every consumer of such a tensor (pygcn_amd/fused.py). */
/* ABI version of the loaded library (GCN_ABI_VERSION). */
int gcn_abi_version(void);

/* Message for the last non-zero return on this thread. */
const char *gcn_last_error(void);

/*
 * Planner, pass 1: count items / chunks / long rows for a HOST rowptr.
 *   item_cost   : target work per row-batch item, in units of (stored entries + rows); <= 0
 *                 selects GCN_DEFAULT_ITEM_COST.
 *   long_thresh : rows with more stored entries than this are chunked; <= 0 selects the
 *                 GCN_DEFAULT_LONG_THRESH.
 */
int gcn_plan_count_host(const void *rowptr_host, int rowptr_is64, int64_t n_rows,
                        int32_t item_cost, int32_t long_thresh, int64_t *n_items,
                        int64_t *n_chunks, int64_t *n_long);

/*
 * Planner, pass 2: fill caller-allocated HOST arrays (sizes from pass 1):
 *   items[2*n_items], chunk_row[n_chunks], chunk_e0[n_chunks], long_row[n_long],
 *   long_chunk0[n_long+1].  The caller copies them to the device and points a gcn_csr_plan
 *   at the copies.
 */
int gcn_plan_fill_host(const void *rowptr_host, int rowptr_is64, int64_t n_rows,
                       int32_t item_cost, int32_t long_thresh, int32_t *items, int64_t n_items,
                       int32_t *chunk_row, int64_t *chunk_e0, int64_t n_chunks,
                       int32_t *long_row, int32_t *long_chunk0, int64_t n_long);

/*
 * The same planner ON THE DEVICE: rowptr is a DEVICE array and never visits
 * the host; the result equals gcn_plan_{count,fill}_host array for array.  Two calls around one
 * 24-byte read:
 *   gcn_plan_count_device  computes the schedule into `workspace` and writes
 *                          counts[0..2] = (n_items, n_chunks, n_long) to DEVICE memory;
 *   (the caller copies the three counts to the host and allocates the plan arrays on the device)
 *   gcn_plan_fill_device   writes items[2*n_items], chunk_row[n_chunks], chunk_e0[n_chunks],
 *                          long_row[n_long], long_chunk0[n_long+1] (DEVICE arrays) from the
 *                          workspace gcn_plan_count_device left behind (same stream / ordered).
 * Greedy segmentation in parallel: per-row item ends by binary search in scanned row costs, item
 * starts by pointer doubling over "next item start" (pygcn_amd/csrc/gcn_plan.hip).  Scratch:
 * gcn_plan_device_workspace_bytes(n_rows) (48 B per row + scan temporaries).
 */
size_t gcn_plan_device_workspace_bytes(int64_t n_rows);
int gcn_plan_count_device(const void *rowptr, int rowptr_is64, int64_t n_rows, int32_t item_cost,
                          int32_t long_thresh, void *workspace, size_t workspace_bytes,
                          int64_t *counts, void *stream);
int gcn_plan_fill_device(const void *rowptr, int rowptr_is64, int64_t n_rows, int32_t long_thresh,
                         const void *workspace, size_t workspace_bytes, int32_t *items,
                         int64_t n_items, int32_t *chunk_row, int64_t *chunk_e0, int64_t n_chunks,
                         int32_t *long_row, int32_t *long_chunk0, int64_t n_long, void *stream);

/* Bytes of DEVICE scratch gcn_spmm_csr() needs for feature width F (0 if no long rows). */
size_t gcn_spmm_workspace_bytes(const gcn_csr_plan *plan, int64_t F);

/*
 * C[n_rows, F] = A · B  (+ bias[F], then ReLU, both optional) on `stream`.
 *
 *   plan       : the CSR matrix A [n_rows, n_cols] and its schedule.
 *   dtype      : GCN_DTYPE_F32 or GCN_DTYPE_BF16 (element type of B and C).
 *   B          : DEVICE [n_cols, F] row-major with leading dimension ldb (elements).
 *   C          : DEVICE [n_rows, F] row-major with leading dimension ldc (elements);
 *                every row is written (rows without stored entries become bias / zero).
 *   bias       : DEVICE fp32 [F] or NULL                      — pygcn/layers.py:35-36.
 *   relu       : non-zero applies max(x, 0) after the bias    — pygcn/models.py:48 (upstream).
 *   workspace  : DEVICE scratch of at least gcn_spmm_workspace_bytes(plan, F) bytes
 *                (may be NULL when that is 0).
 *
 * Forward of pygcn/layers.py:34 when `plan` describes adj; the backward product
 * adj.t() @ grad_output when `plan` describes CSR(adj^T) (see gcn_csr_transpose_*).
 */
int gcn_spmm_csr(const gcn_csr_plan *plan, int dtype, const void *B, int64_t ldb, void *C,
                 int64_t ldc, int64_t F, const float *bias, int relu, void *workspace,
                 size_t workspace_bytes, void *stream);

/*
 * Epilogue applied to every output row inside the kernel's store, in this order:
 *   x = acc + bias[f]            (bias may be NULL)          — pygcn/layers.py:35-36
 *   x = max(x, 0)                if relu                     — F.relu,   pygcn/models.py:48 (upstream)
 *   x = keep ? x * s : 0         if dropout_p > 0            — F.dropout, pygcn/models.py:50 (upstream)
 * The keep bit of element (row, f) is a pure function of (seed, drop_row_base + row, f), identical
 * for every kernel variant.  T = clamp(round(p * 65536), 1, 65535) (p is honoured to 2^-17; p = 1/2
 * exactly); s = 65536 / (65536 - T) = 1 / (keep probability of that T) (ABI 24; before: 1 / (1 - p) of
 * the unquantised p), so E[dropout(x)] = x exactly — the backward scale of a caller must be the
 * same number (pygcn_amd.spmm.dropout_scale); w = Philox4x32-10(counter = (row_lo, row_hi, block, 0), key = (seed_lo, seed_hi)), and
 *   T != 32768 (ABI 22: eight 16-bit fields per call):
 *     block = ((f >> 4) << 1) | ((f >> 2) & 1),   field = (((f >> 3) & 1) << 2) | (f & 3)
 *     keep  = ((w[field >> 1] >> 16 * (field & 1)) & 0xFFFF) >= T
 *   T == 32768, i.e. p = 1/2 — one bit decides (ABI 23: 128 one-bit fields per call):
 *     block = ((f >> 8) << 1) | ((f >> 2) & 1),   index = (((f & 255) >> 3) << 2) | (f & 3)
 *     keep  = (w[index >> 5] >> (index & 31)) & 1
 * Because out > 0 <=> (pre-activation > 0 and kept), no mask is stored: the backward pass is
 * gcn_relu_dropout_backward on the output itself.
 */
typedef struct gcn_epilogue {
    const float *bias;   /* DEVICE fp32 [F] or NULL */
    int32_t relu;
    float dropout_p;     /* in [0, 1); 0 disables dropout */
    uint64_t seed;
    /* Optional hint about the dense operand B (both NULL = none): bit c of the bitmap
     * b_row_nonzero[ceil(n_cols/32)] is clear for rows of B that are entirely zero, *b_nnz_rows
     * counts the rows whose bit is set.  Such rows are
     * not gathered (their products are zero anyway, so the result is unchanged up to summation
     * order); the hint is ignored on the device when it cannot pay (3/4 or more of the rows
     * non-zero for rows of >= 528 bytes, 1/8 or more for narrower rows).  Produced for free by
     * gcn_relu_dropout_backward_colsum — the gradients of a semi-supervised loss
     * (`nll_loss(output[idx_train], ...)`, pygcn/train.py) are non-zero on few rows. */
    const uint32_t *b_row_nonzero;
    const int32_t *b_nnz_rows;
    /* Optional second block of the dense operand (NULL = B is one block): rows b_split, b_split+1,
     * ... of B are read from b2 (row-major, leading dimension ldb2, same dtype) instead.  Lets the
     * sharded path multiply with [own activation rows | received halo rows] without copying the
     * own rows next to the halo buffer. */
    const void *b2;
    int64_t ldb2;
    int64_t b_split;
    /* Optional OUTPUT (NULL = none): byte flags [n_rows], zeroed by the caller; c_row_nonzero[r]
     * is set to 1 where row r of the stored result has a non-zero element.  Lets the consumer of a
     * row-sparse product (the weight / input gradient GEMMs behind A^T · grad) skip the zero rows
     * without another pass over the result. */
    uint8_t *c_row_nonzero;
    /* Non-zero: store log_softmax over each row of (A·B + bias) instead of the row itself — the
     * `F.log_softmax(x, dim=1)` that ends the reference model (pygcn/models.py:52 upstream).
     * The row must live in one wavefront's store: F <= 64 elements, or F a multiple of the 16-byte
     * lane width v (4 fp32 / 8 bf16) with F/v <= 64 and 16-byte aligned operands; not combinable
     * with relu / dropout.  Backward: gcn_log_softmax_backward_colsum. */
    int32_t log_softmax;
    /* Optional DEVICE pointer to the dropout seed (NULL: use `seed`).  The kernel reads the seed
     * when it runs, not when it is enqueued, so a launch captured into a hipGraph draws a fresh
     * mask on every replay if the graph also updates *seed_dev (e.g. a captured increment). */
    const uint64_t *seed_dev;
    /* Optional bitmap over the OUTPUT rows (NULL = all rows wanted): rows whose bit in
     * c_row_select[ceil(n_rows/32)] is clear are not needed by the caller — the kernel may skip
     * their stored entries and leave those rows of C unwritten (contents undefined).  Lets
     * grad_W = (A·X)^T · grad of a first layer be formed from the rows of A·X that meet a non-zero
     * row of grad only (the bitmap gcn_relu_dropout_backward_colsum already produced). */
    const uint32_t *c_row_select;
    /* With c_row_nonzero: rows of the result that are entirely zero are NOT stored (their flag
     * stays 0, their memory is left untouched) — for a consumer that reads the flagged rows only.
     * Rows longer than the plan's long_thresh are always stored. */
    int32_t c_skip_zero_rows;
    /* Added to the row index in the dropout counter (ABI 22): a row-block shard passes the global
     * index of its first row, so the masks of a sharded run are those of the single-GPU run. */
    int64_t drop_row_base;
    /* Optional OUTPUT (NULL = none; ABI 22): one DEVICE float, zeroed by the caller, that receives
     * max |stored value| of the launch (atomic max of the bit patterns: inf / NaN patterns sort
     * above every finite value, so an overflow is never lost) — the bound the scaled GEMMs that
     * consume this product need (gcn_gemm_xw256_f32_h2: x_absmax_bound), without a reduction pass
     * over the result.  One atomic per wavefront at most. */
    float *c_absmax;
} gcn_epilogue;

/* gcn_spmm_csr with the full epilogue (ep may be NULL: plain product). */
int gcn_spmm_csr_ep(const gcn_csr_plan *plan, int dtype, const void *B, int64_t ldb, void *C,
                    int64_t ldc, int64_t F, const gcn_epilogue *ep, void *workspace,
                    size_t workspace_bytes, void *stream);

/*
 * grad_pre[i] = out[i] > 0 ? grad_out[i] * scale : 0 over n_elems contiguous elements
 * (scale = 1 / (1 - p); 1 for a bare ReLU).  DEVICE pointers; grad_pre may alias grad_out.
 * Backward of the fused epilogue above (autograd of F.relu + F.dropout in the reference model).
 */
int gcn_relu_dropout_backward(int dtype, const void *grad_out, const void *out, void *grad_pre,
                              int64_t n_elems, float scale, void *stream);

/*
 * The same backward pass for fp32 or bf16 [n_rows, F] row-major tensors (`dtype`; colsum is always
 * fp32 and, for bf16, sums the rounded values that were stored), producing in the same sweep the
 * column sums of its result: colsum[f] = sum_rows grad_pre[row, f] — the bias gradient of
 * `output + self.bias` (pygcn/layers.py:35-36).  `out == NULL` skips the masking (plain column
 * sums of grad_out; grad_pre is then ignored).  Deterministic (per-block partial rows added in
 * block order, no float atomics).  F must be a multiple of the 16-byte lane width v (4 fp32 / 8
 * bf16) with F/v dividing 256; scratch: gcn_bwd_colsum_workspace_bytes(n_rows, F, dtype).
 * Optional outputs (both or neither; F/v <= 64, GCN_E_BADARG for wider rows — they are never
 * silently left unwritten): bit r of row_bits[ceil(n_rows/32)] is set where
 * row r of the result has a non-zero element, *nnz_rows = how many — the B-operand hint of
 * gcn_epilogue.  skip_zero_rows != 0 (needs those outputs and `out`): rows of the result that are
 * entirely zero are NOT written to grad_pre — for a consumer that reads the flagged rows only
 * (the hinted product below 3/4 resp. 1/8 non-zero rows, the row-compacted GEMMs); at a 5 %
 * labelled share that removes 95 % of the pass's writes.
 */
size_t gcn_bwd_colsum_workspace_bytes(int64_t n_rows, int64_t F, int dtype);
/*
 * Backward of the fused log_softmax epilogue with the same by-products: out = log_softmax(z) row
 * by row, grad_pre = grad_out - exp(out) * rowsum(grad_out), colsum[f] = sum_rows grad_pre[row, f],
 * and the row bitmap / count of grad_pre.  Rows of grad_out that are entirely zero (every vertex
 * outside idx_train) give zero rows without `out` being read.  Same shape rules and scratch as
 * gcn_relu_dropout_backward_colsum, with F/v <= 64 required (a row inside one wavefront).
 */
int gcn_log_softmax_backward_colsum(int dtype, const void *grad_out, const void *out, void *grad_pre,
                                    float *colsum, int64_t n_rows, int64_t F, uint32_t *row_bits,
                                    int32_t *nnz_rows, int skip_zero_rows, void *workspace,
                                    size_t workspace_bytes, void *stream);
int gcn_relu_dropout_backward_colsum(int dtype, const void *grad_out, const void *out, void *grad_pre,
                                     float *colsum, int64_t n_rows, int64_t F, float scale,
                                     uint32_t *row_bits, int32_t *nnz_rows, int skip_zero_rows,
                                     void *workspace, size_t workspace_bytes, void *stream);
/*
 * The same sweep for the gradient of a MEAN NLL LOSS OVER ALL ROWS — `F.nll_loss(output, labels)`,
 * the reference's loss line (pygcn/train.py:153) without its index selection; the fork's live
 * loss likewise reduces over every vertex (pygcn/train.py:151-155).  That gradient has one
 * non-zero per row, grad_out[r][target[r]] = *coef (coef = -upstream / n_rows, DEVICE float), so it
 * is never materialised: grad_pre = *coef * (onehot(target[r]) - exp(out[r])) and its column sums
 * come straight from `out` (log-probabilities, [n_rows, F]) and the label vector `target` (DEVICE
 * int64 [n_rows], entries in [0, F); a NEGATIVE entry marks an ignored row — torch's ignore_index =
 * -100 — whose gradient row is zero; the caller's coef then divides by the rows that count).
 * 2 full-height streams (read out, write grad_pre) instead of 4.  Shape rules and scratch of
 * gcn_log_softmax_backward_colsum.  (ABI 22.)
 */
int gcn_nll_log_softmax_backward_colsum(int dtype, const int64_t *target, const float *coef,
                                        const void *out, void *grad_pre, float *colsum, int64_t n_rows,
                                        int64_t F, void *workspace, size_t workspace_bytes, void *stream);

/*
 * SDDMM on the pattern of `plan`: out_vals[e] = < G[row(e), :], B[col[e], :] > for every stored
 * entry e (fp32 results; G [n_rows, F] and B [n_cols, F] row-major fp32 or bf16, leading dimensions
 * in elements).  The gradient of the adjacency VALUES for a caller that sets adj.requires_grad —
 * PyTorch's `mm` derivative for a sparse first operand, (grad · mat2^T) sampled on self's pattern;
 * the reference never needs it (its adj is a constant: pygcn/train.py:80,123), SURVEY row f4 lists
 * it as optional.  Uses the schedule of `plan` (items, long-row chunks); no workspace, no atomics,
 * deterministic.  (ABI 22.)
 */
int gcn_sddmm_csr(const gcn_csr_plan *plan, int dtype, const void *G, int64_t ldg, const void *B,
                  int64_t ldb, int64_t F, float *out_vals, void *stream);

/*
 * CSR(A^T) on the HOST from CSR(A) on the HOST: stable counting sort by column, so each row of
 * A^T lists its entries in increasing source-row order (deterministic backward sums).
 * rowptr_t[n_cols+1] has the width of rowptr; col_t[nnz]; val_t[nnz].
 */
int gcn_csr_transpose_host(const void *rowptr_host, int rowptr_is64, const int32_t *col,
                           const float *val, int64_t n_rows, int64_t n_cols, void *rowptr_t,
                           int32_t *col_t, float *val_t);

/*
 * Device-side ingest — everything below runs on `stream`, never allocates.
 *
 * gcn_csr_transpose_device: CSR(A) -> CSR(A^T) entirely on the DEVICE (stable radix sort by
 * column carrying (source row, value)); each row of A^T lists its entries in increasing
 * source-row order, exactly like gcn_csr_transpose_host.  All array arguments are DEVICE
 * pointers; rowptr_t has the width of rowptr.  Scratch: gcn_csr_transpose_workspace_bytes().
 * Replaces the transposed view PyTorch re-derives on every backward call of torch.spmm
 * (autograd of pygcn/layers.py:34).
 */
size_t gcn_csr_transpose_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz);
int gcn_csr_transpose_device(const void *rowptr, int rowptr_is64, const int32_t *col,
                             const float *val, int64_t n_rows, int64_t n_cols, int64_t nnz,
                             void *rowptr_t, int32_t *col_t, float *val_t, void *workspace,
                             size_t workspace_bytes, void *stream);

/*
 * COO -> CSR on the DEVICE with duplicate reduction: the adjacency construction of the reference's
 * data recipe (pygcn/utils.py:360-368: scipy `coo_matrix(...)`, symmetrization, `+ I`) and of
 * `sparse_mx_to_torch_sparse_tensor` consumers (utils.py:407-414: int64 [2, nnz] indices, fp32
 * values, uncoalesced).  Entries may come in any order; entries with the same (row, col) are
 * reduced IN STORAGE ORDER with
 *   GCN_REDUCE_SUM  what scipy's coo->csr conversion and torch.spmm on an uncoalesced tensor do;
 *   GCN_REDUCE_MAX  elementwise maximum — `adj + adj.T*(adj.T > adj) - adj*(adj.T > adj)`
 *                   (utils.py:365) is max(adj, adj.T), i.e. MAX over the entries of adj and adj.T.
 * Outputs: rowptr_out[n_rows+1] (int32 or int64), col_out / val_out with capacity nnz (sorted by
 * column inside every row), *nnz_out (DEVICE) = number of distinct entries.  All pointers DEVICE;
 * the caller guarantees 0 <= row < n_rows, 0 <= col < n_cols.  Deterministic, no atomics.
 */
#define GCN_REDUCE_SUM 0
#define GCN_REDUCE_MAX 1
size_t gcn_coo_to_csr_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz);
int gcn_coo_to_csr_device(const int64_t *row, const int64_t *col, const float *val, int64_t nnz,
                          int64_t n_rows, int64_t n_cols, int reduce, void *rowptr_out,
                          int rowptr_is64, int32_t *col_out, float *val_out, int64_t *nnz_out,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * val <- D^-1 · val in place on the DEVICE: every stored entry is divided by the sum of its row;
 * rows that sum to 0 stay 0.  The reference's `normalize(mx)` (pygcn/utils.py:390-397).
 */
int gcn_row_normalize_device(const void *rowptr, int rowptr_is64, float *val, int64_t n_rows,
                             void *stream);

/*
 * Y[M, 256] = X[M, 256] · W[256, 256] — the dense half of the layer, `torch.mm(input, weight)`
 * (pygcn/layers.py:33) and the grad_input GEMM of its backward, for the hidden width 256 of
 * configs C3/C4.  fp32 in / out / accumulate; operands are split into three bf16 parts on the
 * fly and multiplied with six bf16 MFMAs (fp32-level accuracy, see gcn_gemm.hip).  DEVICE
 * pointers; X rows 16-byte aligned; workspace >= gcn_gemm_xw256_workspace_bytes() (384 KiB).
 */
size_t gcn_gemm_xw256_workspace_bytes(void);
int gcn_gemm_xw256_f32(const float *X, int64_t ldx, const float *W, int64_t ldw, float *Y, int64_t ldy,
                       int64_t M, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same product by a cheaper decomposition: both operands are scaled by exact powers of two and
 * split into TWO fp16 parts, three fp16 MFMAs per product (half the matrix work of the 3 x bf16
 * form; error <= ~2^-21 relative per product, normwise ~1e-6 against an fp64 product).  fp16 has
 * little range, so the caller supplies x_absmax_bound: a DEVICE float holding any upper bound of
 * max|X| (the tighter, the more of fp16's 2^17 usable dynamic range is kept below the maximum;
 * elements further below it lose relative — not absolute — precision).  y_absmax: optional DEVICE
 * float, zeroed by the caller, receives max|Y| (atomic max) — the bound a following layer needs.
 * x_rows: optional DEVICE int32 list of M row indices — output row r is then the product of input
 * row x_rows[r] (the gradient GEMMs run on the rows that can be non-zero without a compacting
 * copy); NULL = rows 0 .. M-1.
 * epilogue: optional store-side options (struct gcn_gemm_epilogue below; NULL = plain product);
 * y_absmax reports the maximum of the values actually stored.
 * Workspace >= gcn_gemm_xw256_h2_workspace_bytes().  `torch.mm(input, weight)`, pygcn/layers.py:33.
 */
typedef struct gcn_gemm_epilogue {
    /* FORWARD epilogue, for a layer evaluated as (Â·X)·W + b — pygcn/layers.py:33-36 reassociated,
     * the GEMM being the layer's last stage (Fin <= Fout with a constant X: the product Â·X of the
     * forward pass is then also all the backward pass needs for grad_W, pygcn_amd/fused.py):
     *   y = acc + bias[col]; y = max(y, 0) if relu; y = keep ? y / (1 - p) : 0 if dropout_p > 0.
     * The keep bit is the SAME function of (seed, row, col) as in struct gcn_epilogue — Philox4x32-10 — so
     * a mask does not depend on which kernel stored the element; dropout requires relu. */
    const float *bias;        /* DEVICE fp32 [N] (N = 256 for the fp32 kernel), 16-byte aligned, or NULL */
    int32_t relu;
    float dropout_p;          /* in [0, 1); 0 disables dropout */
    uint64_t seed;
    const uint64_t *seed_dev; /* optional DEVICE seed read at execution time (hipGraph replays) */
    /* BACKWARD mask (excludes the forward epilogue): y = mask_src[input row, col] > 0 ? y *
     * mask_scale : 0 — the backward of a fused ReLU / dropout epilogue (out > 0 encodes ReLU and
     * keep, scale = 1 / (1 - p)) applied to the grad_input GEMM in its own store. */
    const float *mask_src;    /* DEVICE [*, N] in the GEMM's storage type — fp32 for gcn_gemm_xw256_f32_h2, bf16
                               * (passed through this pointer) for gcn_gemm_xw_bf16 — leading dimension ld_mask, or NULL;
                               * rows 16-byte aligned (GCN_E_ALIGN) */
    int64_t ld_mask;
    float mask_scale;
    /* optional DEVICE int32 list [M]: output row r reads mask row mask_rows[r]; NULL = the input
     * row (x_rows[r], or r).  For a COMPACT input whose rows belong to listed rows of a full mask. */
    const int32_t *mask_rows;
    /* added to the row index in the dropout counter (see struct gcn_epilogue; ABI 22) */
    int64_t drop_row_base;
    /* ABI 25, gcn_gemm_xw256_f32_b3 on CONTIGUOUS rows only (x_rows NULL; GCN_E_BADARG otherwise): the result of
     * the forward epilogue as ONE BIT per element — `out > 0`, which is all the backward of ReLU / dropout asks
     * of `out` — 32 bytes per row (DEVICE uint32 [M][8], 8-byte aligned) in the kernel's own lane order (an
     * opaque layout: only mask_bits of the same entry point reads it).  keep_bits_out: written next to Y by a
     * launch with relu and dropout_p in {0, 1/2}.  mask_bits: the backward mask read from such bits (row
     * mask_rows[r], or r) INSTEAD of mask_src — 32 bytes per row instead of 1 KiB; excludes the forward
     * epilogue like mask_src; mask_scale applies. */
    uint32_t *keep_bits_out;
    const uint32_t *mask_bits;
} gcn_gemm_epilogue;

size_t gcn_gemm_xw256_h2_workspace_bytes(void);
int gcn_gemm_xw256_f32_h2(const float *X, int64_t ldx, const int32_t *x_rows, const float *W,
                          int64_t ldw, float *Y, int64_t ldy, int64_t M, const float *x_absmax_bound,
                          float *y_absmax, const gcn_gemm_epilogue *epilogue, void *workspace,
                          size_t workspace_bytes, void *stream);

/*
 * The fp32-EQUIVALENT form of the same product (ABI 24) — what `torch.mm(input, self.weight)`
 * (pygcn/layers.py:33) computes in fp32: both operands split into THREE bf16 parts (a 24-bit
 * significand, fp32's own), six bf16 MFMAs per product (every term down to 2^-16 of the leading
 * one; the dropped ones are <= 2^-24), fp32 accumulation.  bf16 has fp32's exponent range: no
 * scaling and no bound.  The same options as gcn_gemm_xw256_f32_h2 — x_rows, y_absmax, the whole
 * struct gcn_gemm_epilogue.  Two kernels behind it, both deterministic: contiguous rows (x_rows NULL;
 * every epilogue but dropout at p != 1/2) run on 128-row tiles whose stores leave under the next
 * tile's MFMAs (K summed in chunks of 32); a row list runs round 3's pipeline (K in chunks of 16,
 * bit-identical to gcn_gemm_xw256_f32 for the plain product).  The two differ in fp32 summation
 * order only (<= 1e-6 of a row's largest entry).  Workspace >= gcn_gemm_xw256_b3_workspace_bytes().
 */
size_t gcn_gemm_xw256_b3_workspace_bytes(void);
int gcn_gemm_xw256_f32_b3(const float *X, int64_t ldx, const int32_t *x_rows, const float *W,
                          int64_t ldw, float *Y, int64_t ldy, int64_t M, float *y_absmax,
                          const gcn_gemm_epilogue *epilogue, void *workspace, size_t workspace_bytes,
                          void *stream);

/*
 * Y[M, N] = X[M, K] · W[K, N] for bf16 storage (config C5: 128 -> 128): bf16 in / out, fp32
 * accumulate, (K, N) one of (128,128), (128,256), (256,128).  HBM-bound by construction (W resident in LDS, X streamed
 * once, Y written once as 16-byte stores).  `torch.mm(input, weight)` (pygcn/layers.py:33) and
 * the grad_input GEMM of its backward at bf16.  DEVICE pointers; rows 16-byte aligned; workspace
 * >= gcn_gemm_bf16_workspace_bytes(K, N) (0 = unsupported shape).
 * epilogue: optional FORWARD epilogue, struct gcn_gemm_epilogue: bias fp32 [N], relu, dropout (
 * applied to the fp32 accumulators before the rounding to bf16, the order of the SpMM epilogue);
 * its backward-mask fields must be unset (GCN_E_BADARG).  NULL = plain product.
 */
size_t gcn_gemm_bf16_workspace_bytes(int64_t K, int64_t N);
int gcn_gemm_xw_bf16(const void *X, int64_t ldx, const void *W, int64_t ldw, void *Y, int64_t ldy,
                     int64_t M, int64_t K, int64_t N, const gcn_gemm_epilogue *epilogue,
                     void *workspace, size_t workspace_bytes, void *stream);

/*
 * out[256, 256] = Σ_{r < n_list} A[rows_a[r], :]ᵀ ⊗ G[rows_g[r], :]  — the weight gradient
 * `inputᵀ · grad_support` of `torch.mm(input, weight)` (pygcn/layers.py:33; autograd of the call
 * at pygcn/train.py:157) for 256-wide layers, fp32 in / out, over a LIST of rows: rows_a / rows_g
 * are DEVICE int32 index lists naming the rows on which the gradient can be non-zero (an identity
 * list for "all rows"), so the operands need not be compacted first; a list must be PADDED to a
 * multiple of 16 entries with valid indices (n_list counts the real ones).  Scaled two-part fp16
 * scheme as gcn_gemm_xw256_f32_h2 (a_absmax_bound / g_absmax_bound: DEVICE floats, upper bounds
 * of max|A|, max|G| over the listed rows); partial products of row slabs are added in slab order
 * (deterministic).  Rows of A and G 16-byte aligned (GCN_E_ALIGN).  Workspace >=
 * gcn_gemm_atg256_workspace_bytes(n_list).
 */
size_t gcn_gemm_atg256_workspace_bytes(int64_t n_list);
int gcn_gemm_atg256_f32(const float *A, int64_t lda, const int32_t *rows_a, const float *G, int64_t ldg,
                        const int32_t *rows_g, int64_t n_list, const float *a_absmax_bound,
                        const float *g_absmax_bound, float *out, int64_t ldo, void *workspace,
                        size_t workspace_bytes, void *stream);
/* The fp32-equivalent form (ABI 24): three bf16 parts per operand, six MFMAs per product, no
 * bounds (see gcn_gemm_xw256_f32_b3); same lists, workspace and determinism. */
int gcn_gemm_atg256_f32_b3(const float *A, int64_t lda, const int32_t *rows_a, const float *G, int64_t ldg,
                           const int32_t *rows_g, int64_t n_list, float *out, int64_t ldo, void *workspace,
                           size_t workspace_bytes, void *stream);
/* ... and (ABI 25) with the BIAS gradient of the same layer as a side result: colsum_g[256] (DEVICE
 * floats) = Σ_{r < n_list} G[rows_g[r], :] — `grad_output.sum(0)` of `output + self.bias`
 * (pygcn/layers.py:36) over the rows on which the gradient can be non-zero — summed from the rows the
 * kernel loads anyway (8 rows at a time, compensated running sums, workgroup partials added in
 * order in double precision: deterministic), instead of in a pass of its own over G. */
int gcn_gemm_atg256_f32_b3_colsum(const float *A, int64_t lda, const int32_t *rows_a, const float *G, int64_t ldg,
                                  const int32_t *rows_g, int64_t n_list, float *out, int64_t ldo, float *colsum_g,
                                  void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same weight gradient for bf16 STORAGE (config C5: 128 -> 128 layers): A [*, K] and G [*, N]
 * are bf16 (row-major, even leading dimensions, 4-byte aligned), accumulation and the [K, N] result
 * are fp32; no scaling and no bounds (bf16 has fp32's range).  (K, N) = (128, 128); the workspace
 * size is 0 for shapes the kernel does not carry (the caller then uses a library GEMM).  Row lists
 * as for gcn_gemm_atg256_f32 (padded to a multiple of 16 entries with valid indices).
 * Deterministic.  (ABI 22.)
 */
size_t gcn_gemm_atg_bf16_workspace_bytes(int64_t n_list, int64_t K, int64_t N);
int gcn_gemm_atg_bf16(const void *A, int64_t lda, const int32_t *rows_a, const void *G, int64_t ldg,
                      const int32_t *rows_g, int64_t n_list, int64_t K, int64_t N, float *out,
                      int64_t ldo, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Rows of a mostly-zero activation as BITMASK + NON-ZERO VALUES — the wire format of the multi-GPU
 * path's compressed hidden-layer halo exchange (pygcn_amd/sharded.py; the input of the second
 * GraphConvolution layer is relu + dropout output, pygcn/models.py:48,50: >= 75 % zeros in
 * training).  The reference has no multi-device code; SURVEY §8(e).  (ABI 22.)
 *   bit j of bits[r*(F/32) + w] is set iff element 32*w + j of row r is non-zero; vals holds the
 *   non-zero elements in row-major order, row r starting at offsets[r].
 * gcn_rows_pack_count : reads rows `rows[r]` (NULL = rows 0..m-1) of src [*, ld] (fp32 or bf16, F
 *   a multiple of 32, ld a multiple of 4, 16-byte aligned base); writes bits [m, F/32] and
 *   counts [m] (non-zeros per row).  The caller scans counts into offsets (int64, exclusive).
 * gcn_rows_pack_values: writes vals [offsets[m-1] + counts[m-1]] (element type of src).
 * gcn_rows_unpack     : dst [m, ldd] (every element written) from bits, offsets, vals.
 * gcn_bits_row_counts : counts [m] = set bits per row of bits [m, words] (the receiver's side:
 *   the wire carries bits and values only; it scans these counts into its own offsets).
 * -0.0 counts as zero (it compares equal to 0) and arrives as +0.0; NaN is non-zero.
 * Deterministic; DEVICE pointers; asynchronous on `stream`.
 */
int gcn_rows_pack_count(int dtype, const void *src, int64_t ld, const int64_t *rows, int64_t m, int64_t F,
                        uint32_t *bits, int32_t *counts, void *stream);
int gcn_rows_pack_values(int dtype, const void *src, int64_t ld, const int64_t *rows, int64_t m, int64_t F,
                         const int64_t *offsets, void *vals, void *stream);
int gcn_rows_unpack(int dtype, const uint32_t *bits, const int64_t *offsets, const void *vals, int64_t m,
                    int64_t F, void *dst, int64_t ldd, void *stream);
int gcn_bits_row_counts(const uint32_t *bits, int64_t m, int64_t words, int32_t *counts, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GCN_SPMM_H */
