/*
 * gcl.h - C ABI of libgcl_hip.so: graphcast-lite's encode-process-decode GNN hot path as
 * hand-written gfx950 (MI355X / CDNA4) HIP kernels.
 *
 * The reference (ArturKKK/graphcast-lite) has no native boundary: its hot path is Python calling
 * torch / torch_geometric ops from src/models.py.  Each entry point below names the reference
 * call site whose stock-op expansion it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  Every function returns 0 on success
 *     or a negative GCL_E* code; gcl_last_error() returns a thread-local message.
 *   - Compute entry points never allocate, never synchronise and never create streams: the caller
 *     passes device pointers it owns (contiguous fp32 / int32), explicit leading dimensions, a
 *     workspace where one is needed (size from the matching *_ws_bytes query) and the hipStream_t
 *     to enqueue on (as void*).  They are re-entrant; the only state is the immutable graph handle.
 *   - Node features are row-major `[B, n, F]`: `ld*` = floats between consecutive rows,
 *     `bs*` = floats between consecutive samples.  B independent samples share one graph.
 *   - Activation chaining: buffers passed between layers hold PRE-activation values.  An entry
 *     point that takes `in_slope` (device pointer to one float, may be NULL) applies
 *     PReLU(x) = max(0,x) + slope*min(0,x) to its input while loading it, and its backward returns
 *     the gradient with respect to the PRE-activation input (and adds the slope gradient into
 *     `d_in_slope`).  This is how the reference's `conv -> shared PReLU -> conv` stacks
 *     (src/models.py:321-330,416-421) and `Linear -> PReLU -> Linear` chains (src/models.py:61-109)
 *     run without a separate activation pass.
 */
#ifndef GCL_H
#define GCL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCL_VERSION 100 /* 0.1.0 */

#define GCL_OK 0
#define GCL_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define GCL_EHIP (-2)     /* HIP runtime error (message has the hipError string) */
#define GCL_ENOMEM (-3)   /* workspace too small / allocation failed in graph_create */
#define GCL_EUNSUPPORTED (-4)

typedef struct gcl_graph gcl_graph_t; /* opaque; owns device CSR arrays */
typedef void* gcl_stream_t;           /* hipStream_t */

int gcl_version(void);
const char* gcl_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Graph handle: reference-layout edge list -> receiver-sorted CSR (+ sender-sorted transpose).
 * Replaces, once per graph instead of once per forward: add_remaining_self_loops + degree
 * scatter_add + pow(-1/2) + two gathers of PyG gcn_norm (GCNConv, src/models.py:419), the
 * remove_self_loops + add_self_loops of GATConv (src/models.py:425,135) and the in-degree count of
 * SimpleConv(aggr="mean") (src/models.py:414).
 * ------------------------------------------------------------------------------------------- */
#define GCL_GRAPH_GCN 0  /* drop self-loops, append one per node, w_e = deg_in[s]^-1/2 * deg_in[r]^-1/2 */
#define GCL_GRAPH_GAT 1  /* drop self-loops, append one per node, no weights */
#define GCL_GRAPH_MEAN 2 /* edges as given, w_e = 1 / max(1, deg_in[r]) */

/* Sizes of the processed edge set for (edge_index, E, n, kind): E' written to *e_out. */
int gcl_graph_count_edges(const int64_t* edge_index, int64_t E, int32_t n, int32_t kind, int64_t* e_out);

/* Host-only CSR construction into caller arrays (no GPU needed; used by gcl_graph_create and by
 * the CPU tests).  edge_index is int64 [2,E] row 0 = sender, row 1 = receiver (src/models.py:12).
 * Outputs (E' = gcl_graph_count_edges):
 *   rowptr[n+1], col[E'] (sender of each slot), w[E'], eperm[E'] (slot -> position in the PyG edge
 *   order "kept edges in input order, then loops 0..n-1"), and the transpose trowptr[n+1],
 *   tcol[E'] (receiver), tw[E'], tslot[E'] (transpose slot -> forward slot).
 * Within a row, slots keep PyG edge order (stable sort), so sums run in the reference's order. */
int gcl_graph_build_host(const int64_t* edge_index, int64_t E, int32_t n, int32_t kind,
                         int32_t* rowptr, int32_t* col, float* w, int32_t* eperm,
                         int32_t* trowptr, int32_t* tcol, float* tw, int32_t* tslot);

int gcl_graph_create(const int64_t* edge_index, int64_t E, int32_t n, int32_t kind, gcl_graph_t** out);
void gcl_graph_destroy(gcl_graph_t* g);
int32_t gcl_graph_num_nodes(const gcl_graph_t* g);
int64_t gcl_graph_num_edges(const gcl_graph_t* g); /* E' */
int32_t gcl_graph_max_in_degree(const gcl_graph_t* g);
/* PyG-order edge list with loops, host int64 [2,E'] (what GATConv returns: src/models.py:135). */
int gcl_graph_export_edges(const gcl_graph_t* g, int64_t* edge_index_out);
/* Device pointer to eperm (int32 [E']), for callers that re-order per-slot data themselves. */
const int32_t* gcl_graph_eperm_device(const gcl_graph_t* g);
/* Source-tile ("halo") layout of one direction (transpose != 0: sender-sorted) for tile height T (64 or 32):
 * out4 = {T (0 when that layout was not built), tiles, largest staged source count per tile (rounded up to 8),
 * 0}.  gcl_graph_create builds it when a tile's edges share their sources well enough (>= 1.6 reads per staged
 * row) - e.g. mesh nodes numbered tile by tile; gcl_aggregate then stages every source row of a tile once in
 * LDS instead of gathering it once per edge (the propagate of src/models.py:419). */
int gcl_graph_halo_info(const gcl_graph_t* g, int32_t transpose, int32_t T, int32_t* out4);

/* ---------------------------------------------------------------------------------------------
 * Dense per-node transform  y = act(x) W^T (+ bias)      [rows, Fin] x [Fout, Fin]^T
 * Replaces nn.Linear (+ the preceding nn.PReLU) inside MLP.forward (src/models.py:106-109) and the
 * `lin` GEMM inside every GCNConv/GATConv (src/models.py:419,425).  fp32 in, fp32 accumulate
 * (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain in k order).
 * ------------------------------------------------------------------------------------------- */
int gcl_linear_fwd(const float* x, int64_t ldx, const float* in_slope, const float* W /*[Fout,Fin]*/,
                   const float* bias /*[Fout] or NULL*/, float* y, int64_t ldy, int64_t rows,
                   int32_t Fin, int32_t Fout, gcl_stream_t stream);

/* dx = (dy W) * PReLU'(x)   [rows, Fin]; adds sum(dy W * min(0,x)) into *d_in_slope when in_slope
 * is given.  ws: gcl_linear_bwd_ws_bytes(rows, Fin, Fout). */
int gcl_linear_bwd_dx(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx,
                      const float* in_slope, float* d_in_slope, float* dx, int64_t lddx,
                      int64_t rows, int32_t Fin, int32_t Fout, void* ws, size_t ws_bytes,
                      gcl_stream_t stream);
/* dW (+)= dy^T act(x), db (+)= colsum(dy) (db may be NULL).  accumulate != 0 adds into dW/db. */
int gcl_linear_bwd_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx,
                      const float* in_slope, float* dW, float* db, int64_t rows, int32_t Fin,
                      int32_t Fout, int32_t accumulate, void* ws, size_t ws_bytes,
                      gcl_stream_t stream);
size_t gcl_linear_bwd_ws_bytes(int64_t rows, int32_t Fin, int32_t Fout);
/* Whole backward of one dense transform in one call: dx (pre-activation gradient), dW, db (NULL: no
 * bias), the slope gradient, and optionally colsum_dx[c] (+)= sum_r dx[r,c] (the bias gradient of
 * the conv layer that produced x).  Uses ONE fused kernel (dY and x read once) when
 * Fout <= 64, Fin <= 96 and rows are 16-B aligned, else the three separate kernels.  Fout itself need not be a
 * multiple of 4 when lddy >= roundup(Fout, 4) and the padding columns of dy hold finite values (zeros): this is how
 * the 33- / 19-wide last decoder conv runs on 16-byte rows without padded copies of its weight.
 * `accumulate` is a bit mask with ONE BIT PER DESTINATION (they belong to different parameters,
 * whose gradients may be in different states): a set bit adds into that destination, a clear bit
 * overwrites it.  *d_in_slope is always added to. */
#define GCL_ACC_DW 1     /* dW */
#define GCL_ACC_DB 2     /* db */
#define GCL_ACC_COLSUM 4 /* colsum_dx */
int gcl_linear_bwd_all(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx,
                       const float* in_slope, float* d_in_slope, float* dx, int64_t lddx, float* dW,
                       float* db, float* colsum_dx, int64_t rows, int32_t Fin, int32_t Fout,
                       int32_t accumulate, void* ws, size_t ws_bytes, gcl_stream_t stream);
size_t gcl_linear_bwd_all_ws_bytes(int64_t rows, int32_t Fin, int32_t Fout);

/* Deferred final pass.  A training step calls gcl_linear_bwd_all once per layer, and each call ends in a small launch
 * that sums its per-block partial records into dW / db / colsum_dx / the slope gradient.  The _deferred form runs the
 * main kernel only and DESCRIBES that pass in *job; gcl_reduce_jobs later runs the passes of many calls in one launch
 * per 16 jobs (the gradients of one optimiser step, src/train.py:232-233, are only read after the whole backward).
 * Rules: every deferred call needs its OWN workspace, alive until gcl_reduce_jobs has run; two pending jobs must not
 * write the same dW / db / colsum_dx (flush first; a shared slope gradient is fine: slopes are summed job after job);
 * job->nparts == 0 means the call took the non-fused path and has already reduced on the spot. */
typedef struct gcl_reduce_job {
  const float* part; /* per-block partial records */
  int64_t pstride;   /* floats between records */
  struct {
    float* out; /* NULL: unused segment */
    int32_t poff, count, pld, cols, ldo, acc;
  } seg[3];           /* dW | db | colsum_dx */
  const double* spart; /* slope partials (one per block) or NULL */
  float* sout;
  int32_t nparts, ns;
} gcl_reduce_job;
int gcl_linear_bwd_all_deferred(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx,
                                const float* in_slope, float* d_in_slope, float* dx, int64_t lddx, float* dW,
                                float* db, float* colsum_dx, int64_t rows, int32_t Fin, int32_t Fout,
                                int32_t accumulate, void* ws, size_t ws_bytes, gcl_stream_t stream,
                                gcl_reduce_job* job);
int gcl_reduce_jobs(const gcl_reduce_job* jobs, int32_t n, gcl_stream_t stream);

/* General form of the three calls above, for wide layers and fused operands (the InteractionNet
 * processor, src/models.py:185-233, and the 256-wide MLPs): any Fin / Fout, the weight may be a
 * column block of a wider matrix (row stride ldw >= Fin, e.g. one third of edge_mlp[0].weight
 * [256, 768]), the activation on the input is selectable, and an addend [rows, Fout] can be summed
 * in the epilogue (residuals; the second half of a split contraction).
 * Shapes that fit the resident-weight-panel kernel run there; wider ones (K or N > 256, or a panel
 * beyond 160 KiB of LDS) run a 128x128-tile contraction that needs Fin % 4 == 0 and 16-B aligned
 * rows.  All use the exact-fp32 matrix instruction. */
#define GCL_ACT_NONE 0
#define GCL_ACT_PRELU 1 /* one learnable slope (nn.PReLU()); needs the slope pointer */
#define GCL_ACT_SILU 2  /* x * sigmoid(x)  (nn.SiLU, "swish": src/models.py:154-163) */
/* y = act(x) W^T + bias + addend */
int gcl_dense_fwd(const float* x, int64_t ldx, int32_t act, const float* slope, const float* W,
                  int64_t ldw, const float* bias, const float* addend, int64_t ldadd, float* y,
                  int64_t ldy, int64_t rows, int32_t Fin, int32_t Fout, gcl_stream_t stream);
/* dx = (dy W) * act'(z) + addend, z = the pre-activation the forward call read as x; PReLU adds
 * its slope gradient into *d_slope (may be NULL).  ws: gcl_linear_bwd_ws_bytes(rows, Fin, Fout). */
int gcl_dense_bwd_dx(const float* dy, int64_t lddy, const float* W, int64_t ldw, const float* z,
                     int64_t ldz, int32_t act, const float* slope, float* d_slope,
                     const float* addend, int64_t ldadd, float* dx, int64_t lddx, int64_t rows,
                     int32_t Fin, int32_t Fout, void* ws, size_t ws_bytes, gcl_stream_t stream);
/* dW[o, c] (+)= sum_r dy[r, o] act(x[r, c]) (row stride lddw), db (+)= colsum(dy) (may be NULL) */
int gcl_dense_bwd_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, int32_t act,
                     const float* slope, float* dW, int64_t lddw, float* db, int64_t rows,
                     int32_t Fin, int32_t Fout, int32_t accumulate, void* ws, size_t ws_bytes,
                     gcl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse aggregation over the CSR  y[b,i,:] = sum_{e in row i} w_e * h[b, col_e, :] (+ bias)
 * Replaces the index_select -> multiply -> scatter_add_ of PyG propagate for GCNConv
 * (src/models.py:419) and SimpleConv mean (src/models.py:414).  transpose != 0 walks the
 * sender-sorted CSR instead (the backward: dh = A_hat^T dy).
 * ------------------------------------------------------------------------------------------- */
int gcl_aggregate(const gcl_graph_t* g, int32_t transpose, const float* h, int64_t ldh, int64_t bsh,
                  const float* bias, float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t F,
                  gcl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GATConv(heads=H, concat=False) attention + aggregation  (src/models.py:425; SparseGATConv :135)
 *   a_s = <h, att_src>, a_d = <h, att_dst> per head; e = LeakyReLU_0.2(a_s[j] + a_d[i]);
 *   softmax over the in-edges of i (max subtracted, +1e-16 in the denominator);
 *   y[i] = mean_heads sum_e alpha_e h[j_e] + bias.
 * h is the output of gcl_linear_fwd with Fout = H*C, laid out [B, n, H*C].
 * alpha (may be NULL in inference) is written per CSR slot: [B, E', H].
 * ------------------------------------------------------------------------------------------- */
int gcl_gat_fwd(const gcl_graph_t* g, const float* h, int64_t ldh, int64_t bsh,
                const float* att_src /*[H*C]*/, const float* att_dst /*[H*C]*/, const float* bias /*[C]*/,
                float* a_src /*[B,n,H]*/, float* a_dst /*[B,n,H]*/, float* alpha /*[B,E',H]*/,
                float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t H, int32_t C,
                gcl_stream_t stream);
/* Gradients of gcl_gat_fwd: dh [B,n,H*C] (grad of the linear output, attention paths included),
 * d_att_src/d_att_dst [H*C] and d_bias [C] (added when accumulate != 0, else overwritten).
 * ws: gcl_gat_bwd_ws_bytes(E', n, B, H, C). */
int gcl_gat_bwd(const gcl_graph_t* g, const float* dy, int64_t lddy, int64_t bsdy,
                const float* h, int64_t ldh, int64_t bsh, const float* att_src, const float* att_dst,
                const float* a_src, const float* a_dst, const float* alpha,
                float* dh, int64_t lddh, int64_t bsdh, float* d_att_src, float* d_att_dst,
                float* d_bias, int32_t accumulate, int32_t B, int32_t H, int32_t C,
                void* ws, size_t ws_bytes, gcl_stream_t stream);
size_t gcl_gat_bwd_ws_bytes(int64_t e_prime, int32_t n, int32_t B, int32_t H, int32_t C);
/* The same two calls with the rows of h read THROUGH a row table (the stage split of src/models.py:837-838 folded
 * into the first processor layer: only the compact rows are transformed, functional.LatSource): row i of sample b is
 * row tab[i] of sample b of h when tab[i] >= 0, the batch-invariant flat row ~tab[i] of h otherwise; a_src / a_dst /
 * alpha / y / dh stay dense [B, n, ..].  One head on a source-tile graph (gcl_gat_tab_ok returns 1), else
 * GCL_EUNSUPPORTED.  gcl_gat_bwd_tab's dh is the gradient of the table-read rows; the caller folds it back. */
int gcl_gat_fwd_tab(const gcl_graph_t* g, const float* h, int64_t ldh, int64_t bsh, const int32_t* tab /*[n]*/,
                    const float* att_src, const float* att_dst, const float* bias, float* a_src, float* a_dst,
                    float* alpha, float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t H, int32_t C,
                    gcl_stream_t stream);
int gcl_gat_bwd_tab(const gcl_graph_t* g, const float* dy, int64_t lddy, int64_t bsdy, const float* h, int64_t ldh,
                    int64_t bsh, const int32_t* tab /*[n]*/, const float* att_src, const float* att_dst,
                    const float* a_src, const float* a_dst, const float* alpha, float* dh, int64_t lddh, int64_t bsdh,
                    float* d_att_src, float* d_att_dst, float* d_bias, int32_t accumulate, int32_t B, int32_t H,
                    int32_t C, void* ws, size_t ws_bytes, gcl_stream_t stream);
int gcl_gat_tab_ok(const gcl_graph_t* g, int64_t ldh, int32_t H, int32_t C);
/* Re-order per-slot attention of ONE sample into PyG edge order: out[eperm[s]*H + k] = alpha[s*H + k]
 * (what SparseGATConv thresholds: src/models.py:136-149). */
int gcl_gat_alpha_to_edge_order(const gcl_graph_t* g, const float* alpha_slots, float* alpha_edges,
                                int32_t H, gcl_stream_t stream);
/* SparseGATConv prune (src/models.py:138-149): keep PyG-order edges with alpha >= threshold.
 * alpha_edges is a DEVICE array [E'] (H = 1).  Writes the surviving edge list, host int64
 * [2, *kept] (capacity E'), synchronising `stream` once.  Wave-ballot + prefix compaction. */
int gcl_gat_prune(const gcl_graph_t* g, const float* alpha_edges, float threshold,
                  int64_t* edge_index_out, int64_t* kept, void* ws, size_t ws_bytes,
                  gcl_stream_t stream);
size_t gcl_gat_prune_ws_bytes(int64_t e_prime);

/* ---------------------------------------------------------------------------------------------
 * PyG LayerNorm(mode="node")  (src/models.py:102-104,368-374): per row, eps inside the sqrt.
 * stats [rows,2] receives (mean, rstd) for the backward.  x may carry an activation (in_slope).
 * ------------------------------------------------------------------------------------------- */
int gcl_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                      float* y, int64_t ldy, float* stats, int64_t rows, int32_t F,
                      gcl_stream_t stream);
/* The same with the output written THROUGH a row map: row (b, i) of the [B][n_per] row space goes to
 * y[b * bsy + pos[i] * ldy] when pos[i] >= 0 and nowhere otherwise (stats are written for every row).  The processor's
 * final LayerNorm writes the rows the decoder reads straight into the decoder's input (src/models.py:860-862). */
int gcl_layernorm_fwd_map(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float* y,
                          int64_t ldy, int64_t bsy, const int32_t* pos /*[n_per]*/, int32_t n_per, float* stats,
                          int64_t rows, int32_t F, gcl_stream_t stream);
int gcl_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                      const float* stats, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                      int32_t accumulate, int64_t rows, int32_t F, void* ws, size_t ws_bytes,
                      gcl_stream_t stream);
/* Same, and colsum_dx[j] (+)= sum over rows of dx[:, j] when colsum_dx != NULL: dx is the dY of the layer below,
 * so this is that layer's bias gradient (GCNConv.bias of the last conv under the stack's LayerNorm,
 * src/models.py:368-374,419) without a second pass over dx.  accumulate: GCL_ACC_DW covers dgamma and dbeta,
 * GCL_ACC_COLSUM the column sums. */
int gcl_layernorm_bwd_cs(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                         const float* stats, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                         float* colsum_dx, int32_t accumulate, int64_t rows, int32_t F, void* ws,
                         size_t ws_bytes, gcl_stream_t stream);
/* Same with dy given through a row map: the rows are B samples of n_per rows, row (b, i) reads
 * dy[b * bsdy + pos[i] * lddy + :] when pos[i] >= 0 and a zero gradient otherwise - the output of the layer was only
 * consumed through a row gather (the decoder reads a subset of the processor's mesh rows, src/models.py:860-862), so
 * the zero-filled dense gradient never has to be formed.  pos == NULL: dy is dense as above. */
int gcl_layernorm_bwd_map(const float* dy, int64_t lddy, int64_t bsdy, const int32_t* pos, int32_t n_per,
                          const float* x, int64_t ldx, const float* gamma, const float* stats, float* dx,
                          int64_t lddx, float* dgamma, float* dbeta, float* colsum_dx, int32_t accumulate,
                          int64_t rows, int32_t F, void* ws, size_t ws_bytes, gcl_stream_t stream);
size_t gcl_layernorm_bwd_ws_bytes(int64_t rows, int32_t F);
/* PyG LayerNorm(mode="graph"): statistics over all n*F elements of each sample, eps added to the
 * std.  stats [B,2] = (mean, 1/(std+eps)). */
int gcl_graphnorm_fwd(const float* x, int64_t ldx, int64_t bsx, const float* gamma, const float* beta,
                      float eps, float* y, int64_t ldy, int64_t bsy, float* stats, int32_t B,
                      int32_t n, int32_t F, void* ws, size_t ws_bytes, gcl_stream_t stream);
int gcl_graphnorm_bwd(const float* dy, int64_t lddy, int64_t bsdy, const float* x, int64_t ldx,
                      int64_t bsx, const float* gamma, const float* stats, float eps, float* dx,
                      int64_t lddx, int64_t bsdx, float* dgamma, float* dbeta, int32_t accumulate,
                      int32_t B, int32_t n, int32_t F, void* ws, size_t ws_bytes, gcl_stream_t stream);
size_t gcl_graphnorm_ws_bytes(int32_t B, int32_t n, int32_t F);

/* Column sums  out[c] (+)= sum_r x[r,c]  (bias gradients of the conv layers). */
int gcl_colsum(const float* x, int64_t ldx, int64_t rows, int32_t F, float* out, int32_t accumulate,
               void* ws, size_t ws_bytes, gcl_stream_t stream);
size_t gcl_colsum_ws_bytes(int64_t rows, int32_t F);

/* ---------------------------------------------------------------------------------------------
 * Input assembly (src/models.py:776-806): out[b] = [ x[b] | grid_static ; 0 | mesh_static ],
 * out is [B, G+M, Cdyn+Cs].  The reference re-allocates the zero block and re-concatenates on
 * every forward.
 * ------------------------------------------------------------------------------------------- */
int gcl_assemble_input(const float* x /*[B,G,Cdyn]*/, const float* grid_static /*[G,Cs]*/,
                       const float* mesh_static /*[M,Cs]*/, float* out, int64_t ldo, int32_t B,
                       int32_t G, int32_t M, int32_t Cdyn, int32_t Cs, gcl_stream_t stream);
/* The same with the last r of the M mesh rows of every sample given whole, tail [B, r, Cdyn + Cs] contiguous (the
 * per-sample folded rows of the compact pipeline: each sample carries different ones).  mesh_static may be NULL when
 * r == M. */
int gcl_assemble_input_tail(const float* x, const float* grid_static, const float* mesh_static, const float* tail,
                            int32_t r, float* out, int64_t ldo, int32_t B, int32_t G, int32_t M, int32_t Cdyn,
                            int32_t Cs, gcl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Residual add + weighted MSE and its gradient (src/train.py:203-213, 85-102).
 *   out = x_last + delta (use_residual) or delta; loss = sum(w (out-y)^2) * inv_wsum;
 *   w[g,c] = node_w[g] * chan_w[c] (either may be NULL = 1);
 *   d_delta = 2 w (out-y) * inv_wsum * grad_scale.
 * x_last / y are addressed as base + b*bs + g*ld + c.  loss_out: one device float (overwritten).
 * out_state (may be NULL) receives `out` for autoregressive roll-forward.
 * ------------------------------------------------------------------------------------------- */
/* loss_prev (device float, may be NULL): *loss_out = *loss_prev + loss of this call - the running sum over the
 * autoregressive steps of one batch (src/train.py:213,231) without a separate add. */
int gcl_wmse_fwd_bwd(const float* delta, int64_t ldd, int64_t bsd, const float* x_last, int64_t ldx,
                     int64_t bsx, const float* y, int64_t ldy_, int64_t bsy, const float* node_w,
                     const float* chan_w, float inv_wsum, float grad_scale, float* d_delta,
                     float* out_state, const float* loss_prev, float* loss_out, int32_t B, int32_t G,
                     int32_t C, void* ws, size_t ws_bytes, gcl_stream_t stream);
size_t gcl_wmse_ws_bytes(int32_t B, int32_t G, int32_t C);

/* torch.optim.Adam step (src/main.py:212, src/train.py:233) over one flat parameter buffer:
 * grad_scale multiplies the gradient first (1/world after the gradient all-reduce). */
int gcl_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                  gcl_stream_t stream);

/* Same update with the step counter kept ON THE DEVICE (step_dev: one int32, incremented by the
 * call; bc_dev: two floats of scratch), so a captured hipGraph of the training step can be replayed
 * without any step-dependent host value baked into it. */
int gcl_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int32_t* step_dev, float* bc_dev,
                      float grad_scale, gcl_stream_t stream);

/* Strided row copy  dst[b, i, 0:F] = src[b, i, 0:F]  (stage glue: src/models.py:837-838,860-862). */
int gcl_copy_rows(const float* src, int64_t lds, int64_t bss, float* dst, int64_t ldd, int64_t bsd,
                  int32_t B, int32_t rows, int32_t F, gcl_stream_t stream);

/* One whole GCNConv layer forward (src/models.py:419) in one kernel, aggregate-first:
 *   y[b,i,:] = (sum_{e in row i} w_e act(x[b, col_e, :])) W^T + bias
 * - the same value as gcl_linear_fwd followed by gcl_aggregate up to fp32 rounding (the aggregation is
 * linear), without the intermediate h in memory (csrc/gcn_layer.hip: wave-independent gather -> LDS tile ->
 * exact-fp32 MFMA -> 16-byte row stores).  Columns [Fout, Fout_store) of y are written as zeros
 * (Fout_store = Fout rounded up to a multiple of 4 keeps 33- / 19-wide outputs on 16-byte rows).
 * Needs Fin % 4 == 0, Fin, Fout_store <= 64, 16-B aligned rows of x and y, and a graph without heavy rows
 * (in-degree <= 64); other shapes use gcl_linear_fwd + gcl_aggregate. */
int gcl_gcn_layer_fwd(const gcl_graph_t* g, const float* x, int64_t ldx, int64_t bsx, int32_t act,
                      const float* slope, const float* W /*[Fout,Fin]*/, const float* bias, float* y,
                      int64_t ldy, int64_t bsy, int32_t B, int32_t Fin, int32_t Fout, int32_t Fout_store,
                      gcl_stream_t stream);
/* Same, computing only the first rows_out rows of every sample (rows beyond are not written): the decoder keeps
 * only the grid rows of its last conv (src/models.py:870-872), whose mesh rows are dead. */
int gcl_gcn_layer_fwd_rows(const gcl_graph_t* g, const float* x, int64_t ldx, int64_t bsx, int32_t act,
                           const float* slope, const float* W /*[Fout,Fin]*/, const float* bias, float* y,
                           int64_t ldy, int64_t bsy, int32_t B, int32_t Fin, int32_t Fout, int32_t Fout_store,
                           int32_t rows_out, gcl_stream_t stream);
/* Same layer with its input rows read THROUGH a row table instead of from a materialised [B, n, Fin] tensor (the
 * stage split of src/models.py:837-838 - `mesh_node_features = encoded[..., G:, :]` - folded into the first processor
 * layer's loads): input row i of sample b is row tab[i] of sample b of x when tab[i] >= 0, and row ~tab[i] of x viewed
 * as ONE flat list of x_rows rows (a batch-invariant row, shared by all samples) when tab[i] < 0.  Source-tile graphs
 * only (gcn_halo_fwd_kernel): GCL_EUNSUPPORTED otherwise; gcl_gcn_layer_fwd_tab_ok returns 1 when the call would run. */
int gcl_gcn_layer_fwd_tab(const gcl_graph_t* g, const float* x, int64_t ldx, int64_t bsx, int64_t x_rows,
                          const int32_t* tab /*[n]*/, int32_t act, const float* slope, const float* W /*[Fout,Fin]*/,
                          const float* bias, float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t Fin, int32_t Fout,
                          int32_t Fout_store, gcl_stream_t stream);
int gcl_gcn_layer_fwd_tab_ok(const gcl_graph_t* g, int64_t ldx, int64_t bsx, int64_t x_rows, int32_t B, int32_t Fin,
                             int32_t Fout);

/* ---------------------------------------------------------------------------------------------
 * Edge-wise glue of the InteractionNet processor (src/models.py:206-236); csrc/interaction.hip.
 * Index arrays are int32 device arrays; rows are [.., D] with D % 4 == 0 and 16-B alignment.
 * ------------------------------------------------------------------------------------------- */
/* out[b,i,:] = sum (mean != 0: mean, empty segment -> 0) over k in [rowptr[i], rowptr[i+1]) of
 * src[b, perm ? perm[k] : k, :].  scatter(edge_update, receivers, reduce="mean")
 * (src/models.py:221) on receiver-sorted edges, and the backward of the x[senders] /
 * x[receivers] gathers (src/models.py:216). */
int gcl_segment_reduce(const float* src, int64_t ld_src, int64_t bs_src, const int32_t* perm,
                       const int32_t* rowptr /*[n+1]*/, int32_t mean, float* out, int64_t ld_out,
                       int64_t bs_out, int32_t B, int32_t n, int32_t D, gcl_stream_t stream);
/* out[b,e,:] = base[b,e,:] + extra[b,e,:] + A[b, ia[e], :] * sa[ia[e]] + C[b, ic[e], :]
 * (every operand optional; base / extra / out contiguous [B,E,D]).  Forward: the sender /
 * receiver terms of the first edge-MLP layer; backward: d(aggregated)[receivers] / deg. */
int gcl_edge_combine(const float* base, const float* extra, const float* A, int64_t lda, int64_t bsa,
                     const int32_t* ia, const float* sa, const float* C, int64_t ldc, int64_t bsc,
                     const int32_t* ic, float* out, int32_t B, int64_t E, int32_t D,
                     gcl_stream_t stream);
/* Elementwise activation where its output must exist in memory (edge_encoder, src/models.py:251):
 * y = act(x);  dx = dy * act'(x), PReLU adds its slope gradient into *d_slope. */
int gcl_act_fwd(const float* x, float* y, int64_t count, int32_t act, const float* slope,
                gcl_stream_t stream);
int gcl_act_bwd(const float* x, const float* dy, float* dx, int64_t count, int32_t act,
                const float* slope, float* d_slope, void* ws, size_t ws_bytes, gcl_stream_t stream);
size_t gcl_act_bwd_ws_bytes(void);

/* Input windows from a device-resident fp16 time series - the reference's on-the-fly loader
 * (src/data/dataloader_chunked.py:179-223: fp16 memmap (T, lon, lat, Ct) -> first C channels ->
 * fp32 -> (x - mean) / std -> (lat, lon)-major transpose -> [G, obs*C] / [G, pred*C]), batched over
 * B window starts t0[b] (device int64).  n_lat = 1, n_lon = N reads the flat (T, N, Ct) layout
 * (:184-197).  Bit-identical to the numpy loader; frames outside [0, T) come back as NaN. */
int gcl_window_pack(const uint16_t* series /* IEEE binary16 */, int64_t T, int32_t n_lon, int32_t n_lat,
                    int32_t Ct, const int64_t* t0, const float* mean, const float* stdv, int32_t C,
                    int32_t obs, int32_t pred, float* X /*[B,G,obs*C]*/, float* Y /*[B,G,pred*C] or NULL*/,
                    int32_t B, gcl_stream_t stream);

/* One autoregressive advance of the observation window, fused (scripts/predict.py:512-535 and the
 * same steps in src/train.py:203-228): step_out = residual ? x_last + delta : delta; static channels
 * (chan_kind 1) carry x_last forward, forcing channels (chan_kind 2) take y_step when it is given;
 * step_out is appended to `out` at column out_off (out may be NULL) and the window
 * state [B,G,obs,C] is shifted by one step into new_state (must not alias state).
 * state / delta / new_state are contiguous; y_step and out are addressed with (ld, bs). */
int gcl_ar_advance(const float* state, const float* delta, int64_t ldd, int64_t bsd, const float* y_step,
                   int64_t ldy, int64_t bsy, const int32_t* chan_kind, float* new_state, float* out,
                   int64_t ldo, int64_t bso, int32_t out_off, int32_t B, int32_t G, int32_t obs, int32_t C,
                   int32_t residual, gcl_stream_t stream);
/* Backward of one autoregressive TRAINING step = gcl_wmse_fwd_bwd (loss of the step, dd = d loss / d pred for a unit
 * upstream gradient) + gcl_ar_advance (src/train.py:203-228), in one pass:
 *   d_delta = g_loss dd + [predicted channel] g_new[last slot]
 *   d_state = shift(g_new) + last slot: [residual] g_loss dd + [static, or predicted with residual] g_new[last slot]
 * g_loss: device float (upstream gradient of the loss sum), g_new: gradient of the advanced window [B,G,obs,C] or NULL
 * (last step), d_state may be NULL (the first window is data).  All tensors contiguous. */
int gcl_ar_step_bwd(const float* dd, const float* g_loss, const float* g_new, const int32_t* chan_kind,
                    int32_t has_y, int32_t residual, float* d_delta, float* d_state, int32_t B, int32_t G,
                    int32_t obs, int32_t C, gcl_stream_t stream);

/* dst[b, i, c] = (i < rows_src && c < F_src) ? src[b, i, c] : 0 for i < rows_dst, c < F_dst: the gradient of a row /
 * column slice (decoder output: grid rows, first 33 / 19 columns - src/models.py:870-872) widened to the sliced
 * tensor's zero-padded layout, in one pass. */
int gcl_pad_rows(const float* src, int64_t lds, int64_t bss, int32_t rows_src, int32_t F_src, float* dst,
                 int64_t ldd, int64_t bsd, int32_t rows_dst, int32_t F_dst, int32_t B, gcl_stream_t stream);
/* hipMemsetAsync(ptr, 0, nbytes) on the stream (optimizer.zero_grad of the flat gradient bucket, src/train.py:166). */
int gcl_zero(void* ptr, size_t nbytes, gcl_stream_t stream);

/* Row gather from up to two sources (stage glue of src/models.py:837-838,860-862 restricted to the
 * rows that matter):  dst[b,i,:] = a[b, map_a[i], :] if map_a[i] >= 0 (map_a NULL = identity), else
 * b[b, map_b[i], :] if map_b[i] >= 0, else 0.  A source with batch stride 0 is broadcast.
 * sum_batch == 1:  dst[0,i,:] = sum_b a[b, map_a[i], :]  (gradient of a broadcast source);
 * sum_batch = R > 1: the same sums, row i stored at dst[i / R, i % R, :] (R rows dealt to each destination sample).
 * Maps are int32 device arrays of length nd; F % 4 == 0 and 16-B aligned rows. */
int gcl_gather2_rows(const float* a, int64_t lda, int64_t bsa, const int32_t* map_a, const float* b,
                     int64_t ldb, int64_t bsb, const int32_t* map_b, float* dst, int64_t ldd, int64_t bsd,
                     int32_t B, int32_t nd, int32_t F, int32_t sum_batch, gcl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GCL_H */
