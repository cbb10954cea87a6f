/*
 * hopmi.h -- C ABI of libhopmi.so: hand-written gfx950 (MI355X) kernels for the HOP
 * generator hot path.
 *
 * The reference (Chenghyyy/HOP-...) is pure Python/PyTorch and has no FFI of its own;
 * each entry point below replaces a run of aten ops inside a reference Python function
 * and cites it.  The host side (Python, ctypes) lives in
 * hop-..._amd/{_lib,ops}.py; INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Conventions (every entry point):
 *   - plain device pointers borrowed for the call; the caller (PyTorch) owns all
 *     memory including workspaces; no allocation, no host sync, no exceptions;
 *   - asynchronous, ordered on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream); safe to capture into a hipGraph;
 *   - returns 0 on success, a negative HOPMI_E* code otherwise; hopmi_last_error()
 *     returns a thread-local message for the last failure;
 *   - all tensors fp32, dense, "channels-last" activations: x[n_slabs][V][64] where a
 *     slab is one (clip b, frame t) pair, i.e. row (b*T + t)*V + v, channel c innermost.
 *
 * State the library holds (all of it; none of it is data-path state -- results depend on the arguments only):
 *   - per host thread: the last error message (hopmi_last_error) and the one-shot measurement hook
 *     (hopmi_time_next_launch: consumed by that thread's next launch of a kernel that honours it);
 *   - per process, read-only after first use: a table of the DIAGNOSTIC environment knobs (HOPMI_* tile / form selectors,
 *     each read once under a mutex; hopmi_reload_env() forgets the table for probes that sweep a knob) and plan caches
 *     keyed on geometry (occupancy queries, tile plans: pure functions of their key);
 *   - nothing else: every entry point is re-entrant across streams and threads, and workspaces (status words, launch
 *     sequence numbers of the persistent kernels) are the caller's memory.
 * The diagnostic build (`make dbg`: libhopmi_dbg.so, -DHOPMI_CHECK_SPLIT) additionally exports
 * hopmi_debug_set_split_status_<file>(unsigned*): a device buffer into which every fp16 hi/lo split reports an overflow
 * (csrc/common.h); the production library carries neither the symbols nor the checks.
 */
#ifndef HOPMI_H
#define HOPMI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOPMI_OK 0
#define HOPMI_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define HOPMI_ELAUNCH (-2)  /* HIP launch or runtime error */

#define HOPMI_C 64          /* residual = dilation channels (HOP.py:143) */
#define HOPMI_MAX_NODES 48  /* 9 (TED) and 42 (TED-Expressive) are what the reference uses */

const char* hopmi_version(void);
const char* hopmi_last_error(void);
/* Tuning knobs (HOPMI_WN_GRID, HOPMI_WN_MAXMT, HOPMI_WN_BWD_GRID, HOPMI_GCN_*) are read from the environment once per
 * process; this forgets the cached values so that the next call reads them again (used by the sweep probes). */
void hopmi_reload_env(void);

/* Stream-capture hygiene for the recorded training step (hopmi/graph.py; the step being recorded is the reference's
 * train_eval/train_llm.py:9-98).  An exception raised while a step is being recorded leaves `stream` inside a capture; whether
 * that capture can still be ended depends on its state, which the caller must look at BEFORE walking into hipStreamEndCapture:
 *   hopmi_stream_capture_status: *status = 0 not capturing, 1 capture active (healthy), 2 capture invalidated by an illegal call.
 *   hopmi_stream_capture_abandon: ends whatever capture `stream` is in and destroys the graph it yields, if any; returns 0 when
 *     the stream is out of capture mode afterwards (also when the runtime reports the capture as invalidated, which is the
 *     expected answer for state 2), HOPMI_ELAUNCH otherwise.  Neither call launches or allocates anything. */
int hopmi_stream_capture_status(void* stream, int* status);
int hopmi_stream_capture_abandon(void* stream);

/* Measurement hook: the next hopmi_wn_layer_fwd call of this host thread records the two hipEvent_t EXACTLY around its
 * layer kernel (hipExtLaunchKernelGGL start/stop events: the dispatch's own begin/end timestamps, what a profiler's
 * kernel trace reports), then the hook clears itself.  Pass NULLs to clear.  bench.py uses it for the live roofline. */
int hopmi_time_next_launch(void* start_event, void* stop_event);
/* An empty kernel launched with a WaveNet-layer kernel's shape; honours hopmi_time_next_launch.  Its measured duration is
 * the floor of the dispatch-event timing method (~3.9 us on MI355X / ROCm 7.2; tools/probes/launch_floor.hip). */
int hopmi_noop_launch(void* stream);

/* ---- graph convolution: model/gwnet.py:24-46 (gcn.forward) + :8-14 (nconv) + :16-22 (linear)
 *
 *   h = Wm . [x ; x A1 ; x A2] + bm        with A1 = adp, A2 = adp @ adp
 *
 * i.e. per slab s:  h[s] = X[s] W0^T + (A1^T X[s]) W1^T + (A2^T X[s]) W2^T + bm,
 * X[s] is V x 64, Wm is the reference's mlp.mlp.weight viewed [64][192] = [W0|W1|W2].
 * The reference computes x2 = (x A) A; passing A2 = A @ A (a V x V product done by the
 * caller, so autograd sees it) is the same contraction re-associated.
 */
/* The V x V matrices are the same for all workgroups and all 8 WaveNet layers of a forward pass
 * (gwnet.py:161-164 computes adp once per forward): hopmi_gcn_prepare writes their zero-padded
 * on-chip images into `prep` (hopmi_gcn_prep_floats(V) floats) once; fwd/bwd take `prep`. */
size_t hopmi_gcn_prep_floats(int V);
int hopmi_gcn_prepare(const float* A1, const float* A2, float* prep, int V, void* stream);
int hopmi_gcn_fwd(const float* x, const float* prep, const float* Wm, const float* bm,
                  float* h, int n_slabs, int V, void* stream);

/* Backward of the above w.r.t. everything (autograd of gwnet.py:33-46):
 *   dx[s]  = G0 + A1 G1 + A2 G2,  [G0|G1|G2] = dh[s] Wm
 *   dAk    = sum_s X[s] Gk[s]^T          (the V x V "joint" reduction)
 *   dWm    = sum_rows dh^T [x ; xA1 ; xA2]     dbm = sum_rows dh
 * `ws` must hold hopmi_gcn_bwd_ws_floats(n_slabs, V) floats; it receives per-workgroup
 * partials that a second, fixed-order pass sums (bitwise reproducible, no atomics).
 */
size_t hopmi_gcn_bwd_ws_floats(int n_slabs, int V);
int hopmi_gcn_bwd(const float* x, const float* dh, const float* prep, const float* Wm,
                  float* dx, float* dA1, float* dA2, float* dWm, float* dbm, float* ws,
                  int n_slabs, int V, void* stream);

/* ---- weight images of the fused WaveNet layers (once per forward pass, all layers in one launch)
 *
 * The two channel contractions of a layer (gated TCN, gwnet.py:186-200; graph-conv mlp, gwnet.py:44-46) run as
 * three-term split-bf16 products on v_mfma_f32_16x16x32_bf16: a = a_hi + a_lo (two bf16 numbers, 2^-17 relative),
 * a b ~= a_hi b_hi + a_lo b_hi + a_hi b_lo, fp32 accumulation -- fp32-class accuracy (~1.5e-5 per product, inside the
 * 1e-3 bar with two orders of margin) at 5.3x the rate of the exact-fp32 MFMA.  hopmi_wn_prepare_weights splits the
 * layers' weights into the MFMA A-operand fragments a wave loads:
 *   wf[l], wg[l]  [64][64][1][2]  filter_convs[l].weight / gate_convs[l].weight exactly as nn.Conv2d holds them
 *   Wm[l]         [64][192]       gconv[l].mlp.mlp.weight
 *   image         hopmi_wn_weight_image_bytes(n_layers) bytes (112 KiB per layer); layer l's image starts at
 *                 l * hopmi_wn_weight_image_bytes(1).
 * wf / wg / Wm are HOST arrays of n_layers device pointers (n_layers <= 8). */
size_t hopmi_wn_weight_image_bytes(int n_layers);
int hopmi_wn_prepare_weights(const float* const* wf, const float* const* wg, const float* const* Wm, int n_layers,
                             void* image, void* stream);

/* ---- one fused WaveNet layer, forward: model/gwnet.py:181-237 (gated dilated TCN :186-200, the part of
 *      the skip path that reaches the output :209-220, graph conv :224-231, residual :233, BatchNorm2d
 *      batch statistics :237).  Activations channels-last [B][T][V][64].
 *
 *   xin      [B][T_in][V][64]  previous layer's PRE-BatchNorm output (or the start-conv output)
 *   scsh_in  [128]             scale[64], shift[64] applied to xin on load (previous layer's BatchNorm as
 *                              an affine map; ones / zeros for the first layer)
 *   wimg                       this layer's weight image (hopmi_wn_prepare_weights);  bf, bg [64] the TCN biases
 *   prep, bm                   as hopmi_gcn_fwd (used when do_gcn)
 *   y        [B][T_out][V][64] gcn(u) + bm + r^[t+d], pre-BatchNorm (nullable; T_out = T_in - dilation)
 *   fs       [B][T_out][V][128] tanh and sigmoid gate values (nullable).  The training forward does not store them: the
 *            backward regenerates them per layer with a gate-only call (do_gcn = 0, y = utail = NULL, fs given)
 *   utail    u of the last 4 frames, row (b, f, v) at utail[((b*4 + f)*V + v)*utail_ld .. +64] (nullable: not stored)
 *   ws       nullable; hopmi_wn_layer_ws_floats(...) floats receiving per-workgroup sum / sum-of-squares of y
 *            (training-mode BatchNorm statistics, do_gcn only), to be finalised by hopmi_wn_bn_finalize.
 */
size_t hopmi_wn_layer_ws_floats(int B, int T_in, int V, int dilation);
int hopmi_wn_layer_fwd(const float* xin, const float* scsh_in, const void* wimg, const float* bf, const float* bg,
                       const float* prep, const float* bm, float* y, float* fs, float* utail,
                       int utail_ld, float* ws, int B, int T_in, int V, int dilation, int do_gcn, void* stream);

/* BatchNorm2d training-mode finalisation (gwnet.py:237) from the partials of the layer call with the same
 * (B, T_in, V, dilation): fixed-order sums (16 workgroups, the last to arrive combines them in index order) ->
 * mean_rstd_out [192] (mean, rstd, unbiased variance), scsh_out [128] = the scale / shift the next layer applies on
 * load, running_mean / running_var [64] updated in place (nullable) with torch semantics (biased variance normalises,
 * unbiased feeds running_var).  ws is the layer call's workspace (the kernel uses its tail as scratch).
 * hopmi_wn_bn_replay applies the same running-statistics update once more from a mean_rstd_out (the step's second
 * generator forward reuses the audio branch: model/HOP.py:186-196 would run it, and update the statistics, again). */
int hopmi_wn_bn_finalize(const float* ws, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, float* scsh_out, float* mean_rstd_out,
                         int B, int T_in, int V, int dilation, void* stream);
int hopmi_wn_bn_replay(const float* mean_rstd, float* running_mean, float* running_var, float momentum, void* stream);

/* ---- the whole WaveNet stack, training-mode forward, as ONE persistent launch: model/gwnet.py:181-237 for all layers
 *      (what n_layers x (hopmi_wn_layer_fwd + hopmi_wn_bn_finalize) compute; csrc/wavenet_stack.hip).  Training-mode BatchNorm
 *      makes every layer a chip-wide dependency (the statistics of layer i over all clips feed layer i + 1); inside this launch
 *      the per-workgroup partial sums are exchanged through memory (data-tagged 8-byte granules, two levels, fixed-order
 *      sums: bitwise reproducible) while the next layer's activations are already being loaded, so the seam costs three
 *      memory hops instead of a kernel boundary plus a finalisation launch.
 *
 *   x0        [B][T_in][V][64]   start-conv output;  dilations[n_layers];  T shrinks by dilations[l] per layer
 *   wimg      hopmi_wn_prepare_weights image of the n_layers layers;  bf, bg, bm, gamma, beta: n_layers pointers to [64]
 *   running_mean / running_var: n_layers pointers (entries nullable) updated in place with torch semantics
 *   y         n_layers - 1 pointers: y[l] [B][T_l][V][64] receives layer l's pre-BatchNorm output (the backward's and the next
 *             layer's input); the last layer's output is dead (gwnet.py:240) and not stored
 *   utail     [B][4][V][utail_ld]: layer l's gated activations of the last 4 frames at channel offset 64 l
 *   scsh_out  [n_layers][128]  scale | shift of BN_l (what layer l + 1 applies on load);  mean_rstd_out [n_layers][192]
 *             mean | rstd | unbiased variance
 *   ws        hopmi_wn_stack_ws_bytes(...) bytes, ZERO before the first launch and reused from launch to launch (it holds the
 *             launch sequence number the hand-off tags are made of); int word [32] is the status word (non-zero: a wait timed
 *             out, results invalid; zero the whole workspace before using it again).
 *   The grid (hopmi_wn_stack_grid, <= one workgroup per CU) must be resident at once: do not launch beside kernels that hold
 *   CUs indefinitely.  hopmi_wn_stack_grid returns 0 when the configuration is not supported (use the per-layer calls).
 *   Honours hopmi_time_next_launch. */
int hopmi_wn_stack_grid(int B, int T_in, int V, const int* dilations, int n_layers);
size_t hopmi_wn_stack_ws_bytes(int B, int T_in, int V, const int* dilations, int n_layers);
int hopmi_wn_stack_fwd(const float* x0, const void* wimg, const float* const* bf, const float* const* bg, const float* prep,
                       const float* const* bm, const float* const* gamma, const float* const* beta,
                       float* const* running_mean, float* const* running_var, float momentum, float eps, float* const* y,
                       float* utail, int utail_ld, float* scsh_out, float* mean_rstd_out, void* ws, int B, int T_in, int V,
                       const int* dilations, int n_layers, void* stream);

/* ---- storage type of the graph-wavenet activations (BASELINE.json configs 2 / 4: bf16).  The `_dt` forms of the graph-conv,
 *      WaveNet-layer and WaveNet-stack entry points take `dtype` (0 = fp32, 1 = bf16) for the ACTIVATION tensors that enter and
 *      leave the kernels -- x / h / dh / dx of hopmi_gcn_*, xin / y / utail of the forward, xin / y / dutail of the backward, x0 /
 *      y[l] / utail of the stack -- read as the previous GEMM leaves them and written as the next one reads them (no cast launches
 *      at the block's boundary, half the bytes).  Arithmetic, statistics, weights, scale / shift, the tanh / sigmoid diagnostic
 *      output fs and the gradients between layers (P0 / P1) stay fp32; y is rounded once, when it is stored.  Every other
 *      argument is as in the untyped form, which is the dtype = 0 case. */
int hopmi_gcn_fwd_dt(const void* x, const float* prep, const float* Wm, const float* bm, void* h, int n_slabs, int V, int dtype,
                     void* stream);
int hopmi_gcn_bwd_dt(const void* x, const void* dh, const float* prep, const float* Wm, void* dx, float* dA1, float* dA2, float* dWm,
                     float* dbm, float* ws, int n_slabs, int V, int dtype, void* stream);
int hopmi_wn_layer_fwd_dt(const void* xin, const float* scsh_in, const void* wimg, const float* bf, const float* bg,
                          const float* prep, const float* bm, void* y, float* fs, void* utail, int utail_ld, float* ws, int B,
                          int T_in, int V, int dilation, int do_gcn, int dtype, void* stream);
int hopmi_wn_stack_fwd_dt(const void* x0, const void* wimg, const float* const* bf, const float* const* bg, const float* prep,
                          const float* const* bm, const float* const* gamma, const float* const* beta,
                          float* const* running_mean, float* const* running_var, float momentum, float eps, void* const* y,
                          void* utail, int utail_ld, float* scsh_out, float* mean_rstd_out, void* ws, int B, int T_in, int V,
                          const int* dilations, int n_layers, int dtype, void* stream);
int hopmi_wn_layer_bwd_dt(const void* xin, const float* scsh_in, const float* fs, const float* wf, const float* wg,
                          const float* prep, const float* Wm, const float* P0n, const float* P1n, int d_next,
                          const void* y, const float* bn_coef, const void* dutail, int dutail_ld,
                          const float* gamma_prev, const float* mean_rstd_prev,
                          float* P0, float* P1, float* dwf, float* dwg, float* dbtcn, float* dWm, float* dbm,
                          float* dA1, float* dA2, int accumulate_dA, float* dgamma_prev, float* dbeta_prev,
                          float* coef_prev, float* ws,
                          int B, int T_in, int V, int dilation, int do_gcn, int dtype, void* stream);

/* Backward of one fused WaveNet layer (autograd of gwnet.py:181-237), see csrc/wavenet_bwd.hip.
 *   xin, scsh_in, fs, wf, wg, prep, Wm : as in / saved by the forward
 *   P0n, P1n [B][T_out - d_next][V][64]: gradient w.r.t. this layer's BatchNorm output as written by the NEXT
 *            layer's backward (its tap-0 / tap-1 contributions), d_next = that layer's dilation   (do_gcn only)
 *   y, bn_coef [3][64]: this layer's pre-BN output and the dy = ca*dx^ + cb*y + ck coefficients that the next
 *            layer's backward call produced (coef_prev there)                                     (do_gcn only)
 *   dutail: gradient w.r.t. this layer's skip-tail block, rows of stride dutail_ld
 *   gamma_prev, mean_rstd_prev: BatchNorm_{i-1} (nullable for the first layer)
 * Outputs: P0, P1 [B][T_out][V][64] (for the previous layer / the start conv), dwf, dwg [64][64][1][2] (the conv
 *   weights' own layout), dbtcn [128] = d(bf) | d(bg), dWm [64][192], dbm [64], dA1, dA2 [V][V] (do_gcn only;
 *   accumulate_dA != 0 ADDS into them, so one pair of buffers collects all layers), dgamma_prev, dbeta_prev [64],
 *   coef_prev [3][64].
 *   ws: hopmi_wn_layer_bwd_ws_floats(...) floats.  Two launches (layer kernel + fixed-order reduce).
 *   The kernel keeps five tile images and both mix images in LDS: V <= 42 (hopmi_wn_layer_bwd_ws_floats returns 0 and
 *   hopmi_wn_layer_bwd HOPMI_EINVAL for larger graphs; hopmi_gcn_bwd covers those). */
size_t hopmi_wn_layer_bwd_ws_floats(int B, int T_in, int V, int dilation);
int hopmi_wn_layer_bwd(const float* xin, const float* scsh_in, const float* fs, const float* wf, const float* wg,
                       const float* prep, const float* Wm, const float* P0n, const float* P1n, int d_next,
                       const float* y, const float* bn_coef, const float* dutail, int dutail_ld,
                       const float* gamma_prev, const float* mean_rstd_prev,
                       float* P0, float* P1, float* dwf, float* dwg, float* dbtcn, float* dWm, float* dbm, float* dA1,
                       float* dA2, int accumulate_dA, float* dgamma_prev, float* dbeta_prev, float* coef_prev, float* ws,
                       int B, int T_in, int V, int dilation, int do_gcn, void* stream);

/* ---- reprogramming cross-attention: model/HOP.py:289-299 (ReprogrammingLayer.reprogramming)
 *   o[n][h][:] = sum_s dropout(softmax_s(scale * q[n][h][:] . k[s][h][:])) v[s][h][:]
 *   q, o [N][H][E] with N = B*L flat query rows; k, v [S][H][E] shared by the batch; E must be 128;
 *   lse [N][H] = log-sum-exp of the scaled scores (saved for the backward).
 *   Dropout keeps probability (n, h, s) iff hash(seed', n, h, s) >= p_drop * 2^32 (stateless, so the
 *   backward regenerates the same mask); p_drop = 0 disables it.  seed' = seed + *seed_dev when seed_dev (a device
 *   pointer, nullable) is given: the stream position then lives in device memory, so a captured hipGraph of the
 *   training step draws fresh masks on every replay (the host adds to *seed_dev inside the graph).  Every seeded
 *   entry point below takes the same (seed, seed_dev) pair.
 *   Both contractions run on v_mfma_f32_16x16x32_bf16 with every operand carried as a hi + lo bf16 pair (three terms,
 *   ~2^-16 relative per product, fp32 accumulation and fp32 softmax).  ws: hopmi_reprog_attn_ws_bytes(S, H, E) bytes, the
 *   bf16 operand images of k and v, rebuilt by every call (two small launches in front of the main kernel).
 */
size_t hopmi_reprog_attn_ws_bytes(int S, int H, int E);
int hopmi_reprog_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, void* ws,
                          int N, int S, int H, int E, float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream);

/* The same with the storage type of q, k, v and o as an argument (dtype 0 = fp32, 1 = bf16; lse stays fp32): under bf16
 * autocast (BASELINE.json configs 2, 4) the tensors are read as they leave the projection GEMMs and o is written as the
 * output projection reads it -- no cast launches.  A bf16 operand has no lo part: its terms are not issued (two MFMA
 * terms per product instead of three; the scaled q and the probabilities keep their lo parts). */
int hopmi_reprog_attn_fwd_dt(const void* q, const void* k, const void* v, void* o, int dtype, float* lse, void* ws,
                             int N, int S, int H, int E, float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream);

/* Backward of the above: d_o [N][H][E] -> dq [N][H][E] and PARTIAL dk, dv [R][S][H][E] with
 * R = hopmi_reprog_attn_bwd_splits() (the query rows are split R ways over workgroups; the caller adds the
 * R slabs in order).  delta [N][H] = sum_e d_o * o (one small reduction by the caller).  ws:
 * hopmi_reprog_attn_bwd_ws_bytes(N, S, H, E) bytes (bf16 operand images of k, v, q, d_o, rebuilt by every call).  Four
 * image launches + two main launches (same split-bf16 arithmetic as the forward); every output element has one owner:
 * no atomics, bitwise reproducible. */
int hopmi_reprog_attn_bwd_splits(void);
size_t hopmi_reprog_attn_bwd_ws_bytes(int N, int S, int H, int E);
int hopmi_reprog_attn_bwd(const float* q, const float* k, const float* v, const float* d_o, const float* lse,
                          const float* delta, float* dq, float* dk, float* dv, void* ws, int N, int S, int H, int E,
                          float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream);

/* ... with the storage type of q, k, v, d_o and dq as an argument (dk, dv partial slabs, lse and delta stay fp32). */
int hopmi_reprog_attn_bwd_dt(const void* q, const void* k, const void* v, const void* d_o, int dtype, const float* lse,
                             const float* delta, void* dq, float* dk, float* dv, void* ws, int N, int S, int H, int E,
                             float scale, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream);

/* ---- bias gradient of a trainable linear layer: out[N] = sum over the M rows of x [M][N] (what the backward of
 *      torch.nn.functional.linear -- align_layer, the projections, the GRU input projections, the gwnet 1x1 convs,
 *      HOP.py:104-173,262-265; gwnet.py:65,117,129-134 -- computes with a generic reduction).  dtype of x: 0 = fp32, 1 = bf16; out fp32.
 *      ws: hopmi_colsum_ws_floats(M, N) floats (partial sums of 128-row chunks; none for M <= 128).  Fixed summation
 *      order: bitwise reproducible. ---- */
size_t hopmi_colsum_ws_floats(int M, int N);
int hopmi_colsum(const void* x, int dtype, int M, int N, float* out, float* ws, void* stream);

/* ---- self-attention of the frozen BERT encoder (HOP.py:204 -> transformers BertSelfAttention.forward; replaces
 *      the transpose_for_scores copies + the library scaled-dot-product attention + the output re-layout) ----
 *   qkv   [B][L][3][H][64]  output of the fused Q|K|V projection (bias added), L <= 64, head dim 64
 *   out   [B][L][H*64]      dropout(softmax(q k^T / 8)) v, laid out for the output projection
 *   backward: d_out [B][L][H*64] -> dqkv [B][L][3][H][64] (gradient of the projection output); probabilities are
 *   recomputed, nothing is saved.  Dropout keeps (b*L + l, h, key) iff hash(seed, b*L + l, h, key) >= p_drop * 2^32
 *   (same stateless hash as hopmi_reprog_attn_fwd).  One workgroup per (b, h): no atomics, reproducible. */
int hopmi_bert_attn_fwd(const float* qkv, float* out, int B, int L, int H, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream);
/* ... (round 5, fp32) also writing `out`'s fp16 hi / lo operand image (hopmi_rows_image_f16_bytes(B L, 64 H)) and its [2][B L] row
 * scales for the attention-output GEMM (hopmi_gemm_f16x2_ab_ep).  A row spans the H workgroups of its clip, so the scale is one per
 * clip from a bound: |dropout(P) V| <= max |V[clip]| / (1 - p); max |V| is read from v_rowmax = the QKV product's c_rowmax
 * [tiles][B L] (hopmi_gemm_f16x2 / _ab_ep), V's columns being tiles vt0 .. vt1 - 1. */
int hopmi_bert_attn_fwd_im(const float* qkv, float* out, const float* v_rowmax, int vt0, int vt1, void* image, float* scales, int B, int L,
                           int H, float p_drop, unsigned seed, const unsigned* seed_dev, void* stream);
int hopmi_bert_attn_bwd(const float* qkv, const float* d_out, float* dqkv, int B, int L, int H, float p_drop,
                        unsigned seed, const unsigned* seed_dev, void* stream);

/* ---- fused element-wise epilogues of the frozen BERT block (HOP.py:204 -> transformers BertIntermediate /
 *      BertSelfOutput / BertOutput), forward and backward w.r.t. activations (the LLM is frozen, HOP.py:90-91).
 *   bias_gelu:  out[M][N] = gelu_erf(x + bias)                     dx = dy * gelu'(x + bias)
 *   bias_dropout_residual_layernorm:
 *       z = dropout(x + bias) + res[row % res_rows];  out = LayerNorm_eps(z) * gamma + beta
 *       saves xhat [M][D] and rstd [M] (nullable in inference);  backward returns dx (through the dropout
 *       mask, regenerated from the same (seed,row,col) hash) and dres.  N, D multiples of 4, D <= 1024. */
int hopmi_bias_gelu_fwd(const float* x, const float* bias, float* out, int M, int N, void* stream);
int hopmi_bias_gelu_bwd(const float* x, const float* bias, const float* dy, float* dx, int M, int N, void* stream);
int hopmi_bias_dropout_residual_layernorm_fwd(const float* x, const float* bias, const float* res, int res_rows,
                                              const float* gamma, const float* beta, float* out, float* xhat,
                                              float* rstd, int M, int D, float eps, float p_drop, unsigned seed, const unsigned* seed_dev,
                                              void* stream);
int hopmi_bias_dropout_residual_layernorm_bwd(const float* dout, const float* xhat, const float* rstd,
                                              const float* gamma, float* dx, float* dres, int M, int D,
                                              float p_drop, unsigned seed, const unsigned* seed_dev, void* stream);

/* ---- storage-typed forms of the frozen BERT's operators (g1: bf16 configurations).  `dtype` = 0 (fp32) or 1 (bf16) is the
 *      storage type of the tensors that sit between two GEMMs -- x / out / dy / dx of bias + GELU, qkv / out / d_out / dqkv of the
 *      self-attention, x (GEMM output) / out_t (next GEMM's input) / dout_t / dx of bias + dropout + residual + LayerNorm, whose
 *      residual stream (res, out, dout, dres) and saved xhat / rstd stay fp32.  All arithmetic is fp32; with dtype 1 the values are
 *      those of the fp32 forms under torch autocast, rounded to bf16 at the store instead of by a cast kernel.  The untyped entry
 *      points above are these with dtype 0 (and out_t / dout_t NULL). */
int hopmi_bias_gelu_fwd_dt(const void* x, const float* bias, void* out, int M, int N, int dtype, void* stream);
int hopmi_bias_gelu_bwd_dt(const void* x, const float* bias, const void* dy, void* dx, int M, int N, int dtype, void* stream);
int hopmi_bias_dropout_residual_layernorm_fwd_dt(const void* x, const float* bias, const float* res, int res_rows,
                                                 const float* gamma, const float* beta, float* out, void* out_t, float* xhat,
                                                 float* rstd, int M, int D, float eps, float p_drop, unsigned seed,
                                                 const unsigned* seed_dev, int dtype, void* stream);
int hopmi_bias_dropout_residual_layernorm_bwd_dt(const float* dout, const void* dout_t, const float* xhat, const float* rstd,
                                                 const float* gamma, void* dx, float* dres, int M, int D, float p_drop,
                                                 unsigned seed, const unsigned* seed_dev, int dtype, void* stream);
/* The same two operators also emitting the fp16-form GEMM operand scales of what they hand to the next GEMM (hopmi_gemm_f16x2's
 * a_scales [2][M] of `out` / of `dx`; row_scales NULL = the _dt forms): the rows are in registers here, a hopmi_row_scales pass over
 * them afterwards costs a launch and a re-read. */
int hopmi_bias_dropout_residual_layernorm_fwd_rs(const void* x, const float* bias, const float* res, int res_rows,
                                                 const float* gamma, const float* beta, float* out, void* out_t, float* xhat,
                                                 float* rstd, float* row_scales, int M, int D, float eps, float p_drop, unsigned seed,
                                                 const unsigned* seed_dev, int dtype, void* stream);
int hopmi_bias_dropout_residual_layernorm_bwd_rs(const float* dout, const void* dout_t, const float* xhat, const float* rstd,
                                                 const float* gamma, void* dx, float* dres, float* row_scales, int M, int D, float p_drop,
                                                 unsigned seed, const unsigned* seed_dev, int dtype, void* stream);
/* ... and (round 5) the operand IMAGE itself: `image` (nullable; needs row_scales) receives `out` / `dx` times the row's power of two
 * as fp16 hi / lo images [2][M][D] -- hopmi_rows_image_f16's layout, the A operand of hopmi_gemm_f16x2_ab(_ep): the GEMM behind the
 * LayerNorm (QKV, FFN-in; in the backward the attention-output and FFN-out gradients) then stages both operands by LDS-DMA and
 * splits nothing in its k-loop (59 vs 79 us at N = 2304, 77 vs 99 at N = 3072, M = 4352).  `row_norms` (nullable): [M] 2-norms of the
 * rows (rounded up) -- what hopmi_gemm_f16x2_ab_img's a-priori bound is made of. */
int hopmi_bias_dropout_residual_layernorm_fwd_im(const void* x, const float* bias, const float* res, int res_rows,
                                                 const float* gamma, const float* beta, float* out, void* out_t, float* xhat,
                                                 float* rstd, float* row_scales, void* image, float* row_norms, int M, int D, float eps,
                                                 float p_drop, unsigned seed, const unsigned* seed_dev, int dtype, void* stream);
int hopmi_bias_dropout_residual_layernorm_bwd_im(const float* dout, const void* dout_t, const float* xhat, const float* rstd,
                                                 const float* gamma, void* dx, float* dres, float* row_scales, void* image,
                                                 float* row_norms, int M, int D, float p_drop, unsigned seed, const unsigned* seed_dev,
                                                 int dtype, void* stream);
int hopmi_bert_attn_fwd_dt(const void* qkv, void* out, int B, int L, int H, float p_drop, unsigned seed, const unsigned* seed_dev,
                           int dtype, void* stream);
int hopmi_bert_attn_bwd_dt(const void* qkv, const void* d_out, void* dqkv, int B, int L, int H, float p_drop, unsigned seed,
                           const unsigned* seed_dev, int dtype, void* stream);

/* ---- bidirectional GRU layer recurrence: model/HOP.py:166-167,248 (decoder nn.GRU, hidden 350) and
 *      model/multimodal_context_net.py:236-237,257 (discriminator nn.GRU, hidden 64); torch.nn.GRU
 *      semantics, gate order r,z,n, h0 = 0.
 *
 *   gi    [B][T][2][3H]  input projections x W_ih^T + b_ih of both directions (one GEMM by the caller)
 *   whh   [2][3H][H], bhh [2][3H]
 *   y     [B][T][2H]     layer output, forward direction in [:H], reverse in [H:]  (torch layout)
 *   gates [B][T][2][4H]  saved for the backward: r, z, n and (W_hn h + b_hn)
 * ws: nullable workspace of hopmi_gru_ws_bytes(B, T, H) bytes.  With it, and when every workgroup of the layer can
 * be resident at once (8 * ceil(2 ceil(B/16) / 8) * ceil(H/32) <= #CUs), the layer is ONE persistent launch: each
 * workgroup owns 16 batch rows x 32 hidden units for all T steps, keeps its W_hh fragments in registers (split into
 * hi + lo bf16 pairs; the product runs as three v_mfma_f32_16x16x32_bf16 terms, ~2^-16 relative per product, fp32
 * accumulation) and the workgroups of a (direction, batch group) hand h_t over through y itself: y is filled with a
 * "not written yet" NaN pattern before the launch and a consumer re-loads whatever still reads as that pattern
 * (bounded).  The int at index [size - 16] of ws is a status word, non-zero if a hand-off ever timed out (y is then
 * garbage).  Otherwise one exact-fp32 launch per time step is enqueued (both directions per launch).  y must be
 * 8-byte aligned.
 */
size_t hopmi_gru_ws_bytes(int B, int T, int H);
int hopmi_gru_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* gates, void* ws,
                  int B, int T, int H, void* stream);

/* Back-propagation through time of the above.  dy [B][T][2H] -> dgi [B][T][2][3H] (gradient w.r.t. the
 * input projections) and dgh [B][T][2][3H] (w.r.t. the recurrent pre-activations); the caller turns them
 * into dx, dW_ih, db_ih, dW_hh, db_hh with four GEMMs / reductions.  whhT [2][H][3H] is W_hh transposed
 * per direction; ws holds hopmi_gru_bwd_ws_floats(B, H) floats (used by the per-step form); ws2 (nullable) =
 * hopmi_gru_ws_bytes(B, T, H) bytes enables the persistent one-launch form exactly as in hopmi_gru_fwd (the hand-off
 * array is dgh). */
size_t hopmi_gru_bwd_ws_floats(int B, int H);
int hopmi_gru_bwd(const float* dy, const float* y, const float* gates, const float* whhT,
                  float* dgi, float* dgh, float* ws, void* ws2, int B, int T, int H, void* stream);

/* ---- the generator losses of one training step: train_eval/train_llm.py:46-79
 *   huber   = smooth_l1(out / 0.1, target / 0.1) * 0.1 (mean);  pose_b = sum_f smooth_l1(out[b] / .05, out_rand[b] / .05) * .05;
 *   z_b = mean_j |z_context[b] - z_rand[b]|;  div_reg = mean_b max(-pose_b / (z_b + 1e-5), -1000);
 *   kld = -0.5 mean(1 + logvar - mu^2 - exp(logvar));  total = w_reg huber + w_div div_reg + w_kld kld.
 *   out, target, out_rand [B][F]; z_context, z_rand, mu, logvar [B][Z].  out_rand / z_context / z_rand are nullable
 *   together (no div_reg term), mu / logvar together (no kld term).  vals [4] = huber, div_reg, kld, total.
 *   ws: hopmi_hop_losses_ws_floats(B) floats, kept for the backward.  Backward: gradient of total times the device scalar
 *   *g w.r.t. out (d_out [B][F]), mu and logvar (d_mu, d_logvar [B][Z], nullable with mu); target, out_rand, z_context and
 *   z_rand take none (the reference detaches them).  Fixed-order reductions: bitwise reproducible. */
size_t hopmi_hop_losses_ws_floats(int B);
int hopmi_hop_losses_fwd(const float* out, const float* target, const float* out_rand, const float* z_context,
                         const float* z_rand, const float* mu, const float* logvar, int B, int F, int Z, float w_reg,
                         float w_div, float w_kld, float* vals, float* ws, void* stream);
int hopmi_hop_losses_bwd(const float* out, const float* target, const float* out_rand, const float* mu, const float* logvar,
                         const float* ws, const float* g, int B, int F, int Z, float w_reg, float w_kld, float* d_out,
                         float* d_mu, float* d_logvar, void* stream);

/* The same two entry points with the storage type of the input projections as an argument (0 = fp32, 1 = bf16): under bf16
 * autocast gi comes straight from the library GEMM and dgi goes straight into its backward; everything else stays fp32. */
int hopmi_gru_fwd_dt(const void* gi, int gi_dtype, const float* whh, const float* bhh, float* y, float* gates, void* ws,
                     int B, int T, int H, void* stream);
int hopmi_gru_bwd_dt(const float* dy, const float* y, const float* gates, const float* whhT, void* dgi, int dgi_dtype,
                     float* dgh, float* ws, void* ws2, int B, int T, int H, void* stream);

/* hopmi_gru_fwd_dt over TWO input-projection tensors in one launch (round 6): batch rows [0, B1) read gi1 (B1,T,2,3H), rows
 * [B1, B) read gi2 (B - B1,T,2,3H); y (B,T,2H) and gates (B,T,2,4H) are whole.  The decoder of a train_llm step runs twice on the
 * same weights -- the graded forward and the no-grad forward of the diversity regulariser (train_llm.py:42,58; HOP.py:248) -- and a
 * time step of the recurrence is a hand-off round trip around very little arithmetic: the second batch rides in the same launch
 * (persistent kernel with 32 rows per workgroup when 16-row tiles would not fit the chip).  Where no persistent form applies the
 * call runs as two hopmi_gru_fwd_dt calls; results are those of the two calls either way (fp32 class). */
int hopmi_gru_fwd_pair_dt(const void* gi1, const void* gi2, int B1, int gi_dtype, const float* whh, const float* bhh, float* y,
                          float* gates, void* ws, int B, int T, int H, void* stream);

/* Adam over a list of fp32 tensors in ONE launch (csrc/adam.hip; train_llm.py:86 `model_optim.step()`; arithmetic of torch's
 * _fused_adam_ without weight decay / amsgrad / maximize).  `tensors`: device array of {float* p; const float* g; float* m;
 * float* v; long long n} (40 bytes each); `items`: device array of n_items (tensor index, chunk index) int pairs, chunk =
 * hopmi_adam_chunk() elements; `step`: the optimizer's device-side step counter, ALREADY advanced for this step (bias corrections
 * 1 - beta^step, taken in double like 1 - beta: in fp32 1 - 0.999f is off by 1.3e-5).  p, m, v are updated in place. */
int hopmi_adam_chunk(void);
int hopmi_adam_multi(const void* tensors, const void* items, int n_items, double lr, double beta1, double beta2, double eps,
                     const float* step, void* stream);

/* The two operands of a layer's backward that are re-arrangements of forward tensors, in one launch: whhT (2,H,3H) = whh (2,3H,H)
 * transposed per direction (the whhT argument of hopmi_gru_bwd), and hprev (B,T,2,H) = y shifted one step along each direction's
 * processing order, zero at its first step (dW_hh = sum_{b,t} dgh^T hprev; multimodal_context_net.py:35 -> torch.nn.GRU backward). */
int hopmi_gru_bwd_operands(const float* y, const float* whh, float* hprev, float* whhT, int B, int T, int H, void* stream);

/* ---- BatchNorm1d of the discriminator's pre_conv (multimodal_context_net.py:226-234: nn.BatchNorm1d(16) / (8) between the
 *      Conv1d layers) on channels-last rows x [M = B*T][C <= 64], one launch forward and one backward.
 *   training != 0: batch statistics over the M rows (biased variance for the normalisation), running_mean / running_var advanced
 *     by `momentum` (unbiased variance) when not NULL, save_mean_rstd [2][C] written when not NULL; y = (x - mean) rstd gamma + beta.
 *   training == 0: y from the running statistics.  y may be NULL (statistics update only).
 *   backward: dx (NULL: parameter gradients only), dgamma [C], dbeta [C] from x, dy and the saved mean / rstd. */
int hopmi_bn_cl_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                    float* save_mean_rstd, int M, int C, float eps, float momentum, int training, void* stream);
int hopmi_bn_cl_bwd(const float* x, const float* dy, const float* gamma, const float* save_mean_rstd, float* dx, float* dgamma,
                    float* dbeta, int M, int C, void* stream);

/* ---- fp32 GEMM against frozen weights on the bf16 matrix cores (the frozen BERT's linears, HOP.py:90-91,204 ->
 *      transformers BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput nn.Linear calls)
 *   C[M][N] = A[M][K] . Bt[N][K]^T (+ bias[N]);  A, C fp32 row-major.
 * Every operand is carried as `parts` bf16 numbers and the product taken as the terms a_i b_j with i + j < parts on
 * v_mfma_f32_16x16x32_bf16, fp32 accumulation:  parts = 3 keeps all 24 significand bits (6 terms; the dropped ones are below
 * an fp32 multiply's rounding error: fp32-equivalent), parts = 2 gives 2^-16-class products (3 terms) at twice the rate.
 * The weight operand is split once (frozen weights): hopmi_gemm_split_prepare(W [N][K]) -> image of
 * hopmi_gemm_split_image_bytes(N, K, parts) bytes; for the activation gradient dX = dY . W pass the image of W^T.
 * N % 128 == 0, K % 32 == 0, any M. */
size_t hopmi_gemm_split_image_bytes(int N, int K, int parts);
int hopmi_gemm_split_prepare(const float* W, int N, int K, int parts, void* image, void* stream);
int hopmi_gemm_split(const float* A, const void* Bimage, const float* bias, float* C, int M, int N, int K, int parts,
                     void* stream);

/* The same product with an activation epilogue (BertIntermediate, transformers modeling_bert.py: `intermediate_act_fn(dense(x))`,
 * gelu = x/2 (1 + erf(x / sqrt 2)), and its gradient), so that the M x 3072 tensor is not re-read and re-written by a launch of its
 * own.  epilogue 0: C = A.Bt^T + bias (= hopmi_gemm_split);  1: h = A.Bt^T + bias, C = gelu(h), C2 = h when C2 is not NULL (the
 * backward's operand);  2: C = (A.Bt^T + bias) * gelu'(aux), aux [M][N] = the h of the forward.  Bit-identical to hopmi_gemm_split
 * followed by hopmi_bias_gelu_fwd / _bwd (the same expressions on the same fp32 values). */
int hopmi_gemm_split_ep(const float* A, const void* Bimage, const float* bias, float* C, float* C2, const float* aux, int M, int N,
                        int K, int parts, int epilogue, void* stream);

/* The fp16 hi/lo form of the same product (round 4): every operand carried as TWO scaled fp16 numbers, THREE MFMA terms
 * (hi hi + hi lo + lo hi on v_mfma_f32_16x16x32_f16), fp32-equivalent like the six-term bf16 form at half the matrix work
 * (csrc/gemm.hip).  Any M, N and any K % 4 == 0: the weight image is padded to whole tiles.
 *   hopmi_gemm_f16x2_prepare(W [N][K], N, K, image): the image of W, the Bt operand of y = x W^T (for dX = dY W the caller
 *     prepares the image of the row-major W^T, [K][N]).  One launch; one power-of-two scale per image row.  Frozen weights: once;
 *     trainable weights: once per optimizer step (hopmi/ops.py caches by owner parameter + version counter).
 *     hopmi_gemm_split_prepare / _image_bytes with parts = 16 are the same calls.
 *   a_parts = 0: a_scales [2][M] = {s_row, 1 / s_row}, the activations' power-of-two scale PER ROW, written by
 *     hopmi_row_scales(A, M, K, a_scales) -- or by whichever kernel produced A (hopmi_bias_dropout_residual_layernorm_*_rs).
 *   a_parts = P > 0: a_scales [P][M] = partial row maxima of |A| (one per workgroup of the producer that held a piece of the row),
 *     reduced and turned into scales in this kernel's prologue -- what c_rowmax of another hopmi_gemm_f16x2 call is.
 *   c_rowmax (nullable): [hopmi_gemm_f16x2_tiles_n(N)][M] receives, per column tile, the row maxima of |C| as written (after the
 *     epilogue): the a_scales / a_parts = tiles_n of the GEMM that consumes C (BertIntermediate -> BertOutput.dense and its backward).
 *   epilogue / C2 / aux as hopmi_gemm_split_ep.
 * Replaces: the HF BERT linears behind model/HOP.py:204 (built at run_ted.py:177-195) and the generator's own nn.Linear / GRU input
 * projections (HOP.py:118,130-134,166-167,259-265: align_layer, beat, gru weight_ih, the reprogramming projections). */
size_t hopmi_gemm_f16x2_image_bytes(int N, int K);
int hopmi_gemm_f16x2_prepare(const float* W, int N, int K, void* image, void* stream);
int hopmi_row_scales(const float* A, int M, int K, float* scales, void* stream);
int hopmi_gemm_f16x2_tiles_n(int N);
int hopmi_gemm_f16x2(const float* A, const float* a_scales, int a_parts, const void* Bimage, const float* bias, float* C, float* C2,
                     const float* aux, float* c_rowmax, int M, int N, int K, int epilogue, void* stream);

/* The fp16 form with BOTH operands as images, every tile staged by LDS-DMA (no staging registers, no split in the k-loop; csrc/gemm.hip:
 * gemm_split_ab_kernel<2, ., true>): hopmi_rows_image_f16 writes the activations' scaled fp16 hi / lo images [2][M][K] and their
 * [2][M] row-scale pairs in one pass (what hopmi_row_scales + the in-kernel split do together), hopmi_gemm_f16x2_ab multiplies
 * (K % 32 == 0, any N; bias epilogue only).  Bit-identical to hopmi_gemm_f16x2; faster where one 128 x 128 tile per CU covers the
 * problem (N = 768 at M = 4352: 24 vs 30 us, K = 3072: 71 vs 92). */
size_t hopmi_rows_image_f16_bytes(int M, int K);     /* tile-blocked, rows padded to 128 (csrc/gemm.hip f16_blk); even K: columns padded
                                                         with zeros to the 32-wide k-step (8-byte loads when K % 32 != 0) */
int hopmi_rows_image_f16(const float* A, int M, int K, void* image, float* scales, void* stream);
int hopmi_gemm_f16x2_ab(const void* Aimage, const float* a_scales, const void* Bimage, const float* bias, float* C, int M, int N,
                        int K, void* stream);
/* ... with hopmi_gemm_f16x2's epilogues (0 bias, 1 GELU (+ C2 = the pre-activation), 2 GELU gradient against aux) and c_rowmax. */
int hopmi_gemm_f16x2_ab_ep(const void* Aimage, const float* a_scales, const void* Bimage, const float* bias, float* C, float* C2,
                           const float* aux, float* c_rowmax, int M, int N, int K, int epilogue, void* stream);
/* ... handing its result on as the NEXT fp16-form GEMM's operand image: out_image (hopmi_rows_image_f16_bytes(M, N), N % 32 == 0) and
 * out_scales [2][M]; C may be NULL (nothing else reads the fp32 values: BertIntermediate's output only feeds BertOutput.dense, the
 * GELU-gradient product only the FFN-in activation gradient).  The row scales come from the bound |C[row][:]| <= row_norm[row] *
 * bound_mul + bound_add (csrc/gemm.hip AbImageOut). */
int hopmi_gemm_f16x2_ab_img(const void* Aimage, const float* a_scales, const void* Bimage, const float* bias, float* C, float* C2,
                            const float* aux, int M, int N, int K, int epilogue, void* out_image, float* out_scales,
                            const float* row_norm, float bound_mul, float bound_add, void* stream);
/* Split-K of the same kernel for a product with few output tiles and a huge contraction: C[M][N] = A Bt^T + row_bias[M][:, None]
 * (row_bias nullable), A = hopmi_rows_image_f16 image, Bt = hopmi_gemm_f16x2_prepare image, any even K, N % 4 == 0.  The k-steps are
 * cut into slabs (about one round of three workgroups per CU), each slab's partial product goes to `workspace`
 * (hopmi_gemm_f16x2_ab_splitk_ws_floats floats) and a second launch adds the slabs in index order (bitwise reproducible).
 * Replaces: the mapping layer's forward S = mapping_layer(word_embeddings^T)^T = W_map E + b (model/HOP.py:116,200; 1500 x 768
 * outputs, K = 30522 -- the last large library GEMM of the fp32 step: 534 -> 3xx us). */
size_t hopmi_gemm_f16x2_ab_splitk_ws_floats(int M, int N, int K);
int hopmi_gemm_f16x2_ab_splitk(const void* Aimage, const float* a_scales, const void* Bimage, const float* row_bias, float* C, int M,
                               int N, int K, float* workspace, void* stream);

/* The weight gradient of a linear layer in the same arithmetic (round 5; csrc/gemm_tn.hip):  C[N][K] (+)= A[M][N]^T . B[M][K]  -- both
 * operands ACTIVATIONS (A = dY, B = X, row-major with leading dimensions lda / ldb), contraction over their M rows; `batch`
 * independent products at element strides batch_stride_* (the two directions of a GRU's recurrent gradient, HOP.py:166-167).
 * a_rows / b_rows: the operands' [2][M] row-scale pairs as hopmi_row_scales (or a fused producer) writes them -- the kernel takes the
 * smallest row scale of each as the operand's ONE power-of-two scale (rows are the contraction index here).  Shapes with few output
 * tiles split their rows over workgroups: `ws` = hopmi_gemm_f16x2_tn_ws_floats(M, N, K, batch) floats (0: not needed), summed in
 * index order by a second launch (bitwise reproducible).  accumulate != 0: C += .
 * Replaces: dW = dY^T X of every nn.Linear / nn.GRU weight of the generator (autograd of HOP.py:116-134,166-167,259-265). */
size_t hopmi_gemm_f16x2_tn_ws_floats(int M, int N, int K, int batch);
int hopmi_gemm_f16x2_tn(const float* A, int lda, long long batch_stride_a, const float* a_rows, const float* B, int ldb,
                        long long batch_stride_b, const float* b_rows, float* C, int ldc, long long batch_stride_c, float* ws, int M, int N,
                        int K, int batch, int accumulate, void* stream);
/* ... also leaving a_colsum [N] = the column sums of A (batch == 1): with A = dY that is the bias gradient of the same linear, a
 * by-product of the rows the first k-tile column of workgroups stages anyway (replaces a hopmi_colsum pass: 2 launches, one more
 * read of dY).  Fixed summation order: bitwise run-to-run. */
int hopmi_gemm_f16x2_tn_cs(const float* A, int lda, long long batch_stride_a, const float* a_rows, const float* B, int ldb,
                           long long batch_stride_b, const float* b_rows, float* C, int ldc, long long batch_stride_c, float* ws, int M,
                           int N, int K, int batch, int accumulate, float* a_colsum, void* stream);

/* The same product with BOTH operands as part images (Aimage = hopmi_gemm_split_prepare(A, M, K, parts, ...), i.e.
 * [parts][M][K] bf16; a producer may also write that layout itself): nothing is split inside the kernel, every tile is staged
 * by LDS-DMA.  Same arithmetic (the same MFMA terms in the same order) as hopmi_gemm_split: results are bit-identical. */
int hopmi_gemm_split_ab(const void* Aimage, const void* Bimage, const float* bias, float* C, int M, int N, int K, int parts,
                        void* stream);

/* ---- log-mel spectrogram of the audio clips: data_loader/lmdb_data_loader.py:216-218
 *      melspec = librosa.feature.melspectrogram(y, sr=16000, n_fft=1024, hop_length=hop, power=2)   (librosa 0.8.1:
 *      periodic Hann window, center=True with reflect padding, Slaney mel filters);  out = power_to_db(melspec, ref=np.max).T
 *   audio [B][n_samples] -> out [B][1 + n_samples/hop][128]  (amin 1e-10, top_db 80).
 *   The 128 triangular filters come as compact runs: band m covers FFT bins band_start[m] .. +band_len[m], weights at
 *   band_w[band_off[m] ..] (host side: hopmi/feeder.py::mel_filter_runs).  mel_power_ws: B * frames * 128 floats. */
int hopmi_logmel(const float* audio, int B, int n_samples, int hop, const int* band_start, const int* band_len,
                 const int* band_off, const float* band_w, float* mel_power_ws, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HOPMI_H */
