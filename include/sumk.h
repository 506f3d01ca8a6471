/* sumk.h -- C ABI of libsumk.so: MI355X (gfx950) frame-importance scoring kernels.
 *
 * Drop-in boundary for the sequence-scorer hot path of sylvainma/Summarizer.  The reference has NO native
 * layer: its hot path is the implicit ATen dispatch under the nn.Module.forward calls
 * cited per entry point below, so each entry replaces "what torch would run" for one reference call site.
 * Python binds this header with ctypes (summarizer_amd/_lib.py); a maintainer's stub is in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; sumk_last_error() gives the message (thread local).
 *  - all matrices are dense row-major fp32 in DEVICE memory, BORROWED for the duration of the call.
 *  - a batch of videos is PACKED: frames of video s are rows [seq_off[s], seq_off[s+1]) of every per-frame
 *    matrix.  seq_off is given twice: a host copy (grid sizing, workspace carving) and a device copy
 *    (read by the kernels).  n_rows == seq_off[n_seq].
 *  - no hidden allocation: the caller owns the workspace; query its size first.
 *  - `stream` is a hipStream_t (passed as void*); calls only enqueue work and are graph-capturable.
 *  - D (feature size) and H (LSTM hidden size) must be multiples of 4.
 */
#ifndef SUMK_H
#define SUMK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUMK_OK 0
#define SUMK_ERR_ARG (-1)       /* bad shape / null pointer / unsupported size */
#define SUMK_ERR_WORKSPACE (-2) /* workspace too small */
#define SUMK_ERR_HIP (-3)       /* HIP runtime error (message has hipGetErrorString) */

const char* sumk_last_error(void);
int sumk_version(void);
/* number of HIP devices visible; <0 if the runtime cannot be initialised (the library never falls back to CPU) */
int sumk_device_count(void);

/* GEMM arithmetic selectors (see sumk_vasnet_opts.precision) */
#define SUMK_PRECISION_FP32 0
#define SUMK_PRECISION_BF16X3 1
#define SUMK_PRECISION_BF16X6 2   /* x = x1+x2+x3 exactly, 6 bf16 MFMAs per product: fp32-grade results */
#define SUMK_PRECISION_BF16 3     /* plain bf16 operands (one plane), ONE bf16 MFMA per product, fp32 accumulate: the mixed-precision
                                     TRAINING arithmetic of BASELINE config 2 (storage, master weights and optimiser stay fp32) */
#define SUMK_PRECISION_MAX 3

/* ------------------------------------------------------------------------------------------------ VASNet
 * Weights of summarizer/models/vasnet.py:56-66, each as stored by nn.Linear ([out][in]).
 * The SAME layer_norm (ln_w, ln_b) is applied twice (vasnet.py:137 and :143). */
typedef struct sumk_vasnet_weights {
  const float* Wk; const float* Wq; const float* Wv; /* (D,D)  K/Q/V.weight            vasnet.py:57-59 */
  const float* Wo;                                   /* (D,D)  attention_head_projection vasnet.py:60  */
  const float* W1; const float* b1;                  /* (D,D),(D)  k1                   vasnet.py:64   */
  const float* w2; const float* b2;                  /* (D),(1)    k2                   vasnet.py:65   */
  const float* ln_w; const float* ln_b;              /* (D)        layer_norm           vasnet.py:54   */
} sumk_vasnet_weights;

typedef struct sumk_vasnet_opts {
  float scale;          /* logits multiplier, 1/sqrt(D) by default      vasnet.py:34,119 */
  float eps;            /* LayerNorm epsilon                            vasnet.py:18,54  */
  int32_t ignore_self;  /* diag -> -inf                                 vasnet.py:121-122 */
  int32_t aperture;     /* -1 = global; w>=0: |i-j|>w -> -inf, and in-band logits with e*e==0 -> -inf
                           (tril*triu==0 quirk)                         vasnet.py:124-127 */
  /* training-mode dropout (vasnet.py:53,130,136,142): p=0 disables (eval).  Keep-masks are a pure function
     of (seed, site, element index) -- see DESIGN.md "Dropout" -- so backward regenerates them. */
  float dropout_p;
  uint64_t seed;
  /* arithmetic of the GEMMs: SUMK_PRECISION_FP32 (native fp32 MFMA, the default everywhere), SUMK_PRECISION_BF16X6 (operands
     split exactly into three bf16 planes, 6 bf16 MFMAs per product: fp32-grade results, DESIGN.md "bf16x6") or
     SUMK_PRECISION_BF16X3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate: ~2^-16 relative
     per product, scores stay within ~1e-5 of the fp32 path -- DESIGN.md "bf16x3").  Storage is fp32 either way. */
  int32_t precision;
  /* NULL, or a device word the kernels ADD to `seed` when they run: the masks of a call are then a function of (seed + *seed_dev),
     read at execution time.  For steps captured into a HIP graph: kernel arguments are frozen at capture, the device word is not --
     the caller increments it between replays (after the backward pass of a step) and every replay draws fresh masks. */
  const uint64_t* seed_dev;
  /* NULL, or the batch's problem tables prebuilt by sumk_vasnet_build_tables for the SAME (D, seq_off, training, precision): the call
     then launches no table-setup kernel (5-7 us per call: 5 % of a single-video forward).  The tables depend on the batch geometry
     only -- not on weights, inputs or the workspace -- so a caller builds them once per distinct batch (SeqBatch in kernels.py) and
     may share them between forward, backward and any number of calls ON ONE STREAM AT A TIME (they hold the tickets of the
     in-launch split-K launches, which every launch leaves zero).  Wrong tables give wrong results, not errors: device memory. */
  void* tables;
  /* NULL, or bf16(x) (n_rows, D) written by sumk_cast_f32_bf16 from the SAME x this call (forward and backward) is given: the mixed-precision
     training step (training != 0, SUMK_PRECISION_BF16 on the bf16-source kernels) then reads it instead of casting x again -- features
     are constant over the epochs of a run while the weights change every step, and x is 70 % of the elements the per-step cast kernel
     converts.  Ignored by every other mode.  Not allowed together with pos_table (which changes x in place). */
  const void* x16;
  /* NULL, or -- for inference in SUMK_PRECISION_BF16X6 / BF16X3 -- the "KB planes" (below) of the SAME x (n_rows x D; 3 resp. 2 planes)
     written by sumk_split_planes, and the weight-plane block written by sumk_vasnet_wplanes_build for the SAME weights and plane count.
     With both given (and D % 256 == 0, n_rows >= 256, no pos_table) the three row-wise GEMMs of the call -- K/Q/V projection, output
     projection, k1 (vasnet.py:114-116,132,138) -- run on the plane-aware wide kernel (csrc/gemm_pw.hip): no fp32 -> bf16 split inside any
     k-loop.  Same arithmetic as without them (the planes are the roundings the in-loop kernels make), faster.  x planes are constant per
     dataset, weight planes per weight change; stale planes give wrong results, not errors.  The per-video products follow: T <= 320 on the
     plane strips (csrc/attn_pw.hip), every video of the batch with T >= 1536 (round 6) as launches of the plane GEMM itself -- both need the
     larger workspace sumk_vasnet_workspace_bytes_for reports for the precision; anything else keeps the in-loop kernels between plane GEMMs. */
  const void* xplanes;
  const void* wplanes;
} sumk_vasnet_opts;


/* Bytes of workspace sumk_vasnet_forward / _backward need for this batch (seq_off_host has n_seq+1 entries). */
size_t sumk_vasnet_workspace_bytes(int32_t D, int32_t n_seq, const int32_t* seq_off_host, int32_t training);
/* The same for a known arithmetic: only the mixed-precision training step (training != 0, SUMK_PRECISION_BF16) needs the bf16
 * operand shadows at the end of the workspace (~ +25 %); every other mode is content with this smaller size.  Which kernels a step
 * runs on never depends on the workspace it is handed: a bf16 training step given the smaller workspace is refused
 * (SUMK_ERR_WORKSPACE), so a forward and a backward pass cannot end up on different paths. */
size_t sumk_vasnet_workspace_bytes_for(int32_t D, int32_t n_seq, const int32_t* seq_off_host, int32_t training, int32_t precision);

/* Weight-plane block of sumk_vasnet_opts::wplanes: bytes (0 = D or n_planes not eligible: D % 256, n_planes 2 or 3), and the build --
 * planes of [Wq; Wk; Wv] (Wvo != NULL: [Wq; Wk; Wvo], the folded path of sumk_vasnet_forward_folded), of Wo, of k1's weight with the
 * LayerNorm gain folded in (W1 diag(ln_w)), and the three per-column vectors of the fused tail.  256-byte aligned device buffer; redo
 * after every weight change. */
size_t sumk_vasnet_wplanes_bytes(int32_t D, int32_t n_planes);
int sumk_vasnet_wplanes_build(int32_t D, const sumk_vasnet_weights* w, const float* Wvo, int32_t n_planes, void* out, size_t out_bytes,
                              void* stream);

/* Problem tables of a batch, separately (sumk_vasnet_opts::tables): size, and the one-time build (the setup kernels of a call, run into
 * `tables` instead of the workspace).  256-byte aligned device buffer. */
size_t sumk_vasnet_tables_bytes(int32_t D, int32_t n_seq, const int32_t* seq_off_host);
int sumk_vasnet_build_tables(int32_t D, int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev, int32_t training,
                             int32_t precision, void* tables, size_t tables_bytes, void* stream);

/* Replaces VASNet.forward (vasnet.py:92-148) for a packed batch: x (n_rows,D) -> scores (n_rows,), sigmoid
 * outputs in (0,1).  If pos_rows != NULL, x[r,:] += pos_table[pos_rows[r],:] is applied IN PLACE first
 * (vasnet.py:106-112 mutates the caller's tensor the same way).  When training != 0 the workspace keeps the
 * intermediates sumk_vasnet_backward consumes. */
int sumk_vasnet_forward(float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                        const int32_t* seq_off_dev, const sumk_vasnet_weights* w,
                        const sumk_vasnet_opts* opts, const float* pos_table, const int32_t* pos_rows,
                        float* scores, void* workspace, size_t workspace_bytes, int32_t training,
                        void* stream);

/* Inference with the value and output projections folded: Wvo (D,D) = Wo . Wv, computed once per weight change by the caller
 * (sumk_gemm_nn(Wo, Wv, Wvo, D, D, D)).  (alpha V) Wo^T = alpha (X Wvo^T): the out-projection GEMM of vasnet.py:132 disappears
 * (18 % of the step's FLOPs); results equal sumk_vasnet_forward's up to fp32 re-association (~1e-6 on scores).  Opt-in
 * (VASNet(fold_vo=True)): the default path keeps the reference's operation order. */
int sumk_vasnet_forward_folded(float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                               const int32_t* seq_off_dev, const sumk_vasnet_weights* w, const float* Wvo,
                               const sumk_vasnet_opts* opts, const float* pos_table, const int32_t* pos_rows,
                               float* scores, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ BiLSTM
 * One bidirectional LSTM layer (torch.nn.LSTM semantics: gates i,f,g,o; h0=c0=0), as used by DSN
 * (summarizer/models/dsn.py:23-27,45) and sLSTM (summarizer/models/sumgan.py:27-32,43).
 * dir 0 = forward, dir 1 = reverse.  Output h (n_rows, 2H) = [h_fwd || h_rev] per frame. */
typedef struct sumk_lstm_layer_weights {
  const float* w_ih[2]; /* (4H, In)  weight_ih_l{k}[_reverse] */
  const float* w_hh[2]; /* (4H, H)   weight_hh_l{k}[_reverse] */
  const float* b_ih[2]; /* (4H)      bias_ih_l{k}[_reverse]   */
  const float* b_hh[2]; /* (4H)      bias_hh_l{k}[_reverse]   */
  /* Optional, all three or none (zero-initialise the struct otherwise): inference in SUMK_PRECISION_BF16X6 / BF16X3 then runs the
     input projection of dsn.py:45 / sumgan.py:43 on the plane-aware wide GEMM (csrc/gemm_pw.hip) -- x_planes: the "KB planes" of x
     (n_rows x In, sumk_split_planes); w_planes: the block sumk_bilstm_wplanes_build wrote for THESE weights (planes of
     [w_ih[0]; w_ih[1]] and the summed biases).  Needs 8 H % 256 == 0, In % 32 == 0, In >= 128, n_rows >= 1024. */
  const void* x_planes;
  const void* w_planes;
} sumk_lstm_layer_weights;

/* Weight-plane block of sumk_lstm_layer_weights::w_planes: bytes (0 = not eligible) and the build (redo after a weight change). */
size_t sumk_bilstm_wplanes_bytes(int32_t In, int32_t H, int32_t n_planes);
int sumk_bilstm_wplanes_build(int32_t In, int32_t H, const sumk_lstm_layer_weights* w, int32_t n_planes, void* out, size_t out_bytes, void* stream);

size_t sumk_bilstm_workspace_bytes(int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host, int32_t training);
/* precision: SUMK_PRECISION_FP32, or SUMK_PRECISION_BF16X3 / SUMK_PRECISION_BF16X6 for the input projection (and, for 256 < H <= 1024, the
 * recurrent product: inference only for BF16X6) in the split-bf16 arithmetics described at sumk_vasnet_opts.  With w->x_planes / w->w_planes given
 * (inference) the projection runs on operand planes in front of the recurrence; SUMK_LSTM_PROJ=1 in the environment computes it INSIDE the persistent
 * recurrence instead (H <= 256, In = 1024, <= 64 videos; csrc/lstm.hip, lstm_persist_proj_kernel -- slower since the recurrence's last speed-up, kept as
 * an A/B path).  The workspace of 256 < H <= 1024 holds the
 * forward recurrence's exchange buffer (1 MB per 64 videos): ask sumk_bilstm_workspace_bytes again after upgrading the library. */
int sumk_bilstm_layer_forward(const float* x, int32_t In, int32_t H, int32_t n_seq,
                              const int32_t* seq_off_host, const int32_t* seq_off_dev,
                              const sumk_lstm_layer_weights* w, float* h_out,
                              void* workspace, size_t workspace_bytes, int32_t training, int32_t precision, void* stream);

/* Per-frame head shared by DSN (dsn.py:34-36,46: Linear(2H,1)+Sigmoid) and sLSTM (sumgan.py:33-34,44-45):
 * scores[r] = sigmoid(dot(h[r,:F], w) + b[0]). */
int sumk_frame_head_forward(const float* h, int32_t n_rows, int32_t F, const float* w, const float* b,
                            float* scores, void* stream);

/* Gradients of sum_r dscores[r]*scores[r] w.r.t. every weight (same struct, non-const targets) and,
 * optionally (dx != NULL), the input.  Must follow a training-mode forward on the same workspace.
 * Gradients are ACCUMULATED into grads (caller zeroes them), matching autograd's .grad semantics. */
typedef struct sumk_vasnet_grads {
  float* Wk; float* Wq; float* Wv; float* Wo; float* W1; float* b1; float* w2; float* b2; float* ln_w; float* ln_b;
} sumk_vasnet_grads;
int sumk_vasnet_backward(const float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                         const int32_t* seq_off_dev, const sumk_vasnet_weights* w,
                         const sumk_vasnet_opts* opts, const float* dscores, const sumk_vasnet_grads* grads,
                         float* dx, void* workspace, size_t workspace_bytes, void* stream,
                         void* tail_grads_ready_event);
/* tail_grads_ready_event: NULL, or a hipEvent_t recorded on `stream` once the gradients of Wo, W1, b1, w2 and b2 are final
 * (before the attention backward and the Q/K/V weight gradients run): a data-parallel caller overlaps the all-reduce of
 * that part of its gradient bucket with the rest of this call (summarizer_amd/training.py: FlatAdam.reduce_tail_async). */

/* Synchronises `stream` and returns an error if a persistent recurrence kernel that used this workspace reported a
 * timed-out hand-off (outputs invalid).  Per-workspace word: valid until the next call reuses the workspace; the test-suite
 * (SUMK_CHECK=1) calls it after every recurrent layer.  Production code uses sumk_health_check below. */
int sumk_bilstm_check(const void* workspace, int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host,
                      int32_t training, int32_t after_backward, void* stream);
/* Sticky, workspace-free variant: synchronises `stream` and fails if ANY persistent recurrence kernel on the current device
 * timed out since the previous call (then resets the device-side word).  The Python trainers / scorers call it at the host
 * synchronisation points they already have (score D2H in Trainer.test / predict_dataset / StreamingScorer, per-epoch loss). */
int sumk_health_check(void* stream);

typedef struct sumk_lstm_layer_grads {
  float* w_ih[2]; float* w_hh[2]; float* b_ih[2]; float* b_hh[2];
  /* NULL (zero-initialise the struct), or a hipEvent_t the call RECORDS on `stream` once both bias gradients and the REVERSE direction's
     weight gradients (w_ih[1], w_hh[1]) are final -- before the forward direction's weight-gradient GEMMs run.  A data-parallel caller whose
     gradient bucket ends with [reverse direction | head] (the parameter order of DSN, dsn.py:23-36) can all-reduce that half on a side stream
     under the remaining GEMMs (training.FlatAdam.reduce_tail_async).  With the event set the two directions' dW_ih run as two launches
     (reverse first) instead of one. */
  void* tail_ready_event;
} sumk_lstm_layer_grads;
/* dh_out (n_rows,2H) -> accumulates weight grads; dx (n_rows,In) written if non-NULL.  Needs the workspace of
 * a training-mode sumk_bilstm_layer_forward and that call's h_out. */
int sumk_bilstm_layer_backward(const float* x, const float* h_out, const float* dh_out, int32_t In, int32_t H,
                               int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev,
                               const sumk_lstm_layer_weights* w, const sumk_lstm_layer_grads* grads, float* dx,
                               void* workspace, size_t workspace_bytes, int32_t precision, void* stream);

/* ------------------------------------------------------------------------------------------------ GRU cell
 * The reference's optional `DSN(cell="gru")` (dsn.py:28-33, nn.GRU).  One fused element-wise step for B sequences, gate order
 * r, z, n (torch): gx = x_t W_ih^T + b_ih and gh = h_prev W_hh^T + b_hh are (B, 3H) (the host computes them with
 * sumk_linear_forward); mask (B) or NULL: 0 = the sequence has ended (h passes through, no gradient).  rzn (B, 3H) keeps the
 * gate values for sumk_gru_cell_backward, which returns d(gx), d(gh) and the DIRECT part of d(h_prev) (the caller adds dgh . W_hh).
 * Functional path, not a tuned one: DSNTrainer never builds the GRU cell. */
int sumk_gru_cell_forward(const float* gx, const float* gh, const float* h_prev, const float* mask, float* h_out, float* rzn,
                          int32_t B, int32_t H, void* stream);
int sumk_gru_cell_backward(const float* dh, const float* rzn, const float* gh, const float* h_prev, const float* mask,
                           float* dgx, float* dgh, float* dh_prev, int32_t B, int32_t H, void* stream);

/* ------------------------------------------------------------------------------------------------ unidirectional LSTM layer
 * One forward-running nn.LSTM(bidirectional=False) layer with an optional initial state and the final state as an output:
 * the layers of SumGAN's eLSTM / dLSTM / cLSTM (summarizer/models/sumgan.py:48-115,185-210).  Packed batch as above;
 * h0 / c0 / h_last / c_last / dh_last / dc_last / dh0 / dc0 are (n_seq, H) and may be NULL (zeros / not wanted). */
typedef struct sumk_lstm_dir_weights { const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh; } sumk_lstm_dir_weights;
typedef struct sumk_lstm_dir_grads { float* w_ih; float* w_hh; float* b_ih; float* b_hh; } sumk_lstm_dir_grads;
size_t sumk_lstm_workspace_bytes(int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host, int32_t training);
int sumk_lstm_layer_forward(const float* x, int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host,
                            const int32_t* seq_off_dev, const sumk_lstm_dir_weights* w, const float* h0, const float* c0,
                            float* h_out, float* h_last, float* c_last, void* workspace, size_t workspace_bytes,
                            int32_t training, int32_t precision, void* stream);
/* Gradients ACCUMULATE into grads; dx (n_rows, In) written if non-NULL; dh_out may be NULL (no per-step upstream gradient).
 * Needs the workspace of a training-mode sumk_lstm_layer_forward, that call's h_out and the same c0. */
int sumk_lstm_layer_backward(const float* x, const float* h_out, const float* dh_out, const float* dh_last, const float* dc_last,
                             int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev,
                             const sumk_lstm_dir_weights* w, const float* c0, const sumk_lstm_dir_grads* grads, float* dx,
                             float* dh0, float* dc0, void* workspace, size_t workspace_bytes, int32_t precision, void* stream);

/* Step-wise decoder = SumGAN's dLSTM (sumgan.py:74-115): an n_layers-deep forward-running LSTM (input size == H) whose
 * input at step t is its own top-layer output of step t-1 (zeros at t = 0), started from (h0, c0) (n_layers, n_seq, H) or
 * NULL.  out (n_rows, H) = top-layer outputs in time order (the module flips them, sumgan.py:114).  w / grads: n_layers
 * entries.  backward: needs the forward's workspace and out; grads ACCUMULATE; dh0 / dc0 (n_layers, n_seq, H) or NULL. */
size_t sumk_lstm_decoder_workspace_bytes(int32_t H, int32_t n_layers, int32_t n_seq, const int32_t* seq_off_host);
int sumk_lstm_decoder_forward(int32_t H, int32_t n_layers, int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev,
                              const sumk_lstm_dir_weights* w, const float* h0, const float* c0, float* out, void* workspace,
                              size_t workspace_bytes, void* stream);
int sumk_lstm_decoder_backward(int32_t H, int32_t n_layers, int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev,
                               const sumk_lstm_dir_weights* w, const float* c0, const float* out, const float* dout,
                               const sumk_lstm_dir_grads* grads, float* dh0, float* dc0, void* workspace, size_t workspace_bytes,
                               void* stream);

/* Dense layer y (M,N) = x (M,K) w^T + b for the small Linear layers around the LSTM stacks (sumgan.py:58-59,84); w is (N,K)
 * as in nn.Linear, b may be NULL.  backward: dx (M,K) written if non-NULL, dw (N,K) / db (N) ACCUMULATED if non-NULL. */
size_t sumk_linear_workspace_bytes(int32_t N, int32_t K);
int sumk_linear_forward(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t N, int32_t K,
                        void* workspace, size_t workspace_bytes, int32_t precision, void* stream);
int sumk_linear_backward(const float* x, const float* w, const float* dy, int32_t M, int32_t N, int32_t K, float* dx, float* dw,
                         float* db, void* workspace, size_t workspace_bytes, int32_t precision, void* stream);

/* dh[r,:] = ds[r]*s(1-s)*w ; dw += sum_r ds*s(1-s)*h[r,:] ; db += sum_r ds*s(1-s) */
size_t sumk_frame_head_workspace_bytes(int32_t F);
int sumk_frame_head_backward(const float* h, const float* scores, const float* dscores, int32_t n_rows,
                             int32_t F, const float* w, float* dh, float* dw, float* db, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ Transformer scorer
 * The reference's Transformer-encoder scorer (summarizer/models/transformer.py:74-103): n_layers stock
 * nn.TransformerEncoderLayer (post-norm, ReLU, dim_feedforward F), final norm = the SHARED layer_norm (also applied after
 * k1), optional extra residual (more_residuals), k1 + ReLU + LN + k2 + sigmoid.  Weights as torch stores them. */
typedef struct sumk_tf_layer_weights {
  const float* in_proj_w;  const float* in_proj_b;    /* (3D,D),(3D)  self_attn.in_proj_{weight,bias} */
  const float* out_proj_w; const float* out_proj_b;   /* (D,D),(D)    self_attn.out_proj              */
  const float* lin1_w; const float* lin1_b;           /* (F,D),(F)    linear1                          */
  const float* lin2_w; const float* lin2_b;           /* (D,F),(D)    linear2                          */
  const float* norm1_w; const float* norm1_b; const float* norm2_w; const float* norm2_b;   /* (D) */
} sumk_tf_layer_weights;
typedef struct sumk_tf_head_weights {
  const float* ln_w; const float* ln_b;               /* (D) layer_norm (== transformer_encoder.norm)  transformer.py:47,50,100 */
  const float* k1_w; const float* k1_b;               /* (D,D),(D)                                     transformer.py:52 */
  const float* k2_w; const float* k2_b;               /* (D),(1)                                       transformer.py:53 */
} sumk_tf_head_weights;
typedef struct sumk_tf_opts {
  float layer_eps;        /* LayerNorm eps inside the encoder layers (torch default 1e-5)                       */
  float final_eps;        /* eps of the shared layer_norm (constructor argument `epsilon`)    transformer.py:47  */
  int32_t more_residuals; /* encoder_out += x                                                  transformer.py:94  */
  float layer_dropout_p;  /* training: dropout inside the encoder layers (0.1)                 transformer.py:49  */
  float head_dropout_p;   /* training: dropout after relu(k1) (0.5)                            transformer.py:46,99 */
  uint64_t seed;          /* keep-masks are a pure function of (seed, site, element), as for VASNet               */
  int32_t precision;      /* SUMK_PRECISION_*, as in sumk_vasnet_opts                                             */
  const void* wplanes;    /* inference in BF16X6 / BF16X3: the weights as bf16 planes (sumk_transformer_wplanes_build, 256-byte
                             aligned) -> every projection of the stack runs on the plane GEMM; NULL: the in-loop split kernels */
} sumk_tf_opts;
size_t sumk_transformer_workspace_bytes(int32_t D, int32_t F, int32_t n_heads, int32_t n_layers, int32_t n_seq,
                                        const int32_t* seq_off_host, int32_t training);
/* the same plus the two activation-plane buffers of the plane path (inference, BF16X6 / BF16X3, D and F multiples of 256,
 * at least 128 frames); equal to sumk_transformer_workspace_bytes otherwise */
size_t sumk_transformer_workspace_bytes_for(int32_t D, int32_t F, int32_t n_heads, int32_t n_layers, int32_t n_seq,
                                            const int32_t* seq_off_host, int32_t training, int32_t precision);
/* Weight planes of the encoder stack and k1 (transformer.py:49-53: in_proj, out_proj, linear1, linear2 of every layer; k1) for
 * n_planes = 3 (BF16X6) or 2 (BF16X3); 0 bytes = not eligible.  Rebuild after every weight change. */
size_t sumk_transformer_wplanes_bytes(int32_t D, int32_t F, int32_t n_layers, int32_t n_planes);
int sumk_transformer_wplanes_build(int32_t D, int32_t F, int32_t n_layers, const sumk_tf_layer_weights* layers,
                                   const sumk_tf_head_weights* head, int32_t n_planes, void* out, size_t out_bytes,
                                   void* stream);
/* x (n_rows,D) packed -> scores (n_rows,).  pos_table/pos_rows as in sumk_vasnet_forward (in-place add, transformer.py:83-89).
 * training != 0 keeps every layer's activations in the workspace for sumk_transformer_backward. */
int sumk_transformer_forward(float* x, int32_t D, int32_t F, int32_t n_heads, int32_t n_layers, int32_t n_seq,
                             const int32_t* seq_off_host, const int32_t* seq_off_dev,
                             const sumk_tf_layer_weights* layers, const sumk_tf_head_weights* head,
                             const sumk_tf_opts* opts, const float* pos_table, const int32_t* pos_rows, float* scores,
                             void* workspace, size_t workspace_bytes, int32_t training, void* stream);
/* Gradient targets, same shapes as the weights; ACCUMULATED into (caller zeroes them). */
typedef struct sumk_tf_layer_grads {
  float* in_proj_w; float* in_proj_b; float* out_proj_w; float* out_proj_b; float* lin1_w; float* lin1_b;
  float* lin2_w; float* lin2_b; float* norm1_w; float* norm1_b; float* norm2_w; float* norm2_b;
} sumk_tf_layer_grads;
typedef struct sumk_tf_head_grads { float* ln_w; float* ln_b; float* k1_w; float* k1_b; float* k2_w; float* k2_b; } sumk_tf_head_grads;
/* Backward of sum_r dscores[r]*scores[r] (loss.backward(), transformer.py:163); must follow a training-mode forward on the
 * same workspace.  dx (n_rows,D) is written if non-NULL. */
int sumk_transformer_backward(const float* x, int32_t D, int32_t F, int32_t n_heads, int32_t n_layers, int32_t n_seq,
                              const int32_t* seq_off_host, const int32_t* seq_off_dev,
                              const sumk_tf_layer_weights* layers, const sumk_tf_head_weights* head,
                              const sumk_tf_opts* opts, const float* dscores, const sumk_tf_layer_grads* layer_grads,
                              const sumk_tf_head_grads* head_grads, float* dx, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ------------------------------------------------------------------------------------------------ DSN reward
 * DSNTrainer.compute_reward (dsn.py:185-236) for E episodes of one or more packed videos:
 * actions (E, n_rows) of 0/1 floats -> reward (E, n_seq).  Zero picks -> 0 (dsn.py:199-203); one pick ->
 * r_div = 0 (dsn.py:211-214) and r_rep from that pick (the reference itself raises IndexError there). */
size_t sumk_dsn_reward_workspace_bytes(int32_t D, int32_t n_seq, const int32_t* seq_off_host, int32_t n_episodes);
int sumk_dsn_reward(const float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                    const int32_t* seq_off_dev, const float* actions, int32_t n_episodes, int32_t far_sim,
                    int32_t temp_dist_thre, float* reward, void* workspace, size_t workspace_bytes, void* stream);
/* The loss glue of DSNTrainer.train for a packed batch (dsn.py:113-140), per video v:
 *   loss_per_video[v] = [ beta (mean_t p_t - eps_target)^2 - sum_e (rewards[e,v] - base[v]) mean_t log P(actions[e,t] | p_t) ] / E
 * with torch.distributions.Bernoulli's log_prob (probabilities clamped to [eps, 1 - eps], eps = float32 epsilon), and its
 * gradient w.r.t. the probabilities given dloss_per_video.  actions (E, n_rows), rewards (E, n_seq), base (n_seq); mean_probs
 * (n_seq) is written by the forward and read by the backward.  E <= 16.  The supervised BCE term of `sup=True` stays outside. */
int sumk_dsn_policy_loss_forward(const float* probs, const float* actions, const float* rewards, const float* base,
                                 int32_t n_seq, int32_t n_rows, const int32_t* seq_off_dev, int32_t n_episodes,
                                 float beta, float eps_target, float* loss_per_video, float* mean_probs, void* stream);
int sumk_dsn_policy_loss_backward(const float* probs, const float* actions, const float* rewards, const float* base,
                                  const float* mean_probs, const float* dloss_per_video, int32_t n_seq, int32_t n_rows,
                                  const int32_t* seq_off_dev, int32_t n_episodes, float beta, float eps_target,
                                  float* dprobs, void* stream);
/* nn.MSELoss per video of a packed batch (vasnet.py:209, transformer.py:161): mse_per_video[v] = mean_t (scores_t - target_t)^2,
 * and dscores_t = 2 (scores_t - target_t) / T_v * dmse_per_video[v]. */
int sumk_segment_mse_forward(const float* scores, const float* target, int32_t n_seq, const int32_t* seq_off_dev,
                             float* mse_per_video, void* stream);
int sumk_segment_mse_backward(const float* scores, const float* target, const float* dmse_per_video, int32_t n_seq,
                              const int32_t* seq_off_dev, float* dscores, void* stream);
/* The trainers' step loss in one launch each way (vasnet.py:209-212: mean over the videos of a step of nn.MSELoss per video):
 * loss[0] = scale * sum_v mse_per_video[v] (scale = 1 / videos of the step; the per-video values are written too and added in video order),
 * dscores_t = 2 (scores_t - target_t) / T_v * scale * dloss[0] (dloss: the device scalar autograd hands the backward).
 * `ticket`: one device word the caller zeroes ONCE and keeps for this batch; every launch leaves it zero (a block per video, the last
 * arriver adds); one launch at a time per word. */
int sumk_segment_mse_mean_forward(const float* scores, const float* target, int32_t n_seq, const int32_t* seq_off_dev, float scale,
                                  float* mse_per_video, float* loss, uint32_t* ticket, void* stream);
int sumk_segment_mse_mean_backward(const float* scores, const float* target, const float* dloss, float scale, int32_t n_seq,
                                   const int32_t* seq_off_dev, float* dscores, void* stream);

/* ------------------------------------------------------------------------------------------------ optimiser
 * torch.optim.Adam(lr, betas, eps, weight_decay) exactly as the trainers construct it (vasnet.py:181,
 * dsn.py:70-73): L2 weight decay folded into the gradient, bias-corrected moments.  One flat launch over
 * n elements; `step` is the 1-based step count.  grad_scale multiplies the gradient first (used for
 * clip_grad_norm_, dsn.py:145, and for the 1/world_size of a data-parallel average). */
int sumk_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                   float grad_scale, void* stream);
/* The same step with NOTHING on the host: `state` is a 16-byte device block the caller zeroes once -- state[0] (int32) counts
 * the optimiser steps and is incremented by the call, state[1..3] are scratch -- so the bias correction follows a counter that
 * lives on the device, and when `sumsq` (device scalar from sumk_sumsq: the squared L2 norm of the UNscaled gradient) is
 * given, torch.nn.utils.clip_grad_norm_(params, max_norm) (dsn.py:145) is folded in without reading the norm back.  No host
 * synchronisation, captures into a HIP graph and replays with the right step count.  Same arithmetic as sumk_adam_step. */
int sumk_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int32_t* state,
                       float grad_scale, const float* sumsq, float max_norm, void* stream);
/* The same, and `grad` is left ZERO: the next step's optimizer.zero_grad() (vasnet.py:210, dsn.py:143) folded into the pass that reads
 * the gradient last -- one 21 MB fill launch less per step of a captured (HIP graph) training step. */
int sumk_adam_step_dev_zero_grad(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, int32_t* state,
                                 float grad_scale, const float* sumsq, float max_norm, void* stream);
/* out[0] += sum of squares of a flat buffer (for clip_grad_norm_), deterministic two-stage reduction.
 * workspace: sumk_sumsq_workspace_bytes() bytes of device scratch. */
size_t sumk_sumsq_workspace_bytes(void);
int sumk_sumsq(const float* v, int64_t n, float* out, void* workspace, void* stream);
/* Flat fp32 <-> bf16 casts (round to nearest even): in the mixed-precision training mode (SUMK_PRECISION_BF16) the gradient
 * bucket crosses the data-parallel all-reduce as bf16 -- 10.5 MB instead of 21 MB for VASNet -- and comes back into the fp32
 * bucket the optimiser reads.  No reference counterpart (the reference has no distributed code). */
int sumk_cast_f32_bf16(const float* src, void* dst_bf16, int64_t n, void* stream);
int sumk_cast_bf16_f32(const void* src_bf16, float* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------ generic
 * fp32 MFMA GEMM (the dominant kernel), exposed for tests and for bench.py's roofline probe:
 * C(M,N) = A(M,K) * B^T  with B given as (N,K) row-major ("NT", both operands K-contiguous). */
int sumk_gemm_nt(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream);
/* any of the three products (layout 0 = NT, 1 = NN, 2 = TN, operands as above) in the selected arithmetic
 * (SUMK_PRECISION_FP32 | SUMK_PRECISION_BF16X3) */
int sumk_gemm_prec(int32_t layout, const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int32_t precision,
                   void* stream);
/* The same three products with bf16 OPERANDS in device memory (fp32 accumulate and output): the GEMM of the mixed-precision
 * training step (csrc/gemm_b16.hip), exposed for tests and probes.  K-contiguous operands need K % 64 == 0, the others rows of
 * whole 16-byte chunks.  workspace == NULL: C = product.  workspace given (layout 2 only, 256-byte aligned, >= 8192 + 4 M N
 * bytes): deterministic split-K over a long K and C += product -- the weight-gradient form of vasnet.py:193-212's backward. */
int sumk_gemm_bf16src(int32_t layout, const void* A_bf16, const void* B_bf16, float* C, int32_t M, int32_t N, int32_t K,
                      void* workspace, size_t workspace_bytes, void* stream);
/* ---- "KB planes": the operand format of the plane-aware wide GEMM behind SUMK_PRECISION_BF16X6 / BF16X3 scoring (csrc/gemm_pw.hip).
 * An fp32 matrix (rows x K, K % 16 == 0) as n_planes = 3 (x = x1 + x2 + x3 exactly) or 2 (hi + lo) bf16 planes, k-blocked:
 *   byte offset of (row, k, plane p) = (((k / 16) * n_planes + p) * 2 + (k / 8) % 2) * 16 * pitch + row * 16 + (k % 8) * 2,
 *   pitch = rows rounded up to 64.   Written once per dataset (x), per weight change (weights) or by the producing kernel's epilogue;
 * no reference counterpart (the reference multiplies fp32 tensors with torch.matmul, vasnet.py:114-131).
 * sumk_planes_bytes: device bytes of one plane array (0 = bad arguments).  sumk_split_planes: fp32 (rows x K, leading dimension ld)
 * -> planes.  sumk_gemm_planes (tests / bench probe): C(M,N) fp32 = A . B^T from two plane arrays built for a_rows / b_rows rows
 * (N % 256 == 0, K % 32 == 0, K >= 128); variant selects a schedule variant of the probe (0 = the product's). */
size_t sumk_planes_bytes(int64_t rows, int32_t K, int32_t n_planes);
int sumk_split_planes(const float* src, int64_t rows, int32_t K, int32_t ld, int32_t n_planes, void* planes, void* stream);
int sumk_gemm_planes(const void* A_planes, int64_t a_rows, const void* B_planes, int64_t b_rows, float* C, int32_t M, int32_t N, int32_t K,
                     int32_t n_planes, int32_t variant, void* stream);
/* Per-video attention on planes (csrc/attn_pw.hip; tests / probes -- sumk_vasnet_forward runs the same two launches): from the KB planes
 * of the (rows x 3 D) matrix [Q | K | V], per video s (frames seq_off[s] .. seq_off[s + 1], at most 320):
 *   alpha = softmax(mask(Q K^T * scale)) (vasnet.py:118-129) -> alpha_planes (rows = frames, k = key index inside the video, row pitch of
 *   the input planes; sumk_attn_planes_alpha_bytes) and, when E != NULL, fp32 alpha as (T x roundup4(T)) blocks back to back;
 *   ctx_planes != NULL: context = alpha V (vasnet.py:131) as KB planes of the (rows x D) matrix.  Synchronises the stream (test entry). */
size_t sumk_attn_planes_alpha_bytes(int64_t rows, int32_t t_max, int32_t n_planes);
int sumk_attn_planes(const void* qkv_planes, int64_t rows, int32_t D, int32_t n_planes, int32_t n_seq, const int32_t* seq_off_host,
                     float scale, int32_t ignore_self, int32_t aperture, float* E, void* alpha_planes, void* ctx_planes, void* stream);
/* C(M,N) = A(M,K) * B(K,N) */
int sumk_gemm_nn(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream);
/* C(M,N) = A^T * B with A given as (K,M), B as (K,N) */
int sumk_gemm_tn(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream);

/* ------------------------------------------------------------------------------------------------ shot selection
 * HOST function (no GPU work).  Replaces knapsack_ortools (summarizer/utils/knapsack.py:5-23): maximise
 * sum(values[i]) subject to sum(weights[i]) <= capacity; selected[i] = 1 for chosen items.  values/weights are the
 * integers the reference builds (knapsack.py:11-15: int(score*1000), int(n_frames_of_segment)). */
int sumk_knapsack_dp(const int64_t* values, const int64_t* weights, int32_t n_items, int64_t capacity,
                     uint8_t* selected);

/* HOST function, natively threaded (no GPU work): the evaluation tail of `Trainer.test` for a batch of videos --
 * upsample (summarizer/utils/eval.py:15-35), generate_summary (eval.py:74-123), evaluate_summary (eval.py:125-165) and
 * evaluate_scores with metric "spearmanr" (eval.py:49-72).  Machine summaries and F-scores are bit-identical to the numpy
 * implementation (numpy's float32 pairwise summation is reproduced); the rank correlation is float64, equal to ~1e-15. */
typedef struct sumk_eval_video {
  const float* scores; int32_t n_steps;        /* in: per-step importance scores                                      */
  const int32_t* picks; int32_t n_picks;       /* in: positions of the sampled frames                                 */
  int32_t n_frames;
  const int32_t* cps; const int32_t* nfps; int32_t n_segs;   /* in: change points (n_segs,2), frames per segment; n_segs 0: skip summary */
  const float* user_summary; int32_t n_users;  /* in: (n_users, n_frames), > 0 means selected                         */
  const double* user_ranks;                    /* in: (n_users, n_frames) average ranks of -user_scores, or NULL: skip correlation */
  float* machine_summary;                      /* out (optional): (sum nfps) 0/1 floats                               */
  int32_t summary_len;                         /* out: sum nfps                                                        */
  double corr, f_avg, f_max;                   /* out: mean Spearman over annotators; mean / max F-score (NaN if skipped) */
  const float* seg_means;                      /* in (optional): (n_segs) float32 segment means already taken on the DEVICE (sumk_eval_device):
                                                  upsampling, the means and the correlation are then skipped (corr is left as given) */
} sumk_eval_video;
/* method: 0 = knapsack (sumk_knapsack_dp), 1 = rank.  n_threads <= 0: min(16, hardware threads). */
int sumk_eval_videos(sumk_eval_video* videos, int32_t n_videos, double proportion, int32_t method, int32_t n_threads);

/* Device-side part of the same tail: the batch's scores stay in HBM; one launch (a block per video)
 * expands them to frame scores, takes the float32 segment means of eval.py:91-94 (numpy's pairwise summation reproduced, so
 * `int(mean * 1000)` and with it the key-shot selection are unchanged) and the mean Spearman correlation of eval.py:49-72.
 * The caller copies seg_means / corr back (one small D2H) and finishes on the host with sumk_eval_videos(seg_means given).
 * All pointers inside the descriptors are DEVICE pointers; picks must be ascending, n_picks <= 4096, n_users <= 32.
 * user_mean[u] / user_ssq[u] = mean and sum of squared deviations of annotator u's ranks (constants of the video). */
typedef struct sumk_eval_dev_video {
  const int32_t* picks; int32_t n_picks; int32_t n_frames; int32_t n_steps;
  int32_t row0;                                /* first row of this video in the packed score vector                   */
  int32_t frame0;                              /* offset of this video's frames in frame_scratch                       */
  const int32_t* cps; int32_t n_segs; int32_t seg0;   /* (n_segs,2) change points; offset of its segments in seg_means */
  const double* user_ranks; const double* user_mean; const double* user_ssq; int32_t n_users;   /* NULL / 0: no correlation (NaN) */
} sumk_eval_dev_video;
int sumk_eval_device(const float* scores_dev, const sumk_eval_dev_video* videos_dev, int32_t n_videos, float* frame_scratch_dev,
                     float* seg_means_dev, double* corr_dev, void* stream);
/* The same tail as two calls, so that the host's key-shot selection (which needs the segment means only) runs UNDER the correlation:
 * _segments = upsample + segment means (a block per video); _spearman = the correlation spread over 8 blocks per video + a final kernel
 * that adds their partial sums in order (scratch: sumk_eval_device_spearman_scratch_bytes(n_videos), 8-byte aligned).  The same
 * descriptors; results equal sumk_eval_device's (segment means bit for bit, correlations to float64 re-association). */
int sumk_eval_device_segments(const float* scores_dev, const sumk_eval_dev_video* videos_dev, int32_t n_videos, float* frame_scratch_dev,
                              float* seg_means_dev, void* stream);
size_t sumk_eval_device_spearman_scratch_bytes(int32_t n_videos);
int sumk_eval_device_spearman(const float* scores_dev, const sumk_eval_dev_video* videos_dev, int32_t n_videos, double* scratch_dev,
                              double* corr_dev, void* stream);

/* ------------------------------------------------------------------------------------------------ data-parallel exchange (RCCL)
 * The gradient all-reduce of data-parallel training as a library call: SUM, in place, over one flat bucket, on the caller's
 * HIP stream (one collective per optimiser step; the reference has no distributed code).  Bootstrap:
 * rank 0 obtains a 128-byte id (sumk_comm_unique_id) and ships it to every rank by any channel it has; every rank then calls
 * sumk_comm_init with its rank -- the communicator binds the CURRENT HIP device.  dtype: 0 = fp32, 1 = bf16 (the
 * mixed-precision mode's gradient bucket).  RCCL is dlopen'ed on first use (never linked): single-GPU users never load it.
 * summarizer_amd/training.py uses torch.distributed by default and this path under SUMK_RCCL_DIRECT=1. */
int sumk_comm_unique_id(uint8_t* id128);
int sumk_comm_init(const uint8_t* id128, int32_t rank, int32_t world, void** comm_out);
int sumk_allreduce_flat(void* comm, void* buf, int64_t n, int32_t dtype, void* stream);
int sumk_comm_destroy(void* comm);

/* ------------------------------------------------------------------------------------------------ feature ingest (host)
 * Packs the (n_rows[i], D) fp32 feature matrices srcs[i] back to back into dst (normally a pinned staging buffer that ONE
 * H2D copy then ships) with a pool of memcpy threads; replaces the per-video host->device uploads of the reference's loops
 * (summarizer/models/__init__.py:47-51, vasnet.py:194-205, dsn.py:98-110).  n_threads <= 0: min(16, hardware threads). */
int sumk_pack_rows(float* dst, const float* const* srcs, const int32_t* n_rows, int32_t n_videos, int32_t D, int32_t n_threads);
/* The same packing with fp32 -> bf16 (round to nearest even, NaN kept) folded into the copy: half the staging and H2D bytes for
 * link-bound hosts ("on-the-fly bf16 conversion").  LOSSY -- the scorer then sees bf16(features);
 * the device side widens with sumk_cast_bf16_f32.  Opt-in: summarizer_amd.ingest.StreamingScorer(stage_dtype="bf16"). */
int sumk_pack_rows_bf16(uint16_t* dst_bf16, const float* const* srcs, const int32_t* n_rows, int32_t n_videos, int32_t D, int32_t n_threads);

/* Per-kernel timing for bench.py's roofline object: launches tagged t are bracketed with hipEvents ON THE LAUNCH STREAM while
 * bit t of the mask passed to sumk_prof_enable is set (0 = off).  sumk_prof_read synchronises and returns the sums. */
#define SUMK_PROF_GEMM_QKV 0
#define SUMK_PROF_GEMM_ALL 1
#define SUMK_PROF_LSTM_REC 2
#define SUMK_PROF_GEMM_QKT 3     /* per-video Q.K^T launches of sumk_vasnet_forward                     */
#define SUMK_PROF_GEMM_PV 4      /* per-video alpha.V launches                                          */
#define SUMK_PROF_GEMM_OPROJ 5   /* output projection (+ residual)                                      */
#define SUMK_PROF_GEMM_K1 6      /* k1 (+ bias, ReLU)                                                   */
#define SUMK_PROF_NTAGS 8
/* The small-batch GEMM form by itself (tests, probes): C = epilogue(A . B) for ONE exact-fp32 problem on 64x64 tiles whose K is cut
 * into `slices` (1 .. 8) slices INSIDE the launch -- every (tile, slice) block stores a partial tile, the last one to arrive adds
 * them in slice order and runs the epilogue (csrc/gemm_lean.hip, SK instances; what sumk_vasnet_forward / _backward launch for batches
 * of <= 1024 frames).  layout 0 NT (A (M,K), B (N,K)), 1 NN (B (K,N)), 2 TN (A (K,M), B (K,N)); epilogue 0 C = alpha acc, 1 acc + R,
 * 2 relu(acc + bias[col]), 4 C += alpha acc.  workspace: sumk_gemm_splitk_workspace_bytes(M, N, slices), 256-byte aligned. */
size_t sumk_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t slices);
int sumk_gemm_splitk(int32_t layout, const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                     int32_t ldc, int32_t slices, int32_t epilogue, const float* R, int32_t ldr, const float* bias, float alpha,
                     void* workspace, size_t workspace_bytes, void* stream);
int sumk_prof_enable(int32_t tag_mask);
int sumk_prof_read(int32_t tag, double* total_ms, int64_t* launches, int32_t reset);
/* Diagnostic (process started with SUMK_GEMM_DBG=2): the last GEMM launch's in-kernel shader-cycle stamps, 4 values per block
 * {whole block, k-loops, epilogues, tiles}; synchronises the device.  Returns SUMK_ERR_ARG when stamping is off. */
int sumk_prof_gemm_stamps(uint64_t* out, int32_t n_blocks);

/* Measurement utility (no reference counterpart; bench.py's `mfma_sustained`): the rate the matrix pipes of the current device SUSTAIN with nothing but
 * MFMAs in flight -- one workgroup of eight waves per CU, four independent accumulators per wave, operands in registers, no memory traffic.  One warm-up
 * launch, one timed launch of `iters` x 16 MFMAs per wave (HIP events on `stream`; synchronises).  kind 0: v_mfma_f32_32x32x16_bf16, 1:
 * v_mfma_f32_32x32x2_f32, 2: v_mfma_f32_16x16x32_bf16; + 4: every MFMA of an iteration reads its own pseudo-random operand registers (the operand buses
 * toggle as under real data: the rate a real kernel can hope for) instead of one constant pair.  *tflops = dense TFLOP/s over the whole chip, *seconds
 * (may be null) = the timed launch. */
int sumk_probe_mfma_rate(int32_t kind, int32_t iters, double* tflops, double* seconds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUMK_H */
