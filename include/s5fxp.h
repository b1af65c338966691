/*
 * s5fxp.h -- C ABI of libs5fxp.so: MI355X (gfx950) kernels for the fixed-point S5 inference
 * path of stevenabreu7/SparseRNNs.
 *
 * This is the drop-in boundary.  The reference is pure Python/JAX and has no
 * FFI of its own, so every entry point below names the reference function it replaces
 * (file:line into /root/reference/sparseRNNs/).  Conventions:
 *   - plain pointers and sizes only; device pointers are raw HIP device addresses
 *     (e.g. torch.Tensor.data_ptr()), `stream` is a hipStream_t passed as void*;
 *   - the caller owns every device buffer; nothing here allocates device memory
 *     (sizes come from the *_bytes query functions);
 *   - every call is asynchronous on `stream` and re-entrant (no global mutable state; the S5FXP_* environment switches
 *     listed under "Environment" below are read once, by s5fxp_model_create, into the handle it returns);
 *   - return value: S5FXP_OK or a negative error code; no exceptions cross the ABI;
 *   - errors that the reference raises from data-dependent values (a negative shift in
 *     fxp_mul's "compute_best", fxparray.py:619-621) are reported through a device status
 *     word, because they are only known on the device.
 * All tensors are int32, row-major, value = data / 2^exp, exactly as fxparray.py:33-38,157.
 */
#ifndef S5FXP_H
#define S5FXP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S5FXP_VERSION 102

enum {
    S5FXP_OK = 0,
    S5FXP_EBADARG = -1,      /* null pointer, non-positive size, unsupported shape */
    S5FXP_ENEGSHIFT = -2,    /* negative/out-of-range shift: the reference raises ValueError (fxparray.py:619-621)
                                or hands XLA an undefined shift (fxparray.py:664-667) */
    S5FXP_EUNSUPPORTED = -3, /* configuration the reference asserts against (fxpmodel.py:429-434,995-999) */
    S5FXP_EHIP = -4,         /* a HIP runtime call failed */
    S5FXP_EWORKSPACE = -5    /* workspace or blob too small */
};

/* rounding modes, fxparray.py:13-17 */
enum { S5FXP_FLOOR = 0, S5FXP_CEIL = 1, S5FXP_ROUND = 2 };

int s5fxp_version(void);
const char *s5fxp_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Op level: one entry point per FxpArray primitive the model uses.
 * ---------------------------------------------------------------------------------------- */

/* fxp_from_fp, fxparray.py:287-307: y = clip(round_mode(x * 2^exp)), x float32. */
int s5fxp_from_fp(const float *x, int32_t *y, int64_t n, int bits, int exp, int round_mode, void *stream);

/* FxpArray.to_float, fxparray.py:72-73. */
int s5fxp_to_float(const int32_t *x, float *y, int64_t n, int exp, void *stream);

/* fxp_change_cfg / fxp_change_exp / fxp_clip, fxparray.py:232-271,310-326,346-357 (signed, FLOOR).
 * Pass new_bits == bits for a pure change_exp; new_exp == exp && new_bits < bits for a pure clip. */
int s5fxp_change_cfg(const int32_t *x, int32_t *y, int64_t n, int bits, int exp, int new_bits, int new_exp,
                     void *stream);

/* fxp_matmul (+ the bias fxp_add of FxpDense.forward), fxparray.py:640-678, fxpmodel.py:352-364.
 * x: (N,K) int32; w: (K,M) int32; bias: (M) int32 or NULL.  y = sat(asr(x@w, x_exp+w_exp-out_exp))
 * [+ change_exp(bias), sat].  flags bit0: apply ReLU (fxpmodel.py:53-63) to the result. */
int s5fxp_dense(const int32_t *x, const int32_t *w, const int32_t *bias, int32_t *y, int64_t N, int K, int M,
                int x_exp, int w_exp, int b_bits, int b_exp, int out_bits, int out_exp, int flags, void *stream);

/* The same layer with a pruned weight (the reference keeps pruned kernels dense with zeros, jaxpruner masks;
 * fxparray.py:662 sums them anyway): the (K,M) kernel stored by output channel -- CSR of kernel^T: rowptr[M+1],
 * colidx[nnz] = k, val[nnz] -- all device int32.  Bit-identical to s5fxp_dense on the densified weight. */
int s5fxp_dense_csr(const int32_t *x, const int32_t *rowptr, const int32_t *colidx, const int32_t *val,
                    const int32_t *bias, int32_t *y, int64_t N, int K, int M, int x_exp, int w_exp, int b_bits, int b_exp,
                    int out_bits, int out_exp, int flags, void *stream);

/* fxp_add with a numeric result_exp, fxparray.py:449-466.  y_len == n (same shape) or a divisor
 * of n (trailing-axis broadcast, e.g. a bias vector). */
int s5fxp_add(const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len, int x_bits, int x_exp,
              int y_bits, int y_exp, int out_bits, int out_exp, int negate_y, void *stream);

/* fxp_mul with a numeric result_exp, fxparray.py:573-637. */
int s5fxp_mul(const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len, int x_exp, int y_exp,
              int out_bits, int out_exp, void *stream);

/* result_exp="compute_best" forms (fxparray.py:420-448 and 601-609): the exponent is chosen on
 * the device from float32 maxima over the whole tensor.  `scratch` is >= 16 bytes of device
 * memory, zeroed by the call; out_exp_dev receives {result_exp, status}.  The host reads it back
 * (one sync, like the reference's own np.any/int() syncs). */
int s5fxp_add_cb(const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len, int x_bits, int x_exp,
                 int y_bits, int y_exp, int out_bits, int32_t *out_exp_dev, void *scratch, void *stream);
int s5fxp_mul_cb(const int32_t *x, const int32_t *y, int32_t *out, int64_t n, int64_t y_len, int x_exp, int y_exp,
                 int out_bits, int32_t *out_exp_dev, void *scratch, void *stream);

/* fxp_relu, fxpmodel.py:27-63.  im == NULL: real max(x,0).  Otherwise the complex form
 * (lexicographic maximum(z,0) through float32). */
int s5fxp_relu(const int32_t *re, const int32_t *im, int32_t *out_re, int32_t *out_im, int64_t n, void *stream);

/* FxpSigmoid.apply, fxpmodel.py:97-144.  lut: 8 host int32 values (fxpmodel.py:89-95). */
int s5fxp_sigmoid(const int32_t *x, int32_t *y, int64_t n, int x_bits, int x_exp, int sig_x_exp, int sig_y_exp,
                  const int32_t *lut_host, void *stream);

/* The sequential diagonal-SSM recurrence, fxpmodel.py:147-208 (make_ssm_step_fn / recurrent_loop,
 * vmapped over the batch at fxpmodel.py:682-704).  bu_*: (B,L,P); a_*: (P); xs_*: (B,L,P).
 * No clip anywhere; int32 wrap.  flags bit0: apply the complex ReLU to the stored states
 * (fxpmodel.py:740-742) instead of storing the raw states. */
int s5fxp_scan(const int32_t *bu_re, const int32_t *bu_im, const int32_t *a_re, const int32_t *a_im, int32_t *xs_re,
               int32_t *xs_im, int B, int L, int P, int a_re_exp, int a_im_exp, int bu_re_exp, int bu_im_exp,
               int x_re_exp, int x_im_exp, int flags, void *stream);

/* The FLOAT model's diagonal-SSM scan, sparseRNNs/model/ssm.py:54-77 (binary operator) + :127
 * (jax.lax.associative_scan over (Lambda_elements, Bu_elements); :166-168 reverse=True for the bidirectional half;
 * vmapped over the batch by the caller of _apply_ssm).  x_t = lambda * x_{t-1} + bu_t in complex64, evaluated as a
 * time-parallel prefix (segment folds, __shfl_up + LDS sweeps over the segment aggregates).  Floating-point prefix sums
 * depend on the combination tree: results agree with the reference's to rounding (tests/: |err| <= 1e-5 * max|x|), not bit
 * for bit.  lambda: (P) complex64 (re, im interleaved); bu, xs: (B,L,P) complex64; x0: (B,P) state before the first
 * step or NULL (zeros -- the reference's scan has no initial element); x_last: (B,P) state after the last step, or NULL.
 * reverse != 0 scans from t = L-1 down to 0. */
int s5fxp_assoc_scan_c64(const float *lambda, const float *bu, float *xs, const float *x0, float *x_last, int B, int L,
                         int P, int reverse, void *stream);

/* ------------------------------------------------------------------------------------------
 * Model level: FxpRegressionModel.forward, fxpmodel.py:1431-1439 (-> 1261-1271 -> 1110-1161).
 * The integer parameters are what FxpRegressionModel.export() emits
 * (fxpmodel.py:368-393,819-847,946-968,1163-1207,1441-1458).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t K, M;
    const int32_t *weight; /* host, (K,M) */
    const int32_t *bias;   /* host, (M) */
    int32_t w_bits, w_exp, b_bits, b_exp, inp_bits, inp_exp, out_bits, out_exp;
} s5fxp_dense_desc;

typedef struct {
    int32_t H, P;
    const int32_t *A_re, *A_im; /* host, (P)   Lambda_bar */
    const int32_t *B_re, *B_im; /* host, (P,H) B_bar */
    const int32_t *C_re, *C_im; /* host, (H,P) C_tilde */
    const int32_t *D;           /* host, (H) */
    int32_t A_re_bits, A_re_exp, A_im_bits, A_im_exp, B_re_bits, B_re_exp, B_im_bits, B_im_exp;
    int32_t C_re_bits, C_re_exp, C_im_bits, C_im_exp, D_bits, D_exp;
    int32_t u_bits, u_exp, Bu_re_bits, Bu_re_exp, Bu_im_bits, Bu_im_exp;
    int32_t x_re_bits, x_re_exp, x_im_bits, x_im_exp, y_bits, y_exp;
} s5fxp_ssm_desc;

typedef struct {
    const int32_t *minus_mean, *invsq_var, *scale, *bias; /* host, (H); scale/bias may be NULL */
    int32_t mean_bits, mean_exp, invsq_var_bits, invsq_var_exp, scale_bits, scale_exp, bias_bits, bias_exp;
} s5fxp_norm_desc;

typedef struct {
    s5fxp_norm_desc norm;
    s5fxp_ssm_desc ssm;
    s5fxp_dense_desc out2;
    int32_t l_bits, l_exp, r_bits, r_exp, res_bits, res_exp; /* mult_gate, fxpmodel.py:1075-1093 */
    int32_t sig_x_exp, sig_y_exp;                            /* fxpmodel.py:1097-1104 */
    int32_t lut[8];                                          /* fxpmodel.py:89-95 */
} s5fxp_layer_desc;

typedef struct {
    int32_t n_layers;
    s5fxp_dense_desc encoder;
    const s5fxp_layer_desc *layers;
    s5fxp_dense_desc decoder;
} s5fxp_model_desc;

typedef struct s5fxp_model s5fxp_model; /* opaque host handle */

/* Optional capture of per-layer intermediates (device pointers, any may be NULL); names follow
 * the reference's sow() keys (fxpmodel.py:653,736,742,793,1120,1126,1135,1145,1153). */
typedef struct {
    int32_t *pre_s5;        /* (N,H)  BatchNorm output                       */
    int32_t *u;             /* (N,H)  SSM input after change_cfg             */
    int32_t *Bu_re, *Bu_im; /* (N,P)                                         */
    int32_t *xs_re, *xs_im; /* (N,P)  raw states                             */
    int32_t *ys;            /* (N,H)  "pre_GLU"                              */
    int32_t *out2;          /* (N,H)                                         */
    int32_t *out2_sigmoid;  /* (N,H)                                         */
    int32_t *post_GLU;      /* (N,H)                                         */
    int32_t *residadd;      /* (N,H)                                         */
} s5fxp_layer_trace;

/* Bytes of device memory the packed parameter blob needs. */
size_t s5fxp_model_blob_bytes(const s5fxp_model_desc *desc);

/* Packs the integer parameters (int8 weight planes for the MFMA path, int32 for the generic one) into
 * `dev_blob` (device, blob_bytes) with a stream-ordered copy and returns a host handle that keeps
 * the scalars.  flags: S5FXP_MODEL_* below.  Returns S5FXP_EUNSUPPORTED for shapes outside the
 * kernels' limits.  Pruned models run the dense kernels on their zero-filled weights (bit-exact; on the MFMA path
 * the weights live in registers and the contraction is ~2 us per kernel): S5FXP_MODEL_FORCE_DENSE is the default
 * behaviour, S5FXP_MODEL_FORCE_CSR is refused (EUNSUPPORTED) -- the CSR kernel exists at op level, s5fxp_dense_csr. */
enum { S5FXP_MODEL_DEFAULT = 0, S5FXP_MODEL_FORCE_DENSE = 1, S5FXP_MODEL_FORCE_CSR = 2, S5FXP_MODEL_FORCE_GENERIC = 4 };
int s5fxp_model_create(const s5fxp_model_desc *desc, void *dev_blob, size_t blob_bytes, int flags, void *stream,
                       s5fxp_model **out);
void s5fxp_model_destroy(s5fxp_model *m);

/* Device workspace for one forward of B sequences of L frames. */
size_t s5fxp_workspace_bytes(const s5fxp_model *m, int B, int L);

/* Number of int32 words in the device status buffer, and its layout:
 *   [0] error bits (S5FXP_ST_*), [1] decoder output exponent,
 *   [2] which kernels this forward ran: S5FXP_PATH_GENERIC (one-lane / VALU kernels, any int32 operands) or
 *       S5FXP_PATH_FUSED (the int8-MFMA tile kernels + the quad / pair recurrence kernels),
 *   [8 + 8*l + 0..4] layer l: exponents chosen by the 4 BatchNorm ops and the residual add,
 *   [8 + 8*l + 5]    layer l: the recurrence kernel that was enqueued first, coded as s5fxp_model_recurrence_kernel
 *                    (5 = the exact 32-bit quad chain of a S5FXP_FWD_EXACT forward),
 *   [8 + 8*l + 6]    layer l: the state slots its kernels ran on: P, or the live states rounded up to a multiple of 32
 *                    when the layer was compacted (s5fxp_model_live_states).
 *   [8 + 8*l + 7]    layer l: the state slots its two recurrence streams hold: [8 + 8*l + 6], or fewer -- the live states
 *                    rounded up to an even number -- when the LDS-fed pair kernel runs a layer compacted to 32 slots. */
#define S5FXP_STATUS_WORDS 128
enum { S5FXP_PATH_GENERIC = 1, S5FXP_PATH_FUSED = 2 };
enum {
    S5FXP_ST_NEGSHIFT = 1,   /* a data-dependent shift came out negative: the reference raises ValueError */
    S5FXP_ST_NEGEXP = 2,     /* a compute_best exponent came out negative (1 << exp fails in the reference) */
    S5FXP_ST_WIDE_STATE = 4, /* informational: an SSM state exceeded 24 bits, the 32-bit C projection ran */
    S5FXP_ST_WIDE_INPUT = 8, /* the input tensor holds values beyond 24 bits: results invalid, re-run with a
                                model created with S5FXP_MODEL_FORCE_GENERIC */
    S5FXP_ST_REDO = 16       /* only with S5FXP_FWD_DEFER_REDO: a state left the fast recurrence's exact range and
                                the exact re-run was NOT enqueued: results invalid, call again with
                                S5FXP_FWD_EXACT */
};

/* s5fxp_forward_opts.flags.  By default a forward is self-contained: next to the fast recurrence it enqueues
 * the exact 32-bit kernels, gated on a device flag, so the output is right whatever the data.  A caller that
 * reads the status words anyway can drop those (normally idle) launches:
 *   DEFER_REDO  do not enqueue the gated exact kernels; S5FXP_ST_REDO in status[0] tells the caller to repeat
 *               the forward with S5FXP_FWD_EXACT.  The recurrence streams are then kept as int16 where the model's
 *               Bu configuration guarantees they fit (half the bytes; a state beyond 16 bits saturates and raises
 *               S5FXP_ST_REDO like any other state outside the fast kernels' range);
 *   EXACT       skip the fast recurrence, run the exact kernels only;
 *   NO_PAIR     (with DEFER_REDO) use the quad kernel with int16 streams instead of the pair kernel: its bound on
 *               |state| is the full 16 bits, the pair kernel's is tighter by what the folded Bu needs -- the middle rung
 *               of a caller's pair -> quad -> exact ladder. */
enum { S5FXP_FWD_DEFER_REDO = 1, S5FXP_FWD_EXACT = 2, S5FXP_FWD_NO_PAIR = 4 };

/* Cross-rank hook for the data-dependent exponents (mode A): when not NULL it is
 * called once per compute_best op, after the local float32 maxima (n <= 4 floats, device memory)
 * are complete on `stream`, and must leave their element-wise MAX over all ranks in place,
 * stream-ordered.  NULL = per-shard exponents (mode B). */
typedef int (*s5fxp_allreduce_max_fn)(void *ctx, float *dev_vals, int n, void *stream);

/* Optional knobs of one forward (all-zero = defaults). */
typedef struct {
    s5fxp_allreduce_max_fn allreduce; /* NULL: per-shard exponents */
    void *allreduce_ctx;
    /* Measurement only: 2*n_layers hipEvent_t handles (or NULL).  Fused path: events [2l] / [2l+1] are attached to
     * layer l's recurrence launch (hipExtLaunchKernelGGL start / stop events: the time stamps of that dispatch, as
     * rocprofv3's kernel trace reports them); both must be set.  Generic path: recorded on `stream` immediately
     * before / after the layer's recurrence kernel.  NULL entries are skipped. */
    void **scan_events;
    int32_t flags; /* S5FXP_FWD_* */
    /* Streaming (sparseRNNs/fxpmodel.py:147-172: the step function's carry is an explicit argument; the reference's
     * recurrent_loop always starts it at zero, :196-207).  Device arrays [n_layers][2][B][P] int32 (re plane, im plane
     * per layer) or NULL: state_in = the SSM states the recurrences start from (NULL: zeros), state_out = where the
     * states after the last frame are left (NULL: not wanted).  Feeding a sequence chunk by chunk with the carry gives,
     * per chunk, what the reference computes for that chunk started from that carry -- every chunk is its own
     * compute_best batch, so the exponents (and with them the low bits) can differ from one pass over the whole
     * sequence.  With S5FXP_FWD_DEFER_REDO keep state_in and state_out apart: a forward that comes back with
     * S5FXP_ST_REDO has left an unusable state_out, and its repeat needs the old state_in. */
    const int32_t *state_in;
    int32_t *state_out;
    /* Grouped call (0 or 1: a plain forward).  groups = G > 1: x and y hold G * B sequences -- G independent reference
     * batches of B sequences each, exactly what G calls with B sequences compute (every group is its own compute_best
     * batch: own exponents, own status words, own carry), but enqueued as ONE set of kernel launches (gridDim.y = G on
     * the fused path; a loop over the groups elsewhere).  Everything per-forward is G-fold and contiguous, group after
     * group: workspace >= G * s5fxp_workspace_bytes(m, B, L) (that value is the group stride), status
     * G * S5FXP_STATUS_WORDS words, state_in / state_out [G][n_layers][2][B][P], traces G * n_layers entries.  This is the
     * reference's run_validation loop over batches (sparseRNNs/fxprun.py:53-88) taken several batches at a time: at the
     * N-DNS batch of 32 sequences a launch costs about as much as a quarter of its work. */
    int32_t groups;
    /* Measurement only, like scan_events: 2*n_layers hipEvent_t handles (or NULL) attached to layer l's GATE-kernel launch
     * (C projection + out2 + gate) on the fused path; ignored elsewhere. */
    void **gate_events;
} s5fxp_forward_opts;

/* x: (B,L,d_in) int32 device; y: (B,L,d_out) int32 device; status: S5FXP_STATUS_WORDS int32 device.
 * traces: NULL or n_layers entries (host array of device pointers); opts: NULL or see above. */
int s5fxp_model_forward(const s5fxp_model *m, const int32_t *x, int x_bits, int x_exp, int B, int L, int32_t *y,
                        void *workspace, size_t workspace_bytes, int32_t *status, const s5fxp_layer_trace *traces,
                        const s5fxp_forward_opts *opts, void *stream);

/* Environment (experiments and tests only; read ONCE by s5fxp_model_create and stored in the handle, never by a forward):
 *   S5FXP_NO_PAIR, S5FXP_PAIR_GLOBAL, S5FXP_PAIRL_BLOCKS=16   recurrence kernel choice (see s5fxp_model_recurrence_kernel)
 *   S5FXP_NO_PK16, S5FXP_NO_BN_EXT, S5FXP_NO_COMPACT           unpacked gate epilogues / four-reduction BatchNorm exponents / no
 *                                                              live-state compaction
 *   S5FXP_NO_DEC_RESID                                         the last layer's residual pass as a launch of its own
 *   S5FXP_NO_LIVE_LANES                                        a compacted layer's recurrence streams keep their padding slots
 *   S5FXP_GATE_BN                                              the gate kernel recomputes the SSM input u instead of reading it (slower)
 *   S5FXP_CGATE_FT64, S5FXP_WGS_CGATE32=n                      the gate kernel on 64-frame tiles (six-wave workgroups) / workgroups per launch
 *                                                              of the default 32-frame form
 *   S5FXP_WGS_ENC|DEC|CGATE|BPROJ|RESID=n                      workgroups per launch of the tile kernels
 *   S5FXP_PLANE_SKEW=bytes                                     extra distance between the workspace's planes (multiple of 256)
 *   S5FXP_DEBUG_SYNC                                           synchronise and check after every stage of a forward
 * Results do not depend on any of them. */

/* FxpSequenceLayer.forward, fxpmodel.py:1110-1161, for layer `layer` of a created model -- the unit the reference's
 * verification walks (fxprun.py:583-727).  x: (B,L,H) int32 device with configuration (x_bits, x_exp); y: (B,L,H) int32
 * device, s5fxp_model_layer_out_bits() bits at the exponent the residual compute_best add chose: written to
 * status[8 + 8*layer + 4] and, when y_exp_dev is not NULL, to that device int.  Runs the generic int32 kernels (exact for
 * any int32 operands) whatever path whole forwards of the model take.  workspace / status / trace (ONE entry) / opts as
 * s5fxp_model_forward; opts->state_in / state_out here are [2][B][P], the carry of this layer alone; opts->groups must be
 * 0 or 1. */
int s5fxp_layer_forward(const s5fxp_model *m, int layer, const int32_t *x, int x_bits, int x_exp, int B, int L, int32_t *y,
                        int32_t *y_exp_dev, void *workspace, size_t workspace_bytes, int32_t *status,
                        const s5fxp_layer_trace *trace, const s5fxp_forward_opts *opts, void *stream);
int s5fxp_model_layer_out_bits(const s5fxp_model *m, int layer);
/* States of `layer` whose rows of B_bar are not all zero.  The others receive Bu = 0 at every step and stay (0, 0) from a
 * zero carry (fxpmodel.py:147-172), so their columns of C multiply zeros: when at most P / 2 states are live, forwards on
 * the fused path that neither trace the states nor carry them in or out run the layer's kernels on the fewest groups of 32
 * state slots that hold the live ones (32 or 64 of 128, 32 of 64)
 * (bit-identical; S5FXP_NO_COMPACT at model creation switches it off).  -1: bad argument. */
int s5fxp_model_live_states(const s5fxp_model *m, int layer);

/* Static facts about a created model (for INTEGRATION / debugging). */
int s5fxp_model_out_exp(const s5fxp_model *m);
int s5fxp_model_out_bits(const s5fxp_model *m);
/* 1 if the int8-MFMA kernels were packed for this model (NDNS shapes, <= 8-bit weights, <= 16-bit
 * activations); every forward of such a model runs them, whatever B and L (a sequence's last 4-step block may be
 * partial: L = 3751, the N-DNS clips' native length, or single frames of a stream).  status[2] reports it per forward. */
int s5fxp_model_is_fast(const s5fxp_model *m);
/* Which recurrence kernel an optimistic forward (S5FXP_FWD_DEFER_REDO) runs for `layer`:
 * 0 one lane per state (generic), 1 quad kernel with int32 streams, 2 quad kernel with int16 streams,
 * 3 pair kernel (int32 K stream in, int16 states out; sparseRNNs/fxpmodel.py:147-172 in four instructions per step),
 * 4 the same pair kernel fed through LDS by a helper wave from an int16 Bu stream (the default where it applies).
 * -1: bad argument.  The exact re-run (S5FXP_FWD_EXACT) always uses the 32-bit quad kernel on the MFMA path. */
int s5fxp_model_recurrence_kernel(const s5fxp_model *m, int layer);
/* The bound on |state| up to which that kernel is exact (the consumer of the states checks it on the data and raises
 * S5FXP_ST_REDO / runs the exact kernels beyond it): 32766 at most for the int16-stream kernels, less for the pair
 * kernel when the layer's coefficients and Bu width leave less room.  0: no bound (generic 32-bit recurrence), -1: bad
 * argument. */
int s5fxp_model_recurrence_xmax(const s5fxp_model *m, int layer);

#ifdef __cplusplus
}
#endif
#endif /* S5FXP_H */
