/*
 * svae.h -- C ABI of the MI355X (gfx950) spatial-VAE decoder hot path.
 *
 * The reference (cfframe/spatial-VAE) has no FFI layer: its boundary for this path is the
 * Python callable p_net(x.contiguous(), z) = SpatialGenerator.forward
 * (spatial_vae/models.py:90-132; call sites train_mnist.py:77, train_galaxy.py:115,
 * train_particles.py:102) together with the lines of eval_minibatch around it that build
 * the coordinates and score the output.  Each entry point below names the reference lines
 * it replaces.  INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller, fp32 unless stated, contiguous
 *     row-major in the shapes given (the shapes torch gives the reference's tensors);
 *   - nothing here allocates, synchronises or throws: work is enqueued on `stream` (a
 *     hipStream_t; NULL = the default stream) and the call returns; 0 = success, a negative
 *     SVAE_E_* code otherwise, with text in svae_last_error() (thread-local);
 *   - the library is stateless and re-entrant: one process per GPU, any number of streams;
 *   - `saved` (svae_saved_bytes) carries activations (and the packed weights / per-image tables
 *     built from the parameters, z and the pose) from forward to backward and must stay untouched
 *     in between, as must the parameters, z and the pose themselves; `ws` (svae_workspace_bytes) is
 *     scratch, free to reuse after the call's work has completed on the stream.  Both must be
 *     256-byte aligned.
 */
#ifndef SVAE_H
#define SVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVAE_ABI_VERSION 2
#define SVAE_MAX_HIDDEN 7 /* hidden H x H layers = num_layers - 1 */
#define SVAE_MAX_OUT 4    /* n_out (channels) */

typedef void* svae_stream_t; /* hipStream_t */

enum {
    SVAE_OK = 0,
    SVAE_E_INVALID = -1,   /* bad descriptor / null pointer / unsupported size */
    SVAE_E_WORKSPACE = -2, /* ws_bytes too small or misaligned */
    SVAE_E_LAUNCH = -3     /* HIP reported an error at launch */
};

/* activation = the nn.Module class handed to SpatialGenerator (models.py:58, 77-83) */
enum { SVAE_ACT_TANH = 0, SVAE_ACT_LEAKYRELU = 1, SVAE_ACT_RELU = 2, SVAE_ACT_SIGMOID = 3 };
enum {
    SVAE_FLAG_RESID = 1,    /* resid=True: hidden layers are ResidLinear (models.py:13-21) */
    SVAE_FLAG_BILINEAR = 2, /* bilinear=True (models.py:74-75, 114-121) */
    SVAE_FLAG_SOFTPLUS = 4  /* softplus=True on output channel 0 (models.py:129-130) */
};

/* Constructor arguments of SpatialGenerator (models.py:58-59) plus the batch geometry. */
typedef struct svae_desc {
    int32_t B;      /* images in this call */
    int32_t N;      /* coordinates (pixels) per image */
    int32_t H;      /* hidden_dim */
    int32_t L;      /* num_layers >= 1 (number of hidden activations) */
    int32_t Zd;     /* latent_dim handed to the decoder (0 = no latent_linear) */
    int32_t C;      /* n_out, 1..SVAE_MAX_OUT */
    int32_t in_dim; /* 2, or 5 with expand_coords (models.py:65-67, 99-102) */
    int32_t act;    /* SVAE_ACT_* */
    int32_t flags;  /* SVAE_FLAG_* */
} svae_desc;

/* Parameters exactly as nn.Linear / nn.Bilinear store them (state-dict names in comments). */
typedef struct svae_params {
    const float* coord_w;                   /* coord_linear.weight (H, in_dim) */
    const float* coord_b;                   /* coord_linear.bias   (H) */
    const float* latent_w;                  /* latent_linear.weight (H, Zd); NULL iff Zd == 0 */
    const float* bilinear_w;                /* bilinear.weight (H, in_dim, Zd); NULL unless BILINEAR */
    const float* hidden_w[SVAE_MAX_HIDDEN]; /* layers.<i>[.linear].weight (H, H), L-1 entries */
    const float* hidden_b[SVAE_MAX_HIDDEN]; /* layers.<i>[.linear].bias   (H) */
    const float* out_w;                     /* last Linear weight (C, H) */
    const float* out_b;                     /* last Linear bias   (C) */
} svae_params;

/* Same shapes; every non-NULL entry is OVERWRITTEN with d(loss)/d(parameter). */
typedef struct svae_grads {
    float* coord_w;
    float* coord_b;
    float* latent_w;
    float* bilinear_w;
    float* hidden_w[SVAE_MAX_HIDDEN];
    float* hidden_b[SVAE_MAX_HIDDEN];
    float* out_w;
    float* out_b;
} svae_grads;

/*
 * Where the decoder's coordinates come from.
 *   coords != NULL : explicit (B, N, 2), the generic SpatialGenerator.forward(x, z) entry.
 *   coords == NULL : built on the fly from the shared grid and the per-image pose, replacing
 *                    x.expand + rot + bmm + translate of eval_minibatch (train_mnist.py:26, 42-74;
 *                    train_galaxy.py:73-110; train_particles.py:60-97):
 *                        x''[b,i] = grid[i] @ [[cos t_b, sin t_b], [-sin t_b, cos t_b]] + dx[b]
 *                    theta NULL = no rotation, dx NULL = no translation (dx already times dx_scale).
 */
typedef struct svae_pose {
    const float* coords; /* (B, N, 2) or NULL */
    const float* grid;   /* (N, 2), used when coords == NULL */
    const float* theta;  /* (B) or NULL */
    const float* dx;     /* (B, 2) or NULL */
} svae_pose;

/* Gradient sinks matching svae_pose; NULL entries are skipped. */
typedef struct svae_pose_grads {
    float* dcoords; /* (B, N, 2): d/d(coords) -- allowed in either mode */
    float* dtheta;  /* (B):    sum_i <dx''[b,i], d(rot)/d(theta) grid[i]> */
    float* ddx;     /* (B, 2): sum_i dx''[b,i] */
} svae_pose_grads;

int svae_abi_version(void);
const char* svae_last_error(void);

/* Bytes of the forward->backward carry and of the scratch area for a descriptor. */
size_t svae_saved_bytes(const svae_desc* d);
size_t svae_workspace_bytes(const svae_desc* d);

/*
 * Decoder forward: replaces SpatialGenerator.forward (models.py:90-132) and, when
 * pose->coords == NULL, the coordinate transform in front of it (see svae_pose).
 *   z      (B, Zd)     latent handed to the decoder (ignored when Zd == 0)
 *   y      (B, N, C)   module output: sigmoid(logits), channel 0 through softplus if flagged
 *   logits (B, N, C)   input of the final Sigmoid = output of layers[-2]; may be NULL for
 *                      inference (svae_decoder_backward needs it)
 *   saved              svae_saved_bytes(d) bytes, or NULL for inference-only calls
 */
int svae_decoder_forward(const svae_desc* d, const svae_params* p, const svae_pose* pose, const float* z,
                         float* y, float* logits, void* saved, void* ws, size_t ws_bytes,
                         svae_stream_t stream);

/*
 * The same forward with the per-pixel Bernoulli log-likelihood folded into the output layer's last kernel: replaces
 * SpatialGenerator.forward AND  -F.binary_cross_entropy(y_hat, y) * size  (train_mnist.py:77-81; train_galaxy.py:115-119) in
 * per-image form, with torch's clamps (see svae_bce_loglik) taken on the same fp32 sigmoid value.  No second pass over y.
 *   target (B, N, C); loglik (B); dll_dy (B, N, C) = d(loglik_b)/d(y), may be NULL for inference.
 * Backward: hand dll_dy to svae_decoder_backward as `dy` and the upstream gradient of loglik as `dy_scale`.
 * Refused with SVAE_FLAG_SOFTPLUS (the reference's binary_cross_entropy raises on values above 1).
 */
int svae_decoder_forward_bce(const svae_desc* d, const svae_params* p, const svae_pose* pose, const float* z,
                             const float* target, float* y, float* logits, float* loglik, float* dll_dy, void* saved, void* ws,
                             size_t ws_bytes, svae_stream_t stream);

/*
 * Decoder backward: replaces the autograd replay of the lines above.
 *   logits    (B, N, C) the forward's pre-Sigmoid output (the wrapper keeps it, as autograd
 *             keeps the Sigmoid's result in the reference)
 *   dy        (B, N, C) d(loss)/d(y), y being the module output
 *   dy_scale  (B) or NULL: per-image factor applied to dy (the upstream gradient of a
 *             per-image log-likelihood; lets a fused loss hand over d(loglik_b)/d(y) unscaled)
 *   grads     every non-NULL field is overwritten
 *   dz        (B, Zd) or NULL
 *   pg        coordinate / pose gradients, or NULL
 */
int svae_decoder_backward(const svae_desc* d, const svae_params* p, const svae_pose* pose, const float* z,
                          const float* logits, const float* dy, const float* dy_scale, const void* saved,
                          const svae_grads* grads, float* dz, const svae_pose_grads* pg, void* ws,
                          size_t ws_bytes, svae_stream_t stream);

/*
 * Per-pixel Bernoulli log-likelihood with torch's clamps: replaces
 * -F.binary_cross_entropy(y_hat, y) * size (train_mnist.py:78-81; train_galaxy.py:116-119)
 * in per-image form (the reference's scalar is the mean of loglik over the batch).
 *   y_hat, target (B, n) with n = N*C;  loglik (B);  dll_dy (B, n) = d(loglik_b)/d(y_hat), may be NULL
 */
int svae_bce_loglik(int32_t B, int32_t n, const float* y_hat, const float* target, float* loglik,
                    float* dll_dy, svae_stream_t stream);

/*
 * Gaussian log-likelihood of train_particles.py:102-139, per image, including its layout quirk
 * (with C == 2 the first N entries of the channel-interleaved row are the mean, the last N the
 * log-variance), the optional CTF filter (depth-wise k x k cross-correlation of the mean image,
 * zero padding k/2; train_particles.py:112-119; C == 1 only) and the optional pixel mask
 * (train_particles.py:126-132).
 *   y_params (B, N*C); target (B, N); mask (N) bytes or NULL; ctf (B, k, k) or NULL
 *   loglik (B); dll_dy (B, N*C) may be NULL; ws: svae_gaussian_workspace_bytes(B, N) when ctf != NULL
 */
size_t svae_gaussian_workspace_bytes(int32_t B, int32_t N);
int svae_gaussian_loglik(int32_t B, int32_t N, int32_t C, const float* y_params, const float* target,
                         const uint8_t* mask, const float* ctf, int32_t k, float* loglik, float* dll_dy,
                         void* ws, size_t ws_bytes, svae_stream_t stream);

/*
 * Latent head of eval_minibatch: reparameterisation, pose split and the KL terms, per image.  Replaces
 * train_mnist.py:33-39 (z = exp(logstd)*r + mu), :42-53 / :65-72 (theta = z[:,0]; dx = z[:,off:off+2]*dx_scale;
 * content = z[:,c0:], times z_scale in train_galaxy.py:112 / train_particles.py:99), :61-63 (KL of theta;
 * mu_penalty = 1 keeps train_mnist.py's mu^2 term, 0 is train_galaxy.py:98-99 / train_particles.py:85-86) and
 * :84-85 (unit-normal KL over every remaining latent).  kl[b] is the per-image KL; the reference's kl_div is its mean.
 *   q_out (B, 2*inf_dim) = [z_mu | z_logstd] as InferenceNetwork produces it (models.py:50-52); r (B, inf_dim)
 *   theta (B) iff rotate; dx (B, 2) iff translate; zc (B, inf_dim - rotate - 2*translate) iff non-empty; kl (B)
 * Backward: g_* are d(loss)/d(output) (NULL = zero); g_q_out (B, 2*inf_dim) is overwritten.
 */
typedef struct svae_latent_desc {
    int32_t B;
    int32_t inf_dim;
    int32_t rotate;
    int32_t translate;
    int32_t mu_penalty;
    float dx_scale;
    float z_scale;
    float theta_prior;
} svae_latent_desc;
int svae_latent_forward(const svae_latent_desc* d, const float* q_out, const float* r, float* theta, float* dx,
                        float* zc, float* kl, svae_stream_t stream);
int svae_latent_backward(const svae_latent_desc* d, const float* q_out, const float* r, const float* g_theta,
                         const float* g_dx, const float* g_zc, const float* g_kl, float* g_q_out,
                         svae_stream_t stream);

/*
 * The three scalars eval_minibatch returns, from the per-image terms: out3 = {elbo, log_p_x_g_z, kl_div} =
 * {mean(loglik) - mean(kl), mean(loglik), mean(kl)} (train_mnist.py:81, 86-88: `kl_div.mean()`, `elbo = log_p - kl_div`),
 * and the gradient of any combination of them back to the per-image terms.  g_* are device scalars or null.
 */
int svae_elbo_head_forward(const float* loglik, const float* kl, int32_t B, float* out3, svae_stream_t stream);
int svae_elbo_head_backward(const float* g_elbo, const float* g_logp, const float* g_kl, int32_t B, float* dloglik, float* dkl,
                            svae_stream_t stream);

/*
 * out[c] = sum_r x[r][c] for a row-major (rows, cols) fp32 matrix, fixed summation order: the bias gradient of a Linear
 * layer (the encoder's, whose weight gradients torch computes with hipBLASLt).
 */
int svae_colsum(const float* x, int32_t rows, int32_t cols, float* out, svae_stream_t stream);

/*
 * A Linear layer of the inference network with its activation, for the small row counts of the encoder (one row per image):
 * replaces nn.Linear + activation() of InferenceNetwork.layers (spatial_vae/models.py:31-43, 46-54) and their autograd
 * backward.  fp32 MFMA (v_mfma_f32_16x16x4_f32), one 16 x 16 output tile per wave, bias and activation in the epilogue.
 *   forward : out[m][n] = act( sum_k x[m][k] weight[n][k] + bias[n] );  x (rows, in), weight (out, in) as nn.Linear keeps it,
 *             bias (out) or NULL, out (rows, out);  act = SVAE_ACT_* or SVAE_LINEAR_ACT_NONE
 *   backward: with dpre = dout * act'(out):  dweight (out, in) = dpre^T x,  dbias (out) = column sums of dpre,
 *             dx (rows, in) = dpre weight -- any of the three may be NULL; ONE launch produces all that are asked for.
 *             `out` is the forward's output (act' is taken through it); it may be NULL when act == SVAE_LINEAR_ACT_NONE.
 * Meant for layers whose weights are a few MB; the first layer of the galaxy encoder (49 152 x 5 000) is a real GEMM and
 * stays with the vendor library.
 */
#define SVAE_LINEAR_ACT_NONE (-1)
int svae_linear_forward(const float* x, const float* weight, const float* bias, float* out, int32_t rows, int32_t in_features,
                        int32_t out_features, int32_t act, svae_stream_t stream);
int svae_linear_backward(const float* x, const float* weight, const float* out, const float* dout, int32_t rows,
                         int32_t in_features, int32_t out_features, int32_t act, float* dweight, float* dbias, float* dx,
                         svae_stream_t stream);

/*
 * One Adam update over a flat fp32 parameter buffer: the arithmetic of torch.optim.Adam (amsgrad off, no weight
 * decay) as the reference uses it (optim = torch.optim.Adam(params, lr=lr); optim.step(), train_mnist.py:389,
 * 149), element-wise:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *                      p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
 * `step` is t (1 for the first update).  ATen's fused multi-tensor kernel gives one flat tensor of 0.9 M
 * elements only 14 thread blocks (98 us on MI355X); this is a plain grid over the elements (~5 us).
 * zero_grad != 0 also clears `grad` behind the update (optim.zero_grad(), train_mnist.py:150) in the same pass.
 */
int svae_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, int64_t step, int32_t zero_grad, svae_stream_t stream);

/*
 * Rotation augmentation of the observed images before inference: the reference rotates each image of the
 * minibatch by its own random angle with Pillow, one image at a time on the host (train_galaxy.py:41-54: uint8
 * images, `im.rotate(360*offset/2/pi, resample=Image.BICUBIC)`; train_particles.py:31-43: float32 images).  This is
 * the same resampler for a whole batch resident in HBM, bit-identical to Pillow 12.2's Image.rotate.
 *   y, y_rot  (B, rows*cols, C) fp32, distinct buffers.
 *   matrix    (B, 6) doubles ON THE DEVICE: the inverse affine coefficients Image.rotate derives from the angle
 *             (cos/sin of -radians(angle) rounded to 15 decimals, centre (cols/2, rows/2)); ignored where quarter >= 0.
 *   quarter   (B) int32 on the device: -1 = use the matrix; 0..3 = exact counter-clockwise quarter turns (Pillow's
 *             fast paths for angle % 360 in {0, 90, 180, 270}).
 *   quantize_u8 != 0: samples pass through uint8 as in train_galaxy.py:50-53 ((y*255).astype(uint8) in, /255 out);
 *             0: float32 'F' images as in train_particles.py:40-42.
 */
int svae_rotate_bicubic(const float* y, float* y_rot, const double* matrix, const int32_t* quarter, int32_t B, int32_t rows,
                        int32_t cols, int32_t C, int32_t quantize_u8, svae_stream_t stream);

/*
 * Real-space CTF filter bank, one (n, m) filter per particle: the reference's ctf_filter (spatial_vae/ctf.py:33-56 --
 * closed-form CTF of ctf.py:7-24 on the np.fft.fftfreq grid divided by apix*scale, ifft2, fftshift, real part, negated),
 * which it evaluates per particle with numpy on the host inside the DataLoader path (train_particles.py:420-426).
 *   params   (count, 8) doubles on the device, columns as in ctf.py:29: defocus [um], cs [mm], voltage [kV], apix [A],
 *            bfactor, ampcont [%], dfdiff (unused by the reference), dfang [deg].
 *   filters  (count, n, m) fp32 out -- the `ctf` operand of svae_gaussian_loglik.
 * Arithmetic is in doubles like numpy's; the result agrees with the reference to fp32 rounding.
 * Filters of up to ~80 x 80 are transformed entirely in the LDS of one CU and need no scratch (the workspace size is 0 and
 * ws may be NULL); larger ones keep their intermediate planes in `ws` and only the 2 (n + m) twiddle factors in LDS, so
 * n + m <= 10240 (SVAE_E_INVALID beyond; the reference's numpy code has no size limit, cryo-EM boxes are a few hundred).
 */
size_t svae_ctf_filter_workspace_bytes(int32_t count, int32_t n, int32_t m);
int svae_ctf_filter(const double* params, float* filters, int32_t count, int32_t n, int32_t m, double scale, void* ws,
                    size_t ws_bytes, svae_stream_t stream);

/*
 * How the hidden-layer GEMMs are computed.  SVAE_GEMM_FP32 (default): v_mfma_f32_32x32x2_f32, exact fp32 products.
 * SVAE_GEMM_FP16X3: every operand as two half tensors (hi + lo, power-of-two scaled), three f16 MFMAs per product with
 * fp32 accumulation -- as accurate as an fp32 GEMM, ~1.8x faster per training step; taken for tanh /
 * sigmoid nets whose width is a multiple of 64, fp32 kernels otherwise.  The mode changes svae_saved_bytes and
 * svae_workspace_bytes: set it before sizing buffers, and do not change it between a forward call and its backward call
 * (svae_decoder_backward returns SVAE_E_INVALID when `saved` was planned under another mode or SVAE_FUSE_OUT setting).
 * Without a call the environment variable SVAE_GEMM (fp16x3 | anything else) decides at first use.
 */
#define SVAE_GEMM_FP32 0
#define SVAE_GEMM_FP16X3 1
int svae_gemm_mode_set(int mode);
int svae_gemm_mode_get(void);

/*
 * Optional per-kernel timing (bench.py's roofline figure).  While enabled (on = 1: only the three MFMA
 * GEMM kernels, on = 2: every kernel, 0 = off), kernel launches of
 * this library are bracketed by two HIP events recorded on the launch stream; svae_profile_read
 * synchronises those events and returns, per kernel kind, the summed device time in ms and the
 * number of launches, then clears the records.  Not for use under stream capture.  The reference
 * has no counterpart (it has no profiling at all: SURVEY.md section 5).
 */
#define SVAE_PROF_KINDS 20
int svae_profile_enable(int on);
int svae_profile_read(double* ms_total, int64_t* launches); /* arrays of SVAE_PROF_KINDS */
const char* svae_profile_kind_name(int kind);

/*
 * Which kernel families the calls of this process have dispatched so far: the GEMM mode is a request and the plan decides
 * per geometry (fp16x3 takes the fp32 kernels for ReLU-type activations and odd tile counts; the output-layer backward has a
 * streaming, a split, a rank-1 and a generic fused form), so a caller -- or a test -- can tell which code actually ran.
 * counts[path] = launches of that family since process start (or since the last read with reset != 0); svae_path_name gives
 * the label ("dense_fp32_fwd", "dense_split_fwd", "wgrad_split", "out_bwd_rank1", ...; "" for unused slots).
 */
#define SVAE_PATH_KINDS 16
int svae_path_counts(int64_t* counts, int reset); /* array of SVAE_PATH_KINDS */
const char* svae_path_name(int path);

#ifdef __cplusplus
}
#endif
#endif /* SVAE_H */
