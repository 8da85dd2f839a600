/*
 * salun.h — C-ABI of the MI355X-native SalUn hot path (libsalun.so, gfx950).
 *
 * This is the drop-in boundary.  The reference
 * (OPTML-Group/Unlearn-Saliency) is 100 % Python and has no FFI of its own; each
 * entry point below replaces a *sequence of stock ATen launches* issued from a
 * Python loop over parameter tensors.  The reference sequence every function
 * replaces is cited as  <file>:<lines>  relative to the reference checkout.
 *
 * Conventions (all functions):
 *   - plain pointers and sizes only; no torch / HIP types in the signatures.
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - every pointer marked "dev" is a device pointer owned by the caller; nothing
 *     is retained after return; all work is stream-ordered, nothing synchronises
 *     the host, nothing allocates.  Scratch memory is passed in by the caller
 *     (`ws`, `ws_bytes`), sized by the matching *_workspace_bytes() query.
 *   - return value: 0 (SALUN_OK) or a negative errno-style code; never throws,
 *     never aborts.  salun_strerror() names a code.
 *   - vectors are the *flat* concatenation of all parameter tensors in
 *     named_parameters() order, each tensor row-major.
 *     Pointers should be 16-byte aligned for the fast path; any alignment is
 *     accepted (a scalar path is used).
 *   - masks are uint8 0/1 (1 byte per weight) inside the library; the reference's
 *     on-disk format (int64 0/1) is produced/consumed by the two converters.
 *   - fp32 arithmetic is IEEE round-to-nearest with the operation order stated
 *     per function (compiled with -ffp-contract=off; fused multiply-adds appear
 *     only where written as fma()).  oracle/salun_oracle.c restates exactly the
 *     same order on the CPU, so GPU-vs-oracle comparisons are bit-exact for the
 *     element-wise kernels and for the mask.
 */
#ifndef SALUN_H
#define SALUN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SALUN_OK 0
#define SALUN_EINVAL (-22)   /* bad argument (null pointer, negative size, nk out of range) */
#define SALUN_ENOSPC (-28)   /* workspace too small */
#define SALUN_EIO (-5)       /* HIP launch / runtime error */

#define SALUN_MAX_THRESHOLDS 16

typedef void *salun_stream_t; /* hipStream_t */

/* Library identification: major*10000 + minor*100 + patch. */
int salun_version(void);
/* Static string for a return code. */
const char *salun_strerror(int code);
/* Target ISA the kernels were compiled for ("gfx950"). */
const char *salun_arch(void);
/* Measurement tooling (tools/clock_probe.py): one wave samples the shader-cycle counter against the 100 MHz constant
 * counter `samples` times, `spins` s_sleep(127) apart: out[2i] = shader cycles, out[2i+1] = 100 MHz ticks.  Launched on a
 * stream of its own beside a workload it reads the clock the chip sustains under that workload. */
int salun_clock_probe(unsigned long long *out /*dev, 2*samples*/, int samples, int spins, salun_stream_t stream);

/* ------------------------------------------------------------------ K1 --
 * Saliency accumulation:   acc[i] <- acc[i] + (g[i] * scale)      (2 roundings)
 * Replaces the per-tensor loop  `gradients[name] += param.grad.data`
 *   Classification/generate_mask.py:41-44, DDPM/runners/diffusion.py:992-996,
 *   SD/train-scripts/generate_mask.py:66-69,171-174.
 * scale = 1 for Classification/SD.  For DDPM the per-batch clip of
 * runners/diffusion.py:985-990 is folded in: if `sqnorm` (dev, 1 float, the
 * squared global L2 norm of g from salun_grad_sqnorm) is non-NULL the scale is
 * computed on the device as  min(1, max_norm / (sqrt(*sqnorm) + 1e-6))  — the
 * clip_grad_norm_ coefficient — and `scale` is ignored.
 * Algorithmic traffic: 12 B / element. */
int salun_saliency_accumulate(float *acc /*dev*/, const float *g /*dev*/,
                              double scale, const float *sqnorm /*dev or NULL*/,
                              double max_norm, int64_t n, salun_stream_t stream);

/* ------------------------------------------------------------------ K2 --
 * Global top-k saliency mask for nk thresholds at once.
 * Replaces  abs_ -> -cat(flatten) -> argsort -> argsort -> (ranks < k)
 *   Classification/generate_mask.py:46-80, DDPM/runners/diffusion.py:998-1037,
 *   SD/train-scripts/generate_mask.py:71-106,176-209.
 * For threshold j:  masks_out[j][i] = 1  iff  rank_desc(|acc[i]|) < ks[j], where
 * rank_desc orders by |acc| descending, ties by flat index ascending (the
 * `argsort(stable=True)` reading of the reference) and NaN
 * after every number.  popcount(masks_out[j]) == min(max(ks[j],0), n) exactly.
 * `acc` is NOT modified (the abs is fused).  ks[] and masks_out[] are HOST
 * arrays (read before return); masks_out[j] are device pointers to n bytes.
 * 1 <= nk <= SALUN_MAX_THRESHOLDS.
 * Algorithmic traffic: 4 B read + nk B written per element.
 * Implementation (csrc/salun_topk.hip): for n >= 8192 and 16-B aligned input the vector is read ONCE — a hashed
 * sample brackets every threshold, one streaming pass writes the masks outside the brackets and compacts the ~4 % of
 * elements inside them, and the exact threshold is resolved among those candidates.  k >= n selects everything without
 * a select; exact zeros (a real accumulator holds several per cent of them) are counted instead of compacted, and a
 * threshold that lands among them is finished by a tie pass in flat-index order; candidates a workgroup's slab cannot
 * take (a layer whose magnitudes sit at the threshold) go to a shared spill row.  A persistent full-scan radix select
 * (3 histogram passes + 1 write pass, grid barriers) handles small / unaligned inputs and is the device-side fallback
 * when a bracket misses, heavy ties at a non-zero key overflow a buffer, or a threshold lies among NaNs — the resulting
 * mask is the same function of the input either way.
 * salun_mask_topk_ex takes flags:
 *   SALUN_TOPK_FORCE_FULL_SCAN  skip the single-read route (tests / A-B timing)
 *   SALUN_TOPK_VALUES_ONLY      no masks are written (masks_out may be NULL); only the thresholds are published
 * salun_mask_topk == salun_mask_topk_ex with flags 0. */
#define SALUN_TOPK_FORCE_FULL_SCAN 1u
#define SALUN_TOPK_VALUES_ONLY 2u
size_t salun_mask_topk_workspace_bytes(int64_t n, int nk);
int salun_mask_topk(const float *acc /*dev*/, int64_t n, const int64_t *ks /*host*/,
                    int nk, uint8_t *const *masks_out /*host array of dev ptrs*/,
                    void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);
int salun_mask_topk_ex(const float *acc /*dev*/, int64_t n, const int64_t *ks /*host*/,
                       int nk, uint8_t *const *masks_out /*host array of dev ptrs, or NULL*/,
                       void *ws /*dev*/, size_t ws_bytes, unsigned flags, salun_stream_t stream);
/* Synchronises the stream: which route finished the last call on this ws (1 = single-read, 2 = full scan) and
 * whether a grid barrier of the full scan timed out (*error_out = 1: the masks / thresholds of that call are INVALID).
 * The full scan synchronises its workgroups with a bounded software grid barrier; called directly (small or
 * unaligned inputs, SALUN_TOPK_FORCE_FULL_SCAN) it is launched cooperatively, so the runtime guarantees the
 * co-residency the barrier needs; as the fallback behind the single-read route it is a plain launch.  Either way a
 * time-out is never silent: callers that hand masks to a file or an optimizer check this status (one host sync, the
 * Python wrappers' `check=True`), callers that must not synchronise receive NaN from salun_mask_topk_thresholds. */
int salun_mask_topk_status(const void *ws /*dev*/, int *route_out /*host*/, int *error_out /*host*/,
                           salun_stream_t stream);
/* After salun_mask_topk on the same ws: copies the nk selected thresholds
 * (as fp32 |acc| values; NaN if the k-th element is a NaN OR if the select failed — see salun_mask_topk_status;
 * +inf for k <= 0, -1 for k > n) to a device array, 4*nk bytes.  Used by the proximal step (K9) as the
 * device-resident threshold. */
int salun_mask_topk_thresholds(const void *ws /*dev*/, int nk, float *tau_out /*dev*/,
                               salun_stream_t stream);

/* Mask format converters for the file boundary (reference masks are int64 0/1
 * tensors: `torch.zeros_like(tensor_ranks)`, generate_mask.py:76-79).
 * i64 -> u8 maps any non-zero to 1. */
int salun_mask_u8_to_i64(const uint8_t *m /*dev*/, int64_t *out /*dev*/, int64_t n,
                         salun_stream_t stream);
int salun_mask_i64_to_u8(const int64_t *m /*dev*/, uint8_t *out /*dev*/, int64_t n,
                         salun_stream_t stream);
/* Number of ones -> *count (dev, 1 int64). Workspace: salun_reduce_workspace_bytes(n). */
int salun_mask_popcount(const uint8_t *m /*dev*/, int64_t n, int64_t *count /*dev*/,
                        void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);

/* --------------------------------------------------------------- K3+K4 --
 * Masked SGD-momentum step, one launch for the whole model.
 * Replaces  _apply_mask_to_grads -> torch.optim.SGD.step -> _restore_masked_params
 *   Classification/unlearn/RL.py:11-14, unlearn/impl.py:68-73 (SGD, dampening 0,
 *   nesterov off), RL.py:17-34  (same helpers in GA.py/FT.py/boundary_*.py and
 *   trainer/train.py:58-61,104-107).
 * Where m[i]==1 (or m==NULL: unmasked plain SGD):
 *     d   = fma(wd, p, g)                     (d = g when wd == 0)
 *     buf = first_step ? d : (mu*buf) + d     (mu*buf rounded, then added)
 *     p   = fma(-lr, buf, p)
 * Where m[i]==0:  p is left bit-identical (it equals theta0 by the invariant the
 * reference's restore enforces every step), buf = 0.
 * `mu == 0` means "no momentum": buf is neither read nor written, may be NULL.
 * lr/mu/wd are doubles (Python floats) rounded to fp32 once, as torch does.
 * Algorithmic traffic: 21 B / element (r p,g,buf,m ; w p,buf). */
int salun_masked_sgd_step(float *p /*dev*/, const float *g /*dev*/, float *buf /*dev*/,
                          const uint8_t *m /*dev or NULL*/, double lr, double mu,
                          double wd, int first_step, int64_t n, salun_stream_t stream);

/* ------------------------------------------------------------------ K5 --
 * Squared global L2 norm of the flat gradient: *out = sum_i g[i]^2  (fp32 result,
 * accumulated per thread in fp32 lanes and across threads/blocks in fp64, fixed
 * reduction tree => run-to-run deterministic).
 * Replaces the per-tensor norms of torch.nn.utils.clip_grad_norm_
 *   DDPM/runners/diffusion.py:582-587,985-990.
 * Algorithmic traffic: 4 B / element. */
size_t salun_reduce_workspace_bytes(int64_t n);
int salun_grad_sqnorm(const float *g /*dev*/, int64_t n, float *out /*dev*/,
                      void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);

/* Masked Adam step (clip -> mask -> Adam), one launch for the whole model.
 * Replaces  clip_grad_norm_ -> `param.grad *= mask[name].to(device)` -> Adam.step
 *   DDPM/runners/diffusion.py:582-593 with DDPM/functions/__init__.py:9-18
 *   (Adam, amsgrad off), SD/train-scripts/random_label.py:129-139,
 *   SD/train-scripts/nsfw_removal.py:150-160.
 *     s   = sqnorm ? min(1, max_norm/(sqrt(*sqnorm)+1e-6)) : gscale
 *     ge  = (g*s) * m                 (m==NULL: ge = g*s)
 *     ge  = fma(wd, p, ge)            (only when wd != 0)
 *     m1  = (b1*m1) + ((1-b1)*ge)
 *     v   = (b2*v)  + ((1-b2)*ge)*ge
 *     den = sqrt(v)/sqrt(1-b2^step) + eps
 *     p   = p + (-(lr/(1-b1^step))) * (m1/den)
 * Hyper-parameters are doubles (Python floats); 1-b1, 1-b2, the bias corrections
 * and lr/(1-b1^step) are evaluated on the host in double and only then rounded to
 * fp32, like the Python scalars torch.optim.Adam hands to its fp32 kernels.
 * step is 1-based.  Algorithmic traffic: 29 B / element. */
int salun_masked_adam_step(float *p /*dev*/, const float *g /*dev*/, float *m1 /*dev*/,
                           float *v /*dev*/, const uint8_t *mask /*dev or NULL*/,
                           const float *sqnorm /*dev or NULL*/, double max_norm,
                           double gscale, double lr, double b1, double b2, double eps,
                           double wd, int step, int64_t n, salun_stream_t stream);

/* The same step with its step-dependent scalars taken from DEVICE memory — for callers that replay
 * identical kernel arguments every step (the host cannot pass t):
 *   salun_adam_coefficients: ++(*step) on the stream, then coef = { sqrt(1 - b2^t), -lr / (1 - b1^t) } in the same
 *     double arithmetic as salun_masked_adam_step;
 *   salun_masked_adam_step_coef: salun_masked_adam_step reading those two floats instead of (lr, step). */
int salun_adam_coefficients(int64_t *step /*dev*/, double lr, double b1, double b2, float *coef /*dev [2]*/,
                            salun_stream_t stream);
int salun_masked_adam_step_coef(float *p /*dev*/, const float *g /*dev*/, float *m1 /*dev*/, float *v /*dev*/,
                                const uint8_t *mask /*dev or NULL*/, const float *sqnorm /*dev or NULL*/,
                                double max_norm, double gscale, const float *coef /*dev [2]*/, double b1, double b2,
                                double eps, double wd, int64_t n, salun_stream_t stream);

/* ------------------------------------------------------------------ K6 --
 * Forward diffusion sample:  xt = x0*sqrt_ab[t[b]] + e*sqrt_1mab[t[b]]
 * Replaces  DDPM/functions/losses.py:31-32, runners/diffusion.py:558-559,973-974.
 * x0,e,xt: (B, chw) row-major; sqrt_ab/sqrt_1mab: (T,) tables; t: (B,) int64. */
int salun_qsample(const float *x0 /*dev*/, const float *e /*dev*/,
                  const float *sqrt_ab /*dev*/, const float *sqrt_1mab /*dev*/,
                  const int64_t *t /*dev*/, int64_t T, float *xt /*dev*/, int64_t B,
                  int64_t chw, salun_stream_t stream);

/* Squared-error loss + gradient in one pass over (B, chw) tensors a and b:
 *     per_sample[s] = sum_j (a[s,j]-b[s,j])^2          (optional output)
 *     *loss         = coef * sum_s per_sample[s]
 *     dloss_db[s,j] = -2*coef*(a[s,j]-b[s,j])           (optional output)
 * eps-MSE  `(e-out).square().sum((1,2,3)).mean(0)`  = (a=e, b=out, coef=1/B)
 *   DDPM/functions/losses.py:34-37, runners/diffusion.py:980;
 * nn.MSELoss(pseudo, out) = (a=pseudo, b=out, coef=1/(B*chw))
 *   runners/diffusion.py:507,570, SD random_label.py:113-127.
 * Reduction order is fixed (deterministic).  Workspace: salun_sqerr_workspace_bytes(B, chw). */
size_t salun_sqerr_workspace_bytes(int64_t B, int64_t chw);
int salun_sqerr_loss(const float *a /*dev*/, const float *b /*dev*/, int64_t B,
                     int64_t chw, double coef, float *loss /*dev*/,
                     float *per_sample /*dev or NULL*/, float *dloss_db /*dev or NULL*/,
                     void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);

/* ------------------------------------------------------------------ K7 --
 * Diagonal empirical Fisher accumulation:  F += (tmp*tmp)/n_data ; tmp = 0.
 * Replaces  DDPM/runners/diffusion.py:176-183
 *   (`fisher += tmp**2 / len(dataset)`, then tmp re-zeroed).   16 B / element. */
int salun_fim_square_accumulate(float *F /*dev*/, float *tmp /*dev*/, double n_data,
                                int64_t n, salun_stream_t stream);

/* ------------------------------------------------------------------ K8 --
 * fp32 2-D convolution on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains), NCHW
 * activations, OIHW weights read in place from the flat arena.  Replaces the library convolution calls
 * autograd issues for `nn.Conv2d` forward / backward in the models on the path
 *   (Classification/models/ResNet.py:58-74,218-220; DDPM/models/diffusion.py:54-56,71-73,103-122,154-165,
 *    245-247,325-327) — see DESIGN.md §3 for why (the library picks naive kernels for these fp32 shapes).
 * Supported: square filter R in {1,3}, stride in {1,2}, dilation 1, groups 1; `pad` is the low-side
 * (top/left) padding, the high side is implied by P,Q; Q (forward / weight) resp. W (backward-data) must be
 * a power of two <= 128 and the pixel space must tile (see csrc/salun_conv.hip make_geom).  Unsupported
 * shapes return SALUN_EINVAL and the caller uses the library convolution.
 *   forward        y[N,K,P,Q]  = conv(x[N,C,H,W], w[K,C,R,R]) (+ bias[K] if non-NULL)
 *   backward_data  dx[N,C,H,W] = conv_transpose(dy[N,K,P,Q], w)
 *   backward_weight dw[K,C,R,R] (= or += if accumulate) sum over n,p,q of dy * x; deterministic
 *                  (pixel range split over workgroups -> partials in `ws` -> fixed-order reduce). */
int salun_conv2d_forward(const float *x /*dev*/, const float *w /*dev*/, const float *bias /*dev or NULL*/,
                         float *y /*dev*/, int N, int C, int H, int W, int K, int R, int stride, int pad,
                         int P, int Q, salun_stream_t stream);
/* Forward with the epilogue of a diffusion ResnetBlock (DDPM/models/diffusion.py:113-127 of the reference:
 * `h = conv1(..); h = h + temb_proj(..)[:, :, None, None]` and `return x + h`) folded in:
 *   y = conv2d(x, w) + bias[k] + nbias[n][k] + addend[n][k][p][q]      (each term optional, added in that order —
 * the order the reference's separate adds produce).  addend must not alias y.  nbias / addend are carried by dedicated
 * instantiations of the kernel (stride 1, C a multiple of the staging chunk: 8, or 32 for R = 1, 16-byte aligned
 * weights); any other shape with them returns SALUN_EINVAL. */
int salun_conv2d_forward_fused(const float *x /*dev*/, const float *w /*dev*/, const float *bias /*dev or NULL*/,
                               const float *nbias /*dev [N,K] or NULL*/, const float *addend /*dev [N,K,P,Q] or NULL*/,
                               float *y /*dev*/, int N, int C, int H, int W, int K, int R, int stride, int pad,
                               int P, int Q, void *ws /*dev or NULL*/, size_t ws_bytes, salun_stream_t stream);
/* Under-filled launches (an output small enough for <= 256 workgroups, e.g. the 4x4 level of the DDPM U-Net: half the
 * CUs idle) are split over the reduction channels when a workspace is passed: S <= 8 workgroups per output tile write
 * partial images, a second small kernel adds them in fixed order together with the epilogue terms (deterministic; the
 * rounding differs from the unsplit launch's).  salun_conv2d_data_workspace_bytes: bytes for an OUTPUT of
 * [N, outC, outH, outW] (forward: K, P, Q and the forward stride; backward-data: C, H, W and conv_stride 1); 0 when
 * the launch would not be split.  ws == NULL: never split. */
size_t salun_conv2d_data_workspace_bytes(int N, int outC, int outH, int outW, int R, int conv_stride);
int salun_conv2d_backward_data_ws(const float *dy /*dev*/, const float *w /*dev*/, const float *addend /*dev or NULL*/,
                                  float *dx /*dev*/, int N, int C, int H, int W, int K, int R, int stride, int pad,
                                  int P, int Q, void *ws /*dev or NULL*/, size_t ws_bytes, salun_stream_t stream);
int salun_conv2d_backward_data(const float *dy /*dev*/, const float *w /*dev*/, float *dx /*dev*/, int N, int C,
                               int H, int W, int K, int R, int stride, int pad, int P, int Q,
                               salun_stream_t stream);
/* dx = backward_data(dy, w) + addend  (addend: [N,C,H,W] or NULL; addend == dx accumulates in place).  The residual
 * branch's gradient of a ResNet block is folded into the convolution's epilogue instead of a separate add pass. */
int salun_conv2d_backward_data_add(const float *dy /*dev*/, const float *w /*dev*/, const float *addend /*dev or NULL*/,
                                   float *dx /*dev*/, int N, int C, int H, int W, int K, int R, int stride, int pad,
                                   int P, int Q, salun_stream_t stream);
/* Bias gradient of a convolution: out[k] (= or +=, `accumulate`) sum over n, p, q of dy[n][k][p][q] — what the reference
 * gets from autograd's `sum(dim=(0, 2, 3))` + AccumulateGrad for every biased `nn.Conv2d` of the diffusion U-Nets
 * (DDPM/models/diffusion.py:20-75, SD openaimodel.py `conv_nd`).  One streaming pass, 4 B per element of dy;
 * deterministic (fixed-order partial sums).  `out` may be the parameter's slice of the flat gradient arena. */
size_t salun_channel_sum_workspace_bytes(int N, int K);
int salun_channel_sum(const float *dy /*dev*/, float *out /*dev, K floats*/, int N, int K, int HW, int accumulate,
                      void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);
size_t salun_conv2d_wgrad_workspace_bytes(int N, int C, int K, int R, int P, int Q);
int salun_conv2d_backward_weight(const float *x /*dev*/, const float *dy /*dev*/, float *dw /*dev*/, int N, int C,
                                 int H, int W, int K, int R, int stride, int pad, int P, int Q, int accumulate,
                                 void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);
/* The same with scheduling hints.  SALUN_WGRAD_SHARED: the launch runs on a side stream BESIDE other kernels of the
 * step (resblock.py overlaps backward-weight with backward-data and the normalisation backward): 3x3 / stride 1 layers
 * then keep the register-staged kernel; without the flag they run on the LDS-DMA ring kernel (csrc/salun_conv_ring.hip:
 * 12 - 18 % faster alone).  Both are deterministic; their results differ by the summation order over the pixels. */
#define SALUN_WGRAD_SHARED 1u
int salun_conv2d_backward_weight_ex(const float *x /*dev*/, const float *dy /*dev*/, float *dw /*dev*/, int N, int C,
                                    int H, int W, int K, int R, int stride, int pad, int P, int Q, int accumulate,
                                    unsigned flags, void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);

/* ------------------------------------------------------------------ K8r --
 * 3x3 / stride 1 / pad 1 fp32 convolution, forward and backward-data, with both operands fed by LDS-DMA into a
 * two-stage LDS ring (csrc/salun_conv_ring.hip; round 6) — the same arithmetic as salun_conv2d_forward /
 * salun_conv2d_backward_data for these shapes (bit-identical results: same exact-fp32 FMA chains in the same order),
 * i.e. the BasicBlock convolutions of Classification/models/ResNet.py:77-125 and the ResnetBlock convolutions of
 * DDPM/models/diffusion.py:85-145.  The weights are read from a packed IMAGE of the OIHW tensor (one per direction),
 * rebuilt by salun_conv3x3_pack_weights whenever the parameters changed (once per optimizer step, all layers in one
 * launch):
 *   dgrad = 0: rows = output channels, reduction over input channels   (forward)
 *   dgrad = 1: rows = input channels, reduction over output channels, taps flipped   (backward-data)
 * salun_conv3x3_pack_bytes: size of an image (0: the reduction channel count is not a multiple of 8 — not packable).
 * salun_conv3x3_packed:  y[N,Kout,H,W] = conv3x3(x[N,Cred,H,W], image) (+ bias[Kout]) (+ nbias[N,Kout])
 *   (+ addend[N,Kout,H,W], may alias y); forward: x = input, Cred = C, Kout = K, image dgrad = 0; backward-data:
 *   x = dy, Cred = K, Kout = C, image dgrad = 1.  W in {4, 8, 16, 32}, the pixel space must tile into whole rows /
 *   whole images (SALUN_EINVAL otherwise: the caller uses salun_conv2d_*).  `cfg`: 0 = tile chosen from the problem;
 *   low byte 1..5 pins a tile, bits 8.. the workgroups per CU of the persistent grid (tuning: tools/convring_bench.py). */
size_t salun_conv3x3_pack_bytes(int K, int C, int dgrad);
#define SALUN_PACK_MAX_JOBS 32
typedef struct {
  const float *w;   /* dev, OIHW [K, C, 3, 3] */
  float *img_fwd;   /* dev, salun_conv3x3_pack_bytes(K, C, 0) bytes, or NULL (needs C % 8 == 0) */
  float *img_dgrad; /* dev, salun_conv3x3_pack_bytes(K, C, 1) bytes, or NULL (needs K % 8 == 0) */
  int32_t K, C;
} salun_pack_job_t;
/* Both images of `njobs` layers; one launch per SALUN_PACK_MAX_JOBS layers (`jobs` is a HOST array). */
int salun_conv3x3_pack_weights(const salun_pack_job_t *jobs /*host*/, int njobs, salun_stream_t stream);
int salun_conv3x3_packed(const float *x /*dev*/, const float *img /*dev*/, const float *bias /*dev or NULL*/,
                         const float *nbias /*dev or NULL*/, const float *addend /*dev or NULL*/, float *y /*dev*/,
                         int N, int Cred, int H, int W, int Kout, int cfg, salun_stream_t stream);

/* ------------------------------------------------------------------ K11 --
 * bf16 2-D convolution on the matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulators): the convolutions of the
 * Stable-Diffusion U-Net in its bf16 configuration (BASELINE.json configs[4]) — replaces the library calls autograd
 * issues under autocast for the `conv_nd(2, ...)` modules of
 *   SD/ldm/modules/diffusionmodules/openaimodel.py:98-103 (Upsample), :131-133 (Downsample), :192-231 (ResBlock),
 *   :466-470,:716-720 (input / output heads) and SD/ldm/modules/attention.py:230-247 (proj_in / proj_out).
 * Activations are NHWC bf16 (`uint16_t` = raw bf16 bits): x[N][H][W][C], y[N][OH][OW][K]; the master weights stay
 * fp32 OIHW in the flat arena and `salun_conv2d_bf16_pack_weights` writes the bf16 image wp[K][R*R][C] both
 * data kernels read (backward-data reads it transposed, no second image).  Square filter R in {1,3}, stride in
 * {1,2}, symmetric padding <= R-1, C % 32 == 0, K % 32 == 0, 16-B aligned pointers;
 * anything else returns SALUN_EINVAL (`salun_conv2d_bf16_supported` answers without launching).
 *   forward         y = conv(x, w) + bias[k] + nbias[n][k] + addend[n][oh][ow][k]   (each optional, fp32 / fp32 / bf16)
 *   backward_data   dx = conv_transpose(dy, w) + addend[n][h][w][c]
 *   backward_weight dw[K][C][R][R] fp32 (= or +=): sum over pixels of dy * x, pixel range split over workgroups ->
 *                   fp32 partials in `ws` -> fixed-order reduce (deterministic); db[K] fp32 (optional) = sum of dy. */
int salun_conv2d_bf16_supported(int C, int K, int R, int stride, int pad);
int salun_conv2d_bf16_pack_weights(const float *w /*dev, OIHW*/, uint16_t *wp /*dev*/, int K, int C, int R,
                                   salun_stream_t stream);
/* Every stale weight image of a model in one launch per SALUN_BF16_PACK_MAX_JOBS layers (after an optimizer step all of
 * them are stale; the reference has no counterpart — autocast re-casts each weight on every use,
 * SD/train-scripts/nsfw_removal.py:96-150 under torch.autocast).  `transposed` (R == 1 only): the [C][K] image the Linear
 * layers' input-gradient GEMM reads (what salun_pack_bf16(..., transposed = 1) writes); otherwise wp[K][R*R][C] as above.
 * `jobs` is host memory, read during the call. */
#define SALUN_BF16_PACK_MAX_JOBS 64
typedef struct {
  const float *w;  /* dev, fp32 OIHW [K][C][R][R] */
  uint16_t *wp;    /* dev, bf16 image */
  int32_t K, C, R, transposed;
} salun_bf16_pack_job_t;
int salun_bf16_pack_weights_batch(const salun_bf16_pack_job_t *jobs, int n, salun_stream_t stream);
/* `ws` of forward / backward_data (salun_conv2d_bf16_data_workspace_bytes, may be 0): problems with few output tiles
 * split their reduction over workgroups through fp32 partial tiles there; NULL / too small only disables the split. */
size_t salun_conv2d_bf16_data_workspace_bytes(int N, int H, int W, int C, int K, int R, int stride, int pad);
int salun_conv2d_bf16_forward(const uint16_t *x /*dev*/, const uint16_t *wp /*dev*/, const float *bias /*dev or NULL*/,
                              const float *nbias /*dev or NULL*/, const uint16_t *addend /*dev or NULL*/,
                              uint16_t *y /*dev*/, int N, int H, int W, int C, int K, int R, int stride, int pad,
                              void *ws /*dev or NULL*/, size_t ws_bytes, salun_stream_t stream);
int salun_conv2d_bf16_backward_data(const uint16_t *dy /*dev*/, const uint16_t *wp /*dev*/,
                                    const uint16_t *addend /*dev or NULL*/, uint16_t *dx /*dev*/, int N, int H, int W,
                                    int C, int K, int R, int stride, int pad, void *ws /*dev or NULL*/, size_t ws_bytes,
                                    salun_stream_t stream);
size_t salun_conv2d_bf16_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int stride, int pad);
int salun_conv2d_bf16_backward_weight(const uint16_t *x /*dev*/, const uint16_t *dy /*dev*/, float *dw /*dev*/,
                                      float *db /*dev or NULL*/, int N, int H, int W, int C, int K, int R, int stride,
                                      int pad, int accumulate, void *ws /*dev*/, size_t ws_bytes,
                                      salun_stream_t stream);
/* The same with `dnb` (dev fp32 [N][K] or NULL, N <= 128, always overwritten): the per-image channel sums of dy — the
 * gradient of the forward's `nbias[n][k]` term (a ResBlock's time-embedding projection,
 * SD/ldm/modules/diffusionmodules/openaimodel.py:249-263: `h = h + emb_out[..., None, None]`), from the partial sums the
 * bias gradient is folded from anyway (no fp32 copy of dy, no separate reduction).  db may be NULL then. */
/* The column sums of dy[M][K] (bf16) alone: db[K] (fp32, = or +=; or NULL) and / or dnb[images][K] (fp32, overwritten; or
 * NULL; M = images * pixels, images <= 128) — what the two calls above fold next to dw; for callers that want the
 * per-image sums on another stream than the weight gradient (conv_bf16.py: backward-weight on the side stream, the
 * time-embedding gradient on the compute stream). */
size_t salun_colsum_bf16_workspace_bytes(int K);
int salun_colsum_bf16(const uint16_t *dy /*dev*/, float *db /*dev or NULL*/, float *dnb /*dev or NULL*/, int64_t M, int K,
                      int images, int accumulate, void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);
int salun_conv2d_bf16_backward_weight_ex(const uint16_t *x /*dev*/, const uint16_t *dy /*dev*/, float *dw /*dev*/,
                                         float *db /*dev or NULL*/, float *dnb /*dev or NULL*/, int N, int H, int W, int C,
                                         int K, int R, int stride, int pad, int accumulate, void *ws /*dev*/,
                                         size_t ws_bytes, salun_stream_t stream);

/* ------------------------------------------------------------------ K12 --
 * GroupNorm (+ SiLU) on bf16 NHWC activations with fp32 statistics — the `GroupNorm32` layers of the SD U-Net in its
 * bf16 configuration (SD/ldm/modules/diffusionmodules/util.py:215-217: `super().forward(x.float()).type(x.dtype)`,
 * followed by nn.SiLU in openaimodel.py:192-196,214-221,716-718; attention.py:228 without SiLU).
 *   forward : y = [silu]( gamma*(x-mean)*rstd + beta ), x / y [N][HW][C] bf16, C % 8 == 0, C % G == 0;
 *             mr[N][G][2] = (mean, rstd) and ab[N][C][2] = (gamma*rstd, beta - mean*gamma*rstd) are outputs the
 *             backward consumes (biased variance, as torch.nn.GroupNorm).
 *   backward: dz = dy * silu'(z); dbeta = sum dz; dgamma = sum dz*xhat; dx = rstd*(dz*gamma - (s1 + xhat*s2)/m)
 *             with s1 = sum_group dz*gamma, s2 = sum_group dz*gamma*xhat.  dgamma / dbeta fp32 (= or += if accumulate).
 * Reductions: per-channel fp32 partials over pixel chunks, folded in fp64 in a fixed order (deterministic). */
size_t salun_gn_bf16_workspace_bytes(int N, int C, int HW, int G);
int salun_gn_bf16_forward(const uint16_t *x /*dev*/, const float *gamma /*dev*/, const float *beta /*dev*/,
                          uint16_t *y /*dev*/, float *mr /*dev*/, float *ab /*dev*/, int N, int C, int HW, int G,
                          double eps, int silu, void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);
int salun_gn_bf16_backward(const uint16_t *dy /*dev*/, const uint16_t *x /*dev*/, const float *gamma /*dev*/,
                           const float *mr /*dev*/, const float *ab /*dev*/, uint16_t *dx /*dev*/, float *dgamma /*dev*/,
                           float *dbeta /*dev*/, int N, int C, int HW, int G, int silu, int accumulate, void *ws /*dev*/,
                           size_t ws_bytes, salun_stream_t stream);

/* ------------------------------------------------------------------ K13 --
 * Fused scaled-dot-product attention on bf16 tokens (fp32 softmax / accumulation) — replaces
 * `CrossAttention.forward` of SD/ldm/modules/attention.py:168-192 (q.k^T * scale -> softmax -> @ v, which materialises
 * the (B*heads) x Nq x Nk score tensor) for the SD U-Net's self-attention (4096 / 1024 / 256 / 64 tokens) and
 * cross-attention (77 text tokens), 8 heads of D = 40 / 80 / 160 channels (D in {8,16,32,40,64,80,160} supported).
 * Tensors are [B, tokens, H, D] VIEWS: element (b, t, h, d) at b*bs + t*ld + h*D + d (elements); every (token, head)
 * row must be 16-byte aligned.  `lse` ([B*H][Nq] fp32, log2-domain logsumexp of the scaled scores) is the forward's
 * second output and the backward's input; `dsum` is [B*H][Nq] fp32 scratch; dq / dk / dv are written contiguous.
 * `scale` must be positive and finite (SALUN_EINVAL otherwise): the kernels take the running maximum on the raw
 * scores and apply scale*log2(e) afterwards; every attention of the three models uses 1/sqrt(D). */
int salun_attn_supported(int D);
int salun_attn_forward(const uint16_t *q /*dev*/, const uint16_t *k /*dev*/, const uint16_t *v /*dev*/, uint16_t *o /*dev*/,
                       float *lse /*dev or NULL*/, int B, int H, int Nq, int Nk, int D, long long q_bs, int q_ld,
                       long long k_bs, int k_ld, long long v_bs, int v_ld, long long o_bs, int o_ld, double scale,
                       salun_stream_t stream);
int salun_attn_backward(const uint16_t *q /*dev*/, const uint16_t *k /*dev*/, const uint16_t *v /*dev*/,
                        const uint16_t *o /*dev*/, const uint16_t *d_o /*dev*/, const float *lse /*dev*/,
                        uint16_t *dq /*dev*/, uint16_t *dk /*dev*/, uint16_t *dv /*dev*/, float *dsum /*dev*/, int B, int H,
                        int Nq, int Nk, int D, long long q_bs, int q_ld, long long k_bs, int k_ld, long long v_bs, int v_ld,
                        long long o_bs, int o_ld, long long do_bs, int do_ld, double scale, salun_stream_t stream);

/* ------------------------------------------------------------------ K14 --
 * Token-wise layers of the SD transformer blocks on bf16 tokens, fp32 arithmetic, one pass each:
 *   LayerNorm  (SD/ldm/modules/attention.py:196-216, `nn.LayerNorm(dim)` x3 per BasicTransformerBlock):
 *     forward  y[rows][C] = gamma*(x-mean)*rstd + beta, stats[rows][2] = (mean, rstd) for the backward
 *     backward dx, dgamma / dbeta (fp32, = or +=; per-workgroup partials in `ws`, folded in a fixed order)
 *   GEGLU      (attention.py:37-46: `x, gate = proj(x).chunk(2, dim=-1); x * F.gelu(gate)`, erf form):
 *     forward  out[rows][F] = h[:, :F] * gelu(h[:, F:]);  backward dh[rows][2F] from h and d(out).
 * C % 8 == 0, C <= 2048, F % 8 == 0; x / y / h / out 16-byte aligned; gamma / beta may be any fp32 slice. */
size_t salun_ln_bf16_workspace_bytes(int64_t rows, int C);
int salun_ln_bf16_forward(const uint16_t *x /*dev*/, const float *gamma /*dev*/, const float *beta /*dev*/,
                          uint16_t *y /*dev*/, float *stats /*dev or NULL*/, int64_t rows, int C, double eps,
                          salun_stream_t stream);
int salun_ln_bf16_backward(const uint16_t *dy /*dev*/, const uint16_t *x /*dev*/, const float *gamma /*dev*/,
                           const float *stats /*dev*/, uint16_t *dx /*dev*/, float *dgamma /*dev*/, float *dbeta /*dev*/,
                           int64_t rows, int C, int accumulate, void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);
int salun_geglu_bf16_forward(const uint16_t *h /*dev*/, uint16_t *out /*dev*/, int64_t rows, int F, salun_stream_t stream);
int salun_geglu_bf16_backward(const uint16_t *h /*dev*/, const uint16_t *dy /*dev*/, uint16_t *dh /*dev*/, int64_t rows,
                              int F, salun_stream_t stream);

/* Fused BatchNorm2d (+ residual add) (+ ReLU), NCHW fp32, forward and backward — replaces the
 * bn -> relu / bn -> (+identity) -> relu chains of the classifier blocks
 *   (Classification/models/ResNet.py:108-125,307-309) that run as separate library launches.
 * Semantics of torch.nn.BatchNorm2d: training -> batch statistics (biased variance), running stats updated with
 * `momentum` (unbiased variance); eval -> running statistics.  HW must be a multiple of 4, pointers 16-B aligned.
 *   forward : y = [relu]( gamma*(x-mean)*invstd + beta [+ res] ); save_mean / save_invstd (C floats) are outputs
 *   backward: dz = dy*[y>0] (relu) ; dbeta = sum dz ; dgamma = sum dz*xhat ;
 *             dx = gamma*invstd*(dz - (dbeta + xhat*dgamma)/(N*HW))   (training)   |   gamma*invstd*dz   (eval)
 *             dres = dz  (if non-NULL: gradient of the residual input)
 * Reductions are per-(channel, batch-slice) fp64 partials folded in a fixed order: deterministic.
 * Workspace: salun_bn_workspace_bytes(C). */
size_t salun_bn_workspace_bytes(int C);
int salun_bn_forward(const float *x /*dev*/, const float *res /*dev or NULL*/, float *y /*dev*/,
                     const float *gamma /*dev*/, const float *beta /*dev*/, float *running_mean /*dev or NULL*/,
                     float *running_var /*dev or NULL*/, long long *num_batches_tracked /*dev or NULL: += 1 if training*/,
                     float *save_mean /*dev*/, float *save_invstd /*dev*/,
                     int N, int C, int HW, int training, double momentum, double eps, int relu, void *ws /*dev*/,
                     size_t ws_bytes, salun_stream_t stream);
int salun_bn_backward(const float *dy /*dev*/, const float *y /*dev, needed if relu*/, const float *x /*dev*/,
                      const float *gamma /*dev*/, const float *save_mean /*dev*/, const float *save_invstd /*dev*/,
                      float *dx /*dev*/, float *dres /*dev or NULL*/, float *dgamma /*dev*/, float *dbeta /*dev*/,
                      float *grad_gamma_acc /*dev or NULL: += dgamma*/, float *grad_beta_acc /*dev or NULL: += dbeta*/,
                      int N, int C, int HW, int training, int relu, void *ws /*dev*/, size_t ws_bytes,
                      salun_stream_t stream);

/* Fused GroupNorm (+ SiLU), NCHW fp32, forward and backward — replaces the `nonlinearity(self.norm1(h))` chains of the
 * diffusion U-Nets (DDPM/models/diffusion.py:36-41,118-128,340-352; SD ldm/modules/diffusionmodules/openaimodel.py
 * ResBlock in_layers / out_layers) that run as ~6 library launches forward and ~9 backward per site.
 *   forward : z = [silu]( gamma_c * (x - mean_{n,g}) * rstd_{n,g} + beta_c ),  rstd = 1/sqrt(var_biased + eps);
 *             save_mean / save_rstd (N*G floats) are outputs
 *   backward: dy = dz * silu'(y) with y recomputed from x;  dgamma_c = sum_{n,hw} dy*xhat,  dbeta_c = sum dy;
 *             dx = rstd * (dy*gamma - mean_g(dy*gamma) - xhat * mean_g(dy*gamma*xhat))
 * One workgroup per (image, group); HW must be a power of two >= 4, C % G == 0, pointers 16-B aligned.
 * Reductions are in fixed order (deterministic).  Workspace (backward): salun_gn_workspace_bytes(N, C). */
size_t salun_gn_workspace_bytes(int N, int C);
int salun_gn_forward(const float *x /*dev*/, float *y /*dev*/, const float *gamma /*dev*/, const float *beta /*dev*/,
                     float *save_mean /*dev*/, float *save_rstd /*dev*/, int N, int C, int HW, int G, double eps,
                     int silu, salun_stream_t stream);
int salun_gn_backward(const float *dz /*dev*/, const float *x /*dev*/, const float *gamma /*dev*/,
                      const float *beta /*dev*/, const float *save_mean /*dev*/, const float *save_rstd /*dev*/,
                      float *dx /*dev*/, float *dgamma /*dev*/, float *dbeta /*dev*/,
                      float *grad_gamma_acc /*dev or NULL: += dgamma*/, float *grad_beta_acc /*dev or NULL*/,
                      int N, int C, int HW, int G, int silu, void *ws /*dev*/, size_t ws_bytes,
                      salun_stream_t stream);

/* Backward with the neighbours of a diffusion ResnetBlock folded in: dx (+= addend: the skip branch's gradient,
 * [N,C,HW] or NULL, must not alias dx); nk_sum (N*C floats or NULL) = sum over hw of the dx written, per (image,
 * channel) — the gradient of the per-image channel bias added by the convolution that produced x (the embedding
 * projection); csum / csum_acc (C floats or NULL; need nk_sum) = / += sum over n of nk_sum — that convolution's
 * bias gradient.  All reductions in fixed order. */
int salun_gn_backward_fused(const float *dz /*dev*/, const float *x /*dev*/, const float *gamma /*dev*/,
                            const float *beta /*dev*/, const float *save_mean /*dev*/, const float *save_rstd /*dev*/,
                            const float *addend /*dev or NULL*/, float *dx /*dev*/, float *dgamma /*dev*/,
                            float *dbeta /*dev*/, float *grad_gamma_acc /*dev or NULL*/,
                            float *grad_beta_acc /*dev or NULL*/, float *nk_sum /*dev or NULL*/,
                            float *csum /*dev or NULL*/, float *csum_acc /*dev or NULL*/, int N, int C, int HW, int G,
                            int silu, void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);

/* ------------------------------------------------------------------ K9 --
 * Proximal (soft-threshold) step of RL_proximal — Classification/unlearn/RL_pro.py:52-60:
 *     d = params - init_params ; threshold = -topk(-|d|, ratio)[0][-1] ;
 *     params = where(d > thr, params - thr, where(d < -thr, params + thr, init_params))
 * as three launches on the flat vectors:  salun_param_diff (out = p - p0, 12 B/elem)  ->  salun_mask_topk on `out`
 * with k = n - ratio + 1 (its k-th largest |d| IS the ratio-th smallest; threshold fetched on the device with
 * salun_mask_topk_thresholds)  ->  salun_soft_threshold_step (12 B/elem), which reads the threshold from device
 * memory — no host round trip.  fp32 arithmetic identical to the reference's tensor expressions.
 * A NaN threshold (a failed select) is propagated: every element of `p` becomes NaN, so the next loss is NaN — never
 * the silent "all weights reset to p0" the three-way comparison would otherwise produce. */
int salun_param_diff(const float *p /*dev*/, const float *p0 /*dev*/, float *out /*dev*/, int64_t n,
                     salun_stream_t stream);
int salun_soft_threshold_step(float *p /*dev*/, const float *p0 /*dev*/, const float *tau /*dev, 1 float*/,
                              int64_t n, salun_stream_t stream);

/* ----------------------------------------------------------------- K10 --
 * EWC (Selective-Amnesia) penalty of DDPM train_forget — DDPM/runners/diffusion.py:343-350:
 *     loss += lambda * sum_name sum( fisher[name] * (param - params_mle[name])**2 )
 * One launch over the flat vectors instead of 334 x 4 per-tensor ops + autograd:
 *     g += (lambda*F) * (2*(p - p_star))      (the term's autograd gradient, fp32)
 *     loss_out[0] = lambda * S, loss_out[1] = S = sum F (p - p_star)^2   (fp64 partials, fixed order)
 * Algorithmic traffic: 20 B / element (r p, p_star, F, g; w g).  Workspace: salun_ewc_workspace_bytes(n). */
size_t salun_ewc_workspace_bytes(int64_t n);
int salun_ewc_penalty_grad(const float *p /*dev*/, const float *p_star /*dev*/, const float *F /*dev*/,
                           float *g /*dev*/, double lambda, float *loss_out /*dev, 2 floats*/, int64_t n,
                           void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);

/* ------------------------------------------------------------------ K0 --
 * Device-resident CIFAR batch assembly (replaces the host DataLoader path
 * Classification/dataset.py:542-556 + main_random.py:38-48: PIL RandomCrop(32,4)
 * + RandomHorizontalFlip + ToTensor + H2D).
 * data: (num, H, W, C) uint8 resident in HBM; idx: (B,) int64 sample indices;
 * crop: (B,2) int32 (dy,dx) offsets in [0, 2*pad] or NULL (= centred, no crop);
 * flip: (B,) uint8 or NULL; out: (B, C, H, W) fp32 = pixel/255 (ToTensor),
 * zero padding outside the image. */
int salun_image_batch(const uint8_t *data /*dev*/, const int64_t *idx /*dev*/,
                      const int32_t *crop /*dev or NULL*/, const uint8_t *flip /*dev or NULL*/,
                      float *out /*dev*/, int64_t B, int H, int W, int C, int pad,
                      salun_stream_t stream);

/* ----------------------------------------------------------------- K15 --
 * fp32 GEMMs on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains; results differ from any other fp32
 * GEMM only by summation order) — the Linear layers and the fp32 attention of the diffusion U-Nets, which the
 * reference runs as library GEMMs behind nn.Linear / torch.bmm / einsum:
 *   DDPM/models/diffusion.py:85-145 (temb / cemb dense layers, temb_cemb_proj of every ResnetBlock), :148-192 (AttnBlock);
 *   SD/ldm/modules/attention.py:37-75,149-200, SD/ldm/modules/diffusionmodules/openaimodel.py:428-520 (fp32 configuration).
 *
 * One call computes a TABLE of problems
 *      C_j[M_j, N_j]  (+)=  alpha * sum_{s in segments(j)} A_s . B_s^T   + bias_j[column]
 * with  A_s(i, k) = A[i * a_i + k * a_k],  B_s(j, k) = B[j * b_j + k * b_k]  (element strides, any sign of "transposed"):
 *   x.W^T (forward), dY.W (input gradient), dY^T.x (weight gradient, `accumulate` adds into the flat gradient) and the
 *   channel-major attention operands of the DDPM are the same kernel without a transposing copy.  Several jobs in one
 *   call share the launch (the DDPM's 22 per-block embedding projections); several segments of one job chain their
 *   reductions (the input gradient of those projections: sum_g dproj_g . W_g).
 * Batch: batch_outer x batch_inner instances, instance (o, i) offsets every A / B / C pointer by
 *   o * strides[0|2|4] + i * strides[1|3|5] elements (a_bo a_bi b_bo b_bi c_bo c_bi); batch_strides may be NULL when
 *   there is one instance.
 * Under-filled launches split the reduction into fp32 partial images in `ws` (salun_gemm_f32_workspace_bytes; with a
 * smaller / NULL workspace the call still works, unsplit) folded in a fixed order: no float atomics, deterministic.
 * Limits: njobs <= SALUN_GEMM_MAX_JOBS, nsegs <= SALUN_GEMM_MAX_SEGS, batch <= 65535. */
#define SALUN_GEMM_MAX_JOBS 32
#define SALUN_GEMM_MAX_SEGS 32
typedef struct {
  const float *A /*dev*/, *B /*dev*/;
  int32_t K;                  /* reduction length of this segment */
  int32_t a_i, a_k, b_j, b_k; /* element strides */
} salun_gemm_seg_t;
typedef struct {
  float *C /*dev*/;
  const float *bias /*dev or NULL: one value per column*/;
  int32_t M, N, ldc;
  int32_t seg0, nseg;  /* this job's segments: segs[seg0 .. seg0 + nseg) */
  int32_t accumulate;  /* 0: C = result, 1: C += result */
} salun_gemm_job_t;
size_t salun_gemm_f32_workspace_bytes(const salun_gemm_job_t *jobs /*host*/, int njobs,
                                      const salun_gemm_seg_t *segs /*host*/, int nsegs,
                                      int batch_outer, int batch_inner);
int salun_gemm_f32(const salun_gemm_job_t *jobs /*host*/, int njobs, const salun_gemm_seg_t *segs /*host*/, int nsegs,
                   int batch_outer, int batch_inner, const int64_t *batch_strides /*host[6] or NULL*/,
                   double alpha, void *ws /*dev or NULL*/, size_t ws_bytes, salun_stream_t stream);
/* out[j] (+)= sum_i x[i * ld + j]: the bias gradient of a Linear layer (the reference: autograd's sum over the token
 * dimension behind nn.Linear).  Two deterministic stages (row slabs, then the slabs in index order). */
size_t salun_colsum_f32_workspace_bytes(int64_t M, int N);
int salun_colsum_f32(const float *x /*dev*/, float *out /*dev*/, int64_t M, int N, int64_t ld, int accumulate,
                     void *ws /*dev*/, size_t ws_bytes, salun_stream_t stream);
/* Row softmax in place: s[r][0..n) <- softmax(s[r][0..n)), rows of leading dimension ld (the P between the two GEMMs of
 * an fp32 attention: DDPM/models/diffusion.py:176, SD attention.py:185).  One wave per row. */
int salun_softmax_rows(float *s /*dev*/, int64_t rows, int n, int ld, salun_stream_t stream);
/* Its backward, written over dp:  dS = scale * P * (dP - sum_j dP_j P_j)  (scale: the attention's 1/sqrt(d), so the two
 * gradient GEMMs that follow need no extra pass). */
int salun_softmax_rows_backward(const float *p /*dev*/, float *dp /*dev*/, int64_t rows, int n, int ld,
                                double scale, salun_stream_t stream);

/* ----------------------------------------------------------------- K16 --
 * bf16 GEMM  y[M, N] = x[M, K] . w[N, K]^T (+ bias[N] fp32) (+ addend[M, N] bf16)  on v_mfma_f32_32x32x16_bf16, fp32
 * accumulation, one rounding to bf16 — the Linear layers of the SD transformer blocks in the bf16 configuration
 * (reference: autocast over SD/ldm/modules/attention.py:37-75 GEGLU / FeedForward, :149-200 to_q / to_k / to_v / to_out).
 * x, w, addend, y: bf16 row-major; both operand tiles go global -> LDS directly (global_load_lds_dwordx4, bank swizzle on
 * the source address).  The input gradient dX = dY . W is the same call on the transposed image (salun_pack_bf16 with
 * transposed = 1).  Requirements: K % 64 == 0, N % 64 == 0, 16-byte aligned pointers (salun_gemm_bf16_supported);
 * SALUN_EINVAL otherwise.  variant: 0 = choose the tile, 1..4 = pin one (A/B measurements). */
int salun_gemm_bf16_supported(int64_t M, int N, int K);
int salun_gemm_bf16_nt(const void *x /*dev bf16*/, const void *w /*dev bf16*/, const float *bias /*dev or NULL*/,
                       const void *addend /*dev bf16 or NULL*/, void *y /*dev bf16*/, int64_t M, int N, int K,
                       int variant, salun_stream_t stream);
/* Weight gradient of the same layers (and of 1x1 stride-1 convolutions: salun_conv2d_bf16_backward_weight routes them
 * here):  dw[Na, Nb] fp32 (+= when accumulate) = dy[M, Na]^T . x[M, Nb]  (reference: autograd of F.linear / F.conv2d under
 * the layers above).  The reduction index M is the slow axis of both operands: tiles go global -> LDS directly and the
 * MFMA operands are gathered with transposing LDS reads.  Split over M into partials in `ws`
 * (salun_gemm_bf16_tn_workspace_bytes) folded in index order — deterministic.  Na, Nb multiples of 32, M < 2^24.
 * variant: 0 = choose, 1..3 = pin a tile (A/B measurements). */
int salun_gemm_bf16_tn_supported(int64_t M, int Na, int Nb);
size_t salun_gemm_bf16_tn_workspace_bytes(int64_t M, int Na, int Nb, int variant);
int salun_gemm_bf16_tn(const void *dy /*dev bf16*/, const void *x /*dev bf16*/, float *dw /*dev*/, int64_t M, int Na, int Nb,
                       int accumulate, int variant, void *ws /*dev or NULL*/, size_t ws_bytes, salun_stream_t stream);
/* fp32 master weights w[N][K] -> the bf16 image the GEMM reads: [N][K] (transposed = 0) or [K][N] (transposed = 1).
 * Once per optimizer step per layer. */
int salun_pack_bf16(const float *w /*dev*/, void *wp /*dev bf16*/, int N, int K, int transposed, salun_stream_t stream);

/* Dropout whose keep decision is a function of (seed, GLOBAL element index) only — replaces
 * `nn.Dropout` in the DDPM ResnetBlock (DDPM/models/diffusion.py:108,124: `h = self.dropout(h)`), which under the
 * reference's nn.DataParallel draws independently per replica (runners/diffusion.py:504).
 *   gi   = (sample_offset + s) * chw + j                      global element index of sample s, position j
 *   bits = splitmix64(key + (gi >> 1)); even gi: bits >> 40, odd gi: (bits >> 8) & 0xFFFFFF
 *   y    = bits >= round(p * 2^24) ? x * (float)(1 / (1 - p)) : 0
 * key = seed + (seed_dev ? *seed_dev : 0)  (seed_dev: optional device word a captured graph advances with
 * salun_u64_add, so that replays draw fresh masks).  A rank holding samples [lo, hi) of a global batch passes
 * sample_offset = lo and gets rows lo..hi-1 of the single-process result; the backward pass is the same call on dy
 * (no mask is stored).  x, y: (n_samples, chw) fp32, y may alias x.  8 B / element. */
int salun_dropout(const float *x /*dev*/, float *y /*dev*/, int64_t n_samples, int64_t chw,
                  int64_t sample_offset, double p, uint64_t seed,
                  const uint64_t *seed_dev /*dev or NULL*/, salun_stream_t stream);
/* *value += inc on the stream (advances a device-resident seed between graph replays). */
int salun_u64_add(uint64_t *value /*dev*/, uint64_t inc, salun_stream_t stream);

/* Counter-based generators shared bit-for-bit with oracle/ (splitmix64 of
 * seed + index; integer arithmetic only, so CPU and GPU agree exactly):
 *   uniform: lo + (hi-lo) * (top 24 bits / 2^24);
 *   normal : mean + std * z, z = (sum of twelve 16-bit chunks of three
 *            splitmix64 outputs - 393210) / 65536  (Irwin-Hall 12: mean 0, var 1);
 *   u8     : byte (index & 7) of splitmix64(seed + (index >> 3)).
 * Used to regenerate identical synthetic inputs on any machine. */
int salun_fill_uniform(float *out /*dev*/, int64_t n, uint64_t seed, double lo, double hi,
                       salun_stream_t stream);
int salun_fill_normal(float *out /*dev*/, int64_t n, uint64_t seed, double mean, double std,
                      salun_stream_t stream);
int salun_fill_u8(uint8_t *out /*dev*/, int64_t n, uint64_t seed, salun_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SALUN_H */
