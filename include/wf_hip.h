/*
 * libwf_hip.so -- C-ABI of the MI355X-native WorldForge guided-denoising hot path.
 *
 * The reference (Westlake-AGI-Lab/WorldForge) is pure Python and has no FFI of its own; the drop-in
 * boundary is therefore the set of tensor operations its sampler executes eagerly through PyTorch
 * (SURVEY.md section 8b).  Every entry point below names the reference statement(s) it replaces
 * (paths relative to /root/reference/wan_for_worldforge):
 *   PIPE  = utils/pipeline_wan_i2v_clean.py
 *   SCHED = utils/scheduling_unipc_multistep_clean.py
 *   DIT   = wan/modules/model.py (+ attention.py)  -- in-tree statement of diffusers' WanTransformer3DModel
 *   VAE   = wan/modules/vae.py                     -- in-tree statement of diffusers' AutoencoderKLWan
 *
 * Conventions
 *   - plain pointers and sizes only; all tensor pointers are DEVICE pointers owned by the caller
 *     (the Python host allocates them with the PyTorch-ROCm caching allocator);
 *   - contiguous row-major tensors unless a stride argument is given;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream;
 *   - return value: 0 = WF_OK, negative = WF_E*; text via wf_last_error(); nothing ever falls back
 *     silently to another implementation (contrast SCHED:1284-1295, 1419-1421);
 *   - no hidden host synchronisation and no allocation inside any call; reductions use a fixed tree
 *     order (no floating-point atomics) so repeated runs are bit-identical;
 *   - `rb*` flags: "this intermediate is a bfloat16 torch tensor in the reference", i.e. its value is
 *     rounded to bf16 (round-to-nearest-even) exactly where eager PyTorch would round it.
 */
#ifndef WF_HIP_H_
#define WF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WF_OK 0
#define WF_EINVAL (-1)   /* bad argument (shape / dtype / alignment) */
#define WF_EHIP (-2)     /* a HIP runtime call or kernel launch failed */
#define WF_ENOTSUP (-3)  /* configuration not built */

#define WF_F32 0
#define WF_BF16 1

/* ---- library ------------------------------------------------------------------------------- */
int wf_version(void);
const char* wf_last_error(void);
/* Number of compute units / name of device `dev` (host query, used by bench/roofline). */
int wf_device_info(int dev, int* n_cu, int* clock_khz, char* name, int name_len);

/* ---- scheduler / injection element-wise set (HBM-bound; latent tensors are [B,16,T,h,w]) -- */

/* PIPE:611  noise_pred = cond + g * (cond - uncond).  Element dtype `dt` for all three. */
int wf_cfg_combine(const void* cond, const void* uncond, void* out, int dt, float g, size_t n, void* stream);

/* SCHED:952-958  x0 = sample - sigma * v.   dt_v / dt_s: dtypes of v and sample; out has the torch-promoted dtype
 * (bf16 only if both are bf16). */
int wf_x0_from_v(const void* sample, int dt_s, const void* v, int dt_v, void* out, float sigma, size_t n,
                 void* stream);

/* SCHED:1083-1098 (predict_x0, bh2)  x_t = c1*x - c2*m0 - c3 * 0.5 * (m1 - m0)/rk   (order 2; m1 == NULL -> order 1)
 *   c1 = sigma_t/sigma_s0, c2 = alpha_t*expm1(-h), c3 = alpha_t*B_h, rk = (lambda_s1-lambda_s0)/h computed by the host in
 *   fp32 exactly as SCHED:1016-1069 does.  out has x's dtype (SCHED:1098). */
int wf_unipc_update(const void* x, int dt_x, const void* m0, int dt_m0, const void* m1, int dt_m1, void* out, float c1,
                    float c2, float c3, float rk, size_t n, void* stream);

/* SCHED:1584  noisy = (1 - sigma) * x0 + sigma * noise.  `one_minus_sigma` and `sigma` are supplied by the host already
 * rounded to x0's dtype (SCHED:1552).  out dtype = promote(x0, noise). */
int wf_add_noise(const void* x0, int dt_x0, const void* noise, int dt_n, void* out, float one_minus_sigma, float sigma,
                 size_t n, void* stream);

/* SCHED:1281 (dir 0: z / istd[c] + mean[c] -> f32 out)  and  SCHED:1385, PIPE:351 (dir 1: (z - mean[c]) * istd[c]).
 * z is [B,C,inner]; mean/istd are HOST arrays of C floats (<= 64). in dtype dt_in, out dtype dt_out. */
int wf_latent_affine(const void* z, int dt_in, void* out, int dt_out, const float* mean, const float* istd, int dir, int B,
                     int C, size_t inner, void* stream);

/* SCHED:1375-1380  fused = (2*ref - 1) * mask + dec * (1 - mask);  ref, dec, out: [B,3,inner] f32, mask: [B,1,inner] f32. */
int wf_blend_pixels(const float* ref, const float* mask, const float* dec, float* out, int B, int C, size_t inner,
                    void* stream);

/* Which pixel columns of the decoded video the blend above can see: wherever mask == 1 (exactly) the result is (2 ref - 1) + dec * 0, i.e.
 * independent of any finite dec.  mask [n_rows, W] f32 (all frames and rows of a job's mask) -> out2 (DEVICE, 2 ints) = {first column with a
 * pixel != 1, last such column + 1}; {W, 0} when every pixel is 1.  What lets the VAE decode only the columns an IRR injection consumes
 * (worldforge_amd/vae.py decode(columns=...)). */
int wf_mask_column_range(const float* mask, size_t n_rows, int W, int* out2, void* stream);

/* PIPE:744 (diffusers VideoProcessor.postprocess_video)  out = clamp(x/2 + 0.5, 0, 1), [C,F,H,W] -> [F,H,W,C]. */
int wf_postprocess_video(const float* x, float* out, int C, int F, int H, int W, void* stream);
/* The two pixel-space statements above as eager PyTorch evaluates them behind a BF16 VAE module -- the dtype the LongCat entry loads its
 * VAE in (longcat_for_worldforge/run_longcat_worldforge_single.py:205): dec / out bf16, the fp32 ref and mask rounded to bf16 first
 * (longcat_video/modules/scheduling_flow_match_euler_discrete.py:1152-1153) and every statement of :1156-1164 rounded to bf16;
 * the post-processing of a bf16 video (pipeline_longcat_video.py:1002) likewise, returned as f32. */
int wf_blend_pixels_bf16(const float* ref, const float* mask, const void* dec, void* out, int B, int C, size_t inner, void* stream);
int wf_postprocess_video_bf16(const void* x, float* out, int C, int F, int H, int W, void* stream);

/* dtype conversion / bf16 rounding of latent-shaped tensors (PIPE:590 .to(bf16), PIPE:708). */
int wf_cast(const void* in, int dt_in, void* out, int dt_out, size_t n, void* stream);

/* SCHED:1410-1412  enc[:, c] = pred[:, c] for the listed channels (host array, n_idx <= C). Tensors [B,C,inner]. */
int wf_channel_swap(void* enc, int dt_enc, const void* pred, int dt_pred, const int* idx, int n_idx, int B, int C,
                    size_t inner, void* stream);

/* ---- resize (SCHED:1316-1324 bilinear align_corners=False for the reference video; SCHED:1355-1362 nearest for the mask).
 * The temporal branches SCHED:1326-1334 / 1364-1371 call F.interpolate with a 3-element size on a 4-D tensor and raise
 * ValueError in the reference, so a frame-count mismatch is an error here too (raised by the Python host). */
int wf_resize_bilinear2d(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, void* stream);
int wf_resize_nearest2d(const float* in, float* out, int N, int Hi, int Wi, int Ho, int Wo, void* stream);

/* ---- front end: soften_mask (infer_worldforge.py:105-150) -------------------------------------------------------------- */
/* mask [F,H,W] f32 (any non-zero = inside) -> out [F,H,W]: inside pixels within transition_distance of the zero region get
 * ramp(d / transition_distance), d = exact Euclidean distance to the nearest zero pixel of the frame (float64 like scipy's EDT);
 * decay_type 0 linear, 1 exponential (1 - e^-3t), 2 sine, 3 cosine.  mask and out may not alias. */
int wf_soften_mask(const float* mask, float* out, int F, int H, int W, int transition_distance, int decay_type, void* stream);

/* ---- DSG (PIPE:664-681) ---------------------------------------------------------------------- */
/* Workspace floats needed by wf_dsg (partials + 8 result floats). */
size_t wf_dsg_workspace_floats(void);
/* One call = reduce (dot, |g|^2, |w|^2 in a fixed tree) -> coefficients on device -> apply:
 *   better = g + omega*sin(theta) * (g - (|g|/(|w|+1e-8))*cos(theta) * w).   g = "good" (post-injection), w = "worse".
 * ws: wf_dsg_workspace_floats() floats; ws[0..7] afterwards = {dot, ng2, nw2, cos, sin, ratio, 0, 0} for inspection. */
int wf_dsg(const void* g, const void* w, void* out, int dt, float omega, size_t n, float* ws, void* stream);
/* CFG-zero (LongCat pipeline_longcat_video.py:374-383 optimized_scale + :875-888): st = <cond,uncond> / (|uncond|^2 + 1e-8) over the
 * whole sample; out = uncond*st + guidance*(cond - uncond*st), negated when negate != 0 (the sign flip of :888).  fp32;
 * ws: wf_dsg_workspace_floats() floats, ws[0..3] afterwards = {dot, |cond|^2, |uncond|^2, st}. */
int wf_cfg_zero(const float* cond, const float* uncond, float* out, float guidance, int negate, size_t n, float* ws, void* stream);

/* ---- FLF metric (SCHED:497-607) ------------------------------------------------------------------ */
/* Temporal-difference motion (SCHED:391-392, 478-479): out[c,t,:] = x[c,t+1,:] - x[c,t,:];  x [C,T,hw] -> out [C,T-1,hw] f32. */
int wf_temporal_diff(const void* x, int dt, float* out, int C, int T, size_t hw, void* stream);
size_t wf_flow_metrics_workspace_floats(int n_channels);
/* For each of n_channels: ref flow [Tm, Cr, hw], chan flow [Tm, Cc, hw] (Cr, Cc in {1,2}; 1 is replicated to 2 as SCHED:531-539),
 * similarity = 1 - (.45*clamp(mEPE/10) + .45*clamp(Fl/.5) + .1*clamp(mAE/30))  -> sim[n_channels] (device floats). */
int wf_flow_metrics(const float* ref_flow, const float* chan_flow, float* sim, int n_channels, int Tm, int Cr, int Cc,
                    size_t hw, float* ws, void* stream);
/* The same with the metric variant selectable: 0 = the Wan scheduler's (above); 1 = the LongCat scheduler's
 * (longcat_video/modules/scheduling_flow_match_euler_discrete.py:172-243): outlier = (epe > 3) OR (epe > 5 % |ref|),
 * similarity = 1 - (.4*clamp(mEPE/10) + .4*clamp(Fl/.5) + .2*clamp(mAE/30)). */
int wf_flow_metrics_variant(const float* ref_flow, const float* chan_flow, float* sim, int n_channels, int Tm, int Cr, int Cc,
                            size_t hw, int variant, float* ws, void* stream);

/* ---- FLF optical flow (SCHED:156-248) ----------------------------------------------------------------------------------
 * What `cv2.calcOpticalFlowFarneback(g1, g2, None, 0.5, 3, 15, 3, 5, 1.2, 0)` (SCHED:220-224) computes for every consecutive frame
 * pair of every channel of a latent tensor, including the reference's preparation of its input: normalisation by the tensor's
 * global min / range (SCHED:376-388, 462-474), x255, truncation to uint8 (SCHED:175), RGB2GRAY of three equal channels.
 * x [C,T,h,w] (f32 / bf16) -> flow [C, T-1, 2, h, w] f32 (x then y displacement, the layout of SCHED:240-244).
 * OpenCV is a third-party dependency absent from the reference tree: parity with a real cv2 is UNPINNED.
 * quant_mode selects the reference's uint8 preparation: 0 = Wan (above); 1 = LongCat-Video (LSCHED:105-121, 290-297): min / range per
 * CHANNEL, normalised in the tensor's own dtype (bf16 arithmetic on a bf16 tensor), then uint8(clip((n + 1) * 127.5, 0, 255)); C <= 64. */
size_t wf_farneback_workspace_bytes(int C, int T, int h, int w);
int wf_farneback_flows(const void* x, int dt, float* flow, int C, int T, int h, int w, int quant_mode, void* ws, void* stream);


/* ---- DiT (wan/modules/model.py; the in-tree statement of diffusers' WanTransformer3DModel) --------------------------- */
#define WF_EPI_BF16 0       /* out bf16 = acc + bias */
#define WF_EPI_BF16_GELU 1  /* out bf16 = gelu_tanh(acc + bias)            (model.py:272) */
#define WF_EPI_F32 2        /* out f32  = acc + bias */
#define WF_EPI_RESID 3      /* out f32 += (acc + bias) * gate[n] (gate NULL -> 1)   (model.py:306, 310, 313) */
#define WF_EPI_F32_ACC 4    /* out f32 += acc + bias */
/* nn.Linear on MFMA (model.py:123-126, 271-273, 332, 456-464; vae.py 1x1 convs): out[M,N] = epi(X[M,K] . W[N,K]^T + bias).
 * X, W bf16 (row strides ldx, ldw); bias/gate f32; K % 8 == 0, N % 4 == 0, 16-byte aligned pointers. */
int wf_gemm_bf16(const void* X, const void* W, const float* bias, void* out, const float* gate, int M, int N, int K, int ldx,
                 int ldw, int ldo, int epilogue, void* stream);
/* `batch` independent small products in one launch: out_b = X_b . W_b^T with problem b at X + b*bsx, W + b*bsw, out + b*bso elements;
 * epilogue WF_EPI_BF16 or WF_EPI_F32, no bias.  The per-head block-score products of the block-sparse gating
 * (bsa_interface.py:181-185: one [n_q, 128] x [n_k, 128]^T per head) and the per-frame score / PV products of the VAE mid-block. */
int wf_gemm_bf16_batched(const void* X, const void* W, void* out, int batch, int M, int N, int K, int ldx, int ldw, int ldo,
                         int64_t bsx, int64_t bsw, int64_t bso, int epilogue, void* stream);
/* wf_gemm_bf16 on fp16 operands (X, W fp16; fp32 accumulation; epilogue WF_EPI_BF16 writes fp16, WF_EPI_F32, WF_EPI_F32_ACC): the
 * 1x1 convolutions and mid-block attention products of the VAE in its fp16 operand formats (see wf_split_f16x3).  acc_scale (round 5;
 * 1 = none): out = epilogue(acc * acc_scale + bias) -- 2^-k for a weight operand stored scaled by 2^k (see "fp16 operand formats"). */
int wf_gemm_f16(const void* X, const void* W, const float* bias, void* out, int M, int N, int K, int ldx, int ldw, int ldo, int epilogue,
                float acc_scale, void* stream);
/* wf_gemm_bf16_batched on fp16 operands (WF_EPI_BF16 writes fp16): the per-frame P . V products of the VAE mid-block attention
 * (vae.py:252-258) of all frames in one launch. */
int wf_gemm_f16_batched(const void* X, const void* W, void* out, int batch, int M, int N, int K, int ldx, int ldw, int ldo, int64_t bsx,
                        int64_t bsw, int64_t bso, int epilogue, void* stream);

/* flash_attention (attention.py:24-130) as used by model.py:149-154 (self) and :220-222 (cross): fused
 * softmax(Q K^T * softmax_scale) V, no mask, head_dim 128.  Q [H][Lq][128], K [H][Lkp][128] (rows >= kv_len zero),
 * Vt [H][Lkp/64][128][64] (wf_v_transpose), O [Lq][ldo] bf16 with head h at columns h*128 (ldo % 8 == 0, O 16-byte aligned: 16-byte stores).
 * accumulate != 0: O += result.
 * seg_len: Lkp for one contiguous K/V; with sequence parallelism K/V are the all-gathered per-rank shards [P][H][seg_len][128]
 * (seg_len % 64 == 0, Lkp = P*seg_len) and key index = seg*seg_len + row.
 * softmax_scale = 0: Q already carries softmax_scale * log2(e) (wf_rmsnorm_heads out_scale) -- the score accumulators then start
 * from -m instead of 0 and hold s - m directly, which takes one VALU instruction per score out of the softmax.
 * seg_stride_bytes (round 5): 0 = the dense [P][H][seg_len][128] form above; otherwise the bytes from one segment's head 0 to the next
 * segment's head 0, the SAME for K and Vt -- each source rank's K shard, V^T shard and norm-bound vector then travel as ONE packed slot
 * [K | V^T | bounds] of an exchange buffer (one collective and one event per source or per chunk; worldforge_amd/parallel.py KVExchange),
 * K and Vt point into slot 0.  kmax_stride: floats between two kmax2 vectors (0 = H, contiguous). */
int wf_attn_fwd(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lkp, int kv_len, int seg_len,
                size_t seg_stride_bytes, int ldo, float softmax_scale, int accumulate, const float* kmax2, int kmax_n, int kmax_stride,
                const float* qmax2, int qmax_n, void* stream);
/* Per-head max over the rows of |x|^2: X bf16 [H][Lp][128] (rows >= L ignored) -> out f32 [H], which the caller zeroes first.
 * Computed for K and for the pre-scaled Q and handed to wf_attn_fwd (softmax_scale = 0 only) as kmax2 / qmax2 -- kmax_n / qmax_n such
 * vectors each, one per shard when the tensor is all-gathered -- it lets the kernel prove by Cauchy-Schwarz that no score of the head
 * can overflow: B = max|q| max|k| bounds |s|, a row's reference max is >= -B, so s - m <= 2B; with 2B <= 100 (exp2 domain) the
 * workgroup runs WITHOUT the running-max tracking of the online softmax (exact: the same m is used for P and the row sum).  The tracked
 * path remains the fallback (larger norms, NaNs) and the only path when either pointer is NULL. */
int wf_head_max_norm2(const void* X, int H, int L, int Lp, float* out, void* stream);
/* The same with the KV sweep split nsplit ways (each split leaves un-normalised partials in the workspace, a merge kernel combines
 * them exactly): fills the chip when Lq is short, e.g. one rank's token shard of the sequence-parallel DiT (Lq = 4096 at 8 ranks:
 * 640 workgroups on 256 CUs).  workspace: wf_attn_split_workspace_bytes(H, Lq, nsplit) bytes, 16-byte aligned. */
size_t wf_attn_split_workspace_bytes(int H, int Lq, int nsplit);
/* WanI2VCrossAttention (model.py:202-229) in ONE launch: O = softmax(Q K1^T s) V1 + softmax(Q K2^T s) V2 for the image context (K1, 257
 * keys) and the text context (K2, 512 keys), where the reference runs two flash_attention calls and adds (model.py:220-227).  K [H][Lk1p +
 * Lk2p][128] holds context 1 (kv_len1 valid rows of Lk1p, zero-padded to whole 64-key tiles) followed by context 2; Vt [H][(Lk1p + Lk2p) /
 * 64][128][64] likewise (wf_v_transpose_seg).  Q is read once and O written once; context 1's normalised result is rounded to bf16 and kept
 * in registers across the seam, so the result is bit-identical to wf_attn_fwd(context 1) followed by wf_attn_fwd(context 2, accumulate). */
int wf_attn_cross2_fwd(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lk1p, int kv_len1, int Lk2p, int kv_len2,
                       int ldo, float softmax_scale, void* stream);
/* Test hook: counters2 = device uint32[2] (or NULL = off, the default).  While set, every workgroup of a pre-scaled-Q launch
 * (wf_attn_fwd / wf_attn_fwd_split with softmax_scale = 0) adds 1 to counters2[0] if it ran the max-tracking body and to counters2[1]
 * if it ran the un-tracked one -- lets a parity test assert WHICH body it compared with the oracle.  Process-global, not thread-safe. */
int wf_attn_debug_body_counter(void* counters2);
int wf_attn_fwd_split(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lkp, int kv_len, int seg_len,
                      size_t seg_stride_bytes, int ldo, float softmax_scale, int accumulate, int nsplit, void* workspace, const float* kmax2,
                      int kmax_n, int kmax_stride, const float* qmax2, int qmax_n, void* stream);
/* The KV sweep in PARTS (round 4; sequence-parallel layers WITHOUT a second CFG branch to hide the K / V^T exchange under -- LongCat-Video,
 * distilled schedules, guidance_scale <= 1; reference idea: wan/distributed/xdit_context_parallel.py:160-176, pipeline_longcat_video.py:
 * 857-866).  The key segments arrive one source rank after the other; a part launch walks only the 64-key tiles [t_begin, t_end) that are
 * there already -- the rank's own shard needs no wait at all -- and leaves un-normalised partials (O f32, reference max, row sum) in the
 * slots `part` ... `part + inner_splits - 1` of an `nparts`-slot workspace of wf_attn_split_workspace_bytes(H, Lq, nparts) bytes
 * (inner_splits >= 1 halves / thirds the window over blockIdx.y so that a short query shard fills whole rounds of workgroups).  Pre-scaled Q only (the form of
 * wf_attn_fwd with softmax_scale = 0); Q / K / Vt / bounds as wf_attn_fwd.  wf_attn_merge combines the slots exactly (the flash combine
 * wf_attn_fwd_split uses) into O once every slot has been written; the result equals the one-launch sweep up to the re-association of the
 * fp32 partial sums.
 * Round 5: nparts up to 12; an optional SECOND window [t_begin2, t_end2) (t_end2 <= t_begin2: none) behind the first -- the keys on the far
 * side of a hole, i.e. of the rank's own segment(s), which an earlier part launch walked without waiting for the exchange: one launch then
 * covers "every peer" of an all-gathered buffer.  seg_stride_bytes / kmax_stride as wf_attn_fwd.
 * Round 6: the two windows are walked as ONE sequence by the same workgroups (the hole must be whole segments): `inner_splits` splits of
 * the joined sequence fill the slots part ... (n tiles in k splits: ceil(n / ceil(n / k)) slots) -- one slot and one prologue / epilogue
 * per peer chunk where round 5 spent two.  O_merge != NULL (the LAST launch of a sweep: part = nparts - 1, one split): the launch folds
 * the slots 0 .. part - 1 into its own result in its epilogue and writes the normalised bf16 rows to O_merge [Lq][ldo] -- no partial, no
 * separate wf_attn_merge pass. */
int wf_attn_fwd_part(const void* Q, const void* K, const void* Vt, int H, int Lq, int Lkp, int kv_len, int seg_len, size_t seg_stride_bytes,
                     int t_begin, int t_end, int t_begin2, int t_end2, int inner_splits, int part, int nparts, void* workspace, void* O_merge,
                     int ldo, const float* kmax2, int kmax_n, int kmax_stride, const float* qmax2, int qmax_n, void* stream);
int wf_attn_merge(void* O, int H, int Lq, int ldo, int accumulate, int nparts, const void* workspace, void* stream);

/* WanLayerNorm (model.py:92-102; eps, no affine) fused with y = ln * (plus_one + mul[c]) + add[c]:
 *   AdaLN modulate (model.py:303, 311, 346): mul = scale e[1]/e[4], add = shift e[0]/e[3], plus_one = 1;
 *   affine LayerNorm (norm3 :262-264, img_emb :356-358): mul = weight, add = bias, plus_one = 0.
 * x f32 [L,C]; out bf16 or f32 [L,C]. */
int wf_ln_modulate(const float* x, const float* mul, const float* add, void* out, int out_dtype, int L, int C, float eps,
                   int plus_one, void* stream);

/* WanRMSNorm over all C channels (model.py:73-89, 142-143, 215-218) + optional 3-axis RoPE (model.py:43-70; cos/sin tables
 * [L][64] f32, NULL -> none), written head-major: in bf16 [L, ld] -> out bf16 [C/128][Lout][128] (rows L..Lout untouched).
 * out_scale (1 = none) multiplies the f32 result in front of the one bf16 rounding: the self-attention Q is produced as
 * q * head_dim^-1/2 * log2(e), and wf_attn_fwd is then called with softmax_scale = 0 ("Q is pre-scaled"). */
int wf_rmsnorm_heads(const void* in, int ld, const float* weight, const float* cos_tab, const float* sin_tab, void* out, int L,
                     int Lout, int C, float eps, float out_scale, void* stream);
/* The same, and in the same pass max_norm2[h] = max over the L rows of |out row of head h|^2 (of the values as stored, after the bf16
 * rounding) -- what wf_head_max_norm2 computes with a second pass over the tensor (rows holding NaN / inf report +inf likewise).
 * ws: wf_rmsnorm_heads_bound_ws_floats(L, C) floats.  C / 128 heads <= 64. */
size_t wf_rmsnorm_heads_bound_ws_floats(int L, int C);
int wf_rmsnorm_heads_bound(const void* in, int ld, const float* weight, const float* cos_tab, const float* sin_tab, void* out, int L, int Lout,
                           int C, float eps, float out_scale, float* ws, float* max_norm2, void* stream);

/* V [L, ld] bf16 (head h at columns h*128) -> Vt [H][Lp/64][128][64] bf16, keys >= L zero-filled. */
int wf_v_transpose(const void* V, int ld, void* Vt, int L, int Lp, int H, void* stream);
/* The same into a destination whose heads are head_stride_tiles 64-key tiles apart (Vt points at the segment's first tile of head 0): the
 * two contexts of wf_attn_cross2_fwd share one buffer. */
int wf_v_transpose_seg(const void* V, int ld, void* Vt, int L, int Lp, int H, int head_stride_tiles, void* stream);

/* model.py:534-537 patch embedding as a GEMM: x bf16 [Cin,T,Hh,Ww] -> tokens bf16 [T*(Hh/2)*(Ww/2), Cin*4] (k = c*4+ph*2+pw). */
int wf_patchify(const void* x, void* tokens, int Cin, int T, int Hh, int Ww, void* stream);
/* model.py:584-607: y f32 [L, 4*Cout] (k = (ph*2+pw)*Cout + c) -> out f32 [Cout,T,Hh,Ww]. */
int wf_unpatchify(const float* y, float* out, int Cout, int T, int Hh, int Ww, void* stream);

/* out = f(a (+ b)); mode 0 SiLU (model.py:463-464), 1 GELU-erf (model.py:357), 2 identity (e = modulation + e0, model.py:298). */
int wf_act(const void* a, int dt_a, const void* b, int dt_b, void* out, int dt_out, int mode, size_t n, void* stream);

/* ---- LongCat-Video DiT companions (longcat_for_worldforge/longcat_video/modules; LCD = longcat_video_dit.py, LCA = attention.py,
 *      LCB = blocks.py, LCR = rope_3d.py).  The LongCat residual stream is bf16 and its AdaLN parameters are per latent frame. ---- */
/* LayerNorm_FP32 (LCB:55-68) fused with y = ln * (plus_one + mul[g][c]) + add[g][c], g = (row0 + row) / rows_per_group (0: one group;
 * row0 = global index of the first row when x is one rank's token shard; group_index (int32 per GLOBAL row, may be NULL) replaces the
 * division when the rows are not in frame order -- the 3D-block token order of the refine pass):
 *   modulate_fp32 with per-frame shift / scale (LCB:133-141 as called at LCD:91, 114 and LCB:165): mul = scale, add = shift,
 *   mod_ld = floats between consecutive frames, rows_per_group = tokens per frame, plus_one = 1;
 *   affine pre_crs_attn_norm (LCD:111): mul = weight, add = bias, rows_per_group = 0, plus_one = 0.   x, out bf16 [L, C]. */
int wf_lc_ln_modulate(const void* x, const float* mul, const float* add, int64_t mod_ld, int rows_per_group, int64_t row0,
                      const int* group_index, int plus_one, void* out, int L, int C, float eps, void* stream);
/* x = bf16(x + gate[(row0 + row) / rows_per_group][c] * y) (LCD:102-104, 117-120); gate NULL: x = bf16(x + y) (LCD:111).
 * x bf16 [L, C] in place; y bf16 [L, ldy]; gate f32 rows gate_ld floats apart. */
int wf_lc_gate_residual(void* x, const void* y, int64_t ldy, const float* gate, int64_t gate_ld, int rows_per_group, int64_t row0,
                        const int* group_index, int L, int C, void* stream);
/* RMSNorm_FP32 over each head's 128 channels (LCB:40-52 as used at LCA:111 and LCA:231) + optional interleaved 3D RoPE
 * (LCR:32-36, 101-120; cos/sin tables [L][64] f32 per rotation pair, NULL -> none), written head-major for wf_attn_fwd:
 * in bf16 [L, ld] (head h at columns h*128) -> out bf16 [H][Lout][128] (rows L..Lout untouched); weight f32 [128].  out_scale multiplies the
 * result in front of its one bf16 rounding (1 = none; softmax_scale * log2(e) for the queries of wf_attn_fwd's softmax_scale = 0 form). */
int wf_lc_norm_heads(const void* in, int64_t ld, const float* weight, const float* cos_tab, const float* sin_tab, void* out, int L,
                     int Lout, int H, float eps, float out_scale, void* stream);
/* FeedForwardSwiGLU gate (LCB:36-37): in bf16 [L, ld] with w1 x in columns [0, Hd) and w3 x in [Hd, 2 Hd) ->
 * out bf16 [L, Hd] = silu(w1 x) * w3 x. */
int wf_lc_swiglu(const void* in, int64_t ld, void* out, int L, int Hd, void* stream);

/* LongCat refine pass input (pipeline_longcat_video.py:1407-1413): stage-1 frames uint8 [F][H0][W0][3] -> bilinear (align_corners) to
 * (H, W) -> / 255 -> linear along the frame axis (the trilinear of :1412 with H, W unchanged) to Fo frames -> * 2 - 1, every step
 * rounded to bf16 as the reference's bf16 tensors.  out f32 [3][Fo][H][W]. */
int wf_refine_upsample_u8(const void* frames_u8, float* out, int F, int H0, int W0, int Fo, int H, int W, void* stream);

/* ---- LongCat block-sparse attention of the 720p refine pass (longcat_video/block_sparse_attention/bsa_interface.py = BSA) --------- */
/* mean_pooling_compression (BSA:169-179): in bf16 [H][L][128] -> out bf16 [H][L/block][128], mean of each block of 64 / 128 tokens. */
int wf_lc_mean_pool_blocks(const void* in, void* out, int H, int L, int block, void* stream);
/* Block selection (BSA:211-224, `torch.topk(score, int((1 - sparsity) * n_k))`) fused with the list building of the sparse kernel:
 * scores bf16 [heads][n_q][ld] (n_k valid columns: the gating products of BSA:181-185) -> for every group of g = 256 / block consecutive
 * query blocks the ascending union of their n_sel best key blocks, entry = physical_block * 2^g + sum_i 2^i [selected by the i-th query
 * block of the group], physical_block = (b / blocks_per_segment) * heads * blocks_per_segment + head * blocks_per_segment + b %
 * blocks_per_segment (the block's position in the all-gathered K / V^T buffers; one segment = the whole sequence on one GPU).
 * lists int32 [heads][ceil(n_q / g)][max_entries], counts int32 [heads][ceil(n_q / g)]; max_entries >= min(g * n_sel, n_k).  Equal
 * scores at the n_sel-th place are taken by ascending block index.  sel_mask (NULL or uint32 [heads][n_q][ceil(n_k / 32)]): bit b of a row
 * = key block b selected by that query block (read back by tests).  n_k <= 2048. */
int wf_bsa_topk_lists(const void* scores, int64_t ld, int heads, int n_q, int n_k, int n_sel, int block, int blocks_per_segment,
                      int* lists, int* counts, int max_entries, uint32_t* sel_mask, void* stream);
/* The cdf selection rule (BSA:226-263, get_select_indices_cdf / get_select_indices_cdf_topk) with the same list building: per query
 * block, the key blocks in descending order of softmax(score / sqrt(128)) as long as their cumulative weight stays <= cdf_threshold
 * (`searchsorted(cdf, threshold, right=True)`: possibly none), at least n_min of them (int((1 - sparsity) * n_k) of the _topk form, 0
 * otherwise).  row_counts int32 [heads][n_q] receives the per-row counts; lists / counts / sel_mask as wf_bsa_topk_lists with
 * max_entries >= n_k.  The chain is evaluated as eager torch evaluates it on the reference's bf16 score tensor -- bf16(score * scale),
 * softmax rounded to bf16, running fp32 sum of the sorted weights rounded to bf16 per element, compared with float32(cdf_threshold) --
 * pinned by tests/golden/g14c_bsa_cdf_bf16.npz (recorded from the reference's function).  Deterministic; for thresholds <= 0.97 the
 * tree scan equals the sequential cumsum exactly, what can differ from a torch run is one bf16 ulp of a weight where exp / the row sum
 * round differently (about one row in a thousand changes its count by one). */
int wf_bsa_cdf_lists(const void* scores, int64_t ld, int heads, int n_q, int n_k, float cdf_threshold, int n_min, int block,
                     int blocks_per_segment, int* lists, int* counts, int max_entries, uint32_t* sel_mask, int* row_counts, void* stream);
/* out[i][:C] = in[index[i]][:C], bf16 rows (16-byte chunks): the two token permutes of BSA:600-610 the refine pass needs per forward
 * (patch tokens into 3D-block order, velocity rows back). */
int wf_gather_rows_bf16(const void* in, int64_t ld_in, const int* index, void* out, int64_t ld_out, int n_rows, int C, void* stream);
/* The sparse attention of BSA:538-560 (flash_attn_bsa_varlen_mask.py:174-285): Q [H][Lq][128], K [H][Lkp][128], Vt [H][Lkp/64][128][64]
 * in 3D-block token order, `block` = 128 or 64 tokens per block; every query block attends to its selected key blocks only.  The
 * selection is given per GROUP of g = 256 / block consecutive query blocks (the 256 query rows of one workgroup): group_lists
 * [H][ceil(Lq/256)][max_entries] int32, entry = physical_block * 2^g + flags (bit i: selected by the i-th query block of the group;
 * physical_block = position of the key block in the K / Vt buffers counted in blocks: ((segment * H + head) * seg_len + offset) / block),
 * the union of the group's lists in any order; group_counts [H][ceil(Lq/256)] entries used.  seg_len = Lkp, or the per-rank shard length
 * when K / Vt are all-gathered shards [P][H][seg_len][128] (as wf_attn_fwd).  O [Lq][ldo] bf16 (block order), head h at columns h*128. */
int wf_attn_bsa_fwd(const void* Q, const void* K, const void* Vt, void* O, int H, int Lq, int Lkp, int seg_len, int ldo,
                    float softmax_scale, const int* group_lists, const int* group_counts, int max_entries, int block, void* stream);

/* ---- 3D causal VAE (wan/modules/vae.py; the in-tree statement of diffusers' AutoencoderKLWan), channels-last ----------- */
/* CausalConv3d / Conv2d as implicit GEMM on MFMA (vae.py:17-36, 76-96, 186-220).  in bf16 [Ti,Hi,Wi,Cin] (Cin % 32 == 0),
 * w bf16 [Cout][kt*kh*kw][Cin], bias f32; out (f32 and/or bf16) [To,Ho,Wo,Cout] (+ resid f32 of the same shape).
 *   out[t,y,x] = sum_taps in[t*st+dt-pt, y*ss+dy-ph, x*ss+dx-pw] . w[tap]   (zero outside the input; ph may be negative when
 *   the input is a row slab that already carries its halo rows -- multi-GPU row sharding)
 * up2: read the input through a nearest-exact 2x spatial upsample (vae.py:78);  tsplit: 'upsample3d' frame interleave
 * (vae.py:134-137): output frame t, channel half h -> frame 1 + 2t + h of a [1+2*To, Ho, Wo, Cout/2] tensor.
 * zero_page: >= 16 bytes of zeros in device memory (enables the DMA-gather ping-pong kernel for large stride-1 layers) or NULL. */
int wf_conv3d_cl(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_bf16, int Ti,
                 int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st, int ss, int pt, int ph,
                 int pw, int up2, int tsplit, const void* zero_page, void* stream);
/* wf_conv3d_cl whose output pixel (t, y, x) of the [To, Ho, Wo] grid lands at pixel (t, sy * y + oy, sx * x + ox) of a [To, out_H, out_W]
 * tensor (resid, if given, is read at the same place).  Lets the nearest-2x upsample + 3 x 3 convolution of Resample (vae.py:76-86) run as
 * its four phases: output (2 y + py, 2 x + px) only ever sees source rows {y - 1 + py, y + py} and columns {x - 1 + px, x + px}, with the
 * 3 x 3 taps that fall on the same source pixel summed beforehand -- 2 x 2 convolutions on the source grid, 4 / 9 of the multiply-adds. */
int wf_conv3d_cl_scatter(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_bf16, int Ti, int Hi,
                         int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st, int ss, int pt, int ph, int pw,
                         const void* zero_page, int out_H, int out_W, int sy, int oy, int sx, int ox, void* stream);
/* The FLOP-heavy layers (3x3x3, stride 1, causal: every ResidualBlock conv, vae.py:186-220) with the input patch resident in LDS:
 * an 8 x 64 pixel tile of one output frame x 96 output channels per workgroup, all 27 taps read the (3 x 10 x 66)-pixel patch of a
 * 16-channel slice from LDS.  Weights in the re-packed layout [27][Cin/16][Cout][16] produced by wf_conv3d_pack333 from
 * [Cout][27][Cin].  in [T,Hi,Wi,Cin] bf16 (Hi = Ho for ph = 1; row slabs carry their halo rows: Hi = Ho + 2, ph = 0),
 * out [T,Ho,Wi,Cout]; Cin % 32 == 0, Cout % 32 == 0.  zero_page: zeros, zero_page_bytes >= wf_conv3d_333_zero_page_bytes(Wi, Cin_stored,
 * layout) of them -- the lanes that stage padding pixels read it at the same wave-uniform slice offset the other lanes read the input at
 * (one 64-bit add per LDS-DMA piece instead of a per-lane select).  Same arithmetic as wf_conv3d_cl.
 * layout 0: in is pixel-major [T,Hi,Wi,Cin_stored]; layout 1: slice-major [T,Hi,Cin_stored/16,Wi,16] (what wf_rms_silu_cl_blocked
 * writes: a patch row of a 16-channel slice is contiguous, so the LDS-DMA gather reads whole cache lines).  Cin_stored = Cin, or
 * 2/3 Cin when the fp32-class three-term operand [hi | lo | hi] (wf_split_bf16x3 side 0) is stored as [hi | lo] and its hi half is
 * read for both K thirds. */
int wf_conv3d_pack333(const void* w, void* w_packed, int Cout, int Cin, void* stream);
size_t wf_conv3d_333_zero_page_bytes(int Wi, int Cin_stored, int layout);
int wf_conv3d_333(const void* in, const void* w_packed, const float* bias, const float* resid, float* out_f32, void* out_bf16, int T,
                  int Hi, int Wi, int Cin, int Ho, int Cout, int ph, const void* zero_page, size_t zero_page_bytes, int layout,
                  int Cin_stored, void* stream);
/* Direct convolution for the thin layers (3->96, 16->384, 96->3, 384->32, 1x1x1 quant convs; vae.py:288, 316, 392, 421, 505-506).
 * in f32 or bf16 channels-last, w f32 [taps][Cin][Cout]; clamp > 0 clamps the output (autoencoder_kl_wan.py:1222). */
int wf_conv3d_small(const void* in, int in_dtype, const float* w, const float* bias, float* out_f32, void* out_bf16, int Ti, int Hi,
                    int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st, int ss, int pt, int ps,
                    float clamp, void* stream);
/* RMS_norm over channels (vae.py:39-54) [+ SiLU]: x f32 [npix, C] -> bf16 and/or f32. */
int wf_rms_silu_cl(const float* x, const float* gamma, void* out_bf16, float* out_f32, size_t npix, int C, int silu, void* stream);
/* Row softmax of the mid-block attention scores (vae.py:252-256): P[m, :N] = softmax(S[m, :N] * scale), P[m, N:ldp] = 0. */
int wf_softmax_rows(const float* S, int lds, void* P, int ldp, int M, int N, float scale, void* stream);
/* bf16 in [R, ld_in] (first C columns) -> out [C, ld_out], columns R..ld_out zero. */
int wf_transpose_bf16(const void* in, int ld_in, void* out, int ld_out, int R, int C, void* stream);
/* fp32-class VAE (the reference loads the VAE with torch_dtype=torch.float32, infer_worldforge.py:185-189; every conv / linear of
 * vae.py then contracts fp32 operands).  The matrix cores take bf16, so an fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi)
 * and a contraction as hi.hi + lo.hi + hi.lo in fp32 accumulators (dropped: lo.lo <= 2^-16 relative, tails <= 2^-17): src f32
 * [rows, C] (row stride ld_src) -> dst bf16 [rows, 3C] (row stride ld_dst); side 0 (activation) = [hi | lo | hi], side 1 (weight) =
 * [hi | hi | lo]: the unchanged conv / GEMM kernels then run on 3C "channels". */
int wf_split_bf16x3(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, int side, void* stream);
/* wf_rms_silu_cl with the side-0 three-term output ([npix, 3C] bf16) written directly. */
int wf_rms_silu_cl_x3(const float* x, const float* gamma, void* out_x3, size_t npix, int C, int silu, void* stream);
/* wf_rms_silu_cl writing the slice-major conv operand of wf_conv3d_333 (layout 1): x f32 [npix, C] (npix = whole rows of W pixels)
 * -> bf16 [npix / W][C/16][W][16]; split != 0: the fp32-class operand stored as [hi | lo] slices, [npix / W][2C/16][W][16].
 * halo_rows = Hs > 0 (row slabs of the sharded VAE, x = [T][Hs][W][C]): the output is the halo-padded operand [T][Hs + 2][...] and row
 * (t, y) is written to (t, y + 1); the two halo rows of every frame are left to the caller (neighbours' rows / zeros). */
int wf_rms_silu_cl_blocked(const float* x, const float* gamma, void* out, size_t npix, int C, int silu, int W, int split, int halo_rows,
                           void* stream);
/* wf_softmax_rows with f32 probabilities (vae.py:252-256 in fp32) / wf_transpose_bf16 on f32. */
int wf_softmax_rows_f32(const float* S, int lds, float* P, int ldp, int M, int N, float scale, void* stream);
int wf_transpose_f32(const float* in, int ld_in, float* out, int ld_out, int R, int C, void* stream);
/* ---- fp16 operand formats of the VAE (round 4) -----------------------------------------------------------------------------------------
 * The reference's VAE is fp32 (infer_worldforge.py:185-189).  The three-term split above on BF16 parts (8-bit significands) leaves
 * ~2^-16 per product; the same split on FP16 parts (11-bit significands: hi = fp16(x), lo = fp16(x - hi), contraction hi.hi + lo.hi +
 * hi.lo on v_mfma_f32_32x32x16_f16, same rate) leaves ~2^-22 -- two decimal digits closer to IEEE fp32 at the same cost; it is the
 * VAE's default ("fp16x3").  fp16 has a 5-bit exponent: values beyond +-65504 do not fit.  The producers raise a sticky flag
 * (wf_f16_overflow_flag) that the host turns into an error; the bf16 split ("bf16x3") remains for such weights.
 * Every *_f16 entry point is its bf16 namesake with fp16 in place of bf16 in the operands (and in a 16-bit output copy).
 * acc_scale (round 5): `lo = fp16(x - hi)` is an fp16 SUBNORMAL for |x| < 2^-3 (absolute floor 2^-25), so a weight of
 * 0.02 was carried to ~2^-18 relative, not 2^-22.  The host therefore stores every weight matrix multiplied by an exact power of two 2^k
 * (its largest magnitude lands in [2^13, 2^14): lo is a normal fp16 for every weight above 2^-17 of the layer's largest) and passes
 * acc_scale = 2^-k: the kernels compute out = acc * acc_scale + bias (+ residual) -- exact, one fma in place of the add. */
int wf_split_f16x3(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, int side, void* stream);
int wf_rms_silu_cl_x3_f16(const float* x, const float* gamma, void* out_x3, size_t npix, int C, int silu, void* stream);
int wf_rms_silu_cl_blocked_f16(const float* x, const float* gamma, void* out, size_t npix, int C, int silu, int W, int split,
                               int halo_rows, void* stream);
int wf_conv3d_cl_f16(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_f16, int Ti,
                     int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st, int ss, int pt, int ph,
                     int pw, int up2, int tsplit, const void* zero_page, float acc_scale, void* stream);
int wf_conv3d_cl_scatter_f16(const void* in, const void* w, const float* bias, const float* resid, float* out_f32, void* out_f16, int Ti,
                             int Hi, int Wi, int Cin, int To, int Ho, int Wo, int Cout, int kt, int kh, int kw, int st, int ss, int pt,
                             int ph, int pw, const void* zero_page, int out_H, int out_W, int sy, int oy, int sx, int ox, float acc_scale,
                             void* stream);
int wf_conv3d_333_f16(const void* in, const void* w_packed, const float* bias, const float* resid, float* out_f32, void* out_f16, int T,
                      int Hi, int Wi, int Cin, int Ho, int Cout, int ph, const void* zero_page, size_t zero_page_bytes, int layout,
                      int Cin_stored, float acc_scale, void* stream);
/* *out = 1 if a producer converted a value beyond the fp16 range (or a NaN) since the last reset; synchronises `stream`. */
int wf_f16_overflow_flag(int* out, int reset, void* stream);
/* The same word copied asynchronously into page-locked host memory behind the work queued on `stream` so far: no synchronisation (the
 * caller records an event behind the call and reads the word once it has fired) -- what AutoencoderKLWan.encode / decode use since
 * round 5, so that a VAE call never stalls the host (SURVEY 8b "no hidden device syncs"). */
int wf_f16_overflow_flag_async(int* pinned_host_out, void* stream);
/* [C, N] f32 -> [N, Cpad] (f32 and/or bf16; channels C..Cpad zero, so thin inputs fill an MFMA K slice);
 * [N, ld] f32 (first C channels) -> [C, N] f32 with optional clamp (autoencoder_kl_wan.py:1222).  N = T*H*W. */
int wf_ncthw_to_cl(const float* in, float* out_f32, void* out_bf16, int C, int Cpad, size_t N, void* stream);
int wf_cl_to_ncthw(const float* in, float* out, int C, int ld, size_t N, float clamp, void* stream);
/* ---- the one-term fp16 operand format of the VAE ("fp16", round 6) -------------------------------------------------------------------
 * What an fp32 convolution keeps of its operands when TF32 is allowed: cuDNN fp32 convolutions are TF32-eligible under PyTorch's
 * defaults (torch.backends.cudnn.allow_tf32 = True; the LongCat entry also sets it explicitly, run_longcat_worldforge_single.py:144-146),
 * i.e. 10 explicit mantissa bits per multiplicand and fp32 accumulation.  fp16 has the same 10 explicit bits: the one-term mode feeds
 * hi = fp16(x) alone to the *_f16 conv / GEMM kernels above (weights stored power-of-two scaled, acc_scale epilogue, range flag) at one
 * third of the three-term mode's matrix work.  These are the producers of that operand; every one raises the range flag of
 * wf_f16_overflow_flag.  wf_cast_f16: src f32 [rows, C] (row stride ld_src) -> dst fp16 [rows, C] (row stride ld_dst), C and the strides
 * multiples of 4.  The others are their bf16 namesakes with an fp16 16-bit output. */
int wf_cast_f16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, void* stream);
/* Any of the operand producers (fmt 0: wf_split_bf16x3, 1: wf_split_f16x3, 2: wf_cast_f16, 3: plain bf16 rounding) writing straight into a
 * halo-padded destination -- the row slabs of the sharded VAE, [T][Hs + 2][W][..] with the slab's own rows at 1 .. Hs of every frame: source
 * row r -> destination row lead_rows + r + (r / group_rows) * gap_rows (group_rows = Hs W, gap_rows = 2 W, lead_rows = W; group_rows = 0: no
 * gaps).  Same values as the stand-alone producers: it replaces producer + copy (worldforge_amd/vae.py _halo_pad_of). */
int wf_operand_rows(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, size_t rows, int C, int fmt, int side, size_t group_rows,
                    size_t gap_rows, size_t lead_rows, void* stream);
int wf_rms_silu_cl_f16(const float* x, const float* gamma, void* out_f16, float* out_f32, size_t npix, int C, int silu, void* stream);
int wf_softmax_rows_f16(const float* S, int lds, void* P, int ldp, int M, int N, float scale, void* stream);
int wf_ncthw_to_cl_f16(const float* in, float* out_f32, void* out_f16, int C, int Cpad, size_t N, void* stream);

/* ---- stage-1 forward warping (vggt/modules/utils_warp.py:863-945, warp_single_img without crack filling) ------------------------- */
/* Depth-guided forward splat of one image to n_cameras new views: un-project (fp64) -> world -> each camera -> project -> nearest pixel,
 * nearest source wins (z-buffer).  image f32 [H][W][3] in [0,1]; depth f32 [H][W] (NaN / <= 0 = invalid, i.e. already filtered by
 * confidence); geometry = 30 doubles {K^-1 (9), K (9), R^T (9), -R^T t (3)} of the source camera; cameras = n x 12 doubles (rows of
 * [R | t] of every new camera).  out_images u8 [n][H][W][3], out_masks u8 [n][H][W] (1 = written), out_depth f32 [n][H][W] (NaN =
 * empty); zbuffer: n*H*W 8-byte words of scratch. */
int wf_warp_splat(const float* image, const float* depth, const double* geometry, const double* cameras, void* out_images, void* out_masks,
                  float* out_depth, void* zbuffer, int n_cameras, int H, int W, void* stream);
/* Depth-aware crack filling of the warped views (vggt/modules/utils_warp.py depth_aware_crack_filling :647-691 with segment_depth_map
 * :506-536, fill_segment_cracks :567-634 (fast outlier test), fill_small_cracks step 1 :390-430, vectorized_depth_estimation :539-564,
 * merge_depth_segments :637-676, as warp_single_img runs them per view :954-985): img u8 [n,H,W,3], mask u8 [n,H,W], depth f32 [n,H,W]
 * (NaN = empty) -> filled image / mask / depth of the same shapes.  min_neighbors: outlier threshold (3 x 3 count incl. the centre);
 * min_valid_neighbors: valid 8-neighbours a newly covered pixel needs; num_segments <= 5.  OpenCV's filter2D / morphologyEx are restated
 * (BORDER_REFLECT_101 correlation; closing that ignores the border): parity with a real cv2 is unpinned.  workspace:
 * wf_crack_fill_workspace_bytes(n, H, W) bytes. */
size_t wf_crack_fill_workspace_bytes(int n, int H, int W);
int wf_crack_fill(const void* img, const void* mask, const float* depth, void* out_img, void* out_mask, float* out_depth, int n, int H,
                  int W, int min_neighbors, int min_valid_neighbors, int num_segments, void* workspace, void* stream);

/* fill_small_cracks in full (vggt/modules/utils_warp.py:386-455) -- what warp_single_img runs INSTEAD of the depth-aware filling for a view
 * with <= 100 splatted depths (:957-962, 973-981): step 1 = 3 x 3 closing + mean of the valid 8-neighbours (>= min_valid_neighbors of them);
 * step 2 (has_depth_conf != 0 and step 1 filled fewer than half of the holes) = the 4-connected hole components of <= min(max_crack_size, 4)
 * pixels in scipy.ndimage.label order, pixel by pixel, from the valid 3 x 3 neighbours whose original_depth (the SOURCE view's map, indexed at
 * the target pixel as the reference does) is within depth_threshold of the pixel's -- sequential, one lane.  One view: img u8 [H][W][3],
 * mask u8 [H][W], original_depth f32 [H][W] (may be NULL when has_depth_conf == 0) -> out_img u8, out_mask u8.  The image is float32 (u8 / 255)
 * through both steps and quantised once ((x * 255) truncated), as the reference.  workspace: wf_fill_small_cracks_workspace_bytes(H, W). */
size_t wf_fill_small_cracks_workspace_bytes(int H, int W);
int wf_fill_small_cracks(const void* img, const void* mask, const float* original_depth, int has_depth_conf, void* out_img, void* out_mask,
                         int H, int W, float depth_threshold, int max_crack_size, int min_valid_neighbors, void* workspace, void* stream);

/* ---- stage-1 of the dynamic-scene path: DepthCrafter point-cloud renderer (DepthCrafter/utils.py, warp_depthcrafter.py:255-288) ------- */
/* project_points_to_image_pytorch (utils.py:103-171): pytorch3d PointsRasterizer(radius, points_per_pixel = 10) -> idx[..., 0] = the
 * covering point of smallest view depth per pixel -> image = features[idx], mask = idx != -1 -> 5 x 5 opening of the mask (morph) ->
 * image zeroed outside it.  points f32 [n][3] (world), features f32 [n][F], drop (NULL or u8 [n]: 1 = point removed by the edge filter);
 * camera = 16 floats in pytorch3d's convention {R' (9, row-major, view = p R' + T'), T' (3), focal' (2), principal' (2)}, i.e. the result
 * of _cameras_from_opencv_projection for the reference's (extrinsic, K, image size) -- computed by the host (worldforge_amd/warp.py).
 * out_image f32 [H][W][F], out_mask u8 [H][W].  pytorch3d / OpenCV are absent from the reference tree: parity with them is unpinned. */
size_t wf_points_render_workspace_bytes(int H, int W);
int wf_points_render(const float* points, const float* features, const void* drop, int n, int F, const float* camera, int H, int W,
                     float radius, int morph, float* out_image, void* out_mask, void* workspace, void* stream);
/* filter_edge_points (utils.py:523-567): out_drop u8 [H][W] = dilate_{2 d + 1}(Sobel magnitude / its maximum > edge_threshold) |
 * (max - min of the depth over the (2 r + 1)^2 window > jump_threshold); depth f32 [H][W].  d = edge_dilation, r = neighbor_radius. */
size_t wf_depth_edge_mask_workspace_bytes(int H, int W);
int wf_depth_edge_mask(const float* depth, int H, int W, double edge_threshold, int edge_dilation, float jump_threshold, int neighbor_radius,
                       void* out_drop, void* workspace, void* stream);

/* ---- measurement aid (bench.py `box_calib_tflops`) -------------------------------------------------------------------------------------
 * One launch of a fixed register-only stream of v_mfma_f32_32x32x16_bf16 (256 workgroups x 4 waves, one per SIMD, 16 accumulator tiles per
 * wave, iters x 16 MFMAs each): what THIS box sustains on the matrix pipe alone under its power limit.  src: 1 MiB of bf16 operand values
 * (N(0,1): the rate depends on the data), sink: >= 4 bytes; *flop (host, may be NULL) receives the launch's flop count.  Asynchronous. */
int wf_calib_mfma(const void* src, float* sink, int iters, double* flop, void* stream);
/* A stream-ordered delay of `us` microseconds (one wave polling the 100 MHz wall clock): the stand-in for a collective's transfer time when
 * one GPU plays one rank of N under a bandwidth model (parallel.LoopbackComm(link model), bench.py --as-rank-of N --emulate-comm). */
int wf_delay_us(double us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WF_HIP_H_ */
