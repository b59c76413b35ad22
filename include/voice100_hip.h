/* voice100_hip.h -- C ABI of libvoice100_hip.so: the MI355X (gfx950) kernels behind the
 * Voice100 non-autoregressive CNN hot path.
 *
 * The reference (kaiidams/voice100 v1.6.0) has no FFI of its own: the path is plain
 * torch.nn modules.  Each entry point below therefore cites the
 * reference call it stands in for; voice100_amd/ (Python, ctypes) mirrors the reference's
 * module interface on top of these.  INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch CUDA tensor storage), 16-byte
 *     aligned, contiguous; activations are fp32 [B][C][T] unless stated otherwise;
 *   - no allocation, no host synchronisation inside: outputs and workspaces ("stats",
 *     "partial") are caller-allocated, `stream` is a hipStream_t (NULL = default stream);
 *   - return value: 0 = launched; 1 = invalid/unsupported shape or mode; 2 = launch error;
 *     3 = a required pointer is NULL.  The Python layer raises RuntimeError on != 0;
 *   - re-entrant; the only global state is the opt-in timing table below; one stream per process is the
 *     intended use.
 */
#ifndef VOICE100_HIP_H
#define VOICE100_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- input-transform modes shared by the depthwise and pointwise kernels ------------------
 *   0 NONE            v
 *   1 AFFINE_RELU6    clamp(v*a[c] + b[c], 0, 6)      BatchNorm affine + ReLU6 of the producer
 *                                                      (asr.py:36-37) applied on load
 *   2 AFFINE2         a[c]*v + b[c]*v2 + c[c]         BatchNorm backward applied on load       */

/* ---- K2 depthwise -------------------------------------------------------------------------
 * y[b,c,t] = sum_j w[c][flip ? K-1-j : j] * in[b,c, t*stride - pad + j], zero padded, `in` being the
 * transformed (and, for upsample U > 1, zero-stuffed by U) input.
 * Replaces nn.Conv1d(groups=C) of ConvBNActivate, voice100/models/asr.py:31-35,49 (forward) and its
 * autograd backward-data (flip=1, pad=K-1-pad; strided layers: stride=1, upsample=S).
 * out_mode: 0 raw + stats(sum y, sum y^2)   1 clamp(y*out_a+out_b,0,6)  (eval-mode BN+ReLU6 folded)
 *           2 y *= [0 < aux*out_a+out_b < 6] + stats(sum y, sum y*aux) (ReLU6 backward)   3 raw
 * stats: [G][C][2] slab, G = v100_dw_num_groups(B, C). */
int v100_dw_num_groups(int B, int C);
int v100_dwconv(const float* x, const float* x2, const float* w, const float* in_a, const float* in_b,
                const float* in_c, int in_mode, float* y, const float* aux, const float* out_a,
                const float* out_b, int out_mode, float* stats, int G, int B, int C, int Tin, int Tout,
                int K, int stride, int pad, int flip, int upsample, int force_generic, void* stream);

/* dw[c][j] = sum_{b,t} g'[b,c,t] * x'[b,c, t*stride - pad + j]; g' / x' are g / x after g_mode / x_mode.
 * Replaces the weight gradient of the same nn.Conv1d(groups=C).  partial: [G][C][K] workspace. */
int v100_dwconv_wgrad(const float* g, const float* g2, const float* ga, const float* gb, const float* gc,
                      int g_mode, const float* x, const float* xa, const float* xb, int x_mode,
                      float* partial, float* dw, int G, int B, int C, int Tin, int Tout, int K, int stride,
                      int pad, int force_generic, void* stream);

/* Backward of the training-mode depthwise stage of an InvertedResidual in ONE call (autograd of asr.py:49 with the
 * BatchNorm backward of the stage after it applied on load and the ReLU6 / BatchNorm backward sums of the stage
 * before it on the way out):   g' = ga*g + gb*g2 + gc,   xin = relu6(xa*xpre + xb),
 *   dxin[b,c,u] = [0 < xa*xpre+xb < 6] * sum_j w[c][j] * g'[b,c,(u + pad - j)/stride]   -> dxin, stats (sum, sum*xpre)
 *   dw[c][j]    = sum_{b,t} g'[b,c,t] * xin[b,c,t*stride - pad + j]                      -> dw (wpartial: [G][C][K])
 * Tin / Tout are the forward conv's input / output lengths.  Stride 1 with a specialised K runs as one fused kernel
 * (the window walk of the data gradient also accumulates the weight gradient); other shapes, or force_split != 0,
 * run v100_dwconv_wgrad + v100_dwconv. */
int v100_dwconv_bwd(const float* g, const float* g2, const float* w, const float* ga, const float* gb,
                    const float* gc, const float* xpre, const float* xa, const float* xb, float* dxin,
                    float* stats, float* wpartial, float* dw, int G, int B, int C, int Tin, int Tout, int K,
                    int stride, int pad, int force_split, void* stream);

/* ---- K1 pointwise (1x1 conv as GEMM on MFMA) ----------------------------------------------
 * Y[b][m][t] = epilogue( sum_k A[m][k] * x'[b][k][t] (+ bias[m]) )
 * Replaces nn.Conv1d(kernel_size=1): asr.py:47 (pw), :51 (pw-linear), :91 (LinearCharDecoder),
 * tts.py:26,77 (heads); with A = W^T it is their backward-data.
 * epi_mode: 0 store   1 store + stats(sum, sum^2)   2 clamp(y*ea+eb,0,6)   3 y*ea+eb (+R)
 *           4 y *= [0 < R*ea+eb < 6] + stats(sum y, sum y*R)   5 y + R
 * stats: [v100_pw_num_parts(B,T)][M][2].  use_bf16: 1 = operands rounded to bf16 (A_bf16 = bf16 copy of A), fp32
 * accumulate; 2 = IEEE fp16 operands (A_bf16 = fp16 copy, v100_weight_prep_f16), inference combinations only
 * (x_mode 0 with epi_mode 0 / 2 / 3, anything else returns the shape status); 0 = exact fp32 MFMA. */
int v100_pw_num_parts(int B, int T);
int v100_pw_gemm(const float* A, const void* A_bf16, const float* X, const float* X2, const float* xa,
                 const float* xb, const float* xc, int x_mode, float* Y, const float* bias, const float* ea,
                 const float* eb, const float* R, int epi_mode, float* stats, int B, int M, int K, int T,
                 int use_bf16, void* stream);
/* The data gradient of Dropout(p) -> Conv1d(k=1) (asr.py:85-94) in one kernel: Y[b][m][t] = keep[b][m][t] ? (sum_k A[m][k] X[b][k][t]) / (1 - p) : 0
 * with keep the byte mask v100_dropout_fwd wrote.  The small-K form only (K <= 32: the vocabulary head's 29 classes; M >= 64, T % 4 == 0):
 * v100_pw_gemm_dropmask_supported tells, anything else returns the shape status.  use_bf16 as v100_pw_gemm (0 / 1). */
int v100_pw_gemm_dropmask_supported(int B, int M, int K, int T, int use_bf16);
int v100_pw_gemm_dropmask(const float* A, const void* A_bf16, const float* X, float* Y, const void* keep, float p, int B, int M, int K,
                          int T, int use_bf16, void* stream);

/* dW[m][k] = sum_{b,t} g'[b][m][t] * x'[b][k][t]   (weight gradient of the same 1x1 conv).
 * partial: [S][M][K] workspace, S = v100_pw_wgrad_splits(B, M, K): S <= B splits the batch, S = B * TS (small weight matrices)
 * also cuts every utterance's t range into TS chunks. */
int v100_pw_wgrad_splits(int B, int M, int K);
int v100_pw_wgrad(const float* G, const float* G2, const float* ga, const float* gb, const float* gc, int g_mode,
                  const float* X, const float* xa, const float* xb, int x_mode, float* partial, float* dW,
                  int S, int B, int M, int K, int T, int use_bf16, void* stream);

/* fp32 [rows][cols] weight -> optional bf16 copy, transposed fp32 copy, transposed bf16 copy. */
int v100_weight_prep(const float* w, int rows, int cols, void* w_bf16, float* wt, void* wt_bf16, void* stream);
/* w16[i] = IEEE half(w[i]): the weight copy of the fp16 inference precision (use_bf16 = 2 below) */
int v100_weight_prep_f16(const float* w, int rows, int cols, void* w16, void* stream);

/* ---- K3 BatchNorm1d (asr.py:36,52; eps 1e-5, momentum 0.1) ----------------------------------
 * finalize_train: slab of (sum x, sum x^2) -> scale = gamma*rstd, shift = beta - mean*scale, saved
 * mean/rstd, running stats (unbiased variance) and num_batches_tracked updated in place. */
int v100_bn_finalize_train(const float* stats, int parts, long long count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                           float eps, float* scale, float* shift, float* save_mean, float* save_rstd, int C, void* stream);
/* eval: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean*scale */
int v100_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, float* scale, float* shift, int C, void* stream);
/* backward: slab of (sum dz, sum dz*a) -> da = p*dz + q*a + r coefficients, dgamma, dbeta */
int v100_bn_bwd_finalize(const float* partial, int parts, long long count, const float* gamma, const float* mean,
                         const float* rstd, float* p, float* q, float* r, float* dgamma, float* dbeta, int C, void* stream);
/* FROZEN statistics under autograd (a block in eval() inside a training model: nn.BatchNorm1d normalises with the running statistics
 * and back-propagates through that fixed affine, asr.py:36,52 in eval mode): frozen_coeffs = eval_coeffs plus (mean, rstd) = (running_mean,
 * 1 / sqrt(running_var + eps)) for the backward; bwd_finalize_frozen: p = gamma * rstd, q = r = 0, dgamma / dbeta as in training. */
int v100_bn_frozen_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          float eps, float* scale, float* shift, float* mean, float* rstd, int C, void* stream);
int v100_bn_bwd_finalize_frozen(const float* partial, int parts, long long count, const float* gamma, const float* mean,
                                const float* rstd, float* p, float* q, float* r, float* dgamma, float* dbeta, int C, void* stream);

/* ---- per-channel elementwise / layout glue --------------------------------------------------*/
/* partial[g][c] = (sum u, sum u*v) over (b in group g, t); v NULL -> sum u*u */
int v100_chan_reduce2(const float* u, const float* v, float* partial, int G, int B, int C, int T, void* stream);
int v100_slab_sum0(const float* partial, int parts, float* out, int C, void* stream);
/* out = A[c]*u + Bc[c]*v + Cc[c]  (block output y = BN3(a3) + x, asr.py:55-59; BN3 backward) */
int v100_chan_affine2(const float* u, const float* v, const float* A, const float* Bc, const float* Cc, float* out,
                      int B, int C, int T, void* stream);
/* out = u*m*s  (nn.Dropout(0.2) with a pre-drawn keep mask, asr.py:90) */
int v100_mul_scale(const float* u, const float* m, float s, float* out, long long n, void* stream);
/* nn.Dropout(p), training mode (asr.py:90), one pass: element i is kept when a 32-bit uniform mixed from (seed, i) is >= p * 2^32;
 * y = keep ? x / (1 - p) : 0, mask[i] = keep (one byte per element, for v100_dropout_bwd: dx = mask ? dy / (1 - p) : 0).
 * Tensor storage must be 16-byte aligned (float4 path); n elements. */
int v100_dropout_fwd(const float* x, long long seed, float p, float* y, void* mask, long long n, void* stream);
int v100_dropout_bwd(const float* dy, const void* mask, float p, float* dx, long long n, void* stream);
/* [B][R][Cc] -> [B][Cc][R]  (torch.transpose(x,1,2) at the model edges, asr.py:111,114; tts.py:177,179) */
int v100_transpose_last2(const float* in, float* out, int B, int R, int Cc, void* stream);
/* out[b][c][t] = table[idx[b][t]][c]  (nn.Embedding + transpose, tts.py:81-83,176-177) and its backward */
int v100_embedding_bct(const long long* idx, const float* table, float* out, int B, int V, int C, int T, void* stream);
int v100_embedding_bwd(const long long* idx, const float* g, float* dtable, int B, int V, int C, int T, void* stream);


/* ---- K8 log-mel batch augmentation ---------------------------------------------------------
 * All of BatchSpectrogramAugumentation.forward (voice100/audio.py:27-108) in one pass over
 * x [B][Tin][F] -> y [B][Tout][F]; the random decisions are drawn by the caller in the reference's
 * order and passed as numbers (0 / 0.f = that op is off).  tm_s / tm_e / tm_a are HOST arrays of
 * n_tmask (<= 3) normalised [start, end) time spans and fill values; len = int32 lengths after the
 * stretch (device).  mix: 1 = mixaudio, 0 = maskaudio (one of the two always runs, audio.py:46-49). */
int v100_augment_fused(const float* x, const int* len, const float* uniform, float* y, int B, int Tin, int Tout,
                       int F, int stretch_rate, float pitch_rate, float amp, int n_tmask, const int* tm_s,
                       const int* tm_e, const float* tm_a, int fm_on, int fm_s, int fm_e, float fm_a, int noise_on,
                       float noise_low, float noise_high, float noise_std, int mix, float log_offset, void* stream);
/* The same pass fed the lengths BEFORE the stretch (len_raw, int32, device): it derives len * stretch_rate / 100 (audio.py:58)
 * itself and also writes it to len_out [B] and (len_out + 1) / 2 -- ConvVoiceEncoder.output_length, asr.py:80-81 -- to half_out [B]
 * (either may be null): the integer tensor ops of a training step folded into the pass that reads the lengths anyway. */
int v100_augment_fused_len(const float* x, const int* len_raw, int* len_out, int* half_out, const float* uniform, float* y,
                           int B, int Tin, int Tout, int F, int stretch_rate, float pitch_rate, float amp, int n_tmask,
                           const int* tm_s, const int* tm_e, const float* tm_a, int fm_on, int fm_s, int fm_e, float fm_a,
                           int noise_on, float noise_low, float noise_high, float noise_std, int mix, float log_offset,
                           void* stream);
/* ... and, for F == 64, the same values TRANSPOSED into yt [B, 64, Tout] by the same pass (null: not written): the layout the encoder
 * takes (torch.transpose(audio, 1, 2), asr.py:111), which otherwise costs a launch that reads y back. */
int v100_augment_fused_len_t(const float* x, const int* len_raw, int* len_out, int* half_out, const float* uniform, float* y,
                             float* yt, int B, int Tin, int Tout, int F, int stretch_rate, float pitch_rate, float amp, int n_tmask,
                             const int* tm_s, const int* tm_e, const float* tm_a, int fm_on, int fm_s, int fm_e, float fm_a,
                             int noise_on, float noise_low, float noise_high, float noise_std, int mix, float log_offset,
                             void* stream);

/* ---- K4 / K7 / K9 glue (csrc/features.hip) -------------------------------------------------
 * shift_copy: out[b][out_coff+c][u*out_mul+out_add] (=|+=) in[b][in_coff+c][u*in_mul+in_add] (0 when that
 * index is out of range) + bias[c], u in [0,n).  Builds the tap-stacked GEMM operand of
 * nn.ConvTranspose1d(512,256,k=5,stride=2,padding=2) (voice100/models/tts.py:22), interleaves its even/odd
 * output phases, and the inverse moves for its backward. */
int v100_shift_copy(const float* in, float* out, const float* bias, int B, int C, int Tin, int Tout, int in_ctot,
                    int in_coff, int out_ctot, int out_coff, int in_mul, int in_add, int out_mul, int out_add, int n,
                    int accumulate, void* stream);
/* frames[b][n][t] = x[b][reflect(t*hop + n + (n_fft-win)/2 - n_fft/2)]: torchaudio Spectrogram(center=True,
 * pad_mode="reflect") framing used by MelSpectrogramAudioTransform, voice100/data_modules.py:276-281 */
int v100_stft_frames(const float* x, float* frames, int B, int N, int T, int hop, int win, int n_fft, void* stream);
/* pw[b][f][t] = spec[b][f][t]^2 + spec[b][F+f][t]^2 */
int v100_power_spectrum(const float* spec, float* pw, int B, int F, int T, void* stream);
/* out[b][t][c] = log(in[b][c][t] + offset)   (data_modules.py:290-291) */
int v100_log_transpose(const float* in, float* out, int B, int C, int T, float offset, void* stream);
/* The whole of MelSpectrogramAudioTransform.forward's arithmetic (voice100/data_modules.py:276-291) in ONE launch: x [B][N] waveform ->
 * out [B][T][n_mels] = log(mel + log_offset), T = 1 + N / hop, torchaudio-0.13.1 MelSpectrogram defaults (center, reflect padding,
 * power 2).  One wave per frame: windowed load, 256-point complex radix-4 FFT of the 512 real samples, real-input split, power,
 * triangular filters, log (csrc/mel.hip).  n_fft = 512 and n_mels <= 64 only (the reference's configuration; 1 otherwise -- the
 * caller then uses the framing / DFT-GEMM / filterbank-GEMM kernels above).  Tables (device, built by the caller in double):
 * window [512] (the win_length Hann window centred in n_fft zeros), tw256 [256][2] / tw512 [257][2] = (cos, -sin)(2 pi k / 256 | 512),
 * mel_start / mel_count [n_mels] = first bin and number of bins (<= 32) of each filter, mel_w [n_mels][32] its weights. */
int v100_log_mel_fused(const float* x, float* out, const float* window, const float* tw256, const float* tw512, const int* mel_start,
                       const int* mel_count, const float* mel_w, int B, int N, int T, int hop, int n_fft, int n_mels, float log_offset,
                       void* stream);
/* AlignTextToAudioModel.predict epilogue (tts.py:197-200, _layers_v1.py:132-138): x [B][T][2+S+Cap] ->
 * f0 = gate(x0 < 0 ? 0 : x1*std+mean), logspc, codeap un-normalised */
int v100_world_unnormalize(const float* x, float* f0, float* logspc, float* codeap, const float* f0_mean, const float* f0_std,
                           const float* ls_mean, const float* ls_std, const float* ca_mean, const float* ca_std,
                           int B, int T, int S, int Cap, void* stream);
/* ---- K12 WORLDLoss, value + gradient in one pass (voice100/models/_layers_v1.py:37-93), with the target preparation of
 * AlignTextToAudioModel._calc_batch_loss (voice100/models/tts.py:203-206: hasf0 = f0 >= 30, WORLDNorm.normalize) --------
 * pred [B][Tp][A], A = 2 + S + Cap: the decoder output as is (hasf0 logit, f0_hat, logspc_hat[S], codeap_hat[Cap] per frame).
 * Targets f0 [B][Tt], logspc [B][Tt][S], codeap [B][Tt][Cap], length [B] int32; frames t < min(Tp, Tt) count
 * (adjust_size, _layers_v1.py:27-34) and of those t < length[b] (generate_padding_mask, :14-24).
 * f0_mean .. ca_std: the six WORLDNorm vectors (then the targets are RAW and hasf0 may be NULL = raw f0 >= 30), or all NULL
 * (targets already normalised; hasf0 [B][Tt] required).  w: [S] logspc weights (use_mel_weights) or NULL = mean over S.
 * l1: 0 squared error, 1 absolute error.  Writes loss[4] = (hasf0, f0, logspc, codeap) terms, each sum / sum(mask), and
 * unit [B][Tp][A] = d loss_term(a) / d pred for a unit upstream gradient.  partial: v100_world_loss_parts(B, Tp) x 4 floats. */
int v100_world_loss_parts(int B, int Tp);
int v100_world_loss(const float* pred, const float* f0, const float* hasf0, const float* logspc, const float* codeap,
                    const int* length, const float* f0_mean, const float* f0_std, const float* ls_mean, const float* ls_std,
                    const float* ca_mean, const float* ca_std, const float* w, float* partial, float* loss, float* unit,
                    int B, int Tp, int Tt, int S, int Cap, int l1, void* stream);
/* dpred[b][t][a] = unit[b][t][a] * gout[term(a)], gout [4] on the device */
int v100_world_loss_bwd(const float* unit, const float* gout, float* dpred, int B, int Tp, int S, int Cap, void* stream);
/* y = max(exp(x) - offset, 0)   (WORLDVocoder.decode, voice100/vocoder.py:99) */
int v100_exp_clip(const float* x, float* y, float offset, long long n, void* stream);

/* ---- WORLD synthesis (csrc/world.hip; SURVEY.md 8f rank 4, first half) -- PARITY PARTIALLY PINNED ----
 * pyworld.decode_aperiodicity + pyworld.synthesize as WORLDVocoder.decode calls them (voice100/vocoder.py:100-101).
 * pyworld 0.3.2 (C++ WORLD) is not in the reference tree: the kernels follow the published algorithm as restated in
 * oracle/world_synth.py.  fft_size 512 (16 kHz): one WAVE per pulse, fp32 with the tables below; any other power of two up to 2048 (1024:
 * the 22.05 kHz models): one workgroup per pulse in fp64, tw256 / tw512 / dc_remover may be NULL.
 *   v100_world_randn_host    HOST function, HOST pointer: the first n values of WORLD's randn() after randn_reseed() (every
 *                            Synthesis call reseeds: one fixed sequence for all utterances; callers upload it once)
 *   v100_world_decode_aperiodicity   coded [rows][nb] dB -> ap [rows][fft_size/2+1]
 *   v100_world_synthesize    f0 [B][T], sp / ap [B][T][257] -- or, instead of ap (then NULL), coded_ap [B][T][nb] in dB, decoded per
 *                            pulse without ever rounding an aperiodicity near 1 to fp32 (the reference decodes in double) --,
 *                            frames [B] int32 (NULL: every utterance has T frames),
 *                            randn_table [table_len >= samples per utterance], tw256 [256][2] / tw512 [257][2] = cos, -sin
 *                            of 2 pi k / 256 and / 512, dc_remover [512] (oracle.world_synth.dc_remover), all built in double
 *                            by the caller -> y [B][Ymax] fp32, Ymax = (int)(T * frame_period_ms * fs / 1000), zero beyond
 *                            an utterance's own length; n_pulses [B] (-1 and a NaN row: more than max_pulses pulses). */
/* ---- channel-major inference layout (round 4; csrc/block.hip v100_ir_fwd_eval with shape[10] == 2) --------------------------
 * Activations [C][B][P], P = (T + 7) & ~7: one [C x (B P)] matrix with every utterance's (padded) row back to back, so a 1x1
 * convolution is ONE GEMM over all B P columns (asr.py:47,51 on 1-second chunks: full 128-column tiles instead of 51 of 128) and a
 * channel's rows are contiguous for the depthwise kernel.  Block inputs / outputs fp32, hidden tensors in the GEMM operand format
 * (bf16, or fp16 at precision "fp16").  v100_bct_to_cm: [B][C][T] fp32 -> [C][B][P] (padding zeroed); v100_cm_to_btc: [C][B][P] ->
 * [B][T][C] (the model-edge transpose of the logits, asr.py:114). */
int v100_bct_to_cm(const float* in, float* out, int B, int C, int T, void* stream);
int v100_cm_to_btc(const float* in, float* out, int B, int C, int T, void* stream);

int v100_world_randn_host(float* host_out, long long n);
int v100_world_decode_aperiodicity(const float* coded, float* ap, long long rows, int nb, int fs, int fft_size, void* stream);
long long v100_world_synth_workspace_bytes(int B, int T, int fs, double frame_period_ms, int fft_size, int max_pulses);
int v100_world_synthesize(const float* f0, const float* sp, const float* ap, const float* coded_ap, int nb, const int* frames, const float* randn_table,
                          long long table_len, const float* tw256, const float* tw512, const float* dc_remover, float* y,
                          int* n_pulses, void* workspace, int B, int T, int fs, double frame_period_ms, int fft_size,
                          int max_pulses, void* stream);

/* ---- K11 dense k-tap conv blocks of the v2 models (csrc/layernorm.hip) -------------------
 * ConvLayerBlock / ConvTransposeLayerBlock, voice100/models/_layers_v2.py:29-89: conv -> LayerNorm over channels
 * -> exact GELU.  The dense convolution runs on the K1 GEMM over an im2col copy (rows tap-major: j*Cin + c);
 * col2im is its adjoint (backward-data), a gather.  ln_gelu_fwd: out = gelu(layer_norm_C(y) * gamma + beta) on
 * [B][C][T] (C <= 1024), saving mean / rstd [B][T]; ln_gelu_bwd: dy plus per-tile partial sums
 * partial[v100_ln_num_parts(B,T)][C][2] = (dgamma, dbeta), summed over parts by v100_slab_sum2. */
int v100_im2col(const float* x, float* cols, int B, int Cin, int Tin, int Tout, int k, int stride, int pad, void* stream);
int v100_col2im(const float* dcols, float* dx, int B, int Cin, int Tin, int Tout, int k, int stride, int pad, void* stream);
/* Stride-1 dense convolutions and the two output phases of ConvTranspose1d(k5,s2,p2) WITHOUT the im2col copy
 * (csrc/pointwise*.hip, "tap-addressed X"): the GEMM's contraction index is k = tap*cx + c and row k is row c of a
 * zero-padded copy Xp [B][cx][Tx] of the input, read from column t + shifts[tap] (0 <= shift <= 15, T + shift <= Tx):
 *   Y[b][m][t]        = sum_{tap,c} A[m][tap*cx + c] * Xp[b][c][t + shifts[tap]]  (+ bias[m]) (+ R[b][m][t], bf16/fp32 only)
 *   dW[m][tap*cx + c] = sum_{b,t<T} G[b][m*Tg + g_off + t] * Xp[b][c][t + shifts[tap]]      (G rows of pitch Tg)
 * nn.Conv1d(Cin,Cout,k,padding=(k-1)/2) forward: Xp = pad(x, (k-1)/2), shifts = 0..k-1, A[m][j*Cin+c] = W[m][c][j];
 * its backward-data is the same GEMM on pad(dy) with the taps reversed.  `shifts` is a HOST array of ntap (<= 8) ints.
 * v100_pad_copy: dst[b][c][lpad + i] = src[b][c][src_off + i*src_step], i < n, zero elsewhere in the Tx-long row
 * (src_step 2 splits a ConvTranspose gradient into its even / odd phases).  v100_pw_taps_supported = 1 when the two
 * GEMM entry points accept the shape at that precision (bf16/fp16 need cx % 64 == 0); otherwise use im2col. */
int v100_pad_copy(const float* src, float* dst, int B, int C, int Tsrc, int src_step, int src_off, int n, int Tx, int lpad, void* stream);
int v100_pw_taps_supported(int B, int M, int cx, int ntap, int T, int Tx, int use_bf16);
int v100_pw_gemm_taps(const float* A, const void* A_bf16, const float* Xp, float* Y, const float* bias, const float* R,
                      int B, int M, int cx, int T, int Tx, int ntap, const int* shifts, int use_bf16, void* stream);
int v100_pw_wgrad_taps(const float* G, int Tg, int g_off, const float* Xp, float* partial, float* dW, int S, int B, int M,
                       int cx, int T, int Tx, int ntap, const int* shifts, int use_bf16, void* stream);
int v100_ln_num_parts(int B, int T);
int v100_slab_sum2(const float* partial, int parts, float* out0, float* out1, int C, void* stream);
int v100_ln_gelu_fwd(const float* y, const float* gamma, const float* beta, float eps, float* out, float* mean,
                     float* rstd, int B, int C, int T, void* stream);
int v100_ln_gelu_bwd(const float* dout, const float* y, const float* gamma, const float* beta, const float* mean,
                     const float* rstd, float* dy, float* partial, int B, int C, int T, void* stream);

/* ---- opt-in kernel timing (bench.py roofline): HIP events on the launch stream for the hot kernels (tags 0-4) and, with tag 5
 * ("other"), in the dispatch packet of every other launch of the library (roofline_step.kernel_ms).
 * tags: 0 depthwise fwd, 1 depthwise bwd, 2 depthwise bwd-weight (stand-alone), 3 pointwise GEMM, 4 pointwise bwd-weight.
 * The depthwise launches are timed with the dispatch packet's own start/stop timestamps (hipExtLaunchKernelGGL: no
 * marker packets, the pair reads what rocprofv3 reports as the kernel's duration); the GEMM tags bracket their launches
 * with hipEventRecord (~3 us of queue time per launch, so time only what is read).
 * enable(mask): bit t of mask switches tag t on (0 = all off); a non-zero mask clears the counters.  read()
 * synchronises the device and returns the summed ms, launch count and (for the depthwise tags) the algorithmic bytes
 * of those launches: fp32 in + out + taps + BN coefficients. */
/* kernel launches issued by the library in this process so far (bench.py: launches per step measured in the run itself) */
long long v100_launch_count(void);
/* dst = src, nbytes (a multiple of 16) in 16-byte pieces: the device-copy yardstick bench.py reports beside the HBM-bound
 * kernels' GB/s ("fraction of both nominal and measured-copy bandwidth") */
int v100_copy_probe(const void* src, void* dst, long long nbytes, void* stream);
int v100_timing_enable(int tag_mask);
int v100_timing_read(int tag, double* ms, long long* count, double* bytes);
/* The depthwise streaming kernels' access pattern as a pure copy (bench.py yardstick only): dst[b][c][:] = src[b][c][:] for
 * row_bytes-byte rows of a [B][C][row_bytes] tensor, one workgroup per channel, one row per wave at a time, nontemporal.
 * row_bytes % 1024 == 0. */
int v100_rows_copy_probe(const void* src, void* dst, int B, int C, int row_bytes, void* stream);

/* ---- "act16": bf16 STORAGE of the big hidden tensors in bf16-operand training (csrc/block.hip) ---------------
 * In bf16 mode the reference under autocast keeps its conv activations in bf16; here the two tensors a block saves for
 * backward (a1 = expand output, a2 = depthwise output, each 4x the block's width) and optionally the two hidden gradients
 * (act16 level 2; level 3 adds the project output a3, saved for backward, and its gradient da3; level 4 a bf16 shadow of the block
 * output for the next block's X operand, v100_chan_affine2_shadow) are stored as bf16 [B][C][P], P = (T + 7) & ~7 (pitched rows: every 4- / 8-sample access stays 8 / 16-byte aligned for any
 * T).  Accumulators, BatchNorm statistics and reductions stay fp32.  io16 masks select which operands are such tensors:
 *   v100_pw_gemm_io:   1 X, 2 X2, 4 Y, 8 R        v100_pw_wgrad_io: 1 G, 2 G2, 4 X
 *   v100_dwconv_*_io:  1 x (first stream), 2 x2, 4 aux (a1), 8 y; 16 = the tensors are CHANNEL-MAJOR [C][B][P] (rows of up to 768
 *                      outputs; the layout probe of round 4: tools/bench_dw_regimes.py --cm, tests/test_gpu_act16.py)
 * Same arithmetic as v100_pw_gemm / v100_pw_wgrad / v100_dwconv / v100_dwconv_bwd (bf16 operands); only the combinations the
 * block executor issues exist, anything else returns 1 (no fallback). */
int v100_pw_gemm_io(const void* A_bf16, const void* X, const void* X2, const float* xa, const float* xb, const float* xc,
                    int x_mode, void* Y, const float* ea, const float* eb, const void* R, int epi_mode, float* stats,
                    int B, int M, int K, int T, int io16, void* stream);
int v100_pw_wgrad_io(const void* G, const void* G2, const float* ga, const float* gb, const float* gc, int g_mode,
                     const void* X, const float* xa, const float* xb, int x_mode, float* partial, float* dW, int S, int B,
                     int M, int K, int T, int io16, void* stream);
int v100_dw_mfma_supported(int K, int stride);
/* Row pitch, in elements, of every 16-bit-stored activation tensor [B][C][P] the _io entry points address: a multiple of 8 (16-byte
 * rows); for B > 1 and T >= 256 a multiple of 64 (rows start on 128-byte lines: a time-stretched 1136-byte row otherwise shares its
 * boundary lines with its neighbours and they are fetched twice).  The ONE rule (csrc/common.h v100_pitch16): callers allocate with it. */
int v100_row_pitch16(int T, int B);
int v100_dwconv_fwd_train_io(const void* a1, const float* w, const float* in_a, const float* in_b, void* a2, float* stats,
                             int G, int B, int C, int T, int K, int io16, void* stream);
int v100_dwconv_bwd_io(const void* g, const void* g2, const float* w, const float* ga, const float* gb, const float* gc,
                       const void* xpre, const float* xa, const float* xb, void* dxin, float* stats, float* wpartial,
                       float* dw, int G, int B, int C, int T, int K, int io16, void* stream);
/* v100_dwconv_bwd_io with every tensor bf16-stored, ONE group, finishing BatchNorm 1's backward itself: the kernel owns the channel's
 * sums (stats [C][2] = sum dz1, sum dz1*a1) after its last row, forms (p, q, r) -> pqr [3][C] and (dgamma, dbeta) from the saved mean /
 * rstd / gamma of BatchNorm 1 and writes the FINISHED gradient da1 = p*dz1 + q*a1 + r (bf16, rounded once) where v100_dwconv_bwd_io
 * writes dz1 -- the operand the expand weight gradient and the expand backward-data GEMM consume (autograd of asr.py:45-49).
 * B <= 32, T <= 512, specialised K only; 1 otherwise (no fallback). */
int v100_dwconv_bwd_da1_io(const void* g, const void* g2, const float* w, const float* ga, const float* gb, const float* gc,
                           const void* xpre, const float* xa, const float* xb, void* da1_out, float* stats, float* dw,
                           const float* bn1_gamma, const float* bn1_mean, const float* bn1_rstd, float* pqr, float* dgamma,
                           float* dbeta, int B, int C, int T, int K, void* stream);
/* the two block-boundary passes with a bf16-stored operand: io16 of v100_chan_reduce2_io: 2 = v is bf16 (sums of dy, dy*a3);
 * of v100_chan_affine2_io: 1 = u is bf16 (y = s3*a3 + t3 (+ x)), 6 = v and out are bf16 (da3 = p*dy + q*a3 + r) */
int v100_chan_reduce2_io(const void* u, const void* v, float* partial, int G, int B, int C, int T, int io16, void* stream);
/* the forward block output (asr.py:55-59) with a bf16 shadow beside it (act16 level 4): out = A*u + Cc (+ v) as fp32 [B][C][T] AND
 * rounded to bf16 in shadow [B][C][(T + 7) & ~7]; u is bf16 (pitched) when u_bf16, else fp32.  The shadow is what the NEXT block's
 * expand GEMM / expand weight gradient read as X (io16 bit PW_IO_X / WG_IO_X): same results, half the operand bytes. */
int v100_chan_affine2_shadow(const void* u, const float* v, const float* A, const float* Cc, float* out, void* shadow,
                             int B, int C, int T, int u_bf16, void* stream);
int v100_chan_affine2_io(const void* u, const void* v, const float* A, const float* Bc, const float* Cc, void* out,
                         int B, int C, int T, int io16, void* stream);
/* 1 when v100_ir_fwd_train / v100_ir_bwd accept shape[10] (act16) != 0 for this block */
int v100_ir_act16_supported(const int* shape);

/* ---- Adam step of every parameter in one launch (torch.optim.Adam as configured by asr.py:169-176 / tts.py:132-135, 239-241:
 * grad += weight_decay * p; m = lerp(m, grad, 1-beta1); v = beta2*v + (1-beta2)*grad^2; p -= lr/(1-beta1^step) * m /
 * (sqrt(v)/sqrt(1-beta2^step) + eps)).  chunks: DEVICE array of nchunks records {int tensor; int count; long long offset}
 * (count <= v100_adam_chunk_elems() elements of one tensor each); params / grads / exp_avg / exp_avg_sq: DEVICE arrays of float
 * pointers indexed by tensor.  Hyper-parameters are doubles: 1 - beta is formed in double, as torch does (1.f - 0.999f is 1.3e-5 off). */
int v100_adam_chunk_elems(void);
int v100_adam_step(const void* chunks, int nchunks, const void* params, const void* grads, const void* exp_avg,
                   const void* exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                   void* stream);

/* ---- K10 log_softmax + CTC (asr.py:148-152: F.log_softmax(-1) then nn.CTCLoss(blank, 'mean', zero_infinity=True)) --
 * logits [B][T][V] fp32, targets [B][Lmax] int64, in_len / tgt_len [B] int32 (device).  Writes nll[b] =
 * -log p(target_b | logits_b) (inf when infeasible) and grad[b][t][c] = d nll_b / d logits[b][t][c] (zero for
 * padded frames and infeasible utterances); the caller forms mean_b(nll_b / tgt_len_b) and scales grad.
 * workspace: v100_ctc_workspace_floats(B, T, Lmax) floats.  Limits: V <= 128, Lmax <= 2047
 * (256 threads x up to 16 lattice states each).  Labels outside [0, V) are treated as blank (the host wrapper can validate
 * them first: VOICE100_CHECK_IDS=1). */
int v100_ctc_workspace_floats(int B, int T, int Lmax);
/* v100_ctc_loss plus the 'mean' reduction in the library: loss[0] = mean over b of (nll_b finite ? nll_b / max(tgt_len_b, 1) : 0),
 * grad = d loss[0] / d logits */
int v100_ctc_loss_mean(const float* logits, const long long* targets, const int* in_len, const int* tgt_len, float* workspace,
                       float* nll, float* loss, float* grad, int B, int T, int V, int Lmax, int blank, void* stream);
int v100_ctc_loss(const float* logits, const long long* targets, const int* in_len, const int* tgt_len, float* workspace,
                  float* nll, float* grad, int B, int T, int V, int Lmax, int blank, void* stream);
/* v100_ctc_loss_mean with the gradient laid out [B, V, T] when grad_bvt != 0 (the logits stay [B, T, V]): the layout of the tensor
 * the reference transposes the logits FROM (asr.py:114), so the backward of that transpose needs no pass over the gradient. */
int v100_ctc_loss_mean_t(const float* logits, const long long* targets, const int* in_len, const int* tgt_len, float* workspace,
                         float* nll, float* loss, float* grad, int grad_bvt, int B, int T, int V, int Lmax, int blank, void* stream);

/* ---- block executor (csrc/block.hip): the whole kernel chain of one InvertedResidual block (asr.py:40-59) per call.
 * shape = {B, Cin, hid, Cout, T, K, stride, residual, bf16, prepped, act16} (11 ints; act16: see "act16" above -- then the
 * caller passes a1 / a2 as bf16 [B][hid][(T+7)&~7]); coef = 12 vectors of `hid` floats (BN scale/shift/mean/rstd
 * of the three BatchNorms) written by forward and read by backward.  Pointer tables (device pointers unless noted):
 *  fwd: x | w1 g1 b1 rm1 rv1 nbt1 | wd g2 b2 rm2 rv2 nbt2 | w3 g3 b3 rm3 rv3 nbt3 | a1 a2 a3 y coef workspace prep   (26)
 *  bwd: x a1 a2 a3 | w1 wd w3 | g1 g2 g3 | coef | dy | dx (NULL = skip) | dW1 dg1 db1 dWd dg2 db2 dW3 dg3 db3 | workspace prep (24)
 * prep = v100_ir_prep_bytes(shape) bytes: bf16 / transposed weight copies written by forward (unless shape.prepped: the
 * caller has filled it, e.g. with v100_ir_prep_batched for the n <= 32 blocks of a stack in one launch; shapes = n x 10
 * ints) and reused by backward.  `shape(s)` and the tables are HOST arrays. */
long long v100_ir_prep_bytes(const int* shape);
long long v100_ir_fwd_workspace_bytes(const int* shape);
/* eval mode (inference): folded BatchNorm coefficients + bf16 weight copies in a caller-owned cache (refill with
 * v100_ir_eval_prep whenever a parameter or running statistic changes), then 3 launches per block and call.
 *  prep ptrs: w1 | g1 b1 rm1 rv1 | g2 b2 rm2 rv2 | w3 | g3 b3 rm3 rv3 | cache   (15)
 *  fwd  ptrs: x | w1 wd w3 | cache | h1 [B,hid,T] h2 [B,hid,T'] y [B,Cout,T']   (8) */
long long v100_ir_eval_cache_bytes(const int* shape);
int v100_ir_eval_prep(const int* shape, const void* const* ptrs, void* stream);
int v100_ir_fwd_eval(const int* shape, const void* const* ptrs, void* stream);
/* n eval-mode blocks back to back in ONE host call (inference at the configs' own sizes is bound by the host's per-call cost, not by
 * the GPU): shapes = n x 11 ints, ptrs = n x 8 pointers, each block's in v100_ir_fwd_eval's order (block i's y is block i+1's x:
 * the caller lays the buffers out).  HOST arrays.  Stops at the first block that fails and returns its code. */
int v100_ir_stack_fwd_eval(int n, const int* shapes, const void* const* ptrs, void* stream);
int v100_ir_prep_batched(const int* shapes, const void* const* w1s, const void* const* w3s, void* const* preps, int n, void* stream);
int v100_ir_fwd_train(const int* shape, const void* const* ptrs, void* stream);
long long v100_ir_bwd_workspace_bytes(const int* shape);
int v100_ir_bwd(const int* shape, const void* const* ptrs, void* stream);

/* ---- stack executor: a run of consecutive InvertedResidual blocks (ConvVoiceEncoder.layers, asr.py:67-76; VoiceDecoder.layers[0:4] /
 * [5:8], tts.py:17-25; TextToAlignTextModel.layers[0:4], tts.py:72-76) forward / backward in ONE host call each: the sequencing of
 * v100_ir_prep_batched + v100_ir_fwd_train per block (resp. v100_ir_bwd in reverse), same kernels and results.
 * desc (HOST ints): {n, B, T, bf16, act16_level, want_last_shadow} then n x {cin, hid, cout, k, stride, residual}.
 * v100_ir_stack_plan fills plan (HOST, 8 per block + 6 long longs): per block the byte offsets into the activation blob of
 * a1, a2, a3, y, y16 (-1: none), coef, prep and the block's output length; then {blob_bytes, bwd_ws_bytes, grad_floats, ...}; returns
 * blob_bytes (-1: bad descriptor).  The caller allocates the blob (kept until backward), the backward workspace and the gradient
 * buffer: grad_floats floats, per block dW1 dg1 db1 dWd dg2 db2 dW3 dg3 db3.
 * params (HOST pointer table, 18 per block): w1 g1 b1 rm1 rv1 nbt1 | wd g2 b2 rm2 rv2 nbt2 | w3 g3 b3 rm3 rv3 nbt3.
 * x16 (may be NULL): bf16 shadow [B][cin][(T + 7) & ~7] of x (act16 level 4). */
long long v100_ir_stack_plan(const int* desc, long long* plan);
int v100_ir_stack_fwd_train(const int* desc, const void* const* params, const void* x, const void* x16, void* blob, void* stream);
int v100_ir_stack_bwd(const int* desc, const void* const* params, const void* x, const void* x16, const void* blob, const void* dy,
                      void* dx, void* grads, void* ws, void* stream);

/* ---- SURVEY 8(f) "next" rows: integer decode / alignment steps on the device (csrc/decode.hip), bit-exact ----
 * greedy CTC decode: argmax per frame (first maximum), collapse repeats, drop blanks (voice100/text.py:99-104);
 * out [B][T] int64 (zero padded), out_len [B]. */
int v100_ctc_greedy_decode(const float* logits, const int* lens, long long* out, int* out_len, int B, int T, int V, int blank, void* stream);
/* ctc_best_path (voice100/models/align.py:18-66) batched: logp [B][T][V] log-probabilities, labels [B][Lmax] int64;
 * back_ws: B*T*(2*Lmax+1) int16; path [B][T] int32 = index into the blank-expanded labels; score [B]. */
int v100_ctc_best_path(const float* logp, const long long* labels, const int* in_len, const int* lab_len, void* back_ws, int* path,
                       float* score, int B, int T, int V, int Lmax, int max_move, void* stream);
/* TextToAlignTextModel.align (voice100/models/tts.py:89-110) batched: text [B][Lmax] int64, align [B][Lmax][2] float64
 * (gap, length), out [B][Tmax] int64 (zero padded; rows longer than Tmax are cut), out_len [B].  out == NULL: only out_len is
 * written (head + int(sum(align)) + tail per utterance), so the caller can size `out` with one read-back. */
int v100_align_expand(const long long* text, const double* align, const int* text_len, long long* out, int* out_len, int B, int Lmax,
                      int Tmax, int head, int tail, void* stream);


/* ---- WORLD analysis (csrc/world_analysis.hip; SURVEY.md 8f rank 4, second half) -- PARITY UNPINNED -------------------------------
 * pyworld.dio / cheaptrick / d4c / code_aperiodicity as WORLDVocoder.encode calls them (voice100/vocoder.py:61-74), in fp64 (the
 * reference converts the waveform to double).  pyworld 0.3.2 (C++ WORLD) is not in the reference tree: the kernels follow the
 * published algorithms as restated in oracle/world_analysis.py.  Waveforms x [B][pitch] fp32 with lengths [B] (samples); frame t of
 * an utterance sits at t * frame_period_ms / 1000 s and an utterance of n samples has v100_world_frames(fs, n, frame_period_ms) =
 * (int)(1000 n / fs / frame_period_ms) + 1 frames; Tmax = that of max_len is the row pitch of every per-frame array.
 *   v100_world_randn_host_f64   HOST function, HOST pointer: WORLD's randn() sequence in double (the safeguard noise of the
 *                               analysis: every call of cheaptrick / d4c restarts it); v100_world_randn_bound(kind, T, fs, fft_size)
 *                               = draws T frames can consume (kind 0 cheaptrick, 1 d4c): the least table_len to pass.
 *   v100_world_dio_bands        number of DIO bands; half_lengths [bands] (HOST) = matlab_round(fs / boundary_f0 / 2) of each.
 *   v100_world_dio              speed = 1 (no decimation).  lowcut [2 hc + 1], hc = matlab_round(fs / 50): the centred taps of
 *                               DesignLowCutFilter; nuttall [bands][4 hmax]: row i = NuttallWindow(4 half_lengths[i]), hmax = the
 *                               largest half length, both built in double by the caller -> f0 [B][Tmax] (0 = unvoiced, 0 beyond an
 *                               utterance's frames).
 *   v100_world_cheaptrick       f0 [B][Tmax] double; twiddle [fft_size/2][2] = cos, -sin of 2 pi k / fft_size; offsets [B][Tmax]
 *                               int64 scratch -> sp [B][Tmax][fft_size/2+1] double and/or logsp = (float) log(sp + log_offset)
 *                               (vocoder.py:70); rows of frames the table or LDS cannot serve come back NaN.
 *   v100_world_d4c              twiddle [1024][2] for the internal 2048-point FFTs, nuttall [window_length] = NuttallWindow(
 *                               (int)(3000 * 2048 / fs) * 2 + 1) -> ap [B][Tmax][fft_size/2+1] double and/or coded (double) /
 *                               coded32 (float) [B][Tmax][bands of 3 kHz] = pyworld.code_aperiodicity of it (vocoder.py:72-73). */
int v100_world_randn_host_f64(double* host_out, long long n);
long long v100_world_randn_bound(int kind, int T, int fs, int fft_size);
int v100_world_frames(int fs, int length, double frame_period_ms);
int v100_world_dio_bands(int fs, double f0_floor, double f0_ceil, double channels_in_octave, int* half_lengths, int max_bands);
long long v100_world_dio_workspace_bytes(int B, int max_len, int fs, double f0_floor, double f0_ceil, double channels_in_octave,
                                         double frame_period_ms);
int v100_world_dio(const float* x, const int* lengths, int B, int max_len, int pitch, int fs, double f0_floor, double f0_ceil,
                   double channels_in_octave, double frame_period_ms, double allowed_range, const double* lowcut,
                   const double* nuttall, double* f0, void* workspace, void* stream);
int v100_world_cheaptrick(const float* x, const int* lengths, const double* f0, int B, int max_len, int pitch, int fs,
                          double frame_period_ms, double q1, int fft_size, const double* randn_table, long long table_len,
                          const double* twiddle, double* sp, float* logsp, double log_offset, long long* offsets, void* stream);
long long v100_world_d4c_workspace_bytes(int B, int max_len, int fs, double frame_period_ms);
int v100_world_d4c(const float* x, const int* lengths, const double* f0, int B, int max_len, int pitch, int fs, double frame_period_ms,
                   double threshold, int fft_size, const double* randn_table, long long table_len, const double* twiddle,
                   const double* nuttall, int window_length, double* ap, double* coded, float* coded32, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VOICE100_HIP_H */
