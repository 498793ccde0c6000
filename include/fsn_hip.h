/*
 * fsn_hip.h - C ABI of libfsn_hip.so, the MI355X (gfx950) implementation of the FullSubNet
 * enhancement path.  Every entry point replaces one call site of the reference (cited as
 * path:line relative to the Audio-WestlakeU/FullSubNet checkout); INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to contiguous fp32 arrays owned by the caller (PyTorch's
 *     caching allocator in practice); the library never allocates device memory;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued and the call returns at once.  The
 *     device the stream belongs to is made current for the duration of the call (and restored), so one
 *     process may drive several GPUs.  The only state the library keeps is one small record per
 *     (device, stream) - an auxiliary stream with fork / join events for the left-over sub-band tiles and
 *     the profiler's events - so calls on DIFFERENT streams (from one or several host threads) are
 *     independent; calls that share a stream must be issued one at a time, like any stream-ordered API;
 *   - return value 0 = OK, < 0 = error; fsn_last_error() gives the thread-local message;
 *   - tensor shapes use the reference's names: B batch, L samples, F = n_fft/2+1 bins,
 *     T = 1 + L/hop frames, la = look_ahead.
 */
#ifndef FSN_HIP_H
#define FSN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSN_OK 0
#define FSN_ERR_ARG (-1)       /* bad shape / null pointer / unsupported configuration */
#define FSN_ERR_WORKSPACE (-2) /* workspace too small                                  */
#define FSN_ERR_LAUNCH (-3)    /* hipLaunch / hipMemsetAsync failed                    */
#define FSN_ERR_TIMEOUT (-4)   /* a persistent kernel ran out of time (see "residency contract" below) */

#define FSN_NORM_OFFLINE_LAPLACE 0    /* audio_zen/model/base_model.py:204-218 */
#define FSN_NORM_CUMULATIVE_LAPLACE 1 /* audio_zen/model/base_model.py:221-251 */
#define FSN_NORM_OFFLINE_GAUSSIAN 2   /* audio_zen/model/base_model.py:295-310 (fsn_norm only) */
#define FSN_NORM_CUMULATIVE_LAYER 3   /* audio_zen/model/base_model.py:312-354 (fsn_norm only) */
#define FSN_NORM_FORGETTING 4         /* audio_zen/model/base_model.py:103-151 (fsn_norm only) */

/* Arithmetic of the three large sub-band kernels.  F32 (the default, what every parity claim and bench.py's
 * `value` refer to): v_mfma_f32_16x16x4_f32, bit-equal to an fmaf chain.  F16X3: opt-in experiment - fp32
 * operands split into two fp16 halves, three 16-bit MFMAs per product block, fp32 accumulation. */
#define FSN_ARITH_F32 0
#define FSN_ARITH_F16X3 1
/* Training entries only (fsn_lstm2_forward_train / fsn_lstm2_backward): the arithmetic of torch.autocast, which the
 * reference trains under (fullsubnet/trainer.py:56, train.toml:5 use_amp = true) - both operands of every LSTM
 * product are rounded to fp16 / bf16 at the matrix core's input, accumulation and everything stored stay fp32. */
#define FSN_ARITH_F16 2
#define FSN_ARITH_BF16 3
/* Flag for the `arith` argument of fsn_lstm2_forward_train / fsn_lstm2_backward(_phase) / their workspace queries, OR-ed to
 * FSN_ARITH_F16 / _BF16 (the SAME value must reach the forward and the backward call of a step): the activated gates the
 * backward pass re-reads are saved in the arithmetic's 16-bit type instead of fp32 - what the vendor LSTM the reference
 * trains on keeps in its reserve space under autocast (fullsubnet/trainer.py:56) - inside the same save buffers; the cell
 * sequence, the hidden sequences and every gradient stay fp32.  Takes effect where the 16-bit arithmetic's own kernels run
 * (lstm_group16_kernels.hip); ignored elsewhere.  Half the save traffic of the two persistent launches. */
#define FSN_ARITH_SAVES16 0x100

const char* fsn_last_error(void);
/* The ABI revision this header describes.  fsn_version() returns the revision the LIBRARY was built with: a caller
 * compares the two once after loading (fullsubnet_amd/_lib.py raises on a mismatch) - argument lists changed between
 * revisions (116: + fsn_lstm_layer_plan_rows; 115: + fsn_train_rows_pieces; 114: + fsn_lstm2_train_is_persistent; 113: + fsn_gru2_forward; 112: + FSN_ARITH_SAVES16; 111: + the composed families' glue entries; 110: fsn_train_dims.norm, fsn_train_den_elems; 100 -> 101 of round 4: fsn_clip_adam_step's found_inf). */
#define FSN_ABI_VERSION 117
int fsn_version(void);

/* ---- STFT / iSTFT : audio_zen/acoustics/feature.py ------------------------------------- */

/* feature.py:9-50  stft(y, n_fft, hop, win) -> (mag, phase, real, imag); phase is never used on
 * the path (inferencer.py:132 discards it) and is not produced.
 * y [B, L];  window [n_fft] (torch.hann_window(n_fft), feature.py:38);  real/imag/mag [B, F, T],
 * any of the three may be NULL.  win_length == n_fft; n_fft = 512 / hop = 256 (every FullSubNet TOML)
 * runs on the radix-8 kernels, any other even n_fft in [16, 4096] / hop in [1, n_fft] (e.g. 512 / 128
 * and 960 / 480 of improved_fullsubnet/model.py:550-557) on a direct fp64 DFT.  Either way the
 * result is the correctly rounded transform of the fp32 windowed frame. */
int fsn_stft(const float* y, int B, int L, int n_fft, int hop, int win_length, const float* window,
             float* real, float* imag, float* mag, void* stream);

/* feature.py:53-91  istft((real, imag), n_fft, hop, win, length, input_type="real_imag").
 * real/imag [B, F, T] -> y [B, length].  workspace >= fsn_istft_workspace_bytes(B, T, n_fft). */
size_t fsn_istft_workspace_bytes(int B, int T, int n_fft);
int fsn_istft(const float* real, const float* imag, int B, int T, int n_fft, int hop, int win_length,
              const float* window, int length, float* y, void* workspace, size_t workspace_bytes,
              void* stream);

/* ---- cIRM mask algebra : audio_zen/acoustics/mask.py ----------------------------------- */

/* mask.py:47-64  decompress_cIRM(mask, K=10, limit=9.9), elementwise over n floats. */
int fsn_decompress_cirm(const float* mask, float* out, size_t n, void* stream);
/* mask.py:32-44  compress_cIRM(mask, K=10, C=0.1). */
int fsn_compress_cirm(const float* mask, float* out, size_t n, void* stream);
/* mask.py:7-29  build_complex_ideal_ratio_mask: four [n] planes -> out [n, 2] (compressed). */
int fsn_build_cirm(const float* noisy_real, const float* noisy_imag, const float* clean_real,
                   const float* clean_imag, float* out, size_t n, void* stream);

/* ---- feature norms : audio_zen/model/base_model.py --------------------------------------- */

/* norm_wrapper (base_model.py:356-372): any of the five norms on x [B, C, F, T] -> y (same shape; y may alias x), as
 * the composed models call them between their SequenceModel blocks.  sample_length: forgetting_norm's argument
 * (192 in the reference's signature; ignored by the others).  eps <= 0: the reference's constant for that norm
 * (1e-5 / fp32 epsilon / 1e-5 / fp32 epsilon / 1e-10); improved_fullsubnet/model.py:124-216 uses fp32 epsilon in its
 * offline norms and passes it.  Statistics are summed exactly (fp64) and then follow the reference's fp32 tensor
 * arithmetic operation by operation. */
#define FSN_NORM_MAX_FRAMES 6144 /* T of one fsn_norm call (a row's per-frame statistics sit in LDS) */
size_t fsn_norm_workspace_bytes(int norm_type, int B, int C, int F, int T);
int fsn_norm(const float* x, float* y, int norm_type, int B, int C, int F, int T, int sample_length, float eps,
             void* workspace, size_t workspace_bytes, void* stream);

/* ---- FullSubNet model : recipes/dns_interspeech_2020/fullsubnet/model.py --------------- */

typedef struct fsn_fullsubnet_cfg {
    int num_freqs;        /* model.py:12  (257)                                   */
    int look_ahead;       /* model.py:13  (2)                                     */
    int sb_num_neighbors; /* model.py:16  (15); fb_num_neighbors must be 0        */
    int fb_hidden;        /* model.py:19  (512), multiple of 64                   */
    int sb_hidden;        /* model.py:20  384 (the persistent kernel's size)      */
    int norm_type;        /* model.py:21  FSN_NORM_*                              */
    int arith;            /* FSN_ARITH_* (0 = fp32 MFMA)                          */
} fsn_fullsubnet_cfg;

/* The 20 tensors of Model.state_dict() in the reference's layout (nn.LSTM: weight_ih [4H, I],
 * weight_hh [4H, H], gate rows i,f,g,o; nn.Linear: weight [O, I]).  SURVEY §8a rows A5 / A9. */
typedef struct fsn_fullsubnet_params {
    const float *fb_w_ih_l0, *fb_w_hh_l0, *fb_b_ih_l0, *fb_b_hh_l0;
    const float *fb_w_ih_l1, *fb_w_hh_l1, *fb_b_ih_l1, *fb_b_hh_l1;
    const float *fb_fc_w, *fb_fc_b;
    const float *sb_w_ih_l0, *sb_w_hh_l0, *sb_b_ih_l0, *sb_b_hh_l0;
    const float *sb_w_ih_l1, *sb_w_hh_l1, *sb_b_ih_l1, *sb_b_hh_l1;
    const float *sb_fc_w, *sb_fc_b;
} fsn_fullsubnet_params;

/* Re-tile the weights into MFMA B-fragment order (done once per checkpoint load, the analogue of
 * nn.LSTM.flatten_parameters(), sequence_model.py:114). */
size_t fsn_fullsubnet_packed_bytes(const fsn_fullsubnet_cfg* cfg);
int fsn_fullsubnet_pack(const fsn_fullsubnet_cfg* cfg, const fsn_fullsubnet_params* params,
                        void* packed, size_t packed_bytes, void* stream);

/* model.py:72-136  Model.forward(noisy_mag [B,1,F,T]) -> compressed cIRM [B,2,F,T]
 * (inference semantics: every sample keeps all F bins, i.e. num_groups_in_drop_band = 1, see
 * SURVEY quirk Q1). */
size_t fsn_fullsubnet_workspace_bytes(const fsn_fullsubnet_cfg* cfg, int B, int T);
int fsn_fullsubnet_forward(const fsn_fullsubnet_cfg* cfg, const void* packed, const float* noisy_mag,
                           int B, int T, float* crm_out, void* workspace, size_t workspace_bytes,
                           void* stream);

/* model.py:85-95 alone: look-ahead pad, norm, full-band model.  fb_output [B, F, T + look_ahead] is the tensor
 * `fb_output` of model.py:95 (stage-level parity checks; the input a batch-sharded full-band stage would
 * all-gather).  workspace >= fsn_fullsubnet_workspace_bytes(cfg, B, T). */
int fsn_fullsubnet_fullband(const fsn_fullsubnet_cfg* cfg, const void* packed, const float* noisy_mag, int B,
                            int T, float* fb_output, void* workspace, size_t workspace_bytes, void* stream);

/* The same model on a contiguous slice [row_begin, row_end) of the B*F flattened (b, f) rows that
 * model.py:121-128 hands to the sub-band LSTM as independent sequences - the multi-GPU partition of the path
 * (one rank per slice, then one all-gather of the slices).  noisy_mag is the WHOLE batch [B,1,F,T]; only the
 * utterances the slice touches are read: their full-band model (model.py:95) and norm statistics (model.py:92,111)
 * are computed whole, the sub-band model only on the slice.  crm_rows [row_end - row_begin][2][T] receives
 * the compressed cIRM of row n = b*F + f at crm_rows[n - row_begin] (= crm[b, :, f, :] of the full result).
 * A slice aligned to utterances does exactly the work of fsn_fullsubnet_forward on those utterances. */
size_t fsn_fullsubnet_rows_workspace_bytes(const fsn_fullsubnet_cfg* cfg, int B, int T, long row_begin,
                                           long row_end);
int fsn_fullsubnet_forward_rows(const fsn_fullsubnet_cfg* cfg, const void* packed, const float* noisy_mag,
                                int B, int T, long row_begin, long row_end, float* crm_rows,
                                void* workspace, size_t workspace_bytes, void* stream);

/* recipes/dns_interspeech_2020/inferencer.py:130-145  Inferencer.full_band_crm_mask:
 * noisy [B, L] -> enhanced [B, L] (stft -> model -> decompress -> complex mask -> istft), all
 * intermediates kept in the frame-major device layout.  crm_out (optional, may be NULL) receives
 * the compressed mask [B, 2, F, T] for parity checks. */
size_t fsn_enhance_workspace_bytes(const fsn_fullsubnet_cfg* cfg, int B, int L, int n_fft, int hop);
int fsn_enhance(const fsn_fullsubnet_cfg* cfg, const void* packed, const float* window,
                const float* noisy, int B, int L, int n_fft, int hop, float* enhanced, float* crm_out,
                void* workspace, size_t workspace_bytes, void* stream);

/* ---- streaming: the model on k more frames with carried state --------------------------------- */

/* Frame-by-frame / chunked form of fsn_fullsubnet_forward for real-time use (not in the reference, which only
 * has the whole-utterance loop of inferencer.py:130-145): `state` carries (h, c) of the four LSTM layers and the
 * running sums of the two cumulative Laplace norms (base_model.py:221-251; norm_type must be
 * FSN_NORM_CUMULATIVE_LAPLACE) from call to call - zero-fill it for a new stream.  mag [B, 1, F, k] are the
 * next k input frames of the model (the caller appends the look_ahead zero frames of model.py:85 at the end of
 * the stream), steps_done the number of frames passed in before this call, crm_out [B, 2, F, k] the model
 * output of exactly these steps (step s is the compressed mask of frame s - look_ahead).  Feeding a stream in
 * any chunking gives the offline result. */
size_t fsn_fullsubnet_stream_state_bytes(const fsn_fullsubnet_cfg* cfg, int B);
size_t fsn_fullsubnet_stream_workspace_bytes(const fsn_fullsubnet_cfg* cfg, int B, int k);
int fsn_fullsubnet_stream_step(const fsn_fullsubnet_cfg* cfg, const void* packed, void* state, size_t state_bytes,
                               int steps_done, const float* mag, int B, int k, float* crm_out, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- training step: one nn.LSTM layer with back-propagation through time ----------------- */

/* audio_zen/model/module/sequence_model.py:52-58 (nn.LSTM, one layer, batch_first, h0 = c0 = 0) as
 * used under autograd by fullsubnet/trainer.py:56-63.  Time-major rows: x [T][N][ldx] (columns
 * I..ldx-1 zero, ldx >= round_up(I,16)), hseq [T][N][H]; w_ih [4H][I], w_hh [4H][H], b_* [4H] in the
 * reference's layout.  N % 16 == 0, H % 64 == 0 (a smaller hidden size is run zero-padded: units
 * with all-zero weights and biases stay at h = c = 0, exactly).  `save` keeps the activated gates and
 * the cell sequence for the backward pass; save == NULL is inference mode (SequenceModel.forward under
 * no_grad, sequence_model.py:106-125): nothing but hseq is kept and, for H = 384, the rows run on the
 * persistent recurrent kernels of the sub-band model - with a narrow input (I <= 32) or as the layer above an
 * equally wide one (I = H = ldx) on the forms that build their input projection themselves, in whole rounds of
 * 2 - 4 row tiles per workgroup (several rounds beyond four tiles per CU); row tiles left over advance step by step
 * beside the launch from their own small projection (round 6). */
size_t fsn_lstm_layer_save_bytes(int T, int N, int H);
size_t fsn_lstm_layer_fwd_workspace_bytes(int T, int N, int I, int H);
int fsn_lstm_layer_forward(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                           const float* b_hh, int T, int N, int I, int H, float* hseq, void* save,
                           size_t save_bytes, void* workspace, size_t workspace_bytes, void* stream);
/* Two stacked layers of equal width H (nn.LSTM(num_layers = 2), sequence_model.py:52-58) forward with saved
 * activations: the result of two fsn_lstm_layer_forward calls (hseq0 / save0 and hseq1 / save1 in that entry's
 * layouts, each save buffer >= fsn_lstm_layer_save_bytes(T, N, H)), ready for two fsn_lstm_layer_backward calls.
 * The full-band shape (H = 512, N <= 64) runs as ONE persistent launch for both layers and all steps
 * (fb_chain_kernel) instead of 2 T launches; other shapes run layer by layer.
 * arith: FSN_ARITH_F32, or FSN_ARITH_F16 / FSN_ARITH_BF16 = the autocast arithmetic the reference trains under
 * (trainer.py:56): on the sub-band shape (the group kernels, 99 % of the step's products) both operands of every
 * product are rounded to 16 bits at the matrix core's input, accumulation and everything stored stay fp32; shapes
 * that run on other kernels compute in fp32 (wider than asked for).  Same rule for fsn_lstm2_backward, which then
 * expects dh1 scaled by the caller's loss scale (GradScaler) like any autocast backward. */
size_t fsn_lstm2_train_workspace_bytes(int T, int N, int I, int H, int arith);
/* 1 when BOTH directions of this shape run as persistent launches (forward: the group kernel - H = 384, 17 - 32 input
 * columns, 96+ row tiles in whole 64-row clusters up to 8 left-over tiles - or the chain; backward likewise): the shapes on
 * which the 16-bit arithmetic has kernels of its own.  Callers with more rows than one launch holds split them into such
 * pieces (the rows of a stack are independent sequences: sequence_model.py:52-58; fullsubnet_amd.train.lstm2_train_chunks). */
int fsn_lstm2_train_is_persistent(int T, int N, int I, int H);
int fsn_lstm2_forward_train(const float* x, long ldx, const float* w_ih0, const float* w_hh0, const float* b_ih0,
                            const float* b_hh0, const float* w_ih1, const float* w_hh1, const float* b_ih1,
                            const float* b_hh1, int T, int N, int I, int H, float* hseq0, float* hseq1, void* save0,
                            void* save1, size_t save_bytes, void* workspace, size_t workspace_bytes, int arith,
                            void* stream);
/* Two stacked LSTM layers in inference mode, advanced as a wavefront (layer 1 at step t next to layer 0 at
 * step t + 1): T + 1 dependent launches instead of 2 T.  Either nn.LSTM(num_layers = 2) of one SequenceModel
 * (H1 == H0, sequence_model.py:52-58) or two consecutive single-layer blocks of different widths (the
 * encoder / decoder pairs of fast_fullsubnet/model.py:35-96; layer 1 takes the H0 outputs of layer 0).  For
 * blocks with few rows (the latency-bound regime); hseq1 [T][N][H1] is the hidden sequence of the second layer. */
size_t fsn_lstm2_fwd_workspace_bytes(int T, int N, int I, int H0, int H1);
/* 1 when fsn_lstm2_forward runs this call (T steps, N rows of ldx floats) as ONE persistent launch (equal widths of
 * 384 / 512 with up to 64 rows and 4095 steps: the chain kernel; 384 twice, up to 32 input columns in rows of exactly
 * 16 or 32 floats, 1536 - 2559 or 3584 - 4096 rows in whole 64-row clusters, hidden sequence below 2 GB: the group
 * kernel) - then it is also the better choice above the few-row regime it was made for.  Anything else runs on the
 * generic path of fsn_lstm2_forward (never an error). */
int fsn_lstm2_forward_is_persistent(int T, int N, int I, long ldx, int H0, int H1);
int fsn_lstm2_forward(const float* x, long ldx, const float* w_ih0, const float* w_hh0, const float* b_ih0,
                      const float* b_hh0, const float* w_ih1, const float* w_hh1, const float* b_ih1,
                      const float* b_hh1, int T, int N, int I, int H0, int H1, float* hseq1, void* workspace,
                      size_t workspace_bytes, void* stream);

/* improved_fullsubnet/model.py:402-440 (SubbandModel.forward up to the sequence model) for one band section
 * [lower, upper) of the F-bin inputs noisy / fb_out [B][F][T]: unit u sees the noisy bins lower + u c - n .. (c =
 * sb_center, n = sb_neighbor; model.py:315-400 `_freq_unfold`, reflected at both ends of the spectrum) and the same
 * window of the full-band output (fb_center, fb_neighbor), concatenated and divided by (the mean of the WHOLE section's
 * unfolded input of that utterance + eps): offline_laplace_norm, model.py:124-150 (eps = torch.finfo(float32).eps
 * there).  The result is written for units [unit_lo, unit_hi) in the layout the LSTM entries take: out [T][Np][ldo],
 * row b (unit_hi - unit_lo) + (u - unit_lo), columns beyond the window and rows beyond B (unit_hi - unit_lo) zero
 * (Np <= 65535 rows, ldo <= 240 columns).
 * The unfolded tensor is never formed (three launches instead of nine per section). */
size_t fsn_improved_section_input_workspace_bytes(int B, int F);
int fsn_improved_section_input(const float* noisy, const float* fb_out, int B, int F, int T, int lower, int upper,
                               int sb_center, int sb_neighbor, int fb_center, int fb_neighbor, int unit_lo, int unit_hi,
                               float eps, float* out, int Np, int ldo, void* workspace, size_t workspace_bytes,
                               void* stream);

/* The tensor glue AROUND the models of the composed families (round 5; section_kernels.hip), all bit-identical to the tensor
 * algebra it replaces:
 * fsn_improved_front      improved_fullsubnet/model.py:565-566: mag [B][F][T] -> out [B][F - 1][T] = mag ** fdrc without the
 *                         last bin, contiguous; sqrt_mode 1: fdrc = 0.5 (sqrtf, what torch.pow computes for that exponent),
 *                         0: fdrc = 1.
 * fsn_bft_to_rows         x [B][F][T] -> h [T][Np][Ip], zero beyond (B, F): the layout every LSTM / Linear entry takes
 *                         (audio_zen/model/module/sequence_model.py:106-125 permutes to [B, T, F] for nn.LSTM).
 * fsn_rows_to_bft         o [T][Np][ld] -> y [B][O][T]: the way back (sequence_model.py:123).
 * fsn_improved_mask_apply model.py:438-449 (the sections' outputs re-ordered and concatenated along the bins), :575 F.pad of
 *                         the last bin and :576-577 (mask x noisy real / imaginary part, no complex product) in one pass:
 *                         section i's output o [T][Np][ld] (as fsn_linear_forward writes it: row b units + u, column
 *                         comp center + cc) -> er / ei [B][F][T] at bin lower + u center + cc; bins no section covers: 0. */
typedef struct fsn_mask_section {
    const void* o; /* [T][Np][ld] */
    int Np, ld, lower, units, center;
} fsn_mask_section;
int fsn_improved_front(const float* mag, int B, int F, int T, int sqrt_mode, float* out, void* stream);
int fsn_bft_to_rows(const float* x, int B, int F, int T, float* h, int Np, int Ip, void* stream);
int fsn_rows_to_bft(const float* o, int T, int Np, int ld, int B, int O, float* y, void* stream);
int fsn_improved_mask_apply(int n, const fsn_mask_section* sections, const float* real, const float* imag, int B, int F, int T,
                            float* er, float* ei, void* stream);

/* Two stacked GRU layers of equal width (nn.GRU(num_layers = 2) of a SequenceModel, sequence_model.py:59-66; weights in nn.GRU's
 * layout: w_ih [3H][I], w_hh [3H][H], biases [3H], gate rows r, z, n) with few rows - the full-band model of a GRU FullSubNet -
 * as ONE persistent launch (the chain kernel with the GRU written as a four-gate cell) instead of 2 T per-step launches.
 * fsn_gru2_forward_supported: 1 for H = 384 / 512, N <= 64 rows (a multiple of 16), T <= 4095 on a device that holds the
 * chain's grid; anything else: fsn_gru_layer_forward layer by layer.  x [T][N][ldx] time-major, hseq1 [T][N][H]. */
int fsn_gru2_forward_supported(int T, int N, int H);
size_t fsn_gru2_fwd_workspace_bytes(int T, int N, int I, int H);
int fsn_gru2_forward(const float* x, long ldx, const float* w_ih0, const float* w_hh0, const float* b_ih0, const float* b_hh0,
                     const float* w_ih1, const float* w_hh1, const float* b_ih1, const float* b_hh1, int T, int N, int I, int H,
                     float* hseq1, void* workspace, size_t workspace_bytes, void* stream);

/* Up to eight INDEPENDENT two-layer stacks over the same T frames: the band sections of
 * improved_fullsubnet/model.py:402-449, whose SequenceModels have B x {20, 25, 6, 4} rows and input widths 62 .. 180 at
 * 48 kHz.  Per stack the arguments of fsn_lstm2_forward.  When every stack has H0 = H1 = 384 and together they fill
 * most of the chip (fsn_lstm2_multi_is_persistent: 3/4 .. 1 x CUs / 8 clusters of 64 rows) they run as ONE persistent
 * launch of the group kernel with one weight set per stack; otherwise one stack after the other as fsn_lstm2_forward
 * would (callers with small stacks do better to put them on one stream each).  Results equal fsn_lstm2_forward's per
 * stack within fp32 rounding (the kernels differ in their order of accumulation). */
typedef struct fsn_lstm2_stack {
    const float* x; /* [T][N][ldx], columns I .. ldx-1 zero */
    long ldx;
    const float *w_ih0, *w_hh0, *b_ih0, *b_hh0, *w_ih1, *w_hh1, *b_ih1, *b_hh1;
    int N, I, H0, H1;
    float* hseq1; /* [T][N][H1] out */
} fsn_lstm2_stack;
int fsn_lstm2_multi_is_persistent(int n, const fsn_lstm2_stack* stacks, int T);
size_t fsn_lstm2_multi_workspace_bytes(int n, const fsn_lstm2_stack* stacks, int T);
int fsn_lstm2_forward_multi(int n, const fsn_lstm2_stack* stacks, int T, void* workspace, size_t workspace_bytes,
                            void* stream);

/* Streaming inference (chunked / frame-by-frame processing with carried state - the real-time use the
 * model is designed for; the reference has no such entry point, nn.LSTM's (h_0, c_0) argument is the
 * analogue).  fsn_lstm_layer_pack re-tiles one layer's weights once; fsn_lstm_layer_forward_state then runs
 * T more steps starting from h_state / c_state [N][H], both updated in place, with two launches plus one per
 * step.  workspace >= fsn_lstm_layer_state_workspace_bytes(T, N, H). */
size_t fsn_lstm_layer_packed_bytes(int I, int H);
int fsn_lstm_layer_pack(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int I, int H,
                        void* packed, size_t packed_bytes, void* stream);
size_t fsn_lstm_layer_state_workspace_bytes(int T, int N, int H);
int fsn_lstm_layer_forward_state(const float* x, long ldx, const void* packed, int T, int N, int I, int H,
                                 float* hseq, float* h_state, float* c_state, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* dh [T][N][H] = dLoss/dhseq.  Outputs: dx [T][N][lddx] (may be NULL), dw_ih [4H][I], dw_hh [4H][H],
 * db [4H] (= d b_ih = d b_hh). */
size_t fsn_lstm_layer_bwd_workspace_bytes(int T, int N, int I, int H);
int fsn_lstm_layer_backward(const float* dh, const float* x, long ldx, const float* w_ih, const float* w_hh, int T,
                            int N, int I, int H, const float* hseq, const void* save, float* dx, long lddx,
                            float* dw_ih, float* dw_hh, float* db, void* workspace, size_t workspace_bytes,
                            void* stream);

/* The LAST layer of an inference stack together with the nn.Linear(H, O) that follows it (sequence_model.py:106-125:
 * self.fc_output_layer(self.sequence_model(x)); Fast FullSubNet's bottleneck, fast_fullsubnet/model.py:66-74: 16 384 rows x
 * 384 units -> one value per step).  Where the persistent kernel that forms the projection itself takes the layer
 * (fsn_lstm_layer_fc_supported: H = I = ldx = 384, O = 1 or 2, whole rounds of 2 - 4 row tiles per CU) its fused two-row
 * output layer does the nn.Linear too: the [T][N][H] hidden sequence is neither written nor read back.  out0 / out1:
 * PRE-activation outputs 0 / 1, time-major [T][ldo] (out1 may be NULL when O == 1); the caller applies the block's
 * activation.  Any other shape: FSN_ERR_ARG (fsn_lstm_layer_forward + fsn_linear_forward take it). */
/* Row padding hint for a caller that owns the padding of a stand-alone inference layer's rows: the row count (a multiple of 16,
 * >= N) to allocate so that the persistent kernels take every row with no left-over tiles, where that is the faster plan (e.g. 448
 * row tiles: 224 workgroups x 2 tiles instead of 256 x 1 + 192 tiles step by step); N rounded up to 16 otherwise.  Rows beyond N
 * must be zero-filled inputs (their outputs are not meaningful). */
int fsn_lstm_layer_plan_rows(int N, int H);
int fsn_lstm_layer_fc_supported(int T, int N, int I, long ldx, int H, int O);
size_t fsn_lstm_layer_fc_workspace_bytes(int T, int N, int I, int H);
int fsn_lstm_layer_forward_fc(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                              const float* b_hh, int T, int N, int I, int H, const float* fc_w, const float* fc_b, int O,
                              float* out0, float* out1, long ldo, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of two stacked layers (the counterpart of fsn_lstm2_forward_train): dh1 [T][N][H] = dLoss/dhseq1;
 * outputs as two fsn_lstm_layer_backward calls would give them (dx may be NULL).  The sub-band shape (H = 384, 96+
 * row tiles) runs its back-propagation through time - both layers, all steps, the layer-to-layer dX - as ONE
 * persistent launch (lstm2_group_bptt_kernel) followed by the weight-gradient GEMMs. */
size_t fsn_lstm2_bwd_workspace_bytes(int T, int N, int I, int H, int arith);
int fsn_lstm2_backward(const float* dh1, const float* x, long ldx, const float* w_ih0, const float* w_hh0,
                       const float* w_ih1, const float* w_hh1, int T, int N, int I, int H, const float* hseq0,
                       const float* hseq1, const void* save0, const void* save1, float* dx, long lddx, float* dw_ih0,
                       float* dw_hh0, float* db0, float* dw_ih1, float* dw_hh1, float* db1, void* workspace,
                       size_t workspace_bytes, int arith, void* stream);

/* The same in parts, for callers that have other work waiting on dx (fullsubnet/trainer.py:56-69: the full-band model's
 * backward only needs the sub-band model's input gradient).  `phase` is a sum of 1 = back-propagation through time (the
 * gate gradients stay in the workspace), 4 = dx from them, 2 = the weight- and bias-gradient products from them; 7 =
 * fsn_lstm2_backward; 8 = only what the products need besides the gate gradients (16-bit copies of the hidden sequences:
 * independent of part 1, so it can run on another stream WHILE part 1 runs), 16 (with 2) = a part-8 call has done that.  Parts 2 and 4 take the same arguments and the same workspace as part 1, which nothing else may touch
 * in between; either may be issued on ANOTHER stream, ordered behind part 1 by the caller (an event), so that the products
 * run beside whatever follows dx. */
int fsn_lstm2_backward_phase(const float* dh1, const float* x, long ldx, const float* w_ih0, const float* w_hh0,
                             const float* w_ih1, const float* w_hh1, int T, int N, int I, int H, const float* hseq0,
                             const float* hseq1, const void* save0, const void* save1, float* dx, long lddx, float* dw_ih0,
                             float* dw_hh0, float* db0, float* dw_ih1, float* dw_hh1, float* db1, void* workspace,
                             size_t workspace_bytes, int arith, int phase, void* stream);

/* nn.GRU branch of SequenceModel (sequence_model.py:59-66), one layer, unidirectional, h0 = 0; same
 * conventions as the LSTM layer above with 3H gate rows (r, z, n).  save == NULL: inference.  The two
 * bias gradients differ in the n block (b_hn sits inside r * (W_hn h + b_hn)), hence two outputs.
 * Inference with MANY rows (ABI 117; the sub-band model of a GRU FullSubNet, fullsubnet/model.py:121-128 with
 * sequence_model = "GRU"): H = 384, N >= 18 rows per CU and either I <= 32 (a narrow row-major input) or I = H = ldx (the
 * layer above an equally wide one) run on the LSTM's persistent many-row kernels with the GRU written as a four-gate cell
 * (3/4 of the LSTM's matrix work: the two zero blocks are skipped) - fsn_gru_layer_is_persistent says whether; left-over
 * row tiles advance step by step beside the launch.  Results equal the step form's within fp32 rounding. */
int fsn_gru_layer_is_persistent(int T, int N, int I, long ldx, int H);
size_t fsn_gru_layer_save_bytes(int T, int N, int H);
size_t fsn_gru_layer_fwd_workspace_bytes(int T, int N, int I, int H);
int fsn_gru_layer_forward(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                          const float* b_hh, int T, int N, int I, int H, float* hseq, void* save, size_t save_bytes,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Streaming form: T more steps from the carried state h_state [N][H] (zero-filled for a new stream), updated in
 * place - nn.GRU(x, h_0) is the analogue; workspace >= fsn_gru_layer_fwd_workspace_bytes(T, N, I, H).  Chunked calls
 * give the offline result bit for bit (the same step kernels in the same order) wherever the offline call runs step by
 * step (fsn_gru_layer_is_persistent == 0), within fp32 rounding otherwise. */
int fsn_gru_layer_forward_state(const float* x, long ldx, const float* w_ih, const float* w_hh, const float* b_ih,
                                const float* b_hh, int T, int N, int I, int H, float* hseq, float* h_state,
                                void* workspace, size_t workspace_bytes, void* stream);
size_t fsn_gru_layer_bwd_workspace_bytes(int T, int N, int I, int H);
int fsn_gru_layer_backward(const float* dh, const float* x, long ldx, const float* w_ih, const float* w_hh, int T,
                           int N, int I, int H, const float* hseq, const void* save, float* dx, long lddx,
                           float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, void* workspace,
                           size_t workspace_bytes, void* stream);

/* nn.Linear (sequence_model.py:82-84) of the training step.  x [R][ldx] with ldx = round_up(I,16) and
 * zero padding, w [O][I], b [O] -> y [R][O] (ReLU fused when relu != 0).  Backward: dy [R][lddy]
 * (lddy = round_up(O,16), zero padded) -> dx [R][lddx] (may be NULL), dw [O][I], db [O]. */
size_t fsn_linear_workspace_bytes(int R, int I, int O);
int fsn_linear_forward(const float* x, long ldx, const float* w, const float* b, int R, int I, int O, int relu,
                       float* y, void* workspace, size_t workspace_bytes, void* stream);
int fsn_linear_backward(const float* dy, long lddy, const float* x, long ldx, const float* w, int R, int I, int O,
                        float* dx, long lddx, float* dw, float* db, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ---- training step: loss, gradient clipping and the optimizer ------------------------------- */

/* audio_zen/loss.py:4 (torch.nn.MSELoss(), reduction "mean") as used at fullsubnet/trainer.py:62:
 * loss[0] = mean((input - target)^2); grad_input (may be NULL) = 2 (input - target) / n, i.e. the
 * gradient for an upstream gradient of 1.  Deterministic two-pass reduction (fp64 partials). */
size_t fsn_mse_loss_workspace_bytes(size_t n);
int fsn_mse_loss(const float* input, const float* target, size_t n, float* loss, float* grad_input,
                 void* workspace, size_t workspace_bytes, void* stream);


/* ---- glue of the TRAINING graph (fullsubnet/model.py:72-136 under autograd, fullsubnet/trainer.py:41-71) ----------------
 * Everything between the transforms, the LSTM / Linear entries above and the loss, as kernels (train_glue_kernels.hip):
 * no tensor-algebra kernel of the host framework is left in a training step.  Dimensions: B utterances, F bins, T frames,
 * look_ahead (model.py:13), nb = sb_num_neighbors (model.py:16), groups = num_groups_in_drop_band (model.py:22; band
 * dropping applies when B > 1 and groups > 1, audio_zen/acoustics/feature.py:309-345).  Row order of the sub-band
 * tensors = drop_band's: group i holds samples i, i + g, ... at bins i, i + g, ... (< F - F % g), groups concatenated
 * along the batch axis: row r = b_out Fs + fs.  Time-major tensors: [Tp = T + look_ahead][rows padded][columns padded].
 *
 * fsn_train_rows            Fs (bins per band-dropped sample) and R = B Fs (sub-band rows).
 * norm = FSN_NORM_OFFLINE_LAPLACE (train.toml:82) or FSN_NORM_CUMULATIVE_LAPLACE (train_cumulativeLaplaceNorm.toml:82).
 * fsn_train_fb_input        mag [B][F][T] -> x_tm [Tp][Bp][Fp] = pad(mag) / (mean_b + 1e-5) (model.py:85-95: look-ahead pad,
 *                           base_model.py:204-218 offline Laplace norm over [1, F, Tp]; cumulative: / (running mean over the
 *                           frames so far + EPSILON, base_model.py:221-251)), zeros beyond (B, F): the input of
 *                           fsn_lstm2_forward_train (ldx = Fp); mag_tm [Tp][Bp][Fp] = the padded magnitude itself, for
 *                           fsn_train_sb_input.  Leaves the per-bin sums in `workspace` (same buffer for the whole step).
 * fsn_train_sb_input        freq_unfold(mag, nb) ++ fb_out, / (mean + 1e-5), rows of drop_band (model.py:98-125;
 *                           base_model.py:14-46) -> sb_in [Tp][Rp][32] (columns 2 nb + 2 .. 31 and rows beyond R zero);
 *                           fb_out_tm [Tp][Bp][ld_fb] = the full-band output layer's result (fsn_linear_forward, ReLU);
 *                           den [fsn_train_den_elems] = the divisors (kept for the backward): [B], one per utterance
 *                           (offline; the mean of the unfolded tensor is taken from per-bin sums and window multiplicities) or
 *                           [Tp][Rp], one per unit and frame (cumulative: base_model.py:230-251 sees the units as samples and
 *                           a unit's 2 nb + 2 rows as its frequencies; Rp <= rows rounded up to 64).  The unfolded tensor is
 *                           never formed.
 * fsn_train_sb_input_backward  dx [Tp][Rp][32] (d loss / d sb_in, from fsn_lstm2_backward) -> d_fb [Tp Bp][ld_dfb]: the
 *                           gradient of the full-band output layer's PRE-activation (through the ReLU: fb_out > 0), directly
 *                           (column 2 nb + 1 of the kept rows) and through the mean (cumulative: through the running means of
 *                           this and all later frames of the same unit) - the padded dy fsn_linear_backward takes.  Columns
 *                           2 nb + 2 .. 31 of dx are never read (the dX product does not write them).
 * fsn_train_mask_out        y [Tp][Rp][2] (fsn_linear_forward of the sub-band output layer) -> mask [B][2][Fs][T], the
 *                           look-ahead frames dropped (model.py:129-135).   fsn_train_mask_grad: its adjoint, d_mask ->
 *                           dy [Tp][Rp][ld] (zeros in the look-ahead frames and the padding).
 * fsn_train_cirm_target     compressed cIRM of (noisy, clean) spectra [B][F][T] (mask.py:7-44), band-dropped like the
 *                           prediction, in its layout [B][2][Fs][T] (trainer.py:51-53).
 * fsn_scale_by_scalar       y = x * (*scale), scale a device scalar (the incoming gradient of the loss). */
typedef struct fsn_train_dims {
    int B, F, T, look_ahead, nb, groups;
    int norm; /* FSN_NORM_OFFLINE_LAPLACE / FSN_NORM_CUMULATIVE_LAPLACE (model.py:21 norm_type) */
} fsn_train_dims;
int fsn_train_rows(const fsn_train_dims* dims, int* Fs, int* R);
size_t fsn_train_glue_workspace_bytes(const fsn_train_dims* dims);
size_t fsn_train_den_elems(const fsn_train_dims* dims, int Rp);
int fsn_train_fb_input(const fsn_train_dims* dims, const float* mag, float* x_tm, float* mag_tm, int Bp, int Fp,
                       void* workspace, size_t workspace_bytes, void* stream);
int fsn_train_sb_input(const fsn_train_dims* dims, const float* mag_tm, const float* fb_out_tm, long ld_fb, int Bp, int Fp,
                       float* sb_in, int Rp, float* den, void* workspace, size_t workspace_bytes, void* stream);
int fsn_train_sb_input_backward(const fsn_train_dims* dims, const float* dx, const float* sb_in, int Rp, const float* den,
                                const float* fb_out_tm, long ld_fb, int Bp, float* d_fb, long ld_dfb, void* workspace,
                                size_t workspace_bytes, void* stream);
int fsn_train_mask_out(const fsn_train_dims* dims, const float* y, int Rp, float* mask, void* stream);
int fsn_train_mask_grad(const fsn_train_dims* dims, const float* d_mask, float* dy, int Rp, int ld, void* stream);
/* Time-major rows src [T][N][W] -> dst [n][T][rows][W] (to_pieces = 1; rows beyond N zero) or back (to_pieces = 0: src the
 * pieces, dst the rows; rows beyond N dropped): a batch with more sub-band rows than one persistent training launch holds
 * - the shipped TOMLs train 32 / 48 utterances per process, train.toml:52 - runs as n equal pieces of whole clusters
 * (fullsubnet/model.py:121-128: the rows are independent sequences).  W even, n rows >= N, T <= 65535. */
int fsn_train_rows_pieces(const float* src, float* dst, int T, long N, int W, int rows, int n, int to_pieces, void* stream);
int fsn_train_cirm_target(const fsn_train_dims* dims, const float* noisy_real, const float* noisy_imag,
                          const float* clean_real, const float* clean_imag, float* target, void* stream);
int fsn_scale_by_scalar(const float* x, const float* scale, float* y, size_t n, void* stream);

/* Fast FullSubNet's tensor glue between its LSTM / Linear blocks (fast_fullsubnet/model.py:108-140 real_time_down /
 * up-sampling, :143-202 forward), every tensor TIME-MAJOR [frames][rows][columns] - the layout the LSTM entries take and
 * fsn_linear_forward writes - so that no transposing copy sits between two blocks (fast_glue_kernels.hip):
 * fsn_fast_spec_rows        mag [B][F][T0] (model.py:151 functional.pad(..., [0, look_ahead]) included) -> rows
 *                           [T0 + look_ahead][Bp][Fp] of the mel product (model.py:157), zero beyond (T0, B, F).
 * fsn_fast_norm_rows        offline_laplace_norm (base_model.py:204-218) of x [T][Bp][C]: out = x / (mean over the
 *                           utterance's T x C values + 1e-5), rows beyond B zero (model.py:160, the encoder's input).
 *                           workspace: fsn_fast_glue_workspace_bytes.
 * fsn_fast_bottleneck_input model.py:163-178: unit windows of mel (mel_neighbors) and of the encoder output (enc_neighbors),
 *                           reflected at the band edges (base_model.py:14-46), concatenated, down-sampled in time (frame 0
 *                           kept, then means over blocks of `shrink` frames, a shorter last block over what it has),
 *                           divided by (the utterance's mean of that tensor + 1e-5): out [Ts][Np][Wp], row b num_mels + m,
 *                           zero padding; Ts = fsn_fast_low_rate_frames(T, shrink).  The unfolded tensor is never formed.
 * fsn_fast_decoder_input    model.py:131-140 + :188-190: out [T][Bp][2 num_mels] = encoder output | bottleneck output held
 *                           for `shrink` frames (frame t takes low-rate frame t / shrink; slow[ts ld_slow_frame + row ld_slow_row];
 *                           relu != 0: `slow` is the bottleneck output layer's PRE-activation, fsn_lstm_layer_forward_fc).
 * fsn_fast_mask_out         model.py:200-202: o [T][Bp][ld >= 2F] -> mask [B][2][F][T - look_ahead] (first frames dropped). */
int fsn_fast_low_rate_frames(int T, int shrink);
size_t fsn_fast_glue_workspace_bytes(int T, int B, int num_mels, int shrink);
int fsn_fast_spec_rows(const float* mag, int B, int F, int T0, int look_ahead, float* rows, int Bp, int Fp, void* stream);
int fsn_fast_norm_rows(const float* x, int T, int B, int Bp, int C, float* out, void* workspace, size_t workspace_bytes,
                       void* stream);
int fsn_fast_bottleneck_input(const float* mel, const float* enc, long ld_enc, int T, int B, int Bp, int num_mels,
                              int mel_neighbors, int enc_neighbors, int shrink, float* out, int Np, int Wp, void* workspace,
                              size_t workspace_bytes, void* stream);
int fsn_fast_decoder_input(const float* enc, long ld_enc, const float* slow, long ld_slow_frame, long ld_slow_row, int relu, int T,
                           int B, int Bp, int num_mels, int shrink, float* out, void* stream);
int fsn_fast_mask_out(const float* o, long ld, int T, int B, int Bp, int F, int look_ahead, float* mask, void* stream);

/* fullsubnet/trainer.py:65-69: torch.nn.utils.clip_grad_norm_(parameters, max_norm) followed by
 * torch.optim.Adam.step() (train.py:55-59: lr, betas, eps 1e-8, no weight decay / amsgrad), fused into
 * two multi-tensor launches.  The arrays are HOST arrays of n_tensors device pointers / element
 * counts.  grads are scaled in place by min(1, max_norm / (total_norm + 1e-6)) exactly like
 * clip_grad_norm_ (max_norm <= 0 disables clipping); total_norm_out (device, may be NULL) receives
 * the unclipped 2-norm.  `step` is the 1-based count of this call (bias corrections).
 * A non-finite gradient norm SKIPS the update (parameters, moments and gradients untouched), as GradScaler.step()
 * does in the reference (trainer.py:69), with no host synchronisation: skipped_steps (device, two words, may be
 * NULL; zero it once) counts the skipped updates in word 0 (word 1 is scratch), and a skipped update does not
 * advance Adam's step count - the kernel uses step - skipped for the bias corrections.
 * grad_scale (device scalar, may be NULL = 1): the loss scale the gradients carry (GradScaler, trainer.py:63-69):
 * they are divided by it first (GradScaler.unscale_), so total_norm_out, the clipping and the update see unscaled
 * gradients; an overflowed (inf / NaN) gradient makes the norm non-finite and the update is skipped as above.
 * found_inf (device scalar, may be NULL): GradScaler's inf flag over ALL of the optimizer's parameter groups; non-zero
 * skips this call's update too, so that the groups of one optimizer skip together like scaler.step does. */
#define FSN_ADAM_MAX_TENSORS 32
typedef struct fsn_adam_cfg {
    float lr, beta1, beta2, eps;
    float max_norm;
    int step;
} fsn_adam_cfg;
size_t fsn_clip_adam_workspace_bytes(int n_tensors, const size_t* numel);
int fsn_clip_adam_step(int n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const size_t* numel, const fsn_adam_cfg* cfg,
                       float* total_norm_out, const float* grad_scale, const float* found_inf,
                       unsigned* skipped_steps, void* workspace, size_t workspace_bytes, void* stream);

/* ---- residency contract of the persistent kernels ----------------------------------------------------------
 * Four kernels - the full-band chain (forward, BPTT) and the sub-band group kernel (forward, BPTT) - run a whole
 * recurrence as ONE launch whose workgroups hand results to each other step by step; they make progress only while
 * their whole grid is resident.  What the library guarantees by itself: such a launch is only chosen when the grid
 * fits an idle device (from the compiled kernel's occupancy on THAT device; other shapes take the per-step paths),
 * and the library's own persistent launches from different streams (one process) only run side by side when the
 * whole set is provably placeable in any dispatch order (every launch reports grid and occupancy; a launch waits for
 * earlier ones, oldest first, until sum_j grid_j / (occ_j CUs) stays below the fill level at which some CU state
 * could refuse a workgroup of any kernel of the set - e.g. two 192-workgroup chain launches share the chip, two
 * 448-workgroup group launches take turns).  What it
 * cannot control is bounded instead: a FOREIGN kernel holding CUs (an RCCL collective beside DDP's backward, work
 * of another stream) merely delays the workgroups that found no room until it ends - the resident ones wait for
 * them, by the device's wall clock, up to fsn_set_persistent_timeout_ms (default 20 s).  A wait that runs out means
 * the foreign kernel was itself waiting for this one (two processes sharing a GPU each holding part of it, graph
 * replays of two persistent launches on two streams): the launch then finishes with NaN outputs, never garbage and
 * never a hang, the stream's sticky status is raised, and every later persistent launch on that stream returns
 * FSN_ERR_TIMEOUT until fsn_stream_status_clear.  Callers that cannot rule that situation out switch the
 * persistent kernels off (FSN_PERSISTENT_NEVER: the per-step paths, same results, slower for few rows). */
#define FSN_PERSISTENT_AUTO 0
#define FSN_PERSISTENT_NEVER 1
int fsn_set_persistent_mode(int mode);       /* process-wide */
int fsn_set_persistent_timeout_ms(int ms);   /* process-wide, [1, 3600000] */
/* The sticky status of `stream`: *status_out = status word of the first launch that ran out of time (0 = none),
 * *events_out = number of outputs poisoned since the last clear; either may be NULL.  synchronize != 0 waits for the
 * stream first (the record is written by the device when the launch ends).  Returns FSN_ERR_TIMEOUT when raised. */
int fsn_stream_status(void* stream, int synchronize, unsigned* status_out, unsigned* events_out);
int fsn_stream_status_clear(void* stream);
/* What a RAISED record does to later persistent launches on `stream`.  FSN_TIMEOUT_REFUSE (default): they return
 * FSN_ERR_TIMEOUT until the record is cleared - an inference caller must never consume NaN silently.
 * FSN_TIMEOUT_DEFER: they are launched regardless - for a training step in flight (fullsubnet/trainer.py:41-71:
 * forward, loss, backward are ~10 entry points behind autograd, and under DistributedDataParallel,
 * audio_zen/trainer/base_trainer.py:32, a rank that raised in the middle of backward would leave its peers waiting in
 * the gradient all-reduce): the poisoned outputs are NaN, NaN propagates into every gradient, the fused optimizer
 * skips the update on the device (fsn_clip_adam_step), and the caller reads fsn_stream_status once per step. */
#define FSN_TIMEOUT_REFUSE 0
#define FSN_TIMEOUT_DEFER 1
int fsn_stream_timeout_policy(void* stream, int policy);

/* Per-stage kernel timing of the last fsn_enhance / fsn_fullsubnet_forward call made ON `stream` with
 * profiling enabled for THAT stream (hipEvents on it, kept per (device, stream); forces a sync of them when read).
 * Stage ids are listed by fsn_profile_stage_name(); used by bench.py for the roofline line.  */
int fsn_profile_enable(void* stream, int on);
/* Test hook: a launch of a persistent kernel that ran out of time raises a status word and a follow-up kernel turns
 * its output into NaN instead of leaving garbage.  This entry runs that follow-up kernel on caller data:
 * out[0..n) = NaN iff *status (a device word) != 0, and the stream's sticky status is raised like a real event. */
int fsn_debug_poison_if(const void* status, float* out, size_t n, void* stream);
/* Test hook: a foreign kernel - `workgroups` x 256 threads, lds_bytes of LDS each, ~200 registers per lane when
 * heavy != 0 - that holds its CUs for `ms` milliseconds on `stream`.  sink: one device float (never written). */
int fsn_debug_hog(int workgroups, int lds_bytes, int heavy, float ms, float* sink, void* stream);
/* Test hook: persistent launches admitted so far (process-wide), stream waits inserted between them, and launches
 * that did not report their footprint (must stay 0).  Any pointer may be NULL. */
int fsn_debug_persist_stats(unsigned* launches, unsigned* waits, unsigned* unreported);
/* Test / measurement hook: the two-layer training entries (fsn_lstm2_forward_train / fsn_lstm2_backward) under
 * FSN_ARITH_F16 / _BF16 run the 16-bit arithmetic's own persistent kernels (lstm_group16_kernels.hip) where they
 * apply; on = 0 keeps the fp32-era group kernels under that arithmetic (A/B measurements), on = 2 only the round-3 form
 * of the weight-gradient products (operands converted on the fly), on = 3 layer 0's input-side products (dx, dW_ih0) from
 * fp32 gate gradients as in round 5 (the BPTT launch then stores them), on = 1 restores the default. */
int fsn_debug_g16_kernels(int on);
/* 0: the 16-bit-operand weight-gradient products on 192 x 192 tiles also where the 192 x 384 eight-wave form applies (both
 * give bit-identical partial products; the split count differs).  A/B measurements and tests. */
int fsn_debug_tn16h_wide(int on);
/* Test hooks that need no device.  fsn_debug_persist_set_fits: 1 when the gate would let n persistent launches with the
 * given chip fractions (grid / (occ x CUs)) and occupancies run side by side, 0 when the newest has to wait.
 * fsn_debug_tn_plan: the K splits a weight-gradient product [M x Nc], K rows, would take (*splits) and the bound the
 * scratch of every caller is sized by (*bound >= *splits must hold for every shape). */
int fsn_debug_persist_set_fits(int n, const double* fracs, const int* occs);
int fsn_debug_tn_plan(int M, int Nc, long K, int arith, int* splits, long* bound);
/* Test hook: fsn_enhance / fsn_fullsubnet_forward run a batch of B utterances as one or several calls of the model core
 * (whole rounds of the persistent kernels, a remainder split by a cost model of the plans - the model has no
 * cross-utterance term); sizes[0..n) = their utterance counts, n returned (max_sizes >= 80), -1 on bad arguments. */
int fsn_debug_core_chunks(const fsn_fullsubnet_cfg* cfg, int B, int* sizes, int max_sizes);
/* Diagnostic: how the sub-band model of a B-utterance, T-frame call is spread over the device (of its first chunk when the
 * batch runs as several).  plan[0..8) = sub-band rows, 16-row tiles, row tiles per workgroup of the persistent pair, its
 * workgroups (0: the rows run on the group kernel / step launches), left-over tiles beside it, clusters of the group
 * kernel (0: none), full-band model on the chain kernel (0 / 1), chunks of the batch; with n >= 9 also plan[8] = the rows
 * the persistent pair processes summed over ALL chunks (whole rounds and a remainder have different plans).  bench.py
 * prints it per rank and counts the roofline's FLOPs on the rows the persistent launches actually process. */
int fsn_debug_core_plan(const fsn_fullsubnet_cfg* cfg, int B, int T, int* plan, int n);
int fsn_profile_num_stages(void);
const char* fsn_profile_stage_name(int stage);
int fsn_profile_read(void* stream, float* ms_per_stage, int n);

#ifdef __cplusplus
}
#endif
#endif /* FSN_HIP_H */
