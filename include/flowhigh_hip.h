/* flowhigh_hip.h -- C ABI of libflowhigh_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the FlowHighSR.generate() hot path of resemble-ai/flowhigh.
 * The reference has no FFI layer: its path is stock PyTorch aten ops called from
 * Python.  Each entry point below names the reference call
 * site (file:line under /root/reference/src/flowhigh/) whose arithmetic it replaces.
 * The Python host class `flowhigh_amd.FlowHighSR` binds these through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch-allocated); the
 *     library never allocates, frees, or synchronises; every call only enqueues work
 *     on `stream` (a hipStream_t passed as void*), so calls are graph-capturable;
 *   - all tensors are float32, dense, layouts stated per function;
 *   - return value: 0 = ok, negative = FH_E_* ; fh_last_error() gives the message
 *     (thread-local, valid until the next failing call on the same thread).
 */
#ifndef FLOWHIGH_HIP_H
#define FLOWHIGH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FH_OK 0
#define FH_E_ARG (-1)     /* bad argument / unsupported shape */
#define FH_E_LAUNCH (-2)  /* HIP launch error */

#define FH_ABI_VERSION 5

int fh_abi_version(void);
const char* fh_last_error(void);

/* ------------------------------------------------------------------------------------
 * Grouped implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces: Conv1d / ConvTranspose1d call sites of BigVGAN,
 *   models/bigvgan/models.py:172-194 (conv_pre, ups[i], and the convs1/convs2 of
 *   AMPBlock1 :63-72), including "+ x" (:70) and the "xs / num_kernels" average (:187).
 *
 * One launch runs `n_groups` independent groups (e.g. the three AMP blocks of a stage,
 * or the `u` output phases of a transposed conv).  A group sums `nseg` K-segments, each
 * a (input tensor, packed weight slab, tap list) triple, into one accumulator tile:
 *     out[b, co, n*out_stride + out_phase] =
 *         scale * ( bias[co] + sum_r res[r][b, co, .] +
 *                   sum_seg sum_tap sum_ci  w_seg[tap, ci, co] * x_seg[b, ci, n + tap_off[tap]] )
 * with x read as 0 outside [0, lin).
 *
 * Packed weights: float [cin/ck][ntaps][cout_pad][ck], ck = channel chunk (16, or 8 when some
 * cin % 16 != 0), cout_pad = cout rounded up to the tile height of `tile_cfg` (fh_conv_tile_m),
 * zero padded.  cin % ck == 0 for every segment.
 * --------------------------------------------------------------------------------- */
#define FH_CONV_MAX_TAPS 16
#define FH_CONV_MAX_SEG 3
#define FH_CONV_MAX_HALO 64

typedef struct {
  const float* x;      /* [B, cin, lin] */
  const float* w;      /* packed, see above */
  int32_t cin;
  int32_t ntaps;
  int32_t off_min;     /* min(tap_off) */
  int32_t off_max;     /* max(tap_off); off_max - off_min <= FH_CONV_MAX_HALO */
  int32_t tap_off[FH_CONV_MAX_TAPS];
} fh_conv_seg;

typedef struct {
  fh_conv_seg seg[FH_CONV_MAX_SEG];
  const float* bias;                 /* [cout] or NULL */
  const float* res[FH_CONV_MAX_SEG]; /* each [B, cout, lout] or NULL */
  float* out;                        /* [B, cout, lout] */
  int32_t nseg;
  int32_t nres;
  int32_t cout;
  int32_t cout_pad;
  int32_t lin;
  int32_t lout;
  int32_t n_len;        /* number of n positions (= lout for a plain conv, lin for a transposed-conv phase) */
  int32_t out_stride;
  int32_t out_phase;
  float scale;
} fh_conv_group;

int fh_sizeof_conv_group(void);
/* tile_cfg: 0 = 128x128, 1 = 192x128, 2 = 96x256, 3 = 64x256, 4 = 32x512, 5 = 128x64, 6 = 96x128 (co x n) */
int fh_conv_tile_m(int tile_cfg);
int fh_conv_tile_n(int tile_cfg);
/* groups: device array of n_groups descriptors; all groups share cout_pad and n_len. */
int fh_conv_grouped_f32(const fh_conv_group* groups, int n_groups, int batch, int cout_pad,
                        int n_len, int tile_cfg, int ck, void* stream);

/* ConvTranspose1d(k, stride u, padding (k - u) / 2) with u = `phases` in {2, 3} and k - u even (models/bigvgan/models.py:141-146,179)
 * with ALL output phases of a (co tile, time tile) computed by one block: every group has nseg == phases segments -- segment p =
 * output phase p: the same x, that phase's taps and packed weights (as the per-phase groups of fh_conv_grouped_f32 would have
 * them), an EVEN number of (16-channel chunk, tap) steps -- out_stride == phases, out_phase == 0, n_len input positions, no
 * residuals.  out[b, co, phases * n + p] = scale * (bias[co] + phase p's sum).  Same bits as `phases` groups with strided stores;
 * the block writes phases consecutive floats per position (whole lines).  tile_cfg 3, 4 or 6; channel chunk 16. */
int fh_conv_transpose_fused_f32(const fh_conv_group* groups, int n_groups, int batch, int cout_pad, int n_len,
                                int tile_cfg, int phases, void* stream);

/* ------------------------------------------------------------------------------------
 * The same Conv1d call sites (models/bigvgan/models.py:63-72, "same"-padded, stride 1, square
 * cin x cout residual-stack convs) evaluated with the Winograd minimal-filtering identity
 * F(4,3): the k taps are walked in groups of 3 and every group of 4 outputs costs 6 instead of
 * 12 multiply-adds per (cin, cout) pair.  Exact in exact arithmetic; in fp32 the end-to-end
 * difference to the direct form is ~1e-6 (tests/tools/winograd_numerics.py).
 *   out[b, co, n] = scale * (bias[co] + sum_res res[b, co, n]
 *                            + sum_seg sum_ci sum_{j<k} w[co, ci, j] * x[b, ci, n + (j - center) * dilation])
 * u = host-transformed weights (flowhigh_amd/vocoder.py: pack_wino_weight):
 *   [cin/16][ngrp][6][cout_pad][16], u[., g, xi, co, .] = sum_j G[xi][j] * w[co, ., 3g + j]
 *   (taps past k are zero), cout_pad % (64 or 96, see tile_cfg) == 0, cin % 16 == 0.
 * cin need not equal cout.  With out_stride = u > 1 a group is one output phase of ConvTranspose1d(k, stride u,
 * padding (k - u) / 2) (models/bigvgan/models.py:131-140,179): its taps j = r + p (mod u), ordered by input offset,
 * form a stride-1 correlation with `center` = minus the smallest offset (flowhigh_amd/vocoder.py:
 * transposed_conv_phases), and its outputs interleave with the other phases'.
 */
typedef struct {
  const float* x;      /* [B, cin, len] */
  const float* u;
  int32_t cin;
  int32_t ngrp;        /* ceil(k / 3) */
  int32_t center;      /* (k - 1) / 2 */
  int32_t xlen;        /* 0: x rows are `len` samples long (the group's len).  > 0: x is [B, cin, xlen] and reads past
                          xlen are zero -- a transposed-conv phase whose rows are shorter than its output count, below
                          (plain layout, dilation 1; rows not 16-byte aligned need FH_WINO_NOVL) */
} fh_wino_seg;

typedef struct {
  fh_wino_seg seg[FH_CONV_MAX_SEG];
  const float* bias;                 /* [cout] or NULL */
  const float* res[FH_CONV_MAX_SEG]; /* each [B, cout, len] or NULL */
  float* out;                        /* [B, cout, len] */
  int32_t nseg;
  int32_t nres;
  int32_t cout;
  int32_t cout_pad;
  int32_t len;
  float scale;
  int32_t out_stride;   /* 0 or 1: out[b, co, n]; u > 1: one output phase of a transposed conv (models.py:179),   */
  int32_t out_phase;    /*   out [B, cout, u * len] written at u * n + out_phase (dilation 1, plain layout, no res)  */
  int32_t out_len;      /* 0, or the row length of out when it is not out_stride * len: ConvTranspose1d(k, u, padding
                           (k - u) / 2) with k - u ODD returns u * L + 1 samples (models.py:141-146): its phase groups have
                           len = L + 1 output positions, seg.xlen = L, out_len = u * L + 1; positions u * n + out_phase >=
                           out_len are not written */
  int32_t pad_;
} fh_wino_group;

int fh_sizeof_wino_group(void);
/* tile_cfg: 0 = 64 co x 512 outputs per block, 1 = 96 co x 256 outputs (cout_pad % fh_wino_tile_m == 0);
 * 4 = 64 co x 256 outputs (short rows: less padding of the last block of a dilation phase);
 * 5 = 32 co x 256 outputs (short clips: more, shorter blocks); 6 = 128 co x 256 outputs.  Every shape gives the
 * same bits (same accumulation order); which one is fastest depends on the block count (vocoder.choose_wino_cfg). */
int fh_wino_tile_m(int tile_cfg);
/* Phase-major layout of a [B, C, len] tensor for dilation d: every (batch, channel) row holds its d decimated
 * phases one after the other, x[b, c, p + d u] at row + p * fh_phase_len(len, d) + u, row pitch
 * d * fh_phase_len(len, d) floats.  A dilated conv then reads and writes contiguous runs. */
int fh_phase_len(int len, int dilation);
/* groups: device array; all groups share cout_pad, len and the dilation.  phase_major != 0: x, res and out of
 * every group are phase-major for this dilation (see above), else plain [B, C, len]. */
/* tile_cfg | FH_WINO_BF16X6: the groups' `u` pointers name THREE-PIECE bf16 weights
 * ([C_in/16][tap group][6][cout_pad][3 pieces][16 channels] bf16 = 96 bytes per row and chunk; every fp32 value
 * x = h + m + l, h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)) and the contraction runs on the BF16 matrix cores as
 * six v_mfma_f32_32x32x16_bf16 per 16-channel k-block (pairs hh, hm, mh, mm, hl, lh; fp32 accumulation): fp32-grade
 * products (dropped terms <= 2^-24 |a b|) at 0.375 of the matrix-pipe cycles.  Inputs, outputs and accumulation stay
 * float32; results differ from the fp32-MFMA form by rounding only. */
#define FH_WINO_BF16X6 16
/* tile_cfg | FH_WINO_XCD_RANGES: block -> work mapping by TIME instead of by weight panel: XCD x (block id mod 8) works on
 * the x-th eighth of the time axis of every (group, batch, co tile) panel, the co tiles of the same time tiles next to
 * each other in dispatch order.  The co tiles of a group then read their common input through one L2 while it is
 * resident; every XCD fetches every weight panel.  For launches whose transformed weights are
 * small beside their activations (C <= 384 at batch 1): less HBM traffic, same bits.  Ignored by the ragged entry. */
#define FH_WINO_XCD_RANGES 32
/* tile_cfg | FH_WINO_NOVL: some tensor row of the launch is not 16-byte aligned although `len` % 4 == 0 (a segment with
 * xlen % 4 != 0): the slab loader must not use 16-byte loads.  Same bits. */
#define FH_WINO_NOVL 64
int fh_conv_wino_f32(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len,
                     int dilation, int phase_major, int tile_cfg, void* stream);
/* Ragged form (clips of different lengths, one group per clip and AMP block, batch 1): the grid is laid out for
 * max_len; a "run" is fh_wino_run_len(n_tiles) consecutive output tiles of one (group, co tile) panel
 * (n_tiles = ceil(ceil(max_len / dilation) / fh_wino_tile_n(tile_cfg)) * dilation), run id = panel * runs_per_panel +
 * run-in-panel with panel = group * co_tiles + co tile.  run_map (device int32 [n_runs]) lists the runs that hold
 * real tiles, in launch order (heavy groups first): they are dealt round-robin to the 8 XCDs.
 * layout_flags (0..3): bit 0 = the groups' tensors are phase-major (as phase_major != 0 above); bit 1 = some group's
 * rows are not 16-byte aligned (a len that is not a multiple of 4 in a plain-layout launch): no vector loads. */
int fh_wino_tile_n(int tile_cfg);
int fh_wino_run_len(int n_tiles);
int fh_conv_wino_ragged_f32(const fh_wino_group* groups, int n_groups, int cout_pad, int max_len, int dilation,
                            int layout_flags, int tile_cfg, const int* run_map, int n_runs, void* stream);

/* The same Conv1d call sites in the minimal-filtering form F(5,4) (8 points 0, +-1, +-2, +-1/2, inf; taps walked in
 * groups of 4: 1.6 ceil(k/4) multiply-adds per output and channel pair, 20 % fewer than F(4,3) over k = 3 / 7 / 11), for
 * the wide stages.  Same descriptors with ngrp = ceil(k / 4) and
 *   u = [cin/16][ngrp][8][cout_pad][16], u[., g, xi, co, .] = sum_j G8[xi][j] * w[co, ., 4g + j]
 * (flowhigh_amd/vocoder.py: pack_wino54_weight); ngrp <= 3 (k <= 12), out_stride <= 1, xlen = 0, out_len = 0 (longer
 * kernels run on fh_conv_grouped_f32, transposed-conv phases on fh_conv_wino_f32; the descriptors live in device memory, so
 * the library cannot check this on the host: flowhigh_amd/planner.py does when it builds a descriptor, and the kernel traps
 * on a segment it cannot walk).  Any dilation and any
 * len < 2^24 - 4096 (16.7 M samples = 5.8 min at 48 kHz per row; FH_E_ARG beyond: the kernel finds a sample's phase in fp32): rows
 * that are not 16-byte aligned are read and written with 4-byte accesses, same arithmetic.  tile_cfg: 0 = 128 co x 320 outputs per block,
 * 1 = 96 x 320, 2 = 64 x 320, 3 = 48 x 320 (three 16-row tiles of v_mfma_f32_16x16x4_f32: it sums a chunk's channels in another
 * order than the others, so a caller keeps ONE of {0, 1, 2} / 3 per weight tensor) (cout_pad % fh_wino54_tile_m(tile_cfg) == 0).  Results differ from the F(4,3) form by
 * rounding: 0.8-1.1e-5 per conv against float64 on unit-scale data (F(4,3) 0.6-1.1e-5, direct 4e-7), kernel-fuzz worst cases about a
 * third above F(4,3)'s (tests/tools/winograd_numerics.py, wino_fuzz.py; headroom against the weights' gain: profiles/r05_regime_sweep.txt). */
int fh_wino54_tile_m(int tile_cfg);
int fh_wino54_tile_n(void);
int fh_conv_wino54_f32(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len,
                       int dilation, int phase_major, int tile_cfg, void* stream);
/* Ragged form, as fh_conv_wino_ragged_f32: runs of fh_wino54_run_len(n_tiles) consecutive 320-output tiles,
 * n_tiles = fh_wino54_n_tiles(max_len, dilation, phase-major?): in the plain layout ceil(ceil(len / dilation) / 320) tiles per
 * phase, tile index = block-in-phase * dilation + phase; in the phase-major layout the d phases are tiled as ONE sequence
 * (every phase owns ceil(n / 5) + >= 3 five-output tile slots, rounded up to a multiple of 4; a tile is 64 consecutive slots), a
 * group's real tiles are 0 .. fh_wino54_n_tiles(its len, ...) - 1.  layout_flags bit 0 = phase-major, bit 1 = some group's
 * plain rows are not 16-byte aligned. */
int fh_wino54_n_tiles(int len, int dilation, int phase_major);   /* 320-output blocks per (group, batch, co tile) panel */
int fh_wino54_run_len(int n_tiles);
int fh_conv_wino54_ragged_f32(const fh_wino_group* groups, int n_groups, int cout_pad, int max_len, int dilation,
                              int layout_flags, int tile_cfg, const int* run_map, int n_runs, void* stream);

/* out = ((a + b) + c) * scale over n floats (c may be NULL; n % 4 == 0, 16-byte aligned pointers): the
 * `xs += resblock(x); x = xs / num_kernels` of BigVGAN.forward (models/bigvgan/models.py:183-188) for the
 * stages whose closing conv is not fused (too few blocks to fill the chip at batch 1). */
int fh_mean_f32(const float* a, const float* b, const float* c, float* out, long long n, float scale,
                void* stream);
/* out = (((srcs[0] + srcs[1]) + ...) + srcs[n_srcs-1]) * scale, n floats each (n % 4 == 0, 16-byte aligned); srcs is a
 * HOST array of 1..12 device pointers.  Adds the partial outputs of a conv whose input channels were cut into
 * slices (one fh_wino_group per slice; short clips, vocoder.wino_split_k) in a fixed order; out may be srcs[0]. */
int fh_sum_f32(const float* const* srcs, int n_srcs, float* out, long long n, float scale, void* stream);

/* out = scale * (((p0 + p1) + p2) + ...) for several independent jobs in one launch (the averaging / partial-sum
 * passes of a ragged batch: one job per clip; same order of additions as fh_sum_f32 / fh_mean_f32). */
typedef struct {
  const float* src[12];
  float* out;
  int64_t n;             /* floats, % 4 == 0 */
  int32_t n_src;
  float scale;
} fh_sum_job;
int fh_sizeof_sum_job(void);
int fh_sum_multi_f32(const fh_sum_job* jobs, int n_jobs, long long max_n, void* stream);

/* conv_post + tanh (models/bigvgan/models.py:190-192): x [B, cin, L], w [cin, ksz], bias[1]
 * -> out [B, L] = tanh(bias + sum_ci sum_j w[ci,j] * x[b, ci, t + j - ksz/2]).  ksz odd <= 15. */
int fh_conv_post_tanh_f32(const float* x, const float* w, const float* bias, float* out,
                          int batch, int cin, int len, int ksz, void* stream);

/* ------------------------------------------------------------------------------------
 * Anti-aliased periodic activation, fused: 2x kaiser-sinc upsample -> Snake/SnakeBeta ->
 * 2x low-pass downsample, one read and one write of the tensor.
 * Replaces: Activation1d.forward, models/bigvgan/alias_free_torch/act.py:23-28
 *   (UpSample1d resample.py:25-33, SnakeBeta activations.py:107-120 / Snake :48-59,
 *    DownSample1d -> LowPassFilter1d filter.py:86-95).
 * Grouped like the conv: group g maps x[g] -> y[g] with its own per-channel parameters.
 *   alpha, inv_beta: [channels] already exponentiated if log-scale;
 *   inv_beta = 1 / (beta + 1e-9)  (or 1 / (alpha + 1e-9) for Snake).
 *   up_taps / down_taps: the 12 filter taps stored in the checkpoint.
 * --------------------------------------------------------------------------------- */
typedef struct {
  const float* x;        /* [B, C, L] */
  float* y;              /* [B, C, L] */
  const float* alpha;    /* [C] */
  const float* inv_beta; /* [C] */
  float up_taps[12];
  float down_taps[12];
  int32_t len;           /* fh_act1d_ragged_f32 only: L of THIS group (batch 1) */
  int32_t tile_base;     /* fh_act1d_ragged_f32 only: C * sum over the groups before of ceil(len / fh_act_tile_len()) */
} fh_act_group;

int fh_sizeof_act_group(void);
int fh_act1d_grouped_f32(const fh_act_group* groups, int n_groups, int batch, int channels,
                         int len, void* stream);
/* Same with phase-major tensors (fh_phase_len below): din / dout = dilation (1 .. 16) whose phase-major layout x / y
 * use, 1 = plain [B, C, len].  Used on both sides of a dilated Winograd conv. */
int fh_act1d_grouped_pm_f32(const fh_act_group* groups, int n_groups, int batch, int channels, int len,
                            int din, int dout, void* stream);
/* Ragged batches (clips of different lengths in ONE launch): group g = one clip's [C, groups[g].len] tensor,
 * total_tiles = groups[n-1].tile_base + C * ceil(groups[n-1].len / fh_act_tile_len()).  Same arithmetic per
 * sample as the launches above (replicate padding at each clip's own ends).  all_len_mult4: every group's len is a
 * multiple of 4 (rows 16-byte aligned: vector accesses). */
int fh_act_tile_len(void);
int fh_act1d_ragged_f32(const fh_act_group* groups, int n_groups, int channels, int din, int dout,
                        long long total_tiles, int all_len_mult4, void* stream);
/* Occupancy cap of the activation launches on the CURRENT device: at most `blocks` (2 .. 5) 4-wave blocks per CU, 0 = no cap
 * (7, the default).  A tuning knob, not a semantic one (results are the same bits): on some MI355X boxes the uncapped launch
 * draws enough power that the shader clock drops for the duration of the NEXT launch (a Winograd conv then takes 14 % longer);
 * 3 blocks per CU avoid most of that (the step is 5-7 % faster there) and cost the activation 25 % on boxes without the effect.  No
 * reference counterpart; the host calibrates it once per device (flowhigh_amd/vocoder.py: calibrate_act_occupancy,
 * FH_ACT_BLOCKS). */
int fh_act_set_blocks_per_cu(int blocks);
int fh_act_get_blocks_per_cu(void);

/* ------------------------------------------------------------------------------------
 * Narrow stages (C <= 48 channels): the Conv1d of an AMP block behind its Activation1d, shaped for few channels.
 * Replaces the `xt = conv(xt)` half of one `xt = self.activations[i](x); xt = conv(xt)` pair of an AMP block,
 *   models/bigvgan/models.py:63-72 (AMPBlock1: convs1[dilated] and convs2 "+ x") and :108-117 (AMPBlock2), incl. the
 *   "xs / num_kernels" average of the stage's blocks (:181-187) as K segments of one group:
 *     out[b, co, t] = scale * ( bias[co] + sum_r res[r][b, co, t]
 *                               + sum_seg sum_ci sum_{j<k} w_seg[co, ci, j] * x_seg[b, ci, t + (j - center) * dilation] )
 *   with x_seg read as 0 outside [0, len) (the conv's zero padding).  All tensors are plain [B, C, len] whatever the
 *   dilation.  The conv is evaluated in the Winograd F(5,4) form (fh_conv_wino54_f32): k <= 12 odd,
 *   dilation <= 6, (max_center - center) + 4 ngrp + 3 <= 16 for every segment.
 *   (ABI 3 also had the form with the Activation1d inside the launch -- flags bit 1 clear, activation parameters in the segment:
 *   measured slower than the pair of launches, it left the library with ABI 4.)
 * u = host-transformed weights (flowhigh_amd/packing.py: pack_amp_weight), float
 *   [C/8 chunks][ngrp][ blockA: [8 points][64 lanes][4]  (ceil(C/16) >= 2)  |  blockB: [8 points][64 lanes][2]  (ceil(C/16) odd) ]
 *   lane l = 16 kq + r: blockA[xi][l][2 m + s] = U[g][xi][co = 16 m + r][ci = 8 chunk + 2 kq + s] for the row tiles m = 0, 1,
 *   blockB[xi][l][s] the same for the last row tile of an odd count; U[g][xi] = sum_j G8[xi][j] w[:, :, 4 g + j] (float64 on
 *   the host, taps past k and rows past C zero): 1024 ceil(C/16) floats per (chunk, tap group) stage.
 * Groups may have different `len` (ragged batches).  The work list is a device array of `total_tiles` fh_amp_tile entries, one
 * per (group, batch item, tile of fh_amp_tile_len(dilation) outputs starting at t0), heavy groups first; a group's batch items
 * are [batch, C, len] tensors behind its pointers.
 * max_center: the largest `center` of any segment of the launch (<= 5): it places the samples in the kernel's LDS slabs, the same
 *        for every block.  The launch runs 2 x (CUs of the device) persistent blocks that walk the tile list with stride gridDim.
 * flags: bit 0 = every row of every group is 16-byte aligned (len % 4 == 0: vector accesses; same bits without);
 *        bit 1 = must be set (the conv alone; FH_E_ARG otherwise).
 * --------------------------------------------------------------------------------- */
typedef struct {
  const float* x;        /* [B, C, len]: the conv's input */
  const float* u;        /* transformed weights, layout above */
  int32_t ngrp;          /* ceil(k / 4) */
  int32_t center;        /* (k - 1) / 2 */
} fh_amp_seg;

typedef struct {
  fh_amp_seg seg[FH_CONV_MAX_SEG];
  const float* bias;                 /* [C] or NULL */
  const float* res[FH_CONV_MAX_SEG]; /* each [B, C, len] or NULL */
  float* out;                        /* [B, C, len] */
  int32_t nseg;
  int32_t nres;
  int32_t len;
  int32_t pad0_;
  float scale;
  int32_t pad1_;
} fh_amp_group;

typedef struct {
  int32_t group;       /* index into the launch's groups */
  int32_t batch_item;
  int32_t t0;          /* first output of the tile: a multiple of fh_amp_tile_len(dilation) */
  int32_t len;         /* = groups[group].len */
} fh_amp_tile;

int fh_sizeof_amp_group(void);
int fh_sizeof_amp_tile(void);
int fh_amp_tile_len(int dilation);     /* outputs per block and row: 320, 320, 300, 320, 300, 240 for dilation 1 .. 6; -1 beyond */
int fh_amp_max_channels(void);         /* 48 */
int fh_amp_actconv_f32(const fh_amp_group* groups, int n_groups, const fh_amp_tile* tiles, int total_tiles, int channels,
                       int dilation, int max_center, int flags, void* stream);

/* ------------------------------------------------------------------------------------
 * The same launches in the bf16 x 6 form (conv_form = 'bf16x6'; narrow_bf.hip): the narrow stages' residual-stack convs as a
 * DIRECT implicit GEMM on v_mfma_f32_16x16x32_bf16 -- every input sample is split once into three bf16 pieces (exact), a
 * product is six MFMAs with fp32 accumulation (dropped terms <= 2^-24 |a b|), no Winograd transforms.  Replaces the same
 * reference lines as fh_amp_actconv_f32 (bigvgan/models.py:63-72, :108-117, :181-187).  Same descriptors, read differently:
 *   fh_amp_seg.ngrp = k (taps, 1 .. 11), fh_amp_seg.center = zero-padding on the left in taps (<= 5; "same": (k - 1) / 2);
 *   fh_amp_seg.u = packing.pack_narrow_bf_weight: the channels' octets (8 channels) in slabs of at most three (24 channels: one
 *     slab; 48: two of three octets; balanced otherwise); per slab ceil(k og / 4) k-blocks of 32 = four (tap, octet) pairs,
 *     pair index q = tap og + octet; per k-block [N tile = 16 output channels][piece h, m, l][lane][8 bf16]: lane l holds
 *     w[co = 16 n + (l & 15)][ci = 8 (slab's first octet + octet of pair 4 kb + (l >> 4))  + 0 .. 7][tap of that pair], zero past
 *     the last pair / channel: ceil(C / 16) x 3 KB per k-block;
 *   tiles: t0 a multiple of fh_narrow_tile_len() = 256 outputs per block and row, any dilation 1 .. 6.
 * flags: bit 0 = every row of every group is 16-byte aligned (len % 4 == 0: vector accesses; same bits without).
 * A sample's bits depend on its position in the row only (fixed K order): alone, batched, ragged and chunked runs agree.
 * --------------------------------------------------------------------------------- */
int fh_narrow_tile_len(void);          /* 256 */
int fh_narrow_conv_bf16x6_f32(const fh_amp_group* groups, int n_groups, const fh_amp_tile* tiles, int total_tiles, int channels,
                              int dilation, int flags, void* stream);

/* ------------------------------------------------------------------------------------
 * fp32 MFMA GEMM:  C[M, N] = epilogue( A[M, K] * W[N, K]^T )
 * Replaces every nn.Linear on the path (models/flow.py:239,261; attend.py:170-171;
 * transformer.py:98-104), torch.stft / torch.istft as DFT-by-GEMM
 * (melvoco.py:78-79; postprocessing.py:22-23,39) and the mel projection (melvoco.py:83).
 *   A row-major, row stride lda (floats, % 4 == 0); W row-major [n_pad, K] with
 *   n_pad = N rounded up to 128 (rows >= N zero); K % 32 == 0.
 * epilogue (v = acc + bias[n]):
 *   FH_EPI_LINEAR : C[m, n] = alpha * v + (R ? R[m, n] : 0)              (ldc, ldr strides)
 *   FH_EPI_GEGLU  : packed pairs: columns come in blocks of 64 = 32 "value" + 32 "gate";
 *                   C[m, 32*blk + i] = gelu_erf(gate) * value            (transformer.py:92-95)
 *   FH_EPI_MAG    : pairs (re, im): C = sqrt(re^2 + im^2 + 1e-9)         (melvoco.py:80-81)
 *   FH_EPI_LOGCLAMP: C = log(max(v, 1e-5))                               (modules.py:31-36)
 * --------------------------------------------------------------------------------- */
#define FH_EPI_LINEAR 0
#define FH_EPI_GEGLU 1
#define FH_EPI_MAG 2
#define FH_EPI_LOGCLAMP 3

int fh_gemm_f32(const float* A, int lda, const float* W, const float* bias, const float* R,
                int ldr, float* C, int ldc, int M, int N, int K, float alpha, int epilogue,
                void* stream);

/* The same GEMM in the bf16 x 6 form (gemm_bf.hip; the transformer's linears of a conv_form = 'bf16x6' model): fp32 in / out /
 * accumulate, every product as six v_mfma_f32_32x32x16_bf16 over exact three-piece bf16 splits (dropped terms <= 2^-24 |a b|).
 * A is split by the kernel on its way into LDS; Wp = packing.pack_gemm_bf_weight(W [n_pad, K]): bf16 bit patterns
 * [n_pad / 64][K / 32][piece h, m, l][k-octet 0..3][64 rows][8 bf16] (6 bytes per weight).  Same arguments, epilogues and
 * restrictions as fh_gemm_f32 (flow.py:239,261; attend.py:170-171; transformer.py:98-104). */
int fh_gemm_bf16x6_f32(const float* A, int lda, const float* Wp, const float* bias, const float* R,
                       int ldr, float* C, int ldc, int M, int N, int K, float alpha, int epilogue,
                       void* stream);

/* y[n] = act(bias[n] + sum_k W[n, k] * x[k]),  act: 0 none, 1 SiLU.  One vector (the time
 * embedding is the same for every clip: flow.py:208-211,242; transformer.py:81-83). */
int fh_gemv_f32(const float* W, const float* x, const float* bias, float* y, int N, int K,
                int act, void* stream);

/* LearnedSinusoidalPosEmb (models/pos_emb.py:22-26): out[0:h] = sin(t*w*2pi), out[h:2h] = cos. */
int fh_time_fourier_f32(const float* w, float t, float* out, int half_dim, void* stream);

/* ------------------------------------------------------------------------------------
 * Transformer pointwise / reduction kernels.
 * --------------------------------------------------------------------------------- */
/* ConvPositionEmbed + residual (models/transformer.py:16-46, flow.py:240):
 * y[b,n,c] = x[b,n,c] + gelu_erf(bias[c] + sum_j w[c,j] * x[b, n + j - ksz/2, c]), zero padded.
 * w is passed TRANSPOSED, [ksz, dim] (tap-major), dim % 128 == 0. */
int fh_dwconv_gelu_res_f32(const float* x, const float* w, const float* bias, float* y,
                           int batch, int n, int dim, int ksz, void* stream);

/* AdaptiveRMSNorm / RMSNorm (transformer.py:49-88):
 * y = x / max(||x||, 1e-12) * sqrt(dim) * gamma[c] + (beta ? beta[c] : 0); rows of `dim`. */
int fh_rmsnorm_f32(const float* x, const float* gamma, const float* beta, float* y, int rows,
                   int dim, void* stream);

/* MultiheadRMSNorm on q,k + rotary embedding, in place on the fused qkv buffer
 * (attend.py:144-151,179-184; pos_emb.py:53-59).  qkv [rows = B*n, 3*heads*64];
 * gq, gk [heads, 64]; cos_t, sin_t [n, 32] (angle table, one half).  dim_head must be 64. */
int fh_qknorm_rope_f32(float* qkv, const float* gq, const float* gk, const float* cos_t,
                       const float* sin_t, int batch, int n, int heads, void* stream);

/* softmax(q k^T * scale) v without materialising the score matrix (attend.py:102-139).
 * qkv as above (q | k | v along the feature axis), out [B*n, heads*64].  dim_head 64. */
int fh_attention_f32(const float* qkv, float* out, int batch, int n, int heads, float scale,
                     void* stream);

/* Ragged batches: clips of DIFFERENT lengths packed back to back in the token-major tensors, no padding rows.
 * seg: device int32 [n_seg][2] = (first row, rows) of every clip; max_n = the longest clip.  These are the
 * reference's mask paths -- the key mask of Attend (attend.py:127-128), the mask of ConvPositionEmbed
 * (transformer.py:35-44) -- for the layout in which masked positions simply do not exist; rotary positions
 * restart at 0 in every clip (pos_emb.py:47-51).  Every clip gets the bits it gets alone.
 * cos_t / sin_t must cover max_n positions. */
int fh_attention_seg_f32(const float* qkv, float* out, const int* seg, int n_seg, int max_n, int heads,
                         float scale, void* stream);
int fh_dwconv_gelu_res_seg_f32(const float* x, const float* w, const float* bias, float* y, const int* seg,
                               int n_seg, int max_n, int dim, int ksz, void* stream);
int fh_qknorm_rope_seg_f32(float* qkv, const float* gq, const float* gk, const float* cos_t,
                           const float* sin_t, const int* seg, int n_seg, int max_n, int heads, void* stream);

/* ------------------------------------------------------------------------------------
 * STFT framing, post-processing (postprocessing.py:5-41) and peak normalisation.
 * Packed spectrum layout ("P-layout") used between the DFT GEMMs: per frame 33 blocks of
 * 64 floats = 32 real parts then 32 imaginary parts of bins 32*blk .. 32*blk+31 (bins >= 1025
 * are zero): width 2112.
 * --------------------------------------------------------------------------------- */
/* frames[b*nf + t, k] = pad(audio[b])[hop*t + k] * window[k], k < nfft.
 * pad_mode 0: reflect (melvoco.py:74), 1: zero (torch.stft center=True, pad_mode='constant'). */
int fh_frame_f32(const float* audio, const float* window, float* frames, int batch, int len,
                 int n_frames, int nfft, int hop, int pad, int pad_mode, void* stream);

/* energy[b, f] = sum_t |S[b, t, f]| over the first n_frames frames, f < 1025 (P-layout in). */
int fh_spec_energy_f32(const float* spec, float* energy, int batch, int n_frames, void* stream);

/* get_cutoff_index (postprocessing.py:10-16) / find_cutoff (cfm_superresolution.py:135-140):
 * cr[b] = max{ j in [1, nbins-1] : cumsum(energy[b])[j] < thr * total } or 0; energy [B, nbins]. */
int fh_cutoff_index_f32(const float* energy, int32_t* cr, int batch, int nbins, float thr,
                        void* stream);

/* 2048-point real FFT of framed audio and its inverse (the DFT halves of torch.stft / torch.istft,
 * models/melvoco.py:56-86, postprocessing.py:5-9,39): frames [rows, 2048] ->
 *   mode 0: packed complex spectrum [rows, 2112] (33 blocks of 32 Re + 32 Im, bins >= 1025 zero),
 *   mode 1: magnitudes sqrt(re^2 + im^2 + 1e-9) [rows, 1056] (melvoco.py:81, the input of the mel projection);
 * fh_irfft2048_f32: packed spectrum -> frames (C2R, imaginary parts of DC / Nyquist ignored, 1/N).
 * The butterflies run in float64 (see csrc/fft.hip).
 * twiddles: [1024, 2] DOUBLE = (cos, -sin)(2 pi k / 2048), flowhigh_amd/tables.py: fft_twiddles. */
int fh_rfft2048_f32(const float* frames, const double* twiddles, float* out, int rows, int mode, void* stream);
int fh_irfft2048_f32(const float* spec, const double* twiddles, float* frames, int rows, void* stream);

/* Sampler options of cfm_superresolution.py: mel-domain cutoff energy (:134-159, input to
 * fh_cutoff_index_f32 with thr 0.9995), low-band replacement mel_replace_ops (:146-152), and the
 * independent_cfm_* prior cond * std_1 + eps * std_2 (:226-236).  mel tensors are [B, n, d]. */
int fh_mel_energy_f32(const float* mel, float* energy, int batch, int n, int d, void* stream);
int fh_mel_splice_f32(const float* low, const float* high, const int32_t* cut, float* out, int batch,
                      int n, int d, void* stream);
/* Same for clips of different lengths packed back to back (the reference's masked batches, cfm:162-175,278-279):
 * seg = device int32 [n_seg][2] = (first row, rows) of every clip, mel tensors [sum rows, d], energy [n_seg, d],
 * cut [n_seg]; every clip gets the bits of a call on that clip alone. */
int fh_mel_energy_seg_f32(const float* mel, float* energy, const int32_t* seg, int n_seg, int d, void* stream);
int fh_mel_splice_seg_f32(const float* low, const float* high, const int32_t* cut, float* out,
                          const int32_t* seg, int n_seg, int max_n, int d, void* stream);
int fh_axpby_f32(const float* x, float a, const float* y, float b, float* out, long long n,
                 void* stream);

/* out[b,t,:] = bins < cr[b] ? src : pred   (P-layout, postprocessing.py:36-37). */
int fh_spec_splice_f32(const float* pred, const float* src, const int32_t* cr, float* out,
                       int batch, int n_frames, void* stream);

/* Overlap-add of windowed inverse-DFT frames with window-envelope division, centre trim,
 * zero fill (torch.istft semantics, postprocessing.py:39); also atomically tracks
 * max|y| per clip in peak_bits[b] (float bits as uint32, must be zeroed by the caller). */
int fh_istft_ola_f32(const float* frames, const float* window, float* y, uint32_t* peak_bits,
                     int batch, int n_frames, int len, int nfft, int hop, void* stream);

/* y[b, :] *= target / peak[b]   (postprocessing.py:40; flowhighsr.py:69). */
int fh_peak_scale_f32(float* y, const uint32_t* peak_bits, int batch, int len, float target,
                      void* stream);

/* peak_bits[b] = bits(max_t |x[b,t]|) (caller zeroes peak_bits first). */
int fh_peak_abs_f32(const float* x, uint32_t* peak_bits, int batch, int len, void* stream);

/* ------------------------------------------------------------------------------------
 * Host pre-step on device: scipy.signal.resample_poly(x, up, down) with zero padding
 * (flowhighsr.py:68).  taps: the scipy FIR (already multiplied by `up`), n_taps odd-centred as
 * scipy pads it; y[b, i] = sum_j taps[j*up + (i*down + pre) % up ...] -- see resample_poly_kernel in
 * flowhigh_amd/csrc/frontend.hip.
 * --------------------------------------------------------------------------------- */
int fh_resample_poly_f32(const float* x, const float* taps, float* y, int batch, int len_in,
                         int len_out, int up, int down, int n_taps, int n_pre_remove,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWHIGH_HIP_H */
