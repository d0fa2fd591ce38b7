// Winograd F(5,4) form of the wide residual-stack Conv1d sites of BigVGAN on the fp32 matrix cores.
//
// Replaces the AMPBlock convs (/root/reference/src/flowhigh/models/bigvgan/models.py:36-72: kernel 3 / 7 / 11,
// dilation 1 / 3 / 5, "same" padding) of the stages with >= 96 channels (vocoder.WINO54_MIN_C; an odd multiple of 48 channels
// below that runs the 48-row block of this kernel unless FH_WINO54_H16=0; 24 / 48 channels run amp_fused.hip since round 5).
//
// Minimal filtering F(5,4): 5 outputs of a 4-tap correlation from 8 inputs with 8 multiplies (points 0, +-1, +-2,
// +-1/2, inf), y = A^T [ (G g) .* (B^T d) ].  The k taps are walked in ceil(k/4) groups of 4, so a conv executes
// 1.6 ceil(k/4) multiply-adds per output and channel pair: 1.6 / 3.2 / 4.8 for k = 3 / 7 / 11 against 1.5 / 4.5 / 6.0 of
// the F(4,3) kernel (conv_wino.hip): 20 % fewer matrix instructions over the stack.  fp32 error against float64: 0.8-1.1e-5 per
// conv on unit-scale data (F(4,3): 0.6-1.1e-5, direct: 4e-7) and the same as F(4,3) end to end in the synthetic regime
// (tests/tools/winograd_numerics.py, profiles/r04_winograd_numerics.txt); worst cases of the kernel fuzz are about a third above
// F(4,3)'s (tests/tools/wino_fuzz.py: 4e-5 at |out| ~ 3, x 1.5 above 192 channels); how the headroom to the 1e-4 bar shrinks with
// the weights' gain: profiles/r05_regime_sweep.txt.
//
// Eight transform points = eight waves: block = 8 waves = 2 per SIMD with up to 256 registers each, wave xi owns
// M_xi for (32 MT) output channels x 64 tiles (2 columns of 32) = 320 outputs.  With MT = 4 one B fragment feeds
// 8 MFMAs (4 in the 12-wave F(4,3) shapes): per MFMA 0.625 packed vector + 0.375 LDS instructions against 0.75 + 0.5
// (0.83 + 0.5 in the 96-row shape), for a fifth fewer MFMAs.
// On this chip every instruction a SIMD issues beside v_mfma_f32_32x32x2_f32 costs matrix-pipe time wherever it is
// placed (profiles/r04_wino_kloop_handsched.txt), so the instruction count per MFMA is what sets the K loop's rate.
//   * A (transformed weights [cin/16][tap group][8][cout_pad][16]) goes global -> registers in fragment layout as in
//     conv_wino.hip;
//   * B: the raw 16-channel slab is staged in LDS once per chunk (double buffered, one barrier per chunk), wave w
//     stages channel pair w.  Samples are de-interleaved into 5 planes (local sample v -> plane v % 5, index v / 5),
//     channel pairs interleaved, so that lane `tile` reading sample 5 tile + c is a stride-1, conflict-free
//     ds_read2_b64 (both tile columns in one instruction) with immediate offsets for the channel pair; the slab's
//     alignment slack is absorbed by the WRITER, so the reader's offsets are per-block constants;
//   * every row of B^T (points +-1, +-2, +-1/2: six samples, unit coefficient on the last; 0 and inf: four) is one
//     chain v = x5; v = fma(c_j, x_j, v): five packed FMAs over a k-step pair;
//   * epilogue: the eight M_xi of a 32 x 32 tile meet through LDS, a thread applies A^T for 4 rows x 1 tile (5 outputs
//     each), the 5-sample tiles are re-laid row-major in LDS and leave as aligned 16-byte vectors with bias, residuals
//     and scale applied.
#pragma once
#include "fh_common.h"

#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 v_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 v_bf16x2 __attribute__((ext_vector_type(2)));

// ---- BF = true: the same contraction on the BF16 matrix cores, fp32-grade (round 6) --------------------------------------------
// Every fp32 operand is split EXACTLY into three bf16 pieces, x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m),
// round to nearest even, both subtractions exact in fp32), and a 16-channel k-block is six v_mfma_f32_32x32x16_bf16 over the piece
// pairs (h h), (h m), (h l), (m h), (m m), (l h) into the same fp32 accumulator: 6 x 32 matrix-pipe cycles against 8 x 64 of the
// fp32 form; the dropped pairs are <= 2^-24 |a b| (conv_wino.hip, tools/micro/bf16x6.hip).  The weights arrive pre-split
// (packing.split_bf3 of pack_wino54_weight: [cin/16][tap group][8][cout_pad][3 pieces][16] bf16 = 96 bytes per row and chunk), the
// B operands are transformed exactly as in the fp32 form (same fma chain per element: same V bits) and split in registers.
// Everything beside the MFMAs is written one result per lane and this form's instantiations are compiled with
// -fno-slp-vectorize (conv_wino54_bf.hip): packed fp32 instructions share the matrix cores' fp32 lanes and stall a bf16 MFMA
// (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), plain ones issue in its shadow.  What the loop's shape is worth was
// measured on a stand-alone model of it first (tools/micro/bf54.hip, profiles/r06_bf16x6_coissue.txt).
constexpr int V_A3 = 24;             // floats (96 bytes) per output row and 16-channel chunk of the three-piece weights
__device__ __forceinline__ unsigned v_pack_bf16(float a, float b) {        // v_cvt_pk_bf16_f32
  const v_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float v_bf_lo(unsigned p) {                     // the low bf16 of a pair as a float (v_perm_b32: from
  return __uint_as_float(__builtin_amdgcn_perm(0u, p, 0x01000c0cu));       // `p << 16` the combiner derives a second v_cvt_pk)
}
__device__ __forceinline__ void v_split8(const float (&v)[8], v_bf16x8& h, v_bf16x8& m, v_bf16x8& l) {
  unsigned hp[4], mp[4], lp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    hp[i] = v_pack_bf16(a, b);
    const float ra = a - v_bf_lo(hp[i]), rb = b - __uint_as_float(hp[i] & 0xffff0000u);
    mp[i] = v_pack_bf16(ra, rb);
    const float sa = ra - v_bf_lo(mp[i]), sb = rb - __uint_as_float(mp[i] & 0xffff0000u);
    lp[i] = v_pack_bf16(sa, sb);
  }
  h = __builtin_bit_cast(v_bf16x8, (u32x4){hp[0], hp[1], hp[2], hp[3]});
  m = __builtin_bit_cast(v_bf16x8, (u32x4){mp[0], mp[1], mp[2], mp[3]});
  l = __builtin_bit_cast(v_bf16x8, (u32x4){lp[0], lp[1], lp[2], lp[3]});
}

constexpr int V_CK = 16;             // input channels per chunk
constexpr int V_THREADS = 512;       // 8 waves = 8 transform points
constexpr int V_BT = 64;             // F(5,4) tiles per block (2 columns of 32)
constexpr int V_OUT = 5 * V_BT;      // outputs per block and row
constexpr int V_P = 72;              // samples per (plane, channel pair) row: >= V_BT + (4 * 3 + 7) / 5 + 1
constexpr int V_PAIR = 2 * V_P;      // floats of one channel pair inside a plane
constexpr int V_PP = 8 * V_PAIR;     // plane pitch, floats
constexpr int V_SLAB = 5 * V_PP;     // floats of a slab (one 16-channel chunk)
constexpr int V_BUF = V_SLAB + 128;  // ... of a slab buffer: + one trash slot per lane for samples that are not needed
constexpr int V_XQ = (5 * (V_BT - 1) + 4 * 3 + 7 + 3) / 4 + 1;      // aligned quads a slab can touch (85)
constexpr int V_EP = 36;             // column pitch (floats) of the exchange tiles: conflict-free b128 (conv_wino.hip)
constexpr int V_YP = 168;            // row pitch (floats) of the output staging: 16-byte aligned, 4 * 168 % 64 == 32
constexpr int V_EPI = 2 * 8 * 32 * V_EP;        // exchange tiles of two 32 x 32 sub-tiles (both columns of one mt)
constexpr int V_Y = 2 * 32 * V_YP;
// LDS: [two slab buffers | exchange tiles]; the output staging reuses the slab space (the exchange tiles do not: a wave that
// leaves the K loop writes its accumulators while slower waves still read the slab, and a round's exchange writes need not wait
// for the previous round's stores: 2 MT block barriers in the epilogue instead of 3 MT)
static_assert(V_Y <= 2 * V_BUF, "output staging does not fit the slab space");
constexpr int V_LDS_FLOATS = 2 * V_BUF + V_EPI;
constexpr int V_RUN = 8;             // n-blocks of a panel that run together on one XCD (conv_wino.hip: W_RUN)

// Phase-major rows (dilated convs) are tiled as ONE sequence: the d phases one after the other in a "position" space in which
// every phase owns TS tile slots = its ceil(n / 5) tiles + >= 3 empty ones (multiple of 4, so that a phase starts on a 16-byte
// boundary of the position axis as it does in memory).  Position P = p (5 TS) + u is sample u of phase p; u >= n_p reads as
// zero, which is the conv's padding at both ends of every phase (the empty slots are >= 15 positions, the taps reach <= 10), and
// is not stored.  A block is 64 consecutive tile slots whatever phases they belong to: a dilation-5 conv over 5 x 1 000 samples
// runs 16 blocks per panel, not 5 x 4 (tiles of 320 outputs per phase: 751 us against 500 us undilated at C = 768).  Which block
// computes a tile does not change the tile's arithmetic.
__host__ __device__ inline int v_tile_slots(int len, int dil) { return (((len + dil - 1) / dil + 4) / 5 + 3 + 3) & ~3; }

// Rows of B^T for the points 0, 1, -1, 2, -2, 1/2, -1/2, inf (tests/tools/winograd_numerics.py: toom_cook(5, 4, ...)):
//   v = x[off[5]];  v = fma(coef[j], x[off[j]], v)  for j = 0 .. 4        (coef 0: the slot repeats a real sample)
constexpr int kB8Off[8][6] = {{2, 4, 0, 0, 0, 6}, {1, 2, 3, 4, 5, 6}, {1, 2, 3, 4, 5, 6}, {1, 2, 3, 4, 5, 6},
                              {1, 2, 3, 4, 5, 6}, {1, 2, 3, 4, 5, 6}, {1, 2, 3, 4, 5, 6}, {1, 3, 5, 1, 1, 7}};
__device__ const float kB8Coef[8][5] = {{5.25f, -5.25f, -1.f, 0.f, 0.f},      {1.f, 1.f, -4.25f, -4.25f, 1.f},
                                        {-1.f, 1.f, 4.25f, -4.25f, -1.f},     {0.5f, 0.25f, -2.5f, -1.25f, 2.f},
                                        {-0.5f, 0.25f, 2.5f, -1.25f, -2.f},   {2.f, 4.f, -2.5f, -5.f, 0.5f},
                                        {-2.f, 4.f, 2.5f, -5.f, -0.5f},       {-1.f, 5.25f, -5.25f, 0.f, 0.f}};

// LDS float offsets of the 6 samples of tap group g relative to a lane's base, [point][g][j]: sample c = 4 g + kB8Off[point][j]
// lies in plane c % 5 at index c / 5 (made here: from the run-time point they were ~200 scalar instructions per wave and block)
struct VToff {
  int v[8][3][6];
};
constexpr VToff make_vtoff() {
  VToff t = {};
  for (int x = 0; x < 8; ++x)
    for (int g = 0; g < 3; ++g)
      for (int j = 0; j < 6; ++j) {
        const int c = 4 * g + kB8Off[x][j];
        t.v[x][g][j] = (c % 5) * V_PP + (c / 5) * 2;
      }
  return t;
}
__device__ const VToff kB8Toff = make_vtoff();

struct VSeg {
  const float* x;
  const float* u;
  int cin, ngrp, center;
};
__device__ __forceinline__ VSeg load_vseg(const fh_wino_seg* S) {
  VSeg w;
  w.x = uni(S->x);
  w.u = uni(S->u);
  w.cin = uni(S->cin);
  w.ngrp = uni(S->ngrp);
  w.center = uni(S->center);
  return w;
}

// VL: rows are contiguous and 16-byte aligned (phase-major tensors, or dilation 1 and len % 4 == 0): the slab is fetched
// with 16-byte loads and the outputs leave as 16-byte vectors; else 4-byte accesses (any length, any dilation in the
// plain layout).  Same arithmetic either way.
// launch constants the block -> work mapping divides by (fh_common.h: fh_fastdiv)
struct VDivs {
  fh_fastdiv run_len, runs_per_panel, co_tiles, batch, dil;
};

// H16 (MT = 2 only): the block is 48 output channels, three 16-row tiles of v_mfma_f32_16x16x4_f32 instead of two 32-row tiles of
// the 32x32x2 form -- the C = 48 stage in 64-row blocks spent a quarter of its matrix instructions on padding rows.  The B
// operands are the same registers: lane (tile l31, channel half lh) of the k-step pair (e, e + 1) becomes, after one
// v_permlane16_swap of the pair's two registers, lane (tile l & 15, channel {e, e + 1} x {half 0, 1} = l >> 4) for tiles 0-15 in one
// register and tiles 16-31 in the other; the A lanes read their 4 channels one float later when (l >> 4) is odd, so that elements
// 0 and 2 of a fragment are the channels e + (g & 1) of the two k-step pairs of a half.  The epilogue keeps its two 32-row rounds
// (the second has 16 real rows).  The sum over a chunk's channels runs in another order than in the 32x32x2 form: a conv's
// bits depend on H16, which is fixed per stage (cout_pad), never chosen per launch.
template <int MT, bool VL, bool H16 = false, bool BF = false>
__global__ __attribute__((amdgpu_flat_work_group_size(V_THREADS, V_THREADS), amdgpu_waves_per_eu(2, 2)))
void conv_wino54_kernel(const fh_wino_group* __restrict__ groups, int n_groups, int batch, int co_tiles, int n_tiles,
                        int run_len, int dil, int pm, const int* __restrict__ run_map, int n_runs, VDivs dv) {
  static_assert(!H16 || MT == 2, "the 16-row form is the 48-row block");
  static_assert(!(H16 && BF), "the bf16 x 6 form has 32-row tiles only");
  constexpr int BM = H16 ? 48 : 32 * MT;
  constexpr int MA = H16 ? 3 : MT;                    // A fragments (row tiles) per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];      // V_LDS_FLOATS

  // ---- block -> (panel, n block): panels = (group, batch, co tile), heavy groups first (conv_wino.hip) ----------
  const int panels = n_groups * batch * co_tiles;
  const int runs_per_panel = (int)dv.runs_per_panel.d;
  const int total_runs = panels * runs_per_panel;
  const int bid = blockIdx.x;
  const int slot = bid >> 3;
  const int slot_run = fh_div(slot, dv.run_len);
  int run = slot_run * 8 + (bid & 7);
  if (run_map) {
    if (run >= n_runs) return;
    run = uni(run_map[run]);
  }
  if (run >= total_runs) return;
  const int panel = fh_div(run, dv.runs_per_panel);
  const int ntile = fh_mod(run, panel, dv.runs_per_panel) * run_len + fh_mod(slot, slot_run, dv.run_len);
  if (ntile >= n_tiles) return;
  const int gb = fh_div(panel, dv.co_tiles);
  const int cot = fh_mod(panel, gb, dv.co_tiles);
  const int gi = fh_div(gb, dv.batch);
  const int b = fh_mod(gb, gi, dv.batch);
  const fh_wino_group* __restrict__ G = groups + gi;
  const bool cat = pm != 0;                   // phase-major rows: tiled in the concatenated position space (above)
  const int ntile_d = cat ? 0 : fh_div(ntile, dv.dil);
  const int ph = cat ? 0 : fh_mod(ntile, ntile_d, dv.dil);      // plain layout: phase of the decimated sequence ...
  const int tb = cat ? ntile : ntile_d;                          // ... and 320-output block within it

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);       // transform point of this wave (= channel pair it stages)
  const int l31 = lane & 31, lh = lane >> 5;
  const int co0 = cot * BM;
  const int len = uni(G->len), cout_pad = uni(G->cout_pad), nseg = uni(G->nseg);
  // (the epilogue's fields too: read there, they are a scalar-cache miss -- ~1 us -- with every wave of the CU waiting)
  const int nres = uni(G->nres), cout = uni(G->cout);
  const float scale = G->scale;
  const float* __restrict__ const bias = uni(G->bias);
  const float* const outp = uni((const float*)G->out);
  const float* const resp[3] = {uni(G->res[0]), uni(G->res[1]), uni(G->res[2])};
  // len = lq dil + lr: phase p of a row holds lq + (p < lr) samples
  const int lq = fh_div(len, dv.dil), lr = fh_mod(len, lq, dv.dil);
  const int lmax = lq + (lr > 0);                    // samples of phase 0 = ceil(len / dil)
  const int ps = 5 * (((lmax + 4) / 5 + 3 + 3) & ~3);            // (cat) positions per phase = 5 v_tile_slots(len, dil)
  if (cat ? tb * V_OUT >= dil * ps : tb * V_OUT * dil + ph >= len) return;

  // phase-major tensors (pm): row = dil phases of lp samples, x[p + dil u] at p * lp + u
  const int lp = (lmax + 3) & ~3;
  const int pitch = pm ? dil * lp : len;             // floats per (batch, channel) row, inputs and outputs
  const int nvalid = lq + (ph < lr);                 // (plain layout) real decimated samples of this block's phase
  // aligned position P (a multiple of 4; cat: concatenated space, else decimated index of the plain row at dilation 1) ->
  // float offset of its quad inside a (batch, channel) row, or -1 outside the tensor; nv = real samples from P on in its row.
  // (P / ps per lane: ps is a per-group value, so no host-made multiplier; P < 2^24 is exact in fp32, the product with the
  // rounded reciprocal is within 1 of the quotient, and one step either way corrects it: 12 vector instructions against
  // the ~30 of the generic division, 5 calls per lane and epilogue round)
  const float ps_inv = __builtin_amdgcn_rcpf((float)ps);
  auto locate = [&](int P, int& nv) -> int {
    if (cat) {
      int p = P >= 0 ? (int)((float)P * ps_inv) : dil;
      int u = P - p * ps;
      if (P >= 0 && u < 0) { --p; u += ps; }
      if (P >= 0 && u >= ps) { ++p; u -= ps; }
      nv = p < dil ? lq + (p < lr) - u : 0;
      return (p < dil && u < lp) ? p * lp + u : -1;
    }
    nv = len - P;
    return (P >= 0 && P < len) ? P : -1;
  };

  // this wave's row of B^T: coefficients and sample slots, wave-uniform
  f32x2 bco[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float c = __uint_as_float(uni((int)__float_as_uint(kB8Coef[xi][j])));
    bco[j] = (f32x2){c, c};
  }
  int toff[3][6];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int j = 0; j < 6; ++j) toff[g][j] = uni(kB8Toff.v[xi][g][j]);

  f32x4 acc16[H16 ? 3 : 1][2][2];                     // (H16) [16-row tile][column][tiles 0-15 / 16-31]
#pragma unroll
  for (int i = 0; i < (H16 ? 3 : 1); ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc16[i][j][0] = acc16[i][j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x16 acc[H16 ? 1 : MT][2];
#pragma unroll
  for (int i = 0; i < (H16 ? 1 : MT); ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- slab staging: wave w stages channel pair w; lane handles aligned quads lane and 64 + lane of the pair's two
  // rows (two 16-byte loads each), then 4 ds_write_b64 (channel pair) per quad into the planes.  Local sample
  // v = 4 q + e lands at position w = v - sh (sh = the 0-3 samples between the quad boundary and the first sample
  // needed), plane w % 5, index w / 5: the readers' offsets do not depend on the block's alignment.
  // Everything that does not change from chunk to chunk is made once per segment (setup_seg): the loads' byte offsets
  // inside a chunk's first row (the chunk and the pair's second row are SCALAR offsets of the buffer loads), the LDS
  // positions, and whether the block touches the row's end at all.  Per chunk a lane then issues 4 loads and 8
  // unconditional LDS writes (samples that are not needed go to a per-lane trash slot behind the slab): with address
  // arithmetic and per-sample branches in the chunk loop the staging cost as many vector instructions as the
  // transform of a one-group chunk.
  constexpr int NXS = VL ? 8 : 6;                      // samples per lane and channel: 2 quads, or lane + 64 i, i < 6
  constexpr int NLD = VL ? 2 : 6;                      // loads per lane and channel
  int wofs[NXS];                                       // LDS float offsets of this lane's samples inside a slab buffer
  unsigned voff[NLD];                                  // byte offsets of the loads inside the chunk's first row, or out of range
  int ua = 0;                                          // position of local sample 0 (VL: a multiple of 4)
  int nvq[NLD];                                        // (VL) real samples from each quad's first sample on
  bool tail = false;                                   // (VL) the slab holds positions that are not real samples
  auto setup_seg = [&](const VSeg& S) {
    const int ub = tb * V_OUT - S.center;
    const int sh = VL ? (ub & 3) : 0;
    ua = ub - sh;
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      const int v = VL ? 4 * (lane + 64 * (i >> 2)) + (i & 3) : lane + 64 * i;
      const int w = v - sh;
      wofs[i] = (v < 4 * V_XQ && w >= 0) ? (w % 5) * V_PP + xi * V_PAIR + (w / 5) * 2 : V_SLAB + 2 * lane;
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if constexpr (VL) {
        const int q = lane + 64 * i;
        const int off = locate(ua + 4 * q, nvq[i]);                           // (outside the rows: zero padding)
        voff[i] = (q < V_XQ && off >= 0) ? (unsigned)off * 4u : 0x80000000u;
      } else {
        const int u = ua + lane + 64 * i;                                     // decimated index (plain layout only: phase-major rows are aligned)
        const int pos = u * dil + ph;                                         // position in the clip
        const bool ok = lane + 64 * i < 4 * V_XQ && (unsigned)pos < (unsigned)len;
        voff[i] = ok ? (unsigned)pos * 4u : 0x80000000u;
        nvq[i] = 0;
      }
    }
    if constexpr (VL) {              // block-uniform: does every staged position hold a real sample of one row?
      int nv0;
      tail = locate(ua, nv0) < 0 || nv0 < 4 * V_XQ;
    }
  };
  unsigned xq[2][NXS];                                 // [channel of the pair][sample]
  auto load_x = [&](const VSeg& S, int chunk, bool valid) {
    const __amdgpu_buffer_rsrc_t r =
        make_rsrc(uni(S.x + (size_t)b * S.cin * pitch), valid ? (unsigned)(S.cin * pitch) * 4u : 0u);
    const int so0 = (chunk * V_CK + 2 * xi) * pitch * 4, so1 = so0 + pitch * 4;        // scalar offsets of the pair's rows
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if constexpr (VL) {
        const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(r, voff[i], so0, 0);
        const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(r, voff[i], so1, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xq[0][4 * i + e] = t0[e];
          xq[1][4 * i + e] = t1[e];
        }
      } else {
        xq[0][i] = __builtin_amdgcn_raw_buffer_load_b32(r, voff[i], so0, 0);
        xq[1][i] = __builtin_amdgcn_raw_buffer_load_b32(r, voff[i], so1, 0);
      }
    }
  };
  auto store_x = [&](int buf) {
    float* dst = lds + buf * V_BUF;
    if (tail) {                                        // (last block of a row: zero past the end; block-uniform branch)
#pragma unroll
      for (int i = 0; i < NXS; ++i)
        if ((i & 3) >= nvq[i >> 2]) xq[0][i] = xq[1][i] = 0u;
    }
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      float* q = dst + wofs[i];
      q[0] = __uint_as_float(xq[0][i]);
      q[1] = __uint_as_float(xq[1][i]);
    }
  };

  // A fragments of one step, [mt][half]: half h holds k-steps 4h .. 4h+3 (conv_wino.hip)
  // (H16: row l & 15 of a 16-row tile, channels 8 (g >> 1) + (g & 1) + 4 h .. + 3 with g = l >> 4: a 4-byte aligned 16-byte load)
  u32x4 areg[BF ? 1 : MA][2];
  const int a_lane = H16 ? ((lane & 15) * V_CK + 8 * (lane >> 5) + ((lane >> 4) & 1)) * 4 : (l31 * V_CK + lh * 8) * 4;
  auto load_a_half = [&](int h, const VSeg& S, int chunk, int g, bool valid) {
    const float* up = uni(S.u + ((size_t)((chunk * S.ngrp + g) * 8 + xi) * cout_pad + co0) * V_CK);
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, valid ? BM * V_CK * 4 : 0);
#pragma unroll
    for (int mt = 0; mt < (BF ? 1 : MA); ++mt)
      areg[mt][h] = __builtin_amdgcn_raw_buffer_load_b128(r, a_lane + mt * (H16 ? 16 : 32) * V_CK * 4 + 16 * h, 0, 0);
  };
  // BF: the three pieces of one tap group, [mt][piece]: lane (row l31, half lh) reads its 8 channels of each piece (16 bytes).
  // One register set, refilled behind the group's last MFMA: a second set (request a group ahead) measured no faster on the
  // model loop and costs 12 MT registers the 96 / 128-row blocks do not have (tools/micro/bf54.hip)
  u32x4 a3[BF ? MT : 1][3];
  const int a3_lane = (l31 * V_A3 + lh * 4) * 4;
  // pieces = bit mask of the pieces to request (1 h, 2 m, 4 l)
  auto load_a3 = [&](const VSeg& S, int chunk, int g, bool valid, int pieces = 7) {
    const float* up = uni(S.u + ((size_t)((chunk * S.ngrp + g) * 8 + xi) * cout_pad + co0) * V_A3);
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, valid ? BM * V_A3 * 4 : 0);
    // (the lane offset passes through an empty asm: the 3 MT offsets are then immediates of the loads (row tile 2 on: one add),
    // not 3 MT loop-invariant registers carried through the K loop)
    int o = a3_lane;
    asm volatile("" : "+v"(o));
#pragma unroll
    for (int mt = 0; mt < (BF ? MT : 1); ++mt) {
      const int om = mt < 2 ? o : o + 2 * 32 * V_A3 * 4;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        if (pieces & (1 << pc))
          a3[mt][pc] = __builtin_amdgcn_raw_buffer_load_b128(r, om + (mt & 1) * 32 * V_A3 * 4 + 32 * pc, 0, 0);
    }
  };
  // L2 warm-up of the A tiles of the NEXT chunk (all its tap groups, this wave's xi), as in conv_wino.hip
  // (ordinary loads into two registers that the next prefetch "reads" before it overwrites them -- not the untracked inline-asm
  // load of rounds 2-5, whose late write hit a reused register in every instantiation that spilled: conv_wino.hip)
  unsigned pf[2] = {0u, 0u};
  auto prefetch_a = [&](const VSeg& S, int chunk, bool valid) {
    constexpr int ROWF = BF ? V_A3 : V_CK;                                 // floats per weight row and chunk
    const float* up = uni(S.u + ((size_t)(chunk * S.ngrp * 8 + xi) * cout_pad + co0) * ROWF);
    const unsigned gstride = 8u * (unsigned)cout_pad * ROWF * 4u;          // bytes between tap groups
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, valid ? (unsigned)(S.ngrp - 1) * gstride + BM * ROWF * 4 : 0u);
    asm volatile("" :: "v"(pf[0]), "v"(pf[1]));          // the previous prefetch has landed before its registers are reused
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // (lanes past the tile's lines: out of range; the 96-byte rows of the BF form are 0.75 BM lines, the first 32 lanes' worth
      // of which is touched: enough to start the L2 fill of the tile)
      const unsigned off = l31 < BM * ROWF / 32 ? (unsigned)(2 * j + lh) * gstride + (unsigned)l31 * 128u : 0x80000000u;
      pf[j] = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    }
  };

  // ---- K loop -----------------------------------------------------------------------------------------------------
  // A chunk is a flat sequence of k-step PAIRS p = 4 g + kp (tap group g, channel pair kp of the lane's half): 6
  // ds_read2_b64 (one sample for both columns each), 10 packed FMAs, 4 MT MFMAs.  The samples of pair p + 1 are
  // requested behind the transform of pair p, in the same registers.
  int xbuf = 0;
  const int lane_base = lh * 4 * V_PAIR + l31 * 2;
  auto run_segment = [&](auto gc, const VSeg& S) {
    constexpr int GC = decltype(gc)::value;
    const int nch = S.cin / V_CK;
    setup_seg(S);
    if constexpr (BF) {
      load_a3(S, 0, 0, true);
    } else {
      load_a_half(0, S, 0, 0, true);
      load_a_half(1, S, 0, 0, true);
    }
    load_x(S, 0, true);
    store_x(xbuf);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      const bool has_next = c + 1 < nch;
      const float* xsb = lds + xbuf * V_BUF + lane_base;
      f32x2 xr[6][2];                                   // [sample slot][column] = (k-step 2 kp, 2 kp + 1)
      auto fetch = [&](int p) {
        const int g = p >> 2, kp = p & 3;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float* q = xsb + toff[g][j] + kp * V_PAIR;
          xr[j][0] = *reinterpret_cast<const f32x2*>(q);
          xr[j][1] = *reinterpret_cast<const f32x2*>(q + 64);
        }
      };
      if constexpr (BF) {
        // bf16 x 6 form, one tap group (16-channel k-block) and tile column at a time: the lane's 8 channels -- 4 channel pairs x 6
        // samples from the slab, the fp32 form's fma chain per element -- are split into three pieces and meet the weights' three
        // pieces in 6 MFMAs per 32 x 32 tile.
#pragma unroll
        for (int g = 0; g < GC; ++g) {
          if (g == 0) {
            load_x(S, c + 1, has_next);                  // stored at the end of this chunk
            prefetch_a(S, c + 1, has_next);
          }
          // (one tile column at a time: both columns' samples in flight at once -- one ds_read2_b64 per sample as in the fp32 form --
          // are 24 registers more than the 96-row block has: the allocator then spills a hundred)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            // (nothing crosses from one column's work into the other's: the scheduler otherwise hoists the next column's -- and
            // the next tap group's -- LDS reads and FMAs above this column's MFMAs until the register file is full, and the
            // allocator, left with no room for the 16-register accumulator tuples, spills hundreds.  Overlap of the vector work
            // with the matrix work comes from the SIMD's other wave)
            __builtin_amdgcn_sched_barrier(0);
            float v[8];
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
              f32x2 x[6];
#pragma unroll
              for (int j = 0; j < 6; ++j) x[j] = *reinterpret_cast<const f32x2*>(xsb + toff[g][j] + kp * V_PAIR + 64 * nt);
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                float t = __builtin_fmaf(bco[0][0], x[0][e], x[5][e]);
#pragma unroll
                for (int j = 1; j < 5; ++j) t = __builtin_fmaf(bco[j][0], x[j][e], t);
                v[2 * kp + e] = t;
              }
            }
            v_bf16x8 bh, bm, bl;
            v_split8(v, bh, bm, bl);
            // (piece pairs outermost, row tiles inside: MT independent accumulators between two MFMAs on the same one.  The
            // weights' pieces are used in the order h h h m m l, and each is requested for the NEXT tap group as soon as the group's
            // second column is through with it -- h three pairs, m one pair ahead of the end of the MFMAs, and in the order the next
            // group needs them: one register set, yet most of the L2 round trip runs under this group's matrix work)
            const bool same_chunk = g + 1 < GC;
            const int nc = same_chunk ? c : c + 1, ng = same_chunk ? g + 1 : 0;
            const bool nv = same_chunk || has_next;
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) {
              constexpr int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {0, 1, 2, 0, 1, 0};      // (h h) (h m) (h l) (m h) (m m) (l h)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v_bf16x8, a3[mt][pa[pp]]),
                                                                      pb[pp] == 0 ? bh : pb[pp] == 1 ? bm : bl, acc[mt][nt], 0, 0, 0);
              if (nt == 1 && (pp == 2 || pp == 4 || pp == 5)) load_a3(S, nc, ng, nv, pp == 2 ? 1 : pp == 4 ? 2 : 4);
            }
          }
        }
      } else {
      fetch(0);
#pragma unroll
      for (int p = 0; p < 4 * GC; ++p) {
        const int g = p >> 2, kp = p & 3, h = kp >> 1;
        if (p == 0) {
          load_x(S, c + 1, has_next);                  // stored at the end of this chunk
          prefetch_a(S, c + 1, has_next);
        }
        f32x2 bf[2];                                   // [column] = B values of k-steps 2 kp, 2 kp + 1
        // (inline asm: packed FMAs, and kept out of the MFMA groups below.  The two columns' chains are interleaved so that
        // no packed FMA reads the result of the one issued just before it; VALU result -> MFMA operand needs 2 wait states:
        // the s_nop behind the last one covers both columns, conv_wino.hip)
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(bf[0]) : "s"(bco[0]), "v"(xr[0][0]), "v"(xr[5][0]));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(bf[1]) : "s"(bco[0]), "v"(xr[0][1]), "v"(xr[5][1]));
#pragma unroll
        for (int j = 1; j < 4; ++j) {
          asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(bf[0]) : "s"(bco[j]), "v"(xr[j][0]));
          asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(bf[1]) : "s"(bco[j]), "v"(xr[j][1]));
        }
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(bf[0]) : "s"(bco[4]), "v"(xr[4][0]));
        asm("v_pk_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(bf[1]) : "s"(bco[4]), "v"(xr[4][1]));
        if (p + 1 < 4 * GC) fetch(p + 1);
        if constexpr (H16) {
          float b16[2][2];                             // [column][tiles 0-15 / 16-31]: 4 channels x 16 tiles each
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(bf[nt][0]), __float_as_uint(bf[nt][1]), false, false);
            b16[nt][0] = __uint_as_float(sw[0]);
            b16[nt][1] = __uint_as_float(sw[1]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m16 = 0; m16 < 3; ++m16)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int hf = 0; hf < 2; ++hf)
                acc16[m16][nt][hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(areg[m16][h][2 * (kp & 1)]), b16[nt][hf],
                                                                          acc16[m16][nt][hf], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const int e = 2 * (kp & 1) + k2;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[mt][h][e]), bf[nt][k2],
                                                                   acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (kp & 1) {                                  // this half of the A registers is free: refill it for the next step
          const bool same_chunk = g + 1 < GC;
          load_a_half(h, S, same_chunk ? c : c + 1, same_chunk ? g + 1 : 0, same_chunk || has_next);
        }
      }
      }
      if (has_next) {
        store_x(xbuf ^ 1);
        __syncthreads();
        xbuf ^= 1;
      }
    }
    xbuf ^= 1;                   // the next segment's first slab goes to the buffer nobody is reading
  };

  // Segments are sorted by tap-group count, descending (host: make_wino_group)
  int sg = 0;
  VSeg S0 = load_vseg(&G->seg[0]);
  auto run_all = [&](auto gc) {
    while (sg < nseg && S0.ngrp == decltype(gc)::value) {
      run_segment(gc, S0);
      ++sg;
      if (sg < nseg) S0 = load_vseg(&G->seg[sg]);
    }
  };
  // (1 .. 3 groups of 4 taps: k <= 12, the host's WINO_MAX_K.  Tried: the three row classes of B^T -- point 0, the six
  // +- points, inf -- as compile-time sample positions behind one wave-uniform branch per segment, to drop the address
  // table: the 128 accumulator registers then go through the branch's merge and the compiler spills ~1 900 of them)
  run_all(std::integral_constant<int, 3>{});
  run_all(std::integral_constant<int, 2>{});
  run_all(std::integral_constant<int, 1>{});
  // (a segment of 4 or more tap groups matches no pass above: a caller that bypassed the host's checks would get
  // bias + residual back with FH_OK.  Fail loudly instead.)
  if (sg < nseg) __builtin_trap();

  // ---- epilogue ---------------------------------------------------------------------------------------------------
  const size_t oslab = (size_t)b * cout * pitch;
  const unsigned slab_bytes = (unsigned)cout * (unsigned)pitch * 4u;
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(outp + oslab, slab_bytes);
  const __amdgpu_buffer_rsrc_t rr0 = make_rsrc(nres > 0 ? resp[0] + oslab : nullptr, nres > 0 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rr1 = make_rsrc(nres > 1 ? resp[1] + oslab : nullptr, nres > 1 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rr2 = make_rsrc(nres > 2 ? resp[2] + oslab : nullptr, nres > 2 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, bias ? (unsigned)cout * 4u : 0u);
  float* const E = lds + 2 * V_BUF;                   // [column nt][xi][tile col 32][row, pitch V_EP]: behind the slab buffers
  float* const Y = lds;                               // [column nt][row 32][160 outputs, pitch V_YP]: in the slab space
  // (BF: the thread id the epilogue's geometry is made from passes through an empty asm, so that none of it -- ~40 registers of row /
  // column / offset tables -- is computed above the K loop and carried through it: the bf16 x 6 loop has no registers to spare)
  int etid = tid;
  if constexpr (BF) asm volatile("" : "+v"(etid));
  const int ent = etid >> 8, erq = (etid >> 5) & 7, ecol = etid & 31;     // A^T item: column, row quad, tile
  const int v_first = tb * V_OUT;                     // position of the block's first output
  // store items of this thread: 5 of the 1280 float4 of its column's 32 x 160 sub-tile, the same 5 in every round:
  // where their vectors lie in a row is found once
  int srow[5], scol[5], ioff[5], inv[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int item = (etid & 255) + 256 * i;
    srow[i] = item / 40;
    scol[i] = (item % 40) * 4;
    const int v0 = v_first + ent * 160 + scol[i];                         // position of the vector's first output
    ioff[i] = VL ? locate(v0, inv[i]) : (inv[i] = nvalid - v0, v0);
  }
  // Requests of the store phases: the bias of every round's items now (the K loop's registers are free; a load in a round is
  // an HBM round trip the round waits out), the first residual of a round's items one round ahead -- round 0's here,
  // round r + 1's in front of round r's stores, two register sets: vmcnt counts loads and stores in order, so a load
  // requested BEHIND a round's stores is not there before those stores are acknowledged.
  // (the 128-row tile has no registers for the second set: its residual is requested in the round that adds it)
  constexpr bool AHEAD = MT < 4;
  float bvall[MT][5];
  u32x4 rpre[AHEAD ? 2 : 1][5];
  auto item_geom = [&](int mt, int i, int& nreal, unsigned& soff) {
    const int co = co0 + mt * 32 + srow[i];
    // (mt * 32 + srow < BM: the 48-row block's second round has 16 rows)
    nreal = (co < cout && mt * 32 + srow[i] < BM && ioff[i] >= 0) ? inv[i] : 0;       // real outputs from the vector's first on (<= 0: none)
    soff = ((unsigned)co * (unsigned)pitch + (unsigned)(ioff[i] >= 0 ? ioff[i] : 0)) * 4u;
    return co;
  };
  auto request_res = [&](int mt, u32x4 (&rp)[5]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int nreal;
      unsigned soff;
      item_geom(mt, i, nreal, soff);
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rr0, nreal >= 4 ? soff : 0x80000000u, 0, 0);
    }
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int nreal;
      unsigned soff;
      const int co = item_geom(mt, i, nreal, soff);
      bvall[mt][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rbias, nreal > 0 ? (unsigned)co * 4u : 0x80000000u, 0, 0));
    }
  if (AHEAD && VL && nres > 0) request_res(0, rpre[0]);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    // (no barrier here: E is not the slab, and the readers of the previous round's E are past that round's second barrier)
    if constexpr (H16) {
      // a lane's 4 accumulators of a 16 x 16 tile: rows 4 (l >> 4) .. + 3 of column l & 15; round mt holds the 16-row tiles
      // 2 mt and 2 mt + 1 (the third tile is round 1's rows 0-15: its rows 16-31 are not written and their outputs not stored)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            if (2 * mt + t < 3)
              *reinterpret_cast<f32x4*>(E + ((nt * 8 + xi) * 32 + (lane & 15) + 16 * hf) * V_EP + 16 * t + 4 * (lane >> 4)) =
                  acc16[2 * mt + t][nt][hf];
    } else {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float* ew = E + ((nt * 8 + xi) * 32 + l31) * V_EP + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(ew + 8 * q) =
              (f32x4){acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
      }
    }
    __syncthreads();
    {
      const float* er = E + (ent * 8 * 32 + ecol) * V_EP + 4 * erq;
      f32x4 m[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) m[x] = *reinterpret_cast<const f32x4*>(er + x * 32 * V_EP);
      // (vector index = row of the quad) points 0, 1, -1, 2, -2, 1/2, -1/2, inf
      const f32x4 s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4], s3 = m[5] + m[6], d3 = m[5] - m[6];
      f32x4 y[5];
      y[0] = ((m[0] + s1) + s2) + s3;
      y[1] = __builtin_elementwise_fma((f32x4)(0.5f), d3, __builtin_elementwise_fma((f32x4)(2.f), d2, d1));
      y[2] = __builtin_elementwise_fma((f32x4)(0.25f), s3, __builtin_elementwise_fma((f32x4)(4.f), s2, s1));
      y[3] = __builtin_elementwise_fma((f32x4)(0.125f), d3, __builtin_elementwise_fma((f32x4)(8.f), d2, d1));
      y[4] = __builtin_elementwise_fma((f32x4)(0.0625f), s3, __builtin_elementwise_fma((f32x4)(16.f), s2, s1)) + m[7];
      float* yw = Y + (ent * 32 + 4 * erq) * V_YP + 5 * ecol;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 5; ++q) yw[i * V_YP + q] = y[q][i];
    }
    // Store phase.  Every global access below is an unconditional buffer operation (nothing to do = out-of-range offset)
    // in straight-line code.  (With the bias load and a per-lane "whole vector?" branch inside the item loop the compiler put
    // s_waitcnt vmcnt(0) behind every item's loads: each of the 15 items of a block waited for its load AND for the write
    // acknowledge of the item before it: tools/exp/w54_fixed_cost.py.)
    const float (&bv)[5] = bvall[mt];
    unsigned soff[5];
    int nreal[5];
    bool vec = VL;                                      // every item of this lane is a whole vector, or nothing
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      item_geom(mt, i, nreal[i], soff[i]);
      vec = vec && (nreal[i] >= 4 || nreal[i] <= 0);
    }
    if (VL && nres > 0 && (AHEAD ? mt + 1 < MT : true)) request_res(AHEAD ? mt + 1 : mt, rpre[AHEAD ? (mt + 1) & 1 : 0]);
    const u32x4 (&rp)[5] = rpre[AHEAD ? mt & 1 : 0];
    // one path per WAVE: whole vectors everywhere (all blocks but a row's last, when rows are 16-byte aligned), or 4-byte accesses
    const bool wave_vec = VL && __builtin_amdgcn_ballot_w64(!vec) == 0ull;
    __syncthreads();
    f32x4 yv[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) yv[i] = *reinterpret_cast<const f32x4*>(Y + (ent * 32 + srow[i]) * V_YP + scol[i]);
    if (wave_vec) {
      f32x4 rs[5];
      if (nres > 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
          rs[i] = (f32x4){__uint_as_float(rp[i][0]), __uint_as_float(rp[i][1]), __uint_as_float(rp[i][2]), __uint_as_float(rp[i][3])};
        if (nres > 1) {
          u32x4 t[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) t[i] = __builtin_amdgcn_raw_buffer_load_b128(rr1, nreal[i] >= 4 ? soff[i] : 0x80000000u, 0, 0);
#pragma unroll
          for (int i = 0; i < 5; ++i)
            rs[i] += (f32x4){__uint_as_float(t[i][0]), __uint_as_float(t[i][1]), __uint_as_float(t[i][2]), __uint_as_float(t[i][3])};
        }
        if (nres > 2) {
          u32x4 t[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) t[i] = __builtin_amdgcn_raw_buffer_load_b128(rr2, nreal[i] >= 4 ? soff[i] : 0x80000000u, 0, 0);
#pragma unroll
          for (int i = 0; i < 5; ++i)
            rs[i] += (f32x4){__uint_as_float(t[i][0]), __uint_as_float(t[i][1]), __uint_as_float(t[i][2]), __uint_as_float(t[i][3])};
        }
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        f32x4 o = {yv[i][0] + bv[i], yv[i][1] + bv[i], yv[i][2] + bv[i], yv[i][3] + bv[i]};
        if (nres > 0) o += rs[i];
        o *= scale;
        const u32x4 ou = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
        __builtin_amdgcn_raw_buffer_store_b128(ou, ro, nreal[i] >= 4 ? soff[i] : 0x80000000u, 0, 0);
      }
    } else {                                            // a row ends inside this wave's vectors, or rows are not 16-byte aligned
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        unsigned off[4];
        float rs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (plain layout with a dilation: output v sits at ph + dil v of the row)
          off[q] = q < nreal[i] ? (pm || dil == 1 ? soff[i] + 4u * q : soff[i] + 4u * (unsigned)(ph + (dil - 1) * (v_first + ent * 160 + scol[i]) + dil * q)) : 0x80000000u;
          rs[q] = 0.f;
        }
        if (nres > 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) rs[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr0, off[q], 0, 0));
          if (nres > 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rs[q] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr1, off[q], 0, 0));
          }
          if (nres > 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rs[q] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr2, off[q], 0, 0));
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float o = yv[i][q] + bv[i];
          if (nres > 0) o += rs[q];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o * scale), ro, off[q], 0, 0);
        }
      }
    }
  }
  asm volatile("" :: "v"(pf[0]), "v"(pf[1]));            // (the last prefetch: waited for, never used)
}

// 320-output blocks of a row: per phase in the plain layout, over the concatenated tile slots of all phases in the phase-major one
int wino54_n_tiles(int len, int dilation, int pm) {
  return pm ? fh_cdiv((long long)dilation * v_tile_slots(len, dilation), V_BT) : fh_cdiv(fh_cdiv(len, dilation), V_OUT) * dilation;
}

template <int MT, bool VL, bool H16 = false, bool BF = false>
int launch_wino54(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation, int pm,
                  hipStream_t stream, const int* run_map, int n_runs) {
  constexpr int BM = H16 ? 48 : 32 * MT;
  FH_CHECK_ARG(cout_pad > 0 && cout_pad % BM == 0, "fh_conv_wino54_f32: cout_pad %d not a multiple of %d", cout_pad, BM);
  const int co_tiles = cout_pad / BM;
  const int n_tiles = wino54_n_tiles(len, dilation, pm);
  const long long panels = (long long)n_groups * batch * co_tiles;
  const int run_len = fh_cdiv(n_tiles, fh_cdiv(n_tiles, V_RUN));
  const long long runs = run_map ? (long long)n_runs : panels * fh_cdiv(n_tiles, run_len);
  const long long blocks = (long long)fh_cdiv(runs, 8) * 8 * run_len;
  FH_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fh_conv_wino54_f32: grid too large");
  const VDivs dv = {fh_make_fastdiv((unsigned)run_len), fh_make_fastdiv((unsigned)fh_cdiv(n_tiles, run_len)),
                    fh_make_fastdiv((unsigned)co_tiles), fh_make_fastdiv((unsigned)batch), fh_make_fastdiv((unsigned)dilation)};
  static std::atomic<bool> lds_opt_in[FH_MAX_DEVICES];      // (> 64 KB of dynamic LDS: once per device, conv_wino.hip)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FH_MAX_DEVICES) {
    fh_set_error("fh_conv_wino54_f32: no current HIP device (or ordinal >= %d)", FH_MAX_DEVICES);
    return FH_E_LAUNCH;
  }
  if (!lds_opt_in[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wino54_kernel<MT, VL, H16, BF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       V_LDS_FLOATS * 4);
    if (e != hipSuccess) {
      fh_set_error("fh_conv_wino54_f32: cannot reserve %d bytes of LDS on device %d: %s", V_LDS_FLOATS * 4, dev, hipGetErrorString(e));
      return FH_E_LAUNCH;
    }
    lds_opt_in[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((conv_wino54_kernel<MT, VL, H16, BF>), dim3((unsigned)blocks), dim3(V_THREADS), V_LDS_FLOATS * 4, stream, groups,
                     n_groups, batch, co_tiles, n_tiles, run_len, dilation, pm, run_map, n_runs, dv);
  FH_CHECK_LAUNCH("fh_conv_wino54_f32");
  return FH_OK;
}

}  // namespace
