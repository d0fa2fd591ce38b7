// Winograd F(4,3) conv, second block shape: 4-wave blocks, all SIX transform points of an output in ONE wave.
//
// Same maths, same weights (vocoder.pack_wino_weight), same descriptors and the same bits as conv_wino.hip (replaces
// the AMPBlock convs of /root/reference/src/flowhigh/models/bigvgan/models.py:36-72); what differs is who holds what:
//
//   conv_wino.hip : block = 12 waves = 6 transform points xi x 2 tile halves; the six M_xi of an output sit in six
//                   waves, so the epilogue exchanges every accumulator through LDS (8 block barriers, ~8 us per
//                   block with the matrix pipes idle: tools/wino_trace2.py), and with one 12-wave block per CU nothing
//                   else runs on the CU during its prologue and epilogue.
//   this kernel   : block = 4 waves (one per SIMD), tile 64 co x 64 tiles (256 outputs); wave (mt, nc) owns the
//                   32 co x 32 tiles position (mt, nc) for ALL six xi (6 x 16 accumulator registers).  y = A^T M is
//                   then plain per-lane arithmetic on the wave's own registers: no exchange, no barrier, every wave
//                   stores its outputs on its own.  Three such blocks are resident per CU (3 waves per SIMD, <= 168
//                   VGPRs, 37 KB of LDS each) and are independent, so the prologue, the per-chunk barrier and the
//                   epilogue of one block are covered by the matrix work of the other two.
//
// Per wave and k-step pair (2 MFMA k-steps = 4 input channels of the 16-channel chunk, one tap group):
//   6 ds_read_b64 (samples x0..x5 of the wave's 32 tiles, channel pair interleaved), 12 packed-fp32 instructions (the
//   six rows of B^T with the sub-expressions shared between rows 1/2 and 3/4: the canonical arithmetic written down
//   in conv_wino.hip), 6 eight-byte weight loads (global -> registers in fragment layout, one tap-group ahead in time),
//   12 v_mfma_f32_32x32x2_f32 into six independent accumulators.
// The input slab [16 channels][4 x 64 + halo samples] is staged once per chunk (double buffered, one barrier of the
// block's 4 waves per chunk) de-interleaved into 4 planes (sample i -> plane i & 3), ROTATED so that slab sample 0 is
// the first tap of tile 0 exactly: every B read then has a compile-time LDS offset (no address arithmetic in the loop).
// Vector loads only (16-byte aligned contiguous rows: phase-major tensors, or dilation 1 and len % 4 == 0); other
// launches stay on conv_wino.hip's 64 x 256 tile, which gives the same bits.
#include "fh_common.h"

#include <type_traits>

#include "conv_wino_int.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int V_CK = 16;                 // input channels per chunk
constexpr int V_THREADS = 256;
constexpr int V_BM = 64;                 // output channels per block
constexpr int V_BT = 64;                 // F(4,3) tiles per block (256 outputs)
constexpr int V_P = V_BT + 8;            // plane pitch (samples)
constexpr int V_RP2 = 8 * V_P;           // floats of one channel PAIR (4 planes x V_P x 2 channels interleaved)
constexpr int V_SLAB = (V_CK / 2) * V_RP2;            // floats per slab buffer (18 KB)
constexpr int V_XQ = V_BT + 5;           // absolute quads a slab can touch
constexpr int V_NITEM = (V_XQ + 31) / 32;             // staging items per thread (32 quads per channel pair and item)
constexpr int V_PAD = 8;                 // floats in front of the slabs (rotated samples -3 .. -1 of channel pair 0)

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

__global__ __attribute__((amdgpu_flat_work_group_size(V_THREADS, V_THREADS), amdgpu_waves_per_eu(3, 3)))
void conv_wino2_kernel(const fh_wino_group* __restrict__ groups, int n_groups, int batch, int co_tiles,
                       int n_tiles, int run_len, int dil, int pm, const int* __restrict__ run_map, int n_runs) {
  __shared__ __attribute__((aligned(16))) float lds_raw[V_PAD + 2 * V_SLAB];
  float* const lds = lds_raw + V_PAD;

  // ---- block -> (panel, n block): the mapping of conv_wino.hip (XCD aware, equal runs, optional run map) ----------
  const int panels = n_groups * batch * co_tiles;
  const int runs_per_panel = (n_tiles + run_len - 1) / run_len;
  const int total_runs = panels * runs_per_panel;
  const int bid = blockIdx.x;
  const int slot = bid >> 3;
  int run = (slot / run_len) * 8 + (bid & 7);
  if (run_map) {
    if (run >= n_runs) return;
    run = uni(run_map[run]);
  }
  if (run >= total_runs) return;
  const int panel = uni(run / runs_per_panel);
  const int ntile = uni((run % runs_per_panel) * run_len + (slot % run_len));
  if (ntile >= n_tiles) return;
  const int cot = uni(panel % co_tiles);
  const int gb = uni(panel / co_tiles);
  const int b = uni(gb % batch);
  const fh_wino_group* __restrict__ G = groups + uni(gb / batch);
  const int ph = uni(ntile % dil);            // phase of the decimated sequence
  const int tb = uni(ntile / dil);            // 256-output block within the phase

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 1;                    // 32-row half of the 64 output channels
  const int nc = wave >> 1;                   // 32-tile half of the 64 tiles
  const int l31 = lane & 31, lh = lane >> 5;
  const int co0 = cot * V_BM;
  const int len = uni(G->len), cout_pad = uni(G->cout_pad), nseg = uni(G->nseg);
  if (tb * (4 * V_BT) * dil + ph >= len) return;         // (ragged launches: past this group's row)

  const int lp = ((len + dil - 1) / dil + 3) & ~3;
  const int pitch = pm ? dil * lp : len;             // floats per (batch, channel) row, inputs and outputs

  f32x16 acc[6];
#pragma unroll
  for (int xi = 0; xi < 6; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

  // ---- slab staging: thread (channel pair vp, quad lane vq) handles absolute quads q0 + vq + 32 item ---------------
  // Two aligned 16-byte loads (4 consecutive samples of the pair's two channels), then 4 ds_write_b64 (channel pair)
  // into the 4 planes, rotated by the slab origin: global sample 4 (q0 + q) + e is slab sample i = 4 q + e - s with
  // s = (first sample needed) & 3; i in -3 .. -1 (unused samples of the first quad) fall into the unused last slot
  // of the plane before (V_PAD floats in front of plane 0 of channel pair 0).
  const int vp = tid >> 5, vq = tid & 31;
  u32x4 xq[2];                                         // ONE item in flight: [channel of the pair]
  auto vl_load = [&](const VSeg& S, int chunk, bool valid, int item) {
    const __amdgpu_buffer_rsrc_t r =
        make_rsrc(uni(S.x + (size_t)b * S.cin * pitch), valid ? (unsigned)(S.cin * pitch) * 4u : 0u);
    const int ub = tb * (4 * V_BT) - S.center;                           // first decimated index needed (>= -5)
    const int q0 = (ub - (ub & 3)) >> 2;                                 // floor(ub / 4)
    const int rowlen = pm ? lp : len;
    const int q = vq + 32 * item, qa = q0 + q;
    const bool ok = q < V_XQ && qa >= 0 && 4 * qa < rowlen;              // (outside the row: zero padding)
    const int e0 = (chunk * V_CK + 2 * vp) * pitch + (pm ? ph * lp : 0) + 4 * qa;
    xq[0] = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)e0 * 4u : 0x80000000u, 0, 0);
    xq[1] = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)(e0 + pitch) * 4u : 0x80000000u, 0, 0);
  };
  auto vl_store = [&](const VSeg& S, int buf, int item) {
    const int ub = tb * (4 * V_BT) - S.center;
    const int s = ub & 3;
    const int ua = ub - s;                                               // decimated index of the first quad's sample 0
    const int q = vq + 32 * item;
    const int nvalid = pm ? (len - ph + dil - 1) / dil : len;            // samples of this phase / row
    if (ua + 4 * V_XQ > nvalid) {                                        // last block of the row: zero past the end
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ua + 4 * q + e >= nvalid) { xq[0][e] = 0u; xq[1][e] = 0u; }
    }
    if (q < V_XQ) {
      float* dst = lds + buf * V_SLAB + vp * V_RP2 + 2 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = e - s;                                             // -3 .. 3 (uniform)
        const int off = ((i & 3) * V_P + (i >> 2)) * 2;                  // plane (i mod 4), index floor(i / 4)
        *reinterpret_cast<f32x2*>(dst + off) = (f32x2){__uint_as_float(xq[0][e]), __uint_as_float(xq[1][e])};
      }
    }
  };

  // ---- A fragments: lane (row l31, half lh) holds U[xi][co0 + 32 mt + l31][8 lh + 2 kp + {0, 1}] of a k-step pair ----
  u32x2 areg[1][6];                                    // [xi]: ONE set, refilled right after the MFMAs that read it
  const int a_lane = ((mt * 32 + l31) * V_CK + lh * 8) * 4;
  const unsigned xi_stride = (unsigned)cout_pad * V_CK * 4u;            // bytes between transform points
  auto load_a = [&](int buf, const VSeg& S, int chunk, int g, int kp, bool valid) {
    const float* up = uni(S.u + ((size_t)((chunk * S.ngrp + g) * 6) * cout_pad + co0) * V_CK);
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, valid ? 5u * xi_stride + V_BM * V_CK * 4u : 0u);
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
      areg[buf][xi] = __builtin_amdgcn_raw_buffer_load_b64(r, a_lane + 8 * kp, (int)(xi * xi_stride), 0);
  };
  // L2 warm-up of the A tiles of the NEXT chunk (the register prefetch above is one pair deep: enough for an L2
  // hit, not for HBM, and the blocks of a weight panel run in lockstep).  Wave w touches tap group w: one lane per
  // 128-byte line of the (group, xi) tile [64 co][16], lanes 0-31 xi = 2 j, lanes 32-63 xi = 2 j + 1.
  unsigned pf = 0;
  auto prefetch_a = [&](const VSeg& S, int chunk, bool valid) {
    const float* up = uni(S.u + ((size_t)((chunk * S.ngrp + wave) * 6) * cout_pad + co0) * V_CK);
    const __amdgpu_buffer_rsrc_t r = make_rsrc(up, (valid && wave < S.ngrp) ? 5u * xi_stride + V_BM * V_CK * 4u : 0u);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned off = (unsigned)(2 * j + lh) * xi_stride + (unsigned)l31 * 128u;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(pf) : "v"(off), "s"(r) : "memory");
    }
  };

  // ---- prologue -------------------------------------------------------------------------------------------------
  VSeg S0 = load_vseg(&G->seg[0]);
  load_a(0, S0, 0, 0, 0, true);
  int xbuf = 0;
#pragma unroll
  for (int item = 0; item < V_NITEM; ++item) {
    vl_load(S0, 0, true, item);
    vl_store(S0, 0, item);
  }
  __syncthreads();

  // B reads: slab sample j' = 3 g + j of tile n, channel pair (lane half lh, kp): compile-time offsets from xsb
  auto run_segment = [&](auto gc, const VSeg& S, const VSeg& Sn, bool more_seg) {
    constexpr int GC = decltype(gc)::value;
    constexpr int NP = 4 * GC;                          // k-step pairs of a chunk
    const int nch = S.cin / V_CK;
    for (int c = 0; c < nch; ++c) {
      const bool last_chunk = c == nch - 1;
      const bool has_next = !last_chunk || more_seg;
      const VSeg& Sx = last_chunk ? Sn : S;              // owner of the next chunk
      const int cx = last_chunk ? 0 : c + 1;
      const float* xsb = lds + xbuf * V_SLAB + lh * 4 * V_RP2 + (nc * 32 + l31) * 2;
      f32x2 xr[6];
      auto fetch = [&](int p) {
        const int g = p >> 2, kp = p & 3;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int jj = 3 * g + j;
          xr[j] = *reinterpret_cast<const f32x2*>(xsb + ((jj & 3) * V_P + (jj >> 2)) * 2 + kp * V_RP2);
        }
      };
      fetch(0);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int g = p >> 2, kp = p & 3;
        // staging of the NEXT chunk's slab, one item in flight: item i is requested at pair i NP / V_NITEM and
        // stored where the next one is requested (the last one after the loop)
#pragma unroll
        for (int item = 0; item < V_NITEM; ++item) {
          if (p == item * NP / V_NITEM) {
            if (item > 0 && has_next) vl_store(Sx, xbuf ^ 1, item - 1);
            vl_load(Sx, cx, has_next, item);
            if (item == 0) prefetch_a(Sx, cx, has_next);
          }
        }
        // the six rows of B^T (canonical arithmetic of conv_wino.hip), packed over the k-step pair
        f32x2 bf[6];
        {
          const f32x2 p0 = __builtin_elementwise_fma((f32x2)(-5.f), xr[2], xr[4]);
          const f32x2 p5 = __builtin_elementwise_fma((f32x2)(-5.f), xr[3], xr[5]);
          const f32x2 p12 = __builtin_elementwise_fma((f32x2)(-4.f), xr[2], xr[4]);
          const f32x2 q12 = __builtin_elementwise_fma((f32x2)(-4.f), xr[1], xr[3]);
          const f32x2 p34 = xr[4] - xr[2];
          const f32x2 q34 = xr[3] - xr[1];
          bf[0] = __builtin_elementwise_fma((f32x2)(4.f), xr[0], p0);
          bf[5] = __builtin_elementwise_fma((f32x2)(4.f), xr[1], p5);
          bf[1] = p12 + q12;
          bf[2] = p12 - q12;
          bf[3] = __builtin_elementwise_fma((f32x2)(2.f), q34, p34);
          bf[4] = __builtin_elementwise_fma((f32x2)(-2.f), q34, p34);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (p + 1 < NP) fetch(p + 1);                    // (same registers: the transform has consumed them)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int xi = 0; xi < 6; ++xi)
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[0][xi][k2]), bf[xi][k2], acc[xi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // weights of the next pair (this chunk, or the first pair of the next one) into the registers the MFMAs above
        // have just read: the other two blocks of the CU cover the round trip
        {
          const bool same_chunk = p + 1 < NP;
          const VSeg& Sa = same_chunk ? S : Sx;
          load_a(0, Sa, same_chunk ? c : cx, same_chunk ? (p + 1) >> 2 : 0, same_chunk ? (p + 1) & 3 : 0,
                 same_chunk || has_next);
        }
        (void)g; (void)kp;
      }
      if (has_next) {
        vl_store(Sx, xbuf ^ 1, V_NITEM - 1);
        __syncthreads();
        xbuf ^= 1;
      }
    }
  };

  // Segments are sorted by tap-group count, descending (host: make_wino_group)
  int sg = 0;
  auto run_all = [&](auto gc) {
    while (sg < nseg && S0.ngrp == decltype(gc)::value) {
      const bool more_seg = sg + 1 < nseg;
      const VSeg Sn = load_vseg(&G->seg[more_seg ? sg + 1 : sg]);
      run_segment(gc, S0, Sn, more_seg);
      S0 = Sn;
      ++sg;
    }
  };
  run_all(std::integral_constant<int, 4>{});
  run_all(std::integral_constant<int, 3>{});
  run_all(std::integral_constant<int, 2>{});
  run_all(std::integral_constant<int, 1>{});

  // ---- epilogue: y = A^T M on the wave's own registers, bias + residuals, scale, store -------------------------------
  // Lane (tile n = 32 nc + l31, half lh) holds rows co = co0 + 32 mt + 8 (r >> 2) + (r & 3) + 4 lh, r = 0..15, of
  // every M_xi: 4 outputs per (row, tile) = one 16-byte vector; lanes 0-31 of a store cover 512 contiguous bytes.
  const int nres = uni(G->nres);
  const float scale = G->scale;
  const int cout = uni(G->cout);
  const float* __restrict__ bias = uni(G->bias);
  const int ostride = uni(G->out_stride) > 1 ? uni(G->out_stride) : 1;
  const int ophase = uni(G->out_phase);
  const int opitch = pitch * ostride;
  const size_t oslab = (size_t)b * cout * opitch;
  const unsigned slab_bytes = (unsigned)cout * (unsigned)opitch * 4u;
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(uni((const float*)G->out) + oslab, slab_bytes);
  const __amdgpu_buffer_rsrc_t rr0 = make_rsrc(nres > 0 ? uni(G->res[0]) + oslab : nullptr, nres > 0 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rr1 = make_rsrc(nres > 1 ? uni(G->res[1]) + oslab : nullptr, nres > 1 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rr2 = make_rsrc(nres > 2 ? uni(G->res[2]) + oslab : nullptr, nres > 2 ? slab_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, bias ? (unsigned)cout * 4u : 0u);
  const bool vec = (pm || (dil == 1 && (len & 3) == 0)) && ostride == 1;
  const int v0 = tb * (4 * V_BT) + (nc * 32 + l31) * 4;                    // decimated index of y[0]
  const int row0 = co0 + mt * 32 + 4 * lh;                                 // + 8 q + i
  const bool colok = (v0 + 3) * dil + ph < len;
  const unsigned coloff = (pm ? (unsigned)(ph * lp) : 0u) + (unsigned)v0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // rows row0 + 8 q + {0..3}: bias (4 consecutive floats) and the first residual requested before the arithmetic
    float bq[4];
    u32x4 rq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = row0 + 8 * q + i;
      bq[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rbias, co < cout ? (unsigned)co * 4u : 0x80000000u, 0, 0));
    }
    if (vec && nres > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = row0 + 8 * q + i;
        const bool ok = co < cout && colok;
        rq[i] = __builtin_amdgcn_raw_buffer_load_b128(rr0, ok ? ((unsigned)co * (unsigned)opitch + coloff) * 4u : 0x80000000u, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i;
      const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      float y[4];
      y[0] = m0 + s12 + s34;
      y[1] = fmaf(2.f, d34, d12);
      y[2] = fmaf(4.f, s34, s12);
      y[3] = fmaf(8.f, d34, d12) + m5;
      const int co = row0 + 8 * q + i;
      const bool rowok = co < cout;
      const float bv = bq[i];
      const unsigned rowoff = (unsigned)co * (unsigned)opitch + (pm ? (unsigned)(ph * lp) : 0u);
      if (vec && colok) {
        const unsigned off = rowok ? (rowoff + (unsigned)v0) * 4u : 0x80000000u;
        f32x4 o = {y[0] + bv, y[1] + bv, y[2] + bv, y[3] + bv};
        if (nres > 0) {
          u32x4 t = rq[i];
          f32x4 rs = {__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
          if (nres > 1) {
            t = __builtin_amdgcn_raw_buffer_load_b128(rr1, off, 0, 0);
            rs += (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
          }
          if (nres > 2) {
            t = __builtin_amdgcn_raw_buffer_load_b128(rr2, off, 0, 0);
            rs += (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
          }
          o += rs;
        }
        o *= scale;
        const u32x4 ou = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
        __builtin_amdgcn_raw_buffer_store_b128(ou, ro, off, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = ph + dil * (v0 + e);
          const unsigned off = (rowok && n < len) ? (rowoff + (unsigned)(pm ? v0 + e : n * ostride + ophase)) * 4u
                                                  : 0x80000000u;
          float o = y[e] + bv;
          if (nres > 0) {
            float rs = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr0, off, 0, 0));
            if (nres > 1) rs += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr1, off, 0, 0));
            if (nres > 2) rs += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr2, off, 0, 0));
            o += rs;
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o * scale), ro, off, 0, 0);
        }
      }
    }
  }
  if (pf == 0x7fc12345u) lds_raw[0] = 0.f;      // keeps pf alive; never true for weights
}

}  // namespace

int fh_wino2_launch(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation,
                    int phase_major, hipStream_t stream, const int* run_map, int n_runs) {
  FH_CHECK_ARG(cout_pad > 0 && cout_pad % V_BM == 0, "fh_conv_wino_f32: cout_pad %d not a multiple of %d", cout_pad, V_BM);
  const int co_tiles = cout_pad / V_BM;
  const int n_tiles = fh_cdiv(fh_cdiv(len, dilation), 4 * V_BT) * dilation;
  const long long panels = (long long)n_groups * batch * co_tiles;
  const int run_len = fh_cdiv(n_tiles, fh_cdiv(n_tiles, FH_WINO_RUN));
  const long long runs = run_map ? (long long)n_runs : panels * fh_cdiv(n_tiles, run_len);
  const long long blocks = (long long)fh_cdiv(runs, 8) * 8 * run_len;
  FH_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "fh_conv_wino_f32: grid too large");
  hipLaunchKernelGGL(conv_wino2_kernel, dim3((unsigned)blocks), dim3(V_THREADS), 0, stream, groups, n_groups, batch,
                     co_tiles, n_tiles, run_len, dilation, phase_major, run_map, n_runs);
  FH_CHECK_LAUNCH("fh_conv_wino_f32");
  return FH_OK;
}
