// Narrow stages of BigVGAN (C <= 48): the residual-stack convs, shaped for few channels.  (Round 5 also ran the anti-aliased
// activation in front of the conv inside this launch; that form measured 0.92 ms per step slower than the pair of launches --
// matrix and packed vector instructions share the fp32 ALUs, profiles/r05_amp_ablation.txt -- and left the library in round 6.)
//
// Replaces, per launch, the `xt = conv(xt)` half of one `xt = act(x); xt = conv(xt)` pair of the AMP blocks
// (/root/reference/src/flowhigh/models/bigvgan/models.py:63-72 AMPBlock1, :108-117 AMPBlock2;
//  Activation1d alias_free_torch/act.py:23-28 = UpSample1d resample.py:25-33 -> SnakeBeta activations.py:107-120 ->
//  DownSample1d filter.py:86-95; the convs are "same"-padded, k = 3 / 7 / 11, dilation 1 / 3 / 5), including "+ x"
//  (:70) and the "xs / num_kernels" average over the blocks (:181-187, K segments of one group).
//
// Why: at 24 / 48 channels the conv kernels built for wide stages spent 37 % of a block outside their K loop (eight transform
// points in eight waves meet through LDS) or multiplied 25 % padding rows.  Here the conv is shaped for few channels:
//   * block = 4 waves, 64 F(5,4) tiles = 300-320 outputs of every channel (a multiple of 20 x dilation, so that every block
//     starts on a tile boundary of every dilation phase and on a 16-byte boundary); two blocks per CU, so that one block's
//     prologue / epilogue / barriers run under the other's K loop;
//   * the K loop walks 8-channel chunks.  For the NEXT chunk, wave w loads channel pair w over the block's samples + halo and
//     writes it into the chunk's slab in the Winograd read layout: per dilation phase 5 planes (sample w -> plane w % 5, index w / 5),
//     channel pairs interleaved, so that lane `tile` reads sample 5 tile + e with ONE conflict-free ds_read_b64 for both
//     channels of its pair;
//   * ONE wave owns all 8 transform points of its 16 tiles: v_mfma_f32_16x16x4_f32 with M = 16 output channels, N = 16
//     tiles, K = 4 channels; 8 x ceil(C / 16) accumulator tiles per wave.  B^T (8 x 8) is applied per lane on the packed
//     fp32 ALU over the k-step pair with the even / odd halves of the +- points shared (26 packed ops for 8 points), A^T
//     per lane in the epilogue: no exchange between waves anywhere;
//   * transformed weights [chunk][tap group][point][lane][row tile][2] stream L2 -> LDS one (chunk, tap group) stage ahead
//     (4-12 KB), every wave reads its A fragments from there (the whole set is 25-220 KB per conv: it does not fit beside
//     the slabs, and from global memory it would be one dword per MFMA and lane);
//   * epilogue: A^T, then the 5-output tiles of all dilation phases are re-interleaved through LDS into plain rows and leave
//     as 16-byte vectors with bias, residuals and scale.  No phase-major tensors on either side.
// Every sample's arithmetic depends on its absolute position only (tiles are anchored at multiples of 5 of the decimated
// index, the channel order of the sum is fixed), so a clip gives the same bits alone, in a batch, in a ragged launch and in
// aligned time chunks.
#include "fh_common.h"

#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int F_THREADS = 256;             // 4 waves
constexpr int F_NOUT = 376;                // samples a wave stages per channel and pass: t = tA + j, j < 376
constexpr int F_SLAB = 3840;               // floats of a slab (one 8-channel chunk): 40 d PL <= this for d <= 6
constexpr int F_SLAB_BUF = F_SLAB + 128;   // + one trash pair per lane
constexpr int F_YP = 324;                  // row pitch of the output staging
constexpr int F_MAX_D = 6;
// tuning switches (A/B builds: tools/build_variant.sh)
#ifndef F_A_AHEAD           // A fragments requested one pair of points ahead of their MFMAs
#define F_A_AHEAD 1
#endif

// tiles per dilation phase and plane length (index units) of the slab
__host__ __device__ constexpr int f_ntp(int d) { return 4 * (16 / d); }
__host__ __device__ constexpr int f_tb(int d) { return 5 * d * f_ntp(d); }
__host__ __device__ constexpr int f_pl(int d) { return d == 1 ? 80 : d == 2 ? 48 : f_ntp(d) + 5; }
static_assert(40 * 1 * f_pl(1) <= F_SLAB && 40 * 2 * f_pl(2) <= F_SLAB && 40 * 3 * f_pl(3) <= F_SLAB &&
              40 * 4 * f_pl(4) <= F_SLAB && 40 * 5 * f_pl(5) <= F_SLAB && 40 * 6 * f_pl(6) <= F_SLAB, "slab size");

template <int MA>
constexpr int f_wstage() { return 1024 * MA; }           // floats of one (chunk, tap group) weight stage

constexpr int F_YBUF = 16 * F_YP;          // output staging

template <int MA>
constexpr int f_lds_floats() { return 2 * F_SLAB_BUF + 2 * f_wstage<MA>() + F_YBUF; }

// MA: 16-row output tiles (channels <= 16 MA).  VEC: rows are 16-byte aligned (len % 4 == 0 for every group).
// The conv's input is read as it is: the Activation1d in front of it is a launch of its own (act1d.hip).  (Round 5 also built the
// form with the activation inside this launch: 0.92 ms per step slower, profiles/r05_amp_ablation.txt; it left the library in
// round 6 -- `git show 60fcf48:flowhigh_amd/csrc/amp_fused.hip` has it.)
//
// Persistent blocks: block b works on tiles b, b + gridDim, ... of the launch's flattened (group, batch item, tile) list, and the
// chunk pipeline runs across tile boundaries: the last K step of a tile requests and stages the first chunk (rows, weight stage)
// of the block's NEXT tile and the bias / first residual of its own epilogue, so that no HBM round trip is waited for between
// tiles.  (One block per tile spent more time outside its K loop than inside at 24 channels: 3-9 K steps of ~0.6 us against
// ~6 us of descriptor reads, first loads, barriers and store phases.)
template <int MA, bool VEC>
#ifndef F_WAVES_PER_EU
#define F_WAVES_PER_EU 2
#endif
__global__ __attribute__((amdgpu_flat_work_group_size(F_THREADS, F_THREADS), amdgpu_waves_per_eu(F_WAVES_PER_EU, F_WAVES_PER_EU)))
void amp_actconv_kernel(const fh_amp_group* __restrict__ groups, const fh_amp_tile* __restrict__ tiles, int channels, int d,
                        int total_tiles, int cmax) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const slab0 = lds;
  float* const wbuf0 = lds + 2 * F_SLAB_BUF;
  float* const ybuf = wbuf0 + 2 * f_wstage<MA>();     // output staging of the epilogue

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  d = uni(d);
  cmax = uni(cmax);
  const int NTP = uni(f_ntp(d)), TB = 5 * d * NTP, PL = uni(f_pl(d)), NT = d * NTP;
  const int A = (cmax * d + 3) & ~3;                   // a tile's staged range starts at tA = t0 - A (16-byte aligned)
  const int nch = channels >> 3;

  // ---- tile -> (group, batch item, first output): one 16-byte entry of the host-made tile list ---------------------------
  // (found in the kernel -- prefix search over the groups, two integer divisions -- a tile cost ~150 instructions and a chain
  // of four dependent loads before its first request could leave)
  struct Tile {
    const fh_amp_group* G;
    int len, bb, t0, nseg;
    bool valid;
  };
  auto locate = [&](int tile) {
    Tile T;
    T.valid = tile < total_tiles;
    const fh_amp_tile* e = tiles + (T.valid ? tile : 0);
    T.G = groups + uni(e->group);
    T.bb = uni(e->batch_item);
    T.t0 = uni(e->t0);
    T.len = uni(e->len);
    T.nseg = uni(T.G->nseg);
    return T;
  };
  // ---- slab geometry of this lane (the same for every tile: the launch's largest center `cmax` places the samples) ------
  // writer: sample j of the staged range is t = tA + j: rel = j - A = d u + p, slab sample w = u + cmax of phase p.
  // A lane holds the quads j = 4 (lane + 64 v) + e of its two loads.
  constexpr int NWO = 8;
  int wofs[NWO];
#pragma unroll
  for (int r = 0; r < NWO; ++r) {
    const int j = 4 * (lane + 64 * (r >> 2)) + (r & 3);
    const int rel = j - A;
    int u = rel / d;
    int p = rel - u * d;
    if (p < 0) { p += d; --u; }
    const int w = u + cmax;
    const int q = w / 5;
    const bool ok = j < F_NOUT && w >= 0 && q < PL;
    wofs[r] = ok ? (((p * 5 + (w - 5 * q)) * 4 + wv) * PL + q) * 2 : F_SLAB + 2 * lane;
  }
  // reader: lane (tile n = lane & 15 of this wave's 16, channel pair kq = lane >> 4)
  const int kq = lane >> 4;
  int tl = 16 * wv + (lane & 15);
  const bool tile_ok = tl < NT;
  if (!tile_ok) tl = NT - 1;
  const int tp = tl / NTP, tit = tl - tp * NTP;
  const int rd_base = ((tp * 5 * 4 + kq) * PL + tit) * 2;
  const int ypos = d * 5 * tit + tp;                   // outputs q = 0..4 of the lane's tile at block-relative d (5 tit + q) + tp
  // store items of this thread: vectors item = tid + 256 i of a round's 16 rows x TB / 4, the same in every round and tile
  // (row << 16 | first column, unpacked behind an opaque move in the epilogue: left visible, the compiler makes every round's
  // offsets of every item loop-invariant values and carries ~30 registers of them through the K loop)
  const int vpr = TB >> 2;
  int sitem[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int item = tid + 256 * i;
    const int row = item / vpr;
    sitem[i] = (row << 16) | ((item - row * vpr) * 4);
  }

  // ---- loads --------------------------------------------------------------------------------------------------------
  struct Seg {                                         // one K segment of one tile (wave-uniform)
    const float* x;
    const float* u;
    int ngrp, center, len, bb, tA;
  };
  auto load_seg = [&](const Tile& T, int s, int par) {
    Seg S;
    const fh_amp_seg* P = &T.G->seg[s];
    S.x = uni(P->x);
    S.u = uni(P->u);
    S.ngrp = uni(P->ngrp);
    S.center = uni(P->center);
    S.len = uni(T.len);
    S.bb = uni(T.bb);
    S.tA = uni(T.t0 - A);
    return S;
  };
  u32x4 xq[2][2];                                    // [channel of the pair][vector]: x[tA + 4 f ..], f = lane, lane + 64
  auto load_x = [&](const Seg& S, int chunk, bool valid) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ch = chunk * 8 + 2 * wv + h;
      const __amdgpu_buffer_rsrc_t r = make_rsrc(S.x + ((size_t)S.bb * channels + ch) * (size_t)S.len, valid ? (unsigned)S.len * 4u : 0u);
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int f = lane + 64 * v;
        const int t = S.tA + 4 * f;
        const bool in = f < F_NOUT / 4;
        if (VEC) {
          xq[h][v] = __builtin_amdgcn_raw_buffer_load_b128(r, in ? (unsigned)(t * 4) : 0x80000000u, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            xq[h][v][e] = __builtin_amdgcn_raw_buffer_load_b32(r, in ? (unsigned)((t + e) * 4) : 0x80000000u, 0, 0);
        }
      }
    }
  };
  u32x4 wq[MA];                                      // this thread's 16-byte pieces of the next weight stage
  auto load_w = [&](const float* p, bool valid) {
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p, valid ? (unsigned)f_wstage<MA>() * 4u : 0u);
#pragma unroll
    for (int i = 0; i < MA; ++i) wq[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(tid + 256 * i) * 16u, 0, 0);
  };
  auto store_w = [&](int buf) {
    float* dst = wbuf0 + buf * f_wstage<MA>();
#pragma unroll
    for (int i = 0; i < MA; ++i) *reinterpret_cast<u32x4*>(dst + (tid + 256 * i) * 4) = wq[i];
  };

  // ---- this wave's channel pair of a chunk -> slab (zero outside the row: the loads' out-of-range value) -------------------
  auto stage_pair = [&](int sbuf, const Seg&) {
    float* const sl = slab0 + sbuf * F_SLAB_BUF;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      *reinterpret_cast<f32x2*>(sl + wofs[r]) = (f32x2){__uint_as_float(xq[0][r >> 2][r & 3]), __uint_as_float(xq[1][r >> 2][r & 3])};
  };

  // ---- first tile: its first chunk and weight stage ------------------------------------------------------------------------
  int tile = blockIdx.x;
  Tile T = locate(tile);
  int par = 0, sbuf = 0, wb = 0;
  Seg S = load_seg(T, 0, 0);
  load_w(S.u, true);
  load_x(S, 0, true);
  store_w(0);
  stage_pair(0, S);
  __syncthreads();

  for (;;) {
    const Tile TN = locate(tile + (int)gridDim.x);
    // ---- epilogue state of this tile (the last K step requests round 0's bias and residual) ----------------------------
    const fh_amp_group* __restrict__ const G = (const fh_amp_group*)uni((const float*)T.G);
    const int len = uni(T.len), t0 = uni(T.t0), nseg = uni(T.nseg);
    const int nres = uni(G->nres);
    const float scale = G->scale;
    const size_t oslab = (size_t)uni(T.bb) * channels * (size_t)len;
    const unsigned slab_bytes = (unsigned)channels * (unsigned)len * 4u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(uni((const float*)G->out) + oslab, slab_bytes);
    const __amdgpu_buffer_rsrc_t rr0 = make_rsrc(nres > 0 ? uni(G->res[0]) + oslab : nullptr, nres > 0 ? slab_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rr1 = make_rsrc(nres > 1 ? uni(G->res[1]) + oslab : nullptr, nres > 1 ? slab_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rr2 = make_rsrc(nres > 2 ? uni(G->res[2]) + oslab : nullptr, nres > 2 ? slab_bytes : 0u);
    const float* const bias = uni(G->bias);
    const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, bias ? (unsigned)channels * 4u : 0u);
    float bv[2][5];
    u32x4 rs[2][5];
    // store items: byte offset of the item's vector in round 0 and whether its column lies inside the row, made once per tile
    // behind the K loop (round m adds 16 m rows); the row test is per round
    unsigned soff0[5];
    int irow[5];
    auto epi_setup = [&]() {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        int it_ = sitem[i];
        asm volatile("" : "+v"(it_));                // (opaque: nothing of this is hoisted into the K loop's registers)
        const int row = it_ >> 16, col = it_ & 0xffff;
        irow[i] = (row < 16 && t0 + col < len) ? row : 0x4000;      // 0x4000: never a valid row
        soff0[i] = ((unsigned)row * (unsigned)len + (unsigned)(t0 + col)) * 4u;
      }
    };
    const unsigned round_bytes = 16u * (unsigned)len * 4u;
    auto epi_request = [&](int m) {                  // bias and first residual of round m's items
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int co = 16 * m + irow[i];
        const bool ok = co < channels;
        bv[m & 1][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rbias, ok ? (unsigned)co * 4u : 0x80000000u, 0, 0));
        if (VEC) rs[m & 1][i] = __builtin_amdgcn_raw_buffer_load_b128(rr0, ok ? soff0[i] + (unsigned)m * round_bytes : 0x80000000u, 0, 0);
      }
    };

    // ---- accumulators: [point][row tile]: rows 4 (lane >> 4) .. + 3 of tile column lane & 15 ------------------------
    f32x4 acc[8][MA];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int m = 0; m < MA; ++m) acc[x][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- K loop: stages (segment, chunk, tap group) -------------------------------------------------------------------
    // One code path for every tap-group count (run-time loop over the groups: with one instantiation per count the
    // accumulators passed through three loops in a row and the compiler kept up to three copies of them)
    for (int sidx = 0; sidx < nseg; ++sidx) {
      const int GC = S.ngrp;
      const bool more_seg = sidx + 1 < nseg;
      const bool more = more_seg || TN.valid;        // there is a chunk behind this segment's last (next segment's / next tile's first)
      const Seg SN = more_seg ? load_seg(T, sidx + 1, par) : load_seg(TN, 0, par ^ 1);
      const int shift = cmax - S.center;             // sample e' of this segment's tiles is slab sample (cmax - center) + e'
      for (int c = 0; c < nch; ++c) {
        const bool last_c = c + 1 == nch;
        const float* const sl = slab0 + sbuf * F_SLAB_BUF + rd_base;
        const bool nx = !last_c || more;
        for (int g = 0; g < GC; ++g) {
          const bool last_g = g + 1 == GC;
          // requests: the next weight stage first, then (first tap group) the next chunk's rows: the wait for the weights at the
          // end of this step must not wait for the rows, which come from HBM and are needed one chunk later (vmcnt is in order)
          const bool nw = !last_g || nx;
          const float* wnext = (last_g && last_c) ? SN.u : S.u + (size_t)(c * GC + g + 1) * f_wstage<MA>();
          load_w(wnext, nw);
          if (g == 0) load_x(last_c ? SN : S, last_c ? 0 : c + 1, nx);
          const float* const wl = wbuf0 + wb * f_wstage<MA>();
          // the tile's 8 samples of this tap group, both channels of the pair (one ds_read_b64 each)
          f32x2 xr[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ee = shift + 4 * g + e, q = (ee * 13) >> 6;            // ee / 5 for ee <= 24 (scalar arithmetic)
            xr[e] = *reinterpret_cast<const f32x2*>(sl + ((ee - 5 * q) * 4 * PL + q) * 2);
          }
          // B^T d for the 8 points (0, 1, -1, 2, -2, 1/2, -1/2, inf), both channels of the pair at once
          const f32x2 x0 = xr[0], x1 = xr[1], x2 = xr[2], x3 = xr[3], x4 = xr[4], x5 = xr[5], x6 = xr[6], x7 = xr[7];
          f32x2 V[8];
          V[0] = __builtin_elementwise_fma((f32x2)(-5.25f), x4, __builtin_elementwise_fma((f32x2)(5.25f), x2, x6)) - x0;
          V[7] = __builtin_elementwise_fma((f32x2)(-5.25f), x5, __builtin_elementwise_fma((f32x2)(5.25f), x3, x7)) - x1;
          {
            const f32x2 e = __builtin_elementwise_fma((f32x2)(-4.25f), x4, x6 + x2);
            const f32x2 o = __builtin_elementwise_fma((f32x2)(-4.25f), x3, x1 + x5);
            V[1] = e + o;
            V[2] = e - o;
          }
          {
            const f32x2 e = __builtin_elementwise_fma((f32x2)(-1.25f), x4, __builtin_elementwise_fma((f32x2)(0.25f), x2, x6));
            const f32x2 o = __builtin_elementwise_fma((f32x2)(2.f), x5, __builtin_elementwise_fma((f32x2)(-2.5f), x3, x1 * 0.5f));
            V[3] = e + o;
            V[4] = e - o;
          }
          {
            const f32x2 e = __builtin_elementwise_fma((f32x2)(-5.f), x4, __builtin_elementwise_fma((f32x2)(4.f), x2, x6));
            const f32x2 o = __builtin_elementwise_fma((f32x2)(0.5f), x5, __builtin_elementwise_fma((f32x2)(-2.5f), x3, x1 * 2.f));
            V[5] = e + o;
            V[6] = e - o;
          }
          // A fragments: [point][lane][row tile pair][2] (+ [point][lane][2] for an odd row tile count), requested one pair of
          // points ahead of the MFMAs that take them (a point's two k-steps accumulate back to back into the same tiles)
          f32x4 a01[2][2];
          f32x2 a2[2][2];
          auto read_a = [&](int pp, int set) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int x = 2 * pp + i;
              if (MA >= 2) a01[set][i] = *reinterpret_cast<const f32x4*>(wl + (x * 64 + lane) * 4);
              if (MA & 1) a2[set][i] = *reinterpret_cast<const f32x2*>(wl + (MA >= 2 ? 2048 : 0) + (x * 64 + lane) * 2);
            }
          };
          read_a(0, 0);
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) {
            __builtin_amdgcn_sched_barrier(0);
            if (F_A_AHEAD && pp < 3) read_a(pp + 1, (pp + 1) & 1);
            if (!F_A_AHEAD && pp > 0) read_a(pp, pp & 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int x = 2 * pp + i;
#pragma unroll
              for (int m = 0; m < MA; ++m)
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                  const float a = (MA >= 2 && m < 2) ? a01[pp & 1][i][2 * m + s_] : a2[pp & 1][i][s_];
                  acc[x][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, V[x][s_], acc[x][m], 0, 0, 0);
                }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (nw) store_w(wb ^ 1);
          if (last_g && nx) stage_pair(sbuf ^ 1, last_c ? SN : S);
          __syncthreads();
          wb ^= 1;
        }
        sbuf ^= 1;
      }
      S = SN;
    }

    // ---- epilogue -------------------------------------------------------------------------------------------------
    // (the last step ended on a barrier; the staging is not the slab: the next tile's first chunk is already there)
    epi_setup();
    epi_request(0);
#pragma unroll
    for (int m = 0; m < MA; ++m) {
      f32x4 y[5];
      {
        const f32x4 s1 = acc[1][m] + acc[2][m], d1 = acc[1][m] - acc[2][m], s2 = acc[3][m] + acc[4][m], d2 = acc[3][m] - acc[4][m],
                    s3 = acc[5][m] + acc[6][m], d3 = acc[5][m] - acc[6][m];
        y[0] = ((acc[0][m] + s1) + s2) + s3;
        y[1] = __builtin_elementwise_fma((f32x4)(0.5f), d3, __builtin_elementwise_fma((f32x4)(2.f), d2, d1));
        y[2] = __builtin_elementwise_fma((f32x4)(0.25f), s3, __builtin_elementwise_fma((f32x4)(4.f), s2, s1));
        y[3] = __builtin_elementwise_fma((f32x4)(0.125f), d3, __builtin_elementwise_fma((f32x4)(8.f), d2, d1));
        y[4] = __builtin_elementwise_fma((f32x4)(0.0625f), s3, __builtin_elementwise_fma((f32x4)(16.f), s2, s1)) + acc[7][m];
      }
      if (m > 0) __syncthreads();                    // the previous round's readers are done with the staging
      if (tile_ok) {
        float* yw = ybuf + (4 * kq) * F_YP + ypos;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 5; ++q) yw[r * F_YP + q * d] = y[q][r];
      }
      // the next round's requests in front of this round's stores (vmcnt counts loads and stores in order)
      if (m + 1 < MA) epi_request(m + 1);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const bool ok = 16 * m + irow[i] < channels;
        const unsigned soff = soff0[i] + (unsigned)m * round_bytes;
        const float b_ = bv[m & 1][i];
        const int yrow = sitem[i] >> 16, ycol = sitem[i] & 0xffff;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(ybuf + (yrow < 16 ? yrow : 0) * F_YP + ycol);
        if (VEC) {
          f32x4 o = {yv[0] + b_, yv[1] + b_, yv[2] + b_, yv[3] + b_};
          const u32x4 r0 = rs[m & 1][i];
          if (nres > 0) o += (f32x4){__uint_as_float(r0[0]), __uint_as_float(r0[1]), __uint_as_float(r0[2]), __uint_as_float(r0[3])};
          if (nres > 1) {            // (the stage-closing group only: requested here)
            const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rr1, ok ? soff : 0x80000000u, 0, 0);
            o += (f32x4){__uint_as_float(t1[0]), __uint_as_float(t1[1]), __uint_as_float(t1[2]), __uint_as_float(t1[3])};
          }
          if (nres > 2) {
            const u32x4 t2 = __builtin_amdgcn_raw_buffer_load_b128(rr2, ok ? soff : 0x80000000u, 0, 0);
            o += (f32x4){__uint_as_float(t2[0]), __uint_as_float(t2[1]), __uint_as_float(t2[2]), __uint_as_float(t2[3])};
          }
          o *= scale;
          const u32x4 ou = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
          __builtin_amdgcn_raw_buffer_store_b128(ou, ro, ok ? soff : 0x80000000u, 0, 0);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned off = (ok && t0 + ycol + q < len) ? soff + 4u * q : 0x80000000u;
            float o = yv[q] + b_;
            if (nres > 0) o += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr0, off, 0, 0));
            if (nres > 1) o += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr1, off, 0, 0));
            if (nres > 2) o += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr2, off, 0, 0));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o * scale), ro, off, 0, 0);
          }
        }
      }
    }
    if (!TN.valid) break;
    T = TN;
    tile += (int)gridDim.x;
    par ^= 1;
  }
}

template <int MA, bool VEC>
int launch_amp(const fh_amp_group* groups, const fh_amp_tile* tiles, int channels, int dilation, int total_tiles, int cmax, hipStream_t stream) {
  static std::atomic<int> blocks_per_launch[FH_MAX_DEVICES];      // 0 = LDS opt-in not done yet; else 2 x the device's CUs
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FH_MAX_DEVICES) {
    fh_set_error("fh_amp_actconv_f32: no current HIP device (or ordinal >= %d)", FH_MAX_DEVICES);
    return FH_E_LAUNCH;
  }
  constexpr int bytes = f_lds_floats<MA>() * 4;
  int resident = blocks_per_launch[dev].load(std::memory_order_acquire);
  if (!resident) {
    hipError_t e = hipFuncSetAttribute((const void*)amp_actconv_kernel<MA, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    int cus = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cus <= 0) {
      fh_set_error("fh_amp_actconv_f32: cannot reserve %d bytes of LDS on device %d: %s", bytes, dev, hipGetErrorString(e));
      return FH_E_LAUNCH;
    }
    resident = 2 * cus;                                            // two 4-wave blocks per CU (LDS, registers)
    blocks_per_launch[dev].store(resident, std::memory_order_release);
  }
  const int grid = total_tiles < resident ? total_tiles : resident;
  hipLaunchKernelGGL((amp_actconv_kernel<MA, VEC>), dim3((unsigned)grid), dim3(F_THREADS), bytes, stream, groups, tiles,
                     channels, dilation, total_tiles, cmax);
  FH_CHECK_LAUNCH("fh_amp_actconv_f32");
  return FH_OK;
}

}  // namespace

extern "C" int fh_sizeof_amp_group(void) { return (int)sizeof(fh_amp_group); }
extern "C" int fh_sizeof_amp_tile(void) { return (int)sizeof(fh_amp_tile); }

extern "C" int fh_amp_tile_len(int dilation) { return dilation >= 1 && dilation <= F_MAX_D ? f_tb(dilation) : -1; }

extern "C" int fh_amp_max_channels(void) { return 48; }

extern "C" int fh_amp_actconv_f32(const fh_amp_group* groups, int n_groups, const fh_amp_tile* tiles, int total_tiles, int channels,
                                  int dilation, int max_center, int flags, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && tiles && total_tiles > 0, "fh_amp_actconv_f32: bad sizes");
  FH_CHECK_ARG(channels >= 8 && channels <= 48 && channels % 8 == 0, "fh_amp_actconv_f32: %d channels (8 .. 48, a multiple of 8)", channels);
  FH_CHECK_ARG(dilation >= 1 && dilation <= F_MAX_D, "fh_amp_actconv_f32: dilation %d (1 .. %d)", dilation, F_MAX_D);
  FH_CHECK_ARG(max_center >= 0 && max_center <= 5, "fh_amp_actconv_f32: max_center %d (0 .. 5: kernels of at most 11 taps)", max_center);
  FH_CHECK_ARG(flags == 2 || flags == 3, "fh_amp_actconv_f32: flags %d (bit 0: rows 16-byte aligned; bit 1 must be set: the form with the "
               "Activation1d inside the launch left the library in round 6)", flags);
  const int ma = (channels + 15) / 16;
  const bool vec = flags & 1;
  hipStream_t st = (hipStream_t)stream;
#define FH_AMP_CASE(MA)                                                                                   \
  case MA:                                                                                                \
    return vec ? launch_amp<MA, true>(groups, tiles, channels, dilation, total_tiles, max_center, st)     \
               : launch_amp<MA, false>(groups, tiles, channels, dilation, total_tiles, max_center, st);
  switch (ma) {
    FH_AMP_CASE(1)
    FH_AMP_CASE(2)
    FH_AMP_CASE(3)
  }
#undef FH_AMP_CASE
  return FH_E_ARG;
}
