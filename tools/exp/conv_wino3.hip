// Winograd F(4,3) conv, third block shape: a PERSISTENT 12-wave workgroup per CU made of TWO INDEPENDENT 6-wave teams.
//
// Same maths, weights, descriptors and bits as conv_wino.hip (the AMPBlock convs of
// /root/reference/src/flowhigh/models/bigvgan/models.py:36-72).  Why another shape (tools/wino_trace2.py, round 3):
// a 12-wave block of conv_wino.hip spends ~3 us before and ~8-10 us after its K loop with the matrix pipes of its CU
// idle -- the epilogue moves 256 KB of residuals and outputs through the CU's memory pipe, and neither fewer LDS
// instructions (16-byte exchange: 0 %) nor fewer vector instructions change that; one such block per CU (168 VGPRs x 12
// waves) leaves nothing else to run meanwhile: 10.7 % of all conv time at batch 1 (35 % at C = 48, 22 % at C = 96).
// A kernel with the six transform points in one wave (conv_wino2.hip, no exchange, three independent 4-wave blocks
// per CU) overlaps all of it but needs 1.45 x the non-matrix instructions per MFMA and loses 8 % in the K loop.
//
// Here the 12 waves of a workgroup (3 per SIMD, as before) are two teams of 6 (team = wave / 6, transform point xi =
// wave % 6; every SIMD holds waves of both teams).  A team works on its OWN 64 co x 64 tiles (256 outputs) tile with the
// wave tile, slab layout, K loop and arithmetic of conv_wino.hip's <2, 2> shape (two tile columns 32 tiles apart), its
// own slab buffers, exchange tiles and -- instead of s_barrier -- its own barrier (an LDS counter the team's waves
// spin on).  The teams take their tiles from the two ENDS of their XCD's work list (heavy tap counts at the front,
// light ones at the back), so they are out of step by construction: while one team loads, exchanges and stores, the
// other one's MFMAs keep the SIMDs' matrix pipes busy.  The workgroups are persistent (one per CU); a team that finds
// its XCD's list empty takes tiles from the other XCDs' lists (tails).
//
// Work list = conv_wino.hip's block order for the 64 x 256 tile: slot s of XCD x is what block 8 s + x of that
// kernel would do (runs of a weight panel stay on one XCD).  The two cursors of a list live in one 64-bit word
// (front count | back count << 32) behind the descriptor array: the caller provides FH_WINO3_WS_BYTES zeroed bytes
// there; the last team to leave a launch zeroes them again.
// Vector loads only (16-byte aligned contiguous rows); other launches run conv_wino.hip's 64 x 256 tile (same bits).
#include "fh_common.h"

#include <algorithm>
#include <type_traits>

#include "conv_wino_int.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int T_CK = 16;                 // input channels per chunk
constexpr int T_THREADS = 768;           // 12 waves = 2 teams x 6 transform points
constexpr int T_TEAM = 384;              // threads of a team
constexpr int T_BM = 64;                 // output channels per tile
constexpr int T_BT = 64;                 // F(4,3) tiles per team tile (256 outputs)
constexpr int T_P = T_BT + 8;            // plane pitch (samples)
constexpr int T_RP2 = 8 * T_P;           // floats of one channel PAIR (4 planes x T_P x 2 channels interleaved)
constexpr int T_SLAB = (T_CK / 2) * T_RP2;            // floats per slab buffer (18 KB)
constexpr int T_XQ = T_BT + 5;           // absolute quads a slab can touch
constexpr int T_NITEM = (T_XQ + 47) / 48;             // staging items per thread (48 quads per channel pair and item): 2
constexpr int T_EP = 36;                 // pitch (floats) of a column of the exchange tiles (conv_wino.hip: W_EP)
constexpr int T_EPI = 6 * 32 * T_EP;     // exchange tiles of a team, floats
constexpr int T_TEAM_LDS = 2 * T_SLAB + T_EPI;        // floats of LDS per team (64.5 KB)
constexpr int T_CTRL = 16;               // ints of control words in front (barrier counters, next-item mailboxes)

// rows of B^T: the canonical arithmetic of conv_wino.hip (p = fma(a, x[i], x[j]); q = fma(b, x[k], x[l]); v = fma(c, q, p))
__device__ const int kBt3Off[6][4] = {{2, 4, 0, 0}, {2, 4, 1, 3}, {2, 4, 1, 3}, {2, 4, 1, 3}, {2, 4, 1, 3}, {3, 5, 1, 1}};
__device__ const float kBt3Coef[6][3] = {{-5.f, 0.f, 4.f}, {-4.f, -4.f, 1.f}, {-4.f, -4.f, -1.f},
                                         {-1.f, -1.f, 2.f}, {-1.f, -1.f, -2.f}, {-5.f, 0.f, 4.f}};

struct TSeg {
  const float* x;
  const float* u;
  int cin, ngrp, center;
};
__device__ __forceinline__ TSeg load_tseg(const fh_wino_seg* S) {
  TSeg w;
  w.x = uni(S->x);
  w.u = uni(S->u);
  w.cin = uni(S->cin);
  w.ngrp = uni(S->ngrp);
  w.center = uni(S->center);
  return w;
}

// Barrier of ONE team: every wave adds 1 to the team's LDS counter and waits until all 6 have (the counter only
// grows: `target` is the value after this round).  LDS operations of a wave complete in order, so the release is the
// wave's own s_waitcnt before the add; the other team never touches this counter or this team's LDS regions.
__device__ __forceinline__ void team_sync(int* ctr, int& target) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  target += 6;
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - target < 0)
    __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ __attribute__((amdgpu_flat_work_group_size(T_THREADS, T_THREADS), amdgpu_waves_per_eu(3, 3)))
void conv_wino3_kernel(const fh_wino_group* __restrict__ groups, int n_groups, int batch, int co_tiles,
                       int n_tiles, int run_len, int dil, int pm, const int* __restrict__ run_map, int n_runs,
                       unsigned long long* __restrict__ ws, int slots_per_xcd) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];      // T_CTRL ints + 2 x T_TEAM_LDS floats
  int* const ctrl = reinterpret_cast<int*>(lds_all);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave / 6;                  // 0: takes tiles from the front of a list, 1: from the back
  const int xi = wave - 6 * team;             // transform point of this wave
  const int tt = tid - T_TEAM * team;         // thread in team
  const int l31 = lane & 31, lh = lane >> 5;
  float* const lds = lds_all + T_CTRL + team * T_TEAM_LDS;
  float* const E = lds + 2 * T_SLAB;
  int* const bar = ctrl + team;               // team barrier counter
  int* const mail = ctrl + 4 + 2 * team;      // next work item of the team: (xcd, slot) or (-1, -1)
  int bar_target = 0;
  if (tid < T_CTRL) ctrl[tid] = 0;
  // Matrix-pipe arbitration between the waves of a SIMD is strict (priority, then age), and 6 waves sit 2 / 2 / 1 / 1
  // on the 4 SIMDs: left alone, the older team's PAIRS take their SIMDs' pipes completely, its lone waves idle half
  // the time at the team barrier and the other team starves behind both (measured: -25 % at C = 768).  The lone wave
  // of a team on its SIMD therefore runs at raised priority: per chunk it takes its share W first and leaves 2 W to
  // the other team's pair -- every SIMD serves 3 W per chunk of both teams.
  {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int simd = (int)((hw >> 4) & 3u);
    if (lane == 0) reinterpret_cast<int*>(lds_all + T_CTRL)[wave] = simd;
    __syncthreads();                          // (the only block-wide barriers)
    int mates = 0;
    for (int w = 6 * team; w < 6 * team + 6; ++w) mates += reinterpret_cast<const int*>(lds_all + T_CTRL)[w] == simd;
    __syncthreads();
    if (uni(mates) == 1) __builtin_amdgcn_s_setprio(2);
  }

  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int home = (int)(xcc & 7u);

  const int panels = n_groups * batch * co_tiles;
  const int runs_per_panel = (n_tiles + run_len - 1) / run_len;
  const int total_runs = panels * runs_per_panel;

  const int bo0 = kBt3Off[xi][0], bo1 = kBt3Off[xi][1], bo2 = kBt3Off[xi][2], bo3 = kBt3Off[xi][3];
  const float bc0 = kBt3Coef[xi][0], bc1 = kBt3Coef[xi][1], bc2 = kBt3Coef[xi][2];
  const f32x2 c0 = {bc0, bc0}, c1 = {bc1, bc1}, c2 = {bc2, bc2};

  // Taking the next slot of a list: front (team 0) or back (team 1) of the home XCD's list; when that is empty, of the
  // others (tails).  One lane of the team's first wave.  The request for the home list is issued at the START of a
  // tile and its answer read at the END (the atomic's round trip is covered by the tile); the result goes through the
  // team's mailbox as (xcd, slot), (-1, -1) when every list is empty.
  const unsigned long long inc = team ? (1ull << 32) : 1ull;
  auto decode = [&](unsigned long long old, int x, int& rx, int& rs) {
    const long long f = (long long)(old & 0xffffffffull), bk = (long long)(old >> 32);
    if (f + bk < slots_per_xcd) {
      rx = x;
      rs = team ? slots_per_xcd - 1 - (int)bk : (int)f;
    }
  };
  auto resolve = [&](unsigned long long old_home) {
    int rx = -1, rs = -1;
    decode(old_home, home, rx, rs);
    for (int k = 1; k < 8 && rx < 0; ++k) {
      const int x = (home + k) & 7;
      decode(__hip_atomic_fetch_add(ws + x, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), x, rx, rs);
    }
    mail[0] = rx;
    mail[1] = rs;
  };
  const bool leader = xi == 0 && lane == 0;
  unsigned long long pending = 0;

  if (leader) resolve(__hip_atomic_fetch_add(ws + home, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  team_sync(bar, bar_target);

  for (;;) {
    const int ix = uni(mail[0]), is = uni(mail[1]);
    team_sync(bar, bar_target);               // (everyone has read the mailbox before it is rewritten)
    if (ix < 0) break;
    if (leader) pending = __hip_atomic_fetch_add(ws + home, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- slot -> (panel, n block): block 8 * slot + xcd of conv_wino.hip's 64 x 256 tile ---------------------------
    int run = (is / run_len) * 8 + ix;
    bool valid = true;
    if (run_map) {
      valid = run < n_runs;
      run = valid ? uni(run_map[run]) : 0;
    }
    valid = valid && run < total_runs;
    const int panel = uni(run / runs_per_panel);
    const int ntile = uni((run % runs_per_panel) * run_len + (is % run_len));
    valid = valid && ntile < n_tiles;
    const int cot = uni(panel % co_tiles);
    const int gb = uni(panel / co_tiles);
    const int b = uni(gb % batch);
    const fh_wino_group* __restrict__ G = groups + (valid ? uni(gb / batch) : 0);
    const int ph = uni(ntile % dil);            // phase of the decimated sequence
    const int tb = uni(ntile / dil);            // 256-output block within the phase
    const int co0 = cot * T_BM;
    const int len = uni(G->len), cout_pad = uni(G->cout_pad), nseg = uni(G->nseg);
    valid = valid && tb * (4 * T_BT) * dil + ph < len;
    if (!valid) {                               // padding slot of the list: on to the next item
      if (leader) resolve(pending);
      team_sync(bar, bar_target);
      continue;
    }
    const int lp = ((len + dil - 1) / dil + 3) & ~3;
    const int pitch = pm ? dil * lp : len;             // floats per (batch, channel) row, inputs and outputs

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- loaders (conv_wino.hip's vector form on 384 threads) ----------------------------------------------------
    // thread (pair = tt / 48, vt = tt % 48) handles absolute quads q0 + vt (+ 48): samples 4 Q .. 4 Q + 3 of its two
    // channels = two aligned 16-byte loads, then 4 ds_write_b64 (channel pair) into the 4 planes
    const int vp = tt / 48, vt = tt % 48;
    u32x4 xq[2][2];                                      // [item slot][channel of the pair]
    auto vl_load = [&](const TSeg& S, int chunk, bool ok_seg, int item, int slot) {
      const __amdgpu_buffer_rsrc_t r =
          make_rsrc(uni(S.x + (size_t)b * S.cin * pitch), ok_seg ? (unsigned)(S.cin * pitch) * 4u : 0u);
      const int ub = tb * (4 * T_BT) - S.center;                           // first decimated index needed (>= -5)
      const int q0 = (ub - (ub & 3)) >> 2;                                 // floor(ub / 4)
      const int rowlen = pm ? lp : len;
      const int q = vt + 48 * item, qa = q0 + q;
      const bool ok = q < T_XQ && qa >= 0 && 4 * qa < rowlen;              // (outside the row: zero padding)
      const int e0 = (chunk * T_CK + 2 * vp) * pitch + (pm ? ph * lp : 0) + 4 * qa;
      xq[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)e0 * 4u : 0x80000000u, 0, 0);
      xq[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (unsigned)(e0 + pitch) * 4u : 0x80000000u, 0, 0);
    };
    auto vl_store = [&](const TSeg& S, int buf, int item, int slot) {
      const int ub = tb * (4 * T_BT) - S.center;
      const int ua = ub - (ub & 3);                                        // decimated index of slab sample 0
      const int q = vt + 48 * item;
      const int nvalid = pm ? (len - ph + dil - 1) / dil : len;            // samples of this phase / row
      if (ua + 4 * T_XQ > nvalid) {                                        // last block of the row: zero past the end
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ua + 4 * q + e >= nvalid) { xq[slot][0][e] = 0u; xq[slot][1][e] = 0u; }
      }
      if (q < T_XQ) {
        float* dst = lds + buf * T_SLAB + vp * T_RP2 + 2 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<f32x2*>(dst + e * T_P * 2) = (f32x2){__uint_as_float(xq[slot][0][e]), __uint_as_float(xq[slot][1][e])};
      }
    };
    // A fragments of one step, [mt][half]: half h holds k-steps 4h .. 4h+3
    u32x4 areg[2][2];
    const int a_lane = (l31 * T_CK + lh * 8) * 4;
    auto load_a_half = [&](int h, const TSeg& S, int chunk, int g, bool ok_seg) {
      const float* up = uni(S.u + ((size_t)((chunk * S.ngrp + g) * 6 + xi) * cout_pad + co0) * T_CK);
      const __amdgpu_buffer_rsrc_t r = make_rsrc(up, ok_seg ? T_BM * T_CK * 4 : 0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        areg[mt][h] = __builtin_amdgcn_raw_buffer_load_b128(r, a_lane + mt * 32 * T_CK * 4 + 16 * h, 0, 0);
    };
    // L2 warm-up of the A tiles of the NEXT chunk (all its tap groups, this wave's xi), as in conv_wino.hip
    unsigned pf = 0;
    auto prefetch_a = [&](const TSeg& S, int chunk, bool ok_seg) {
      const float* up = uni(S.u + ((size_t)(chunk * S.ngrp * 6 + xi) * cout_pad + co0) * T_CK);
      const unsigned gstride = 6u * (unsigned)cout_pad * T_CK * 4u;          // bytes between tap groups
      const __amdgpu_buffer_rsrc_t r = make_rsrc(up, ok_seg ? (unsigned)(S.ngrp - 1) * gstride + T_BM * T_CK * 4 : 0u);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned off = (unsigned)(2 * j + lh) * gstride + (unsigned)l31 * 128u;
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(pf) : "v"(off), "s"(r) : "memory");
      }
    };

    // ---- prologue ------------------------------------------------------------------------------------------------
    TSeg S0 = load_tseg(&G->seg[0]);
    load_a_half(0, S0, 0, 0, true);
    load_a_half(1, S0, 0, 0, true);
    int xbuf = 0;
#pragma unroll
    for (int item = 0; item < T_NITEM; ++item) {
      vl_load(S0, 0, true, item, 0);
      vl_store(S0, 0, item, 0);
    }
    team_sync(bar, bar_target);

    // The K loop of a chunk is a flat sequence of k-step PAIRS (conv_wino.hip): pair p = 4 g + kp, 8 MFMAs each; one
    // 16-byte-per-lane LDS instruction fetches one B^T sample for the wave's 2 tile columns (32 tiles = 64 floats
    // apart) x 2 k-steps; the samples of pair p + 1 are requested before the MFMAs of pair p.
    auto run_segment = [&](auto gc, const TSeg& S, const TSeg& Sn, bool more_seg) {
      constexpr int GC = decltype(gc)::value;
      const int nch = S.cin / T_CK;
      for (int c = 0; c < nch; ++c) {
        const bool last_chunk = c == nch - 1;
        const bool has_next = !last_chunk || more_seg;
        const TSeg& Sx = last_chunk ? Sn : S;              // owner of the next chunk
        const int cx = last_chunk ? 0 : c + 1;
        const float* xsb = lds + xbuf * T_SLAB + lh * 4 * T_RP2 + l31 * 2;
        f32x2 xr[2][4][2];                                 // [slot][sample][column] = (k-step 2 kp, 2 kp + 1)
        const int sh = (tb * (4 * T_BT) - S.center) & 3;   // slab starts `sh` samples before the first tap
        auto fetch = [&](int slot, int p) {
          const int j0 = 3 * (p >> 2) + sh, kp = p & 3;
          const int o[4] = {bo0, bo1, bo2, bo3};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* q = xsb + (((j0 + o[r]) & 3) * T_P + ((j0 + o[r]) >> 2)) * 2 + kp * T_RP2;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) xr[slot][r][nt] = *reinterpret_cast<const f32x2*>(q + 64 * nt);
          }
        };
        auto stage = [&](int p) {
          if (p == 0) {
            vl_load(Sx, cx, has_next, 0, 0);
            if (GC == 1) vl_load(Sx, cx, has_next, 1, 1);
            prefetch_a(Sx, cx, has_next);
          }
          if (GC > 1 && p == 4) {
            if (has_next) vl_store(Sx, xbuf ^ 1, 0, 0);
            vl_load(Sx, cx, has_next, 1, 0);
          }
        };
        fetch(0, 0);
#pragma unroll
        for (int p = 0; p < 4 * GC; ++p) {
          const int g = p >> 2, kp = p & 3, h = kp >> 1;
          stage(p);
          if (p + 1 < 4 * GC) fetch((p + 1) & 1, p + 1);
          f32x2 bf[2];                               // [column] = B values of k-steps 2 kp, 2 kp + 1
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            f32x2 qq;
            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(bf[nt]) : "s"(c0), "v"(xr[p & 1][0][nt]), "v"(xr[p & 1][1][nt]));
            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(qq) : "s"(c1), "v"(xr[p & 1][2][nt]), "v"(xr[p & 1][3][nt]));
            if (nt == 0)
              asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(bf[nt]) : "s"(c2), "v"(qq));
            else
              asm("v_pk_fma_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(bf[nt]) : "s"(c2), "v"(qq));
          }
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const int e = 2 * (kp & 1) + k2;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(areg[mt][h][e]), bf[nt][k2],
                                                                   acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (kp & 1) {                              // this half of the A registers is free: refill it for the next step
            const bool same_chunk = g + 1 < GC;
            const TSeg& Sa = same_chunk ? S : Sx;
            load_a_half(h, Sa, same_chunk ? c : cx, same_chunk ? g + 1 : 0, same_chunk || has_next);
          }
        }
        if (has_next) {
          if (GC == 1) vl_store(Sx, xbuf ^ 1, 0, 0);
          vl_store(Sx, xbuf ^ 1, 1, GC == 1 ? 1 : 0);
          team_sync(bar, bar_target);                      // one team barrier per chunk
          xbuf ^= 1;
        }
      }
    };

    // Segments are sorted by tap-group count, descending (host: make_wino_group)
    int sg = 0;
    auto run_all = [&](auto gc) {
      while (sg < nseg && S0.ngrp == decltype(gc)::value) {
        const bool more_seg = sg + 1 < nseg;
        const TSeg Sn = load_tseg(&G->seg[more_seg ? sg + 1 : sg]);
        run_segment(gc, S0, Sn, more_seg);
        S0 = Sn;
        ++sg;
      }
    };
    run_all(std::integral_constant<int, 4>{});
    run_all(std::integral_constant<int, 3>{});
    run_all(std::integral_constant<int, 2>{});
    run_all(std::integral_constant<int, 1>{});

    // ---- epilogue: exchange M_xi through the team's LDS tiles, y = A^T M, bias + residuals, scale, store --------------
    // (conv_wino.hip's epilogue on one team: column-major tiles E[xi][col 32][T_EP], four ds_write_b128 per lane and
    // tile, thread t < 256 of the team = (row quad rq, column col) reads six ds_read_b128 and makes 4 rows x 4 outputs)
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
    const int erq = (tt >> 5) & 7, ecol = tt & 31;
    const bool eact = tt < 256;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        team_sync(bar, bar_target);
        const int corow = co0 + mt * 32 + 4 * erq;                        // + i
        const int v0 = tb * (4 * T_BT) + (nt * 32 + ecol) * 4;            // decimated index of y[0]
        const bool colok = eact && (v0 + 3) * dil + ph < len;
        const unsigned coloff = (pm ? (unsigned)(ph * lp) : 0u) + (unsigned)v0;
        float bpre[4];
        u32x4 rpre[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          bpre[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
              rbias, (eact && corow + i < cout) ? (unsigned)(corow + i) * 4u : 0x80000000u, 0, 0));
        if (vec && nres > 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool ok = colok && corow + i < cout;
            rpre[i] = __builtin_amdgcn_raw_buffer_load_b128(
                rr0, ok ? ((unsigned)(corow + i) * (unsigned)opitch + coloff) * 4u : 0x80000000u, 0, 0);
          }
        }
        {
          float* ew = E + (xi * 32 + l31) * T_EP + 4 * lh;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(ew + 8 * q) =
                (f32x4){acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
        }
        team_sync(bar, bar_target);
        if (eact) {
          const float* er = E + ecol * T_EP + 4 * erq;
          const f32x4 m0 = *reinterpret_cast<const f32x4*>(er), m1 = *reinterpret_cast<const f32x4*>(er + 32 * T_EP),
                      m2 = *reinterpret_cast<const f32x4*>(er + 64 * T_EP), m3 = *reinterpret_cast<const f32x4*>(er + 96 * T_EP),
                      m4 = *reinterpret_cast<const f32x4*>(er + 128 * T_EP), m5 = *reinterpret_cast<const f32x4*>(er + 160 * T_EP);
          const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          const f32x4 y0 = m0 + s12 + s34;
          const f32x4 y1 = __builtin_elementwise_fma((f32x4)(2.f), d34, d12);
          const f32x4 y2 = __builtin_elementwise_fma((f32x4)(4.f), s34, s12);
          const f32x4 y3 = __builtin_elementwise_fma((f32x4)(8.f), d34, d12) + m5;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int co = corow + i;
            const bool rowok = co < cout;
            const float bv = bpre[i];
            const unsigned rowoff = (unsigned)co * (unsigned)opitch + (pm ? (unsigned)(ph * lp) : 0u);
            const float y[4] = {y0[i], y1[i], y2[i], y3[i]};
            if (vec && colok) {
              const unsigned off = rowok ? (rowoff + (unsigned)v0) * 4u : 0x80000000u;
              f32x4 o = {y[0] + bv, y[1] + bv, y[2] + bv, y[3] + bv};
              if (nres > 0) {
                u32x4 t = rpre[i];
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
              for (int q = 0; q < 4; ++q) {
                const int n = ph + dil * (v0 + q);
                const unsigned off = (rowok && n < len) ? (rowoff + (unsigned)(pm ? v0 + q : n * ostride + ophase)) * 4u
                                                        : 0x80000000u;
                float o = y[q] + bv;
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
      }
    }
    if (pf == 0x7fc12345u) ctrl[8] = 0;      // keeps pf alive; never true for weights
    if (leader) resolve(pending);            // the next item into the mailbox
    team_sync(bar, bar_target);
  }

  // ---- leave: the last team of the launch zeroes the cursors for the next launch on these descriptors -------------
  if (leader) {
    const unsigned long long done = __hip_atomic_fetch_add(ws + 8, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == 2ull * gridDim.x - 1ull) {
      for (int k = 0; k < 9; ++k) __hip_atomic_store(ws + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

int fh_wino3_launch(const fh_wino_group* groups, int n_groups, int batch, int cout_pad, int len, int dilation,
                    int phase_major, hipStream_t stream, const int* run_map, int n_runs) {
  FH_CHECK_ARG(cout_pad > 0 && cout_pad % T_BM == 0, "fh_conv_wino_f32: cout_pad %d not a multiple of %d", cout_pad, T_BM);
  const int co_tiles = cout_pad / T_BM;
  const int n_tiles = fh_cdiv(fh_cdiv(len, dilation), 4 * T_BT) * dilation;
  const long long panels = (long long)n_groups * batch * co_tiles;
  const int run_len = fh_cdiv(n_tiles, fh_cdiv(n_tiles, FH_WINO_RUN));
  const long long runs = run_map ? (long long)n_runs : panels * fh_cdiv(n_tiles, run_len);
  const long long slots_per_xcd = (long long)fh_cdiv(runs, 8) * run_len;
  FH_CHECK_ARG(slots_per_xcd > 0 && slots_per_xcd < (1ll << 28), "fh_conv_wino_f32: grid too large");
  static std::atomic<int> n_cus[FH_MAX_DEVICES];
  static std::atomic<bool> lds_opt_in[FH_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FH_MAX_DEVICES) {
    fh_set_error("fh_conv_wino_f32: no current HIP device (or ordinal >= %d)", FH_MAX_DEVICES);
    return FH_E_LAUNCH;
  }
  const int lds_bytes = (T_CTRL + 2 * T_TEAM_LDS) * 4;
  if (!lds_opt_in[dev].load(std::memory_order_acquire)) {
    int cus = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)conv_wino3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess || cus <= 0) {
      fh_set_error("fh_conv_wino_f32: cannot set up the persistent kernel on device %d: %s", dev, hipGetErrorString(e));
      return FH_E_LAUNCH;
    }
    n_cus[dev].store(cus, std::memory_order_relaxed);
    lds_opt_in[dev].store(true, std::memory_order_release);
  }
  // one workgroup per CU; fewer when there are fewer tiles than teams
  const long long wgs = std::min<long long>(n_cus[dev].load(std::memory_order_relaxed), (8 * slots_per_xcd + 1) / 2);
  unsigned long long* ws = (unsigned long long*)(groups + n_groups);       // FH_WINO3_WS_BYTES behind the descriptors
  FH_CHECK_ARG((((size_t)ws) & 7) == 0, "fh_conv_wino_f32: descriptor array not 8-byte aligned");
  hipLaunchKernelGGL(conv_wino3_kernel, dim3((unsigned)wgs), dim3(T_THREADS), lds_bytes, stream, groups, n_groups, batch,
                     co_tiles, n_tiles, run_len, dilation, phase_major, run_map, n_runs, ws, (int)slots_per_xcd);
  FH_CHECK_LAUNCH("fh_conv_wino_f32");
  return FH_OK;
}
