// Narrow stages of BigVGAN (C <= 48) in the bf16 x 6 form: the residual-stack convs as a DIRECT implicit GEMM on
// v_mfma_f32_16x16x32_bf16 (conv_form = 'bf16x6'; the fp32-MFMA Winograd form of the same launches is amp_fused.hip).
//
// Replaces, per launch, the `xt = conv(xt)` half of one `xt = act(x); xt = conv(xt)` pair of the AMP blocks
// (/root/reference/src/flowhigh/models/bigvgan/models.py:63-72 AMPBlock1, :108-117 AMPBlock2: "same"-padded Conv1d, k = 3 / 7 / 11,
//  dilation 1 / 3 / 5), including "+ x" (:70) and the "xs / num_kernels" average over the blocks (:181-187, K segments of one group).
//
// Why not Winograd here: at 24 / 48 channels the F(5,4) form spends more vector instructions on B^T d, A^T m and the phase
// re-interleave than its matrix work is worth (amp_fused.hip: 3.7 vector + 3.8 scalar instructions per MFMA, matrix pipe 0.38-0.48
// busy), and a bf16 x 6 Winograd loop would have to split every transformed value (8 points x channels x tap groups).  The direct
// form splits every input sample ONCE, when it enters LDS, and its K loop is ds_read_b128 + MFMA with no vector arithmetic:
//   * x = h + m + l, h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (exact); a product a b is the six MFMAs h h, h m, m h, h l,
//     l h, m m with fp32 accumulation (dropped terms <= 2^-24 |a b|), weights split on the host (packing.pack_narrow_bf_weight);
//   * block = 4 waves = 256 outputs of every channel; wave = 4 M tiles of 16 outputs x ceil(C / 16) N tiles of 16 channels
//     (M = time: a lane's 4 accumulator registers are 4 consecutive samples of one channel, stored as one 16-byte vector with bias,
//     residuals and scale straight from the registers: no staging, no barrier in the epilogue);
//   * K = (tap, input channel): the slab holds the block's samples + halo of up to 24 channels (3 octets) as three piece planes of
//     16-byte units [piece][octet][sample] (8 channels x bf16), so that the A fragment of (tap, octet) for 16 consecutive outputs is
//     16 consecutive units: one conflict-free ds_read_b128 per piece, any dilation, any tap; a k-block of 32 = four (tap, octet)
//     pairs, one per 16-lane group; 48 channels = two slabs after each other;
//   * weights (B fragments, [k-block][N tile][piece][lane][8 bf16]) come straight from L2 / the CU's L1 one k-block ahead: the whole
//     set is 18-110 KB per conv, the same for every block of the launch, and no wave waits for another inside the K loop.
// A sample's arithmetic depends on its absolute position only (fixed K order), so a clip gives the same bits alone, in a batch, in
// a ragged launch and in time chunks.
#include "fh_common.h"

namespace {

typedef __bf16 nb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 nb_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int NB_THREADS = 256;              // 4 waves
constexpr int NB_NM = 4;                     // M tiles (16 outputs) per wave
constexpr int NB_OUT = 64 * NB_NM;           // outputs per tile and row: 256
constexpr int NB_NSP = NB_OUT + 64;          // samples (16-byte units) per (piece, octet) row of the slab: 256 + 32 + 30 <= 320
constexpr int NB_OG = 3;                     // octets (8 channels) per slab
constexpr int NB_PLANE = NB_OG * NB_NSP;     // units of a piece plane
constexpr int NB_MAX_D = 6;                  // (taps: at most 11, center at most 5 -- NB_NSP; checked where the descriptors are made)

__device__ __forceinline__ unsigned nb_pack(float a, float b) {              // v_cvt_pk_bf16_f32 (round to nearest even)
  const nb_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float nb_lo(unsigned p) { return __uint_as_float(__builtin_amdgcn_perm(0u, p, 0x01000c0cu)); }
__device__ __forceinline__ float nb_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// 8 values -> three 16-byte pieces (8 bf16 each)
__device__ __forceinline__ void nb_split8(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    h[i] = nb_pack(a, b);
    const float ra = a - nb_lo(h[i]), rb = b - nb_hi(h[i]);
    m[i] = nb_pack(ra, rb);
    l[i] = nb_pack(ra - nb_lo(m[i]), rb - nb_hi(m[i]));
  }
}

// MA: N tiles (16 output channels each).  VEC: every row of every group is 16-byte aligned (len % 4 == 0).
template <int MA, bool VEC>
__global__ __launch_bounds__(NB_THREADS) __attribute__((amdgpu_waves_per_eu(2)))
void narrow_bf_kernel(const fh_amp_group* __restrict__ groups, const fh_amp_tile* __restrict__ tiles, int channels, int d, int total_tiles) {
  __shared__ __attribute__((aligned(16))) u32x4 slab[3 * NB_PLANE];

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  d = uni(d);
  const int oct = channels >> 3;
  const int ncg = (oct + NB_OG - 1) / NB_OG;                      // slabs (channel groups) per segment: 1 (<= 24 channels) or 2
  const int og_lo = oct / ncg, og_rem = oct - og_lo * ncg;        // octets per group: og_lo + (group < og_rem)

  for (int tile = blockIdx.x; tile < total_tiles; tile += (int)gridDim.x) {
    const fh_amp_tile* const e = tiles + tile;
    const fh_amp_group* __restrict__ const G = groups + uni(e->group);
    const int bb = uni(e->batch_item), t0 = uni(e->t0), len = uni(e->len);
    const int nseg = uni(G->nseg), nres = uni(G->nres);
    const size_t oslab = (size_t)bb * channels * (size_t)len;
    const unsigned slab_bytes = (unsigned)channels * (unsigned)len * 4u;

    // the first residual is requested HERE, in front of the slab's rows (VEC): both streams are in flight together and the wait
    // for the rows covers it (vmcnt retires in order); lane = (channel 16 na + li, samples t0 + 64 wv + 16 mt + 4 lg .. + 3)
    const __amdgpu_buffer_rsrc_t rr0 = make_rsrc(nres > 0 ? uni(G->res[0]) + oslab : nullptr, nres > 0 ? slab_bytes : 0u);
    u32x4 rs[NB_NM][MA];
    if (VEC) {
#pragma unroll
      for (int na = 0; na < MA; ++na)
#pragma unroll
        for (int mt = 0; mt < NB_NM; ++mt) {
          const int ch = 16 * na + li, t = t0 + 16 * NB_NM * wv + 16 * mt + 4 * lg;
          rs[mt][na] = __builtin_amdgcn_raw_buffer_load_b128(rr0, (ch < channels && t < len) ? (unsigned)(ch * len + t) * 4u : 0x80000000u, 0, 0);
        }
    }

    f32x4 acc[NB_NM][MA];
#pragma unroll
    for (int mt = 0; mt < NB_NM; ++mt)
#pragma unroll
      for (int na = 0; na < MA; ++na) acc[mt][na] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int sidx = 0; sidx < nseg; ++sidx) {
      const fh_amp_seg* const P = &G->seg[sidx];
      const float* const xp = uni(P->x);
      const float* const up = uni(P->u);
      const int k = uni(P->ngrp);                                  // taps (this entry point: fh_amp_seg.ngrp holds k)
      const int H = uni(P->center) * d;                            // slab sample s is t = t0 - H + s
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(xp + oslab, slab_bytes);
      int wbase = 0;                                               // 1 KB units into this segment's weights
      int ob = 0;
      for (int cg = 0; cg < ncg; ++cg) {
        const int og = og_lo + (cg < og_rem ? 1 : 0);
        const int kog = k * og, nb = (kog + 3) >> 2;
        const unsigned wbytes = (unsigned)(nb * MA * 3) * 1024u;
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(reinterpret_cast<const char*>(up) + (size_t)wbase * 1024u, wbytes);
        // ---- weights of the first k-block: requested before the slab (they are in L2; the rows may come from HBM) ------------
        u32x4 bw[2][MA][3];
        auto load_b = [&](int set, int kb) {
#pragma unroll
          for (int na = 0; na < MA; ++na)
#pragma unroll
            for (int p = 0; p < 3; ++p)
              bw[set][na][p] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)lane * 16u, (unsigned)((kb * MA + na) * 3 + p) * 1024u, 0);
        };
        load_b(0, 0);
        // ---- slab: the samples of the group's octets, split, as [piece][octet][sample]; slab sample s is t = t0 - Ha + s ----------
        __syncthreads();                                           // the previous K loop's readers are done
        // VEC: Ha = H rounded up to 4 (16-byte loads; the taps then start Ha - H samples into the slab); else Ha = H
        const int Ha = VEC ? (H + 3) & ~3 : H;
        const int ns = NB_OUT + Ha + H;                            // samples the K loop reads
        if (VEC) {
          // item = (octet, 4 consecutive samples): 8 x 16 bytes from the octet's 8 rows (a wave-load is 1 KB of one row)
          const int o = tid / (NB_NSP / 4), qd = tid - o * (NB_NSP / 4);
          if (o < og) {
            const int t = t0 - Ha + 4 * qd;
            const bool in = t >= 0 && t < len && 4 * qd < ns;
            u32x4 xq[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
              xq[c] = __builtin_amdgcn_raw_buffer_load_b128(rx, in ? (unsigned)((8 * (ob + o) + c) * len + t) * 4u : 0x80000000u, 0, 0);
            u32x4* const dst = slab + o * NB_NSP + 4 * qd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v[8];
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = __uint_as_float(xq[c][e]);
              u32x4 h, m, l;
              nb_split8(v, h, m, l);
              dst[e] = h;
              dst[NB_PLANE + e] = m;
              dst[2 * NB_PLANE + e] = l;
            }
          }
        } else {
          auto stage = [&](int o, int s) {
            const int t = t0 - H + s;
            const bool in = t >= 0 && t < len && s < ns;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
              v[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, in ? (unsigned)((8 * (ob + o) + c) * len + t) * 4u : 0x80000000u, 0, 0));
            u32x4 h, m, l;
            nb_split8(v, h, m, l);
            u32x4* const dst = slab + o * NB_NSP + s;
            dst[0] = h;
            dst[NB_PLANE] = m;
            dst[2 * NB_PLANE] = l;
          };
          for (int o = 0; o < og; ++o) {
            stage(o, tid);
            if (wv == o && H > 0) stage(o, NB_OUT + lane);         // the tail (2 H <= 60 samples): wave o takes octet o's
          }
        }
        __syncthreads();

        // ---- K loop ---------------------------------------------------------------------------------------------------------
        const unsigned mulog = og == 1 ? 65536u : og == 2 ? 32768u : 21846u;       // q / og = (q mulog) >> 16 for q < 2^14
        auto step = [&](int set, int kb) {
          if (kb + 1 < nb) load_b(set ^ 1, kb + 1);
          int q = 4 * kb + lg;
          if (q >= kog) q = 0;                                     // padding of the last k-block: zero weights, any valid sample
          const int tap = (int)(((unsigned)q * mulog) >> 16);
          const int o = q - tap * og;
          const u32x4* const ap = slab + o * NB_NSP + (Ha - H) + tap * d + wv * (16 * NB_NM) + li;
#pragma unroll
          for (int mp = 0; mp < NB_NM; mp += 2) {
            nb_bf16x8 a[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int p = 0; p < 3; ++p) a[i][p] = __builtin_bit_cast(nb_bf16x8, ap[(mp + i) * 16 + p * NB_PLANE]);
            // piece pairs (sample piece, weight piece), small terms first: (l h) (h l) (m m) (m h) (h m) (h h)
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) {
              constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
              for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int na = 0; na < MA; ++na)
                  acc[mp + i][na] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][pa[pp]], __builtin_bit_cast(nb_bf16x8, bw[set][na][pb[pp]]),
                                                                           acc[mp + i][na], 0, 0, 0);
            }
          }
        };
        for (int kb = 0; kb < nb; kb += 2) {
          step(0, kb);
          if (kb + 1 < nb) step(1, kb + 1);
        }
        wbase += nb * MA * 3;
        ob += og;
      }
    }

    // ---- epilogue: lane = (channel 16 na + li, samples t0 + 64 wv + 16 mt + 4 lg .. + 3) ----------------------------------------
    const float scale = G->scale;
    const float* const bias = uni(G->bias);
    const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias, bias ? (unsigned)channels * 4u : 0u);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(uni((const float*)G->out) + oslab, slab_bytes);
    const __amdgpu_buffer_rsrc_t rr1 = make_rsrc(nres > 1 ? uni(G->res[1]) + oslab : nullptr, nres > 1 ? slab_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rr2 = make_rsrc(nres > 2 ? uni(G->res[2]) + oslab : nullptr, nres > 2 ? slab_bytes : 0u);
#pragma unroll
    for (int na = 0; na < MA; ++na) {
      const int ch = 16 * na + li;
      const bool chok = ch < channels;
      const float b_ = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rbias, chok ? (unsigned)ch * 4u : 0x80000000u, 0, 0));
#pragma unroll
      for (int mt = 0; mt < NB_NM; ++mt) {
        const int t = t0 + 16 * NB_NM * wv + 16 * mt + 4 * lg;
        const unsigned off = (unsigned)(ch * len + t) * 4u;
        f32x4 o = acc[mt][na] + (f32x4){b_, b_, b_, b_};
        if (VEC) {
          const unsigned vo = (chok && t < len) ? off : 0x80000000u;
          {
            const u32x4 r = rs[mt][na];                            // (zeros when the group has no residual: range-checked load)
            o += (f32x4){__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3])};
          }
          if (nres > 1) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rr1, vo, 0, 0);
            o += (f32x4){__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3])};
          }
          if (nres > 2) {
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rr2, vo, 0, 0);
            o += (f32x4){__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3])};
          }
          o *= scale;
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])},
                                                 ro, vo, 0, 0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned so = (chok && t + r < len) ? off + 4u * r : 0x80000000u;
            float y = o[r];
            if (nres > 0) y += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr0, so, 0, 0));
            if (nres > 1) y += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr1, so, 0, 0));
            if (nres > 2) y += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr2, so, 0, 0));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y * scale), ro, so, 0, 0);
          }
        }
      }
    }
  }
}

template <int MA, bool VEC>
int launch_narrow(const fh_amp_group* groups, const fh_amp_tile* tiles, int channels, int dilation, int total_tiles, hipStream_t stream) {
  static std::atomic<int> blocks_per_launch[FH_MAX_DEVICES];      // 0 = not asked yet; else resident blocks of the device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FH_MAX_DEVICES) {
    fh_set_error("fh_narrow_conv_bf16x6_f32: no current HIP device (or ordinal >= %d)", FH_MAX_DEVICES);
    return FH_E_LAUNCH;
  }
  int resident = blocks_per_launch[dev].load(std::memory_order_acquire);
  if (!resident) {
    int per_cu = 0, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)narrow_bf_kernel<MA, VEC>, NB_THREADS, 0);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cus <= 0 || per_cu <= 0) {
      fh_set_error("fh_narrow_conv_bf16x6_f32: occupancy query failed on device %d: %s", dev, hipGetErrorString(e));
      return FH_E_LAUNCH;
    }
    resident = per_cu * cus;
    blocks_per_launch[dev].store(resident, std::memory_order_release);
  }
  const int grid = total_tiles < resident ? total_tiles : resident;
  hipLaunchKernelGGL((narrow_bf_kernel<MA, VEC>), dim3((unsigned)grid), dim3(NB_THREADS), 0, stream, groups, tiles, channels, dilation,
                     total_tiles);
  FH_CHECK_LAUNCH("fh_narrow_conv_bf16x6_f32");
  return FH_OK;
}

}  // namespace

extern "C" int fh_narrow_tile_len(void) { return NB_OUT; }

extern "C" int fh_narrow_conv_bf16x6_f32(const fh_amp_group* groups, int n_groups, const fh_amp_tile* tiles, int total_tiles, int channels,
                                         int dilation, int flags, void* stream) {
  FH_CHECK_ARG(groups && n_groups > 0 && tiles && total_tiles > 0, "fh_narrow_conv_bf16x6_f32: bad sizes");
  FH_CHECK_ARG(channels >= 8 && channels <= 48 && channels % 8 == 0, "fh_narrow_conv_bf16x6_f32: %d channels (8 .. 48, a multiple of 8)",
               channels);
  FH_CHECK_ARG(dilation >= 1 && dilation <= NB_MAX_D, "fh_narrow_conv_bf16x6_f32: dilation %d (1 .. %d)", dilation, NB_MAX_D);
  FH_CHECK_ARG(flags == 0 || flags == 1, "fh_narrow_conv_bf16x6_f32: flags %d (bit 0: rows 16-byte aligned)", flags);
  const int ma = (channels + 15) / 16;
  const bool vec = flags & 1;
  hipStream_t st = (hipStream_t)stream;
#define FH_NB_CASE(MA)                                                                              \
  case MA:                                                                                          \
    return vec ? launch_narrow<MA, true>(groups, tiles, channels, dilation, total_tiles, st)        \
               : launch_narrow<MA, false>(groups, tiles, channels, dilation, total_tiles, st);
  switch (ma) {
    FH_NB_CASE(1)
    FH_NB_CASE(2)
    FH_NB_CASE(3)
  }
#undef FH_NB_CASE
  return FH_E_ARG;
}
