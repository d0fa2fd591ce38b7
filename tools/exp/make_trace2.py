"""Writes tools/exp/conv_wino_trace2.hip (generated, git-ignored) = flowhigh_amd/csrc/conv_wino.hip + a trace pointer,
its setter and fine-grained stamps (diagnosis build for tools/wino_trace2.py; never part of the product library, which
carries no debug state):
  python tools/exp/make_trace2.py && tools/build_variant.sh trace2 tools/exp/conv_wino_trace2.hip=conv_wino.hip
Record per wave (24 words, slot = block * 12 + wave, no atomics: 3072 waves hitting one counter at launch cost ~20 us
themselves): 0 id, 1 start, 2 end, 3 before / 4 after the segment descriptor fetch, 16 first A tiles arrived,
17 slab stored, 18 after the prologue barrier (= K loop start), 6 K loop end, 7 + 3 k .. 9 + 3 k (sub-tile k < 3) or
19 .. 21 (k = 3): after the first barrier / after the LDS writes + second barrier / after the arithmetic and stores;
23 wave | shader clocks << 8."""
import re
from pathlib import Path
root = Path(__file__).resolve().parents[2]
s = (root / "flowhigh_amd/csrc/conv_wino.hip").read_text()


def sub(old, new, count=1):
    global s
    assert old in s, old
    s = s.replace(old, new, count)


# the product kernel carries no debug state: the trace pointer, its setter and every stamp are added here
sub('''struct WSeg {''', '''__device__ unsigned long long* g_wino_trace = nullptr;

struct WSeg {''')
sub('''  extern __shared__ __attribute__((aligned(16))) float lds[];      // Cfg::LDS_FLOATS
''', '''  extern __shared__ __attribute__((aligned(16))) float lds[];      // Cfg::LDS_FLOATS
  unsigned long long* const trace = g_wino_trace;
  const unsigned long long t_start = trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const unsigned long long c_start = trace ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long* rec2 = nullptr;
  if (trace) rec2 = trace + 1 + 24 * ((unsigned long long)blockIdx.x * 12ull + (unsigned long long)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
#define STAMP(k) do { if (rec2 && (threadIdx.x & 63) == 0) rec2[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
''')
sub('''  WSeg S0 = load_wseg(&G->seg[0]);''', '''  STAMP(3);
  WSeg S0 = load_wseg(&G->seg[0]);
  STAMP(4);''')
sub('''  int xbuf = 0;
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {''', '''  if (rec2) { __builtin_amdgcn_s_waitcnt(0); }      // diagnosis only: first A tiles have arrived
  STAMP(16);
  int xbuf = 0;
#pragma unroll
  for (int sub = 0; sub < SUBS; ++sub) {''')
sub('''  __syncthreads();

  f32x2 c0 = {bc0, bc0}''', '''  STAMP(17);
  __syncthreads();
  STAMP(18);

  f32x2 c0 = {bc0, bc0}''')
sub('''  run_all(std::integral_constant<int, 1>{});
''', '''  run_all(std::integral_constant<int, 1>{});
  STAMP(6);
''')
sub('''      const int sub = mt * NT + nt;
''', '''      const int sub = mt * NT + nt;
      const int sb_ = sub < 3 ? 7 + 3 * sub : 19;
      STAMP(sb_);       // (first sub-tile: after the barrier that ends the K loop; later ones: after the previous stores)
''')
sub('''      }
      __syncthreads();
      if (eact) {''', '''      }
      __syncthreads();
      STAMP(sb_ + 1);
      if (eact) {''')
sub('''          }
        }
      }
    }
  }
  // (keeps pf alive''', '''          }
        }
      }
      STAMP(sb_ + 2);
    }
  }
  // (keeps pf alive''')
sub('''  if (pf == 0x7fc12345u) __builtin_amdgcn_s_sleep(1);
}
''', '''  if (pf == 0x7fc12345u) __builtin_amdgcn_s_sleep(1);
  if (rec2 && (tid & 63) == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    rec2[0] = (unsigned long long)blockIdx.x | ((unsigned long long)(hw & 0xffffff) << 32) | ((unsigned long long)(xcc & 0xf) << 56);
    rec2[1] = t_start;
    rec2[2] = __builtin_amdgcn_s_memrealtime();
    rec2[23] = (unsigned long long)wave | ((__builtin_amdgcn_s_memtime() - c_start) << 8);
  }
}
''')
sub('''extern "C" int fh_sizeof_wino_group(void)''', '''extern "C" int fh_debug_set_wino_trace(void* buf) {       // (this build only; tools/wino_trace2.py binds it itself)
  unsigned long long* p = (unsigned long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wino_trace), &p, sizeof(p)) == hipSuccess ? FH_OK : FH_E_LAUNCH;
}

extern "C" int fh_sizeof_wino_group(void)''')
(root / "tools/exp/conv_wino_trace2.hip").write_text(s)
print("tools/exp/conv_wino_trace2.hip")
