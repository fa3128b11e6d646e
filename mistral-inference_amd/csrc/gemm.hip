// bf16 MFMA GEMM for the prefill branch (M > 8 tokens): out[M,N] = epilogue(A[M,K] . W[N,K]^T).
//
// Replaces the reference's nn.Linear GEMMs at prefill (transformer_layers.py:66,93,105-106;
// transformer.py:235) with the residual add / SiLU*mul / bf16->fp32 widening fused into the epilogue,
// and (token-grouped form, one launch for all experts) moe.py:28-32's per-expert gather -> expert FFN.
//
// Both operands are K-contiguous ("B^T input"), so A and W fragments are 16-byte row slices.
// 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.  Operands go HBM -> LDS by
// LDS-DMA (`global_load_lds_dwordx4`, double-buffered, one barrier per K step, the next tile's DMA issued before the
// current tile's MFMAs; register staging only when K is not a multiple of 64); the LDS image is XOR-swizzled on the
// 16-byte slot index (slot ^= row & 7) so the ds_read_b128 fragment reads are conflict-free.
//
// SWIGLU form: the B tile holds 64 rows of W1 and the matching 64 rows of W3, every wave accumulates
// both for the same (token, j) and the epilogue is elementwise in registers.
#include <cstdlib>

#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per buffer

__device__ __forceinline__ int lds_off(int row, int slot) { return row * (BK * 2) + ((slot ^ (row & 7)) << 4); }

__device__ __forceinline__ const bf16_t* seg_row(const GemmArgs& g, int r) {
  if (r < g.n0) return g.w0 + (size_t)r * g.K;
  if (r < g.n1) return g.w1 + (size_t)(r - g.n0) * g.K;
  return g.w2 + (size_t)(r - g.n1) * g.K;
}

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + 2 * TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  constexpr int NOUT = (EPI == GEMM_SWIGLU) ? 64 : 128;  // output columns per block

  // ---- block -> (m_tile, n_tile).  Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md, dispatch); every XCD has
  // its own 4 MiB L2.  Blocks are grouped in 8 x 8 supertiles (64 = the blocks one XCD runs at a time) and a
  // supertile stays on ONE XCD, so the blocks running together there share 8 A panels and 8 W panels K-slice by
  // K-slice instead of fetching 64 + 64 (guide T1; a pure speed choice, results do not depend on it).
  const int grouped = g.tile_tab != nullptr;
  const int m_tiles = grouped ? g.max_m_tiles : (g.M + BM - 1) / BM;
  const int n_tiles = (g.N + NOUT - 1) / NOUT;
  const int MS = (m_tiles + 7) >> 3, NS = (n_tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int st = (q >> 6) * 8 + xcd, wi = q & 63;
  if (st >= MS * NS) return;
  const int m_tile = (st % MS) * 8 + (wi & 7), n_tile = (st / MS) * 8 + (wi >> 3);
  if (m_tile >= m_tiles || n_tile >= n_tiles) return;

  // rows of this m-tile: plain = [m_tile*128, +128) of [0, M); grouped = one tile of one expert
  int row0, rows_valid;
  const bf16_t *w0 = g.w0, *w1 = g.w1;
  if (grouped) {
    if (m_tile >= *g.n_tiles_ptr) return;
    const int e = g.tile_tab[m_tile * 4];
    row0 = g.tile_tab[m_tile * 4 + 1];
    rows_valid = g.tile_tab[m_tile * 4 + 2];
    w0 = reinterpret_cast<const bf16_t*>(g.expert_tab[e * 3 + g.w_sel0]);
    if (g.w_sel1 >= 0) w1 = reinterpret_cast<const bf16_t*>(g.expert_tab[e * 3 + g.w_sel1]);
  } else {
    row0 = m_tile * BM;
    rows_valid = min(BM, g.M - row0);
  }

  // ---- global -> LDS staging.
  // GLDS (K % 64 == 0): `global_load_lds_dwordx4` - the DMA writes wave-uniform base + lane * 16, i.e. one instruction
  //   fills 8 consecutive 128-byte tile rows; wave w, instruction j covers rows (4w + j) * 8 .. + 8, lane l sits at
  //   (row + (l >> 3), LDS slot l & 7).  The XOR swizzle the fragment reads expect is applied on the SOURCE address
  //   (the lane fetches global slot (l & 7) ^ (row & 7)); the LDS destination stays linear (guide section 5.4 rule 21).
  //   No staging VGPRs, no ds_write pass.
  // otherwise: through registers; thread owns LDS slot (tid & 7) of rows (tid >> 3) + 32 j and writes it swizzled.
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int slot = GLDS ? ((lane & 7) ^ ((lane >> 3) & 7)) : (tid & 7);
  const bf16_t* arow[4];
  const bf16_t* brow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = GLDS ? ((wu * 4 + j) * 8 + (lane >> 3)) : ((tid >> 3) + 32 * j);
    int m = row0 + min(r, rows_valid - 1);
    if (g.a_gather) m = g.a_gather[m];
    arow[j] = g.a + (size_t)m * g.lda + slot * 8;
    if (EPI == GEMM_SWIGLU) {
      const int jj = min(n_tile * 64 + (r & 63), g.N - 1);
      brow[j] = ((r < 64) ? w0 : w1) + (size_t)jj * g.K + slot * 8;
    } else {
      const int n = min(n_tile * BN + r, g.N - 1);
      brow[j] = (grouped ? w0 + (size_t)n * g.K : seg_row(g, n)) + slot * 8;
    }
  }
  u32x4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int koff = kt * BK;
    const bool ok = koff + slot * 8 < g.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      ra[j] = ok ? ld16(arow[j] + koff) : z;
      rb[j] = ok ? ld16(brow[j] + koff) : z;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (tid >> 3) + 32 * j;
      st16(sA + buf * TILE_BYTES + lds_off(r, slot), ra[j]);
      st16(sB + buf * TILE_BYTES + lds_off(r, slot), rb[j]);
    }
  };
  auto dma_tile = [&](int kt, int buf) {  // GLDS: 8 async 1-KiB pieces per wave (4 of A, 4 of B)
    const int koff = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = (wu * 4 + j) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(arow[j] + koff),
                                       (__attribute__((address_space(3))) void*)(sA + buf * TILE_BYTES + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(brow[j] + koff),
                                       (__attribute__((address_space(3))) void*)(sB + buf * TILE_BYTES + piece), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + BK - 1) / BK;
  if (GLDS) {
    dma_tile(0, 0);
  } else {
    gload(0);
    lstore(0);
  }
  __syncthreads();  // (with LDS-DMA in flight hipcc makes this vmcnt(0) + barrier)
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      if (GLDS) dma_tile(kt + 1, buf ^ 1);  // lands during this step's MFMAs
      else gload(kt + 1);
    }
    const char* a_s = sA + buf * TILE_BYTES;
    const char* b_s = sB + buf * TILE_BYTES;
    // all 16 fragment reads of this K step are issued first; hipcc's lgkmcnt ladder then lets the MFMAs of the first
    // half start while the second half's reads are still in flight
    bf16x8 af[2][4], bfr[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int fs = ks * 4 + (lane >> 4);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int r = wm * 64 + mt * 16 + (lane & 15);
        af[ks][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_s + lds_off(r, fs)));
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        int r;
        if (EPI == GEMM_SWIGLU) r = (nt >> 1) * 64 + wn * 32 + (nt & 1) * 16 + (lane & 15);
        else r = wn * 64 + nt * 16 + (lane & 15);
        bfr[ks][nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(b_s + lds_off(r, fs)));
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][mt], bfr[ks][nt], acc[mt][nt], 0, 0, 0);
    if (!GLDS && kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  acc[mt][nt][r] is tile row wm*64 + mt*16 + (lane>>4)*4 + r, column (lane & 15) of its 16x16 tile
  const bool rope = EPI == GEMM_STORE && g.rope_cs != nullptr;
  int tpos[4][4];
  if (rope) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) tpos[mt][r] = g.tok_pos[row0 + min(wm * 64 + mt * 16 + (lane >> 4) * 4 + r, rows_valid - 1)];
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = wm * 64 + mt * 16 + (lane >> 4) * 4 + r;
      if (rl >= rows_valid) continue;
      const size_t orow = (size_t)(row0 + rl);
      if (EPI == GEMM_SWIGLU) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int j = n_tile * NOUT + wn * 32 + s * 16 + (lane & 15);
          if (j < g.N)
            reinterpret_cast<bf16_t*>(g.out)[orow * g.ldo + j] = f_to_bf(swiglu_bf_fast(acc[mt][s][r], acc[mt][2 + s][r]));
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int n = n_tile * BN + wn * 64 + nt * 16 + (lane & 15);
          if (n >= g.N) continue;
          float y = bf_round(acc[mt][nt][r]);
          if (EPI == GEMM_STORE && rope) {
            // the pair's other half is in the neighbouring lane (column n ^ 1): both lanes form the rotation, each keeps
            // its own component.  (Uniform per 16x16 tile: rope_cols is a multiple of 16, so the exchange is not divergent.)
            const float other = __shfl_xor(y, 1, 64);
            if (n < g.rope_cols) {
              const int i0 = ((n + g.rope_col0) % g.rope_dh) >> 1;
              const float2 cs = *reinterpret_cast<const float2*>(g.rope_cs + ((size_t)tpos[mt][r] * (g.rope_dh >> 1) + i0) * 2);
              float re, im;
              if (n & 1) {
                rope_pair(other, y, cs.x, cs.y, re, im);
                y = bf_round(im);
              } else {
                rope_pair(y, other, cs.x, cs.y, re, im);
                y = bf_round(re);
              }
            }
          }
          if (EPI == GEMM_LOGITS) {
            reinterpret_cast<float*>(g.out)[orow * g.ldo + n] = y;
          } else if (EPI == GEMM_RESIDUAL) {
            reinterpret_cast<bf16_t*>(g.out)[orow * g.ldo + n] = f_to_bf(bf_to_f(g.residual[orow * g.ldo + n]) + y);
          } else {
            reinterpret_cast<bf16_t*>(g.out)[orow * g.ldo + n] = f_to_bf(y);
          }
        }
      }
    }
  }
}

}  // namespace

namespace {

hipError_t launch_gemm128(const GemmArgs& g, hipStream_t s) {
  const int m_tiles = g.tile_tab ? g.max_m_tiles : (g.M + BM - 1) / BM;
  const int nout = (g.epi == GEMM_SWIGLU) ? 64 : 128;
  const int n_tiles = (g.N + nout - 1) / nout;
  const int supertiles = ((m_tiles + 7) / 8) * ((n_tiles + 7) / 8);
  const dim3 grid((unsigned)(((supertiles + 7) / 8) * 64 * 8)), block(256);
  const size_t lds = 4 * TILE_BYTES;
  static int use_glds = -1;  // MI_GEMM_GLDS=0 forces the register-staged path (A/B testing)
  if (use_glds < 0) {
    const char* e = getenv("MI_GEMM_GLDS");
    use_glds = e ? atoi(e) : 1;
  }
  const bool glds = use_glds && (g.K % BK == 0);
#define MI_LAUNCH_GEMM(E)                                                                  \
  if (glds) hipLaunchKernelGGL((gemm_kernel<E, true>), grid, block, lds, s, g);            \
  else hipLaunchKernelGGL((gemm_kernel<E, false>), grid, block, lds, s, g)
  switch (g.epi) {
    case GEMM_STORE: MI_LAUNCH_GEMM(GEMM_STORE); break;
    case GEMM_RESIDUAL: MI_LAUNCH_GEMM(GEMM_RESIDUAL); break;
    case GEMM_SWIGLU: MI_LAUNCH_GEMM(GEMM_SWIGLU); break;
    case GEMM_LOGITS: MI_LAUNCH_GEMM(GEMM_LOGITS); break;
    default: return hipErrorInvalidValue;
  }
#undef MI_LAUNCH_GEMM
  return hipGetLastError();
}

// The same problem restricted to output columns [c0, c1): weight row segments, output and residual shifted.
GemmArgs column_slice(const GemmArgs& g, int c0, int c1) {
  GemmArgs r = g;
  r.N = c1 - c0;
  const size_t esz = (g.epi == GEMM_LOGITS) ? 4 : 2;
  r.out = reinterpret_cast<char*>(g.out) + (size_t)c0 * esz;
  if (g.residual) r.residual = g.residual + c0;
  if (g.rope_cs) {  // (slices start on tile boundaries: multiples of 128, hence of the head size)
    r.rope_col0 = g.rope_col0 + c0;
    r.rope_cols = g.rope_cols - c0 < 0 ? 0 : (g.rope_cols - c0 > r.N ? r.N : g.rope_cols - c0);
  }
  if (g.epi == GEMM_SWIGLU) {  // w0 = W1, w1 = W3, both [N, K]
    r.w0 = g.w0 + (size_t)c0 * g.K;
    r.w1 = g.w1 + (size_t)c0 * g.K;
    r.n0 = r.n1 = r.N;
    return r;
  }
  const bf16_t* seg_w[3] = {g.w0, g.w1, g.w2};
  const int seg_lo[3] = {0, g.n0, g.n1}, seg_hi[3] = {g.n0, g.n1, g.N};
  const bf16_t* w[3] = {nullptr, nullptr, nullptr};
  int end[3] = {0, 0, 0}, k = 0, done = 0;
  for (int i = 0; i < 3; ++i) {
    const int lo = seg_lo[i] > c0 ? seg_lo[i] : c0, hi = seg_hi[i] < c1 ? seg_hi[i] : c1;
    if (hi <= lo) continue;
    w[k] = seg_w[i] + (size_t)(lo - seg_lo[i]) * g.K;
    done += hi - lo;
    end[k++] = done;
  }
  r.w0 = w[0];
  r.w1 = w[1] ? w[1] : w[0];
  r.w2 = w[2] ? w[2] : r.w1;
  r.n0 = end[0];
  r.n1 = k > 1 ? end[1] : r.N;
  return r;
}

}  // namespace

static int split_tail = -1;  // MI_GEMM_TAIL / mi_debug_set_prefill_kernels: 0 no split, 1 tail on the 128 kernel, 2 half-height tiles
void gemm_set_tail_mode(int mode) { split_tail = mode; }

hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
  if (g.K % 8 != 0 || g.M <= 0 || g.N <= 0) return hipErrorInvalidValue;
  static int use_256 = -1;  // MI_GEMM_256=0 keeps every shape on the 128x128 kernel (A/B testing)
  if (use_256 < 0) {
    const char* e = getenv("MI_GEMM_256");
    use_256 = e ? atoi(e) : 1;
  }
  if (g.tile_tab && g.tile_rows == 256) return gemm256_applicable(g) ? launch_gemm256(g, s) : hipErrorInvalidValue;
  if (!use_256 || !gemm256_applicable(g)) return launch_gemm128(g, s);
  if (g.tile_tab) return launch_gemm256(g, s);
  // One 256x256 block per CU: a tile count that is not a multiple of the CU count (256) leaves the last round partly empty
  // (q|k|v of Mistral-7B: 384 tiles = 1.5 rounds cost 2).  When the last round would be under 3/4 full, the columns
  // are split: full rounds on the 256 kernel, the remaining columns on the 128 kernel (two blocks per CU, four times
  // the tiles) - MI_GEMM_TAIL=0 disables the split.  Round 4: when the remaining columns are at most HALF a round of
  // 256 x 256 tiles they run as 128 x 256 tiles of the 256 kernel instead (launch_gemm256_half: twice the blocks, so the
  // half round becomes a full one; q|k|v of Mistral-7B: 256 + 128 tiles -> 256 + 256 blocks) - MI_GEMM_TAIL=1 keeps the
  // 128 x 128 kernel for the tail, 2 (default) prefers the half-height tiles.
  if (split_tail < 0) {
    const char* e = getenv("MI_GEMM_TAIL");
    split_tail = e ? atoi(e) : 2;
  }
  const int nout = (g.epi == GEMM_SWIGLU) ? 128 : 256;
  const int m_tiles = (g.M + 255) / 256, n_tiles = (g.N + nout - 1) / nout;
  const long tiles = (long)m_tiles * n_tiles;
  const int cus = device_cus();
  const int rem = (int)(tiles % cus);
  if (split_tail && tiles > cus && rem != 0 && rem < cus * 3 / 4) {
    int a = cus, b = m_tiles;  // n_first = largest multiple of cus / gcd(cus, m_tiles) not above n_tiles
    while (b) { const int t = a % b; a = b; b = t; }
    const int step = cus / a;
    const int n_first = (n_tiles / step) * step;
    if (n_first > 0 && n_first < n_tiles) {
      const int c0 = n_first * nout;
      hipError_t e = launch_gemm256(column_slice(g, 0, c0), s);
      if (e != hipSuccess) return e;
      const GemmArgs tail = column_slice(g, c0, g.N);
      if (split_tail >= 2 && rem * 2 <= cus && gemm256_half_applicable(tail)) return launch_gemm256_half(tail, s);
      return launch_gemm128(tail, s);
    }
  }
  return launch_gemm256(g, s);
}
