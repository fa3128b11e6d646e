// Nucleus (top-p) sampling of B logits rows in ONE launch: generate.py:151-170 of the reference -
//     probs = softmax(logits / temperature); sort descending; drop every token whose mass-before-it exceeds p;
//     renormalise; draw one (torch.multinomial); gather its index
// plus the logprob bookkeeping of generate.py:134-136 (log_softmax of the UNSCALED logits at the drawn token).  The
// reference runs ~10 torch launches per token for it (softmax, sort of the whole vocabulary, cumsum, compare, masked fill,
// sum, divide, multinomial, two gathers); here it is one block per row and no sort:
//
//   * kept set.  Token at sorted position k is kept iff mass_before(k) <= p * total.  The kept positions are a prefix, so
//     the question is a SELECT, not a sort: find the last position whose mass-before is within the target.  Logits map
//     to monotone 32-bit keys; a pass builds a (count, mass) histogram of <= 1024 bins over the current key range in LDS,
//     a block scan over the bins from the top finds the bin holding that position, and the range narrows to that bin - at
//     most 4 passes over the (L2-resident) row until the range is ONE key value v*: tokens above v* are kept, and of the
//     tokens tied at v* the first n_keep by ascending index (torch.sort leaves the order of ties unspecified; ascending
//     index = a stable sort).
//   * draw.  u ~ U[0, 1) from Philox4x32-10 keyed by (seed; decode step counter of the workspace, row): the counter lives
//     on the device, so a hipGraph replay draws fresh numbers.  r = u * kept mass; the drawn position is again "the last
//     position whose mass-before is <= r" - the same select - and one more pass finds the index of the k-th token (by
//     ascending index) tied at that value.
//   * masses are 40-bit FIXED-POINT integers (exp(x/T - max) in [0, 1] scaled by 2^40): integer sums do not depend on the
//     order in which the LDS atomics land, so the sample is bit-reproducible for a given (seed, step) - a float histogram
//     would flip boundary decisions from run to run.  A token with probability < 2^-40 of the mode has mass 0 (it would be
//     drawn once in 10^12 samples).
//
// The sample is written where the greedy sample goes (token buffer that the next step reads as its input ids, logprob,
// history rings): the decode loop of generate() at temperature > 0 is then ONE native call per token as well.
#include "common.cuh"
#include "kernels.h"

namespace {

constexpr int ST = 1024;     // threads per row
constexpr int NBIN = 1024;   // histogram bins per pass
constexpr float QSCALE = 1099511627776.0f;  // 2^40

__device__ __forceinline__ uint32_t key_of(float x) {  // monotone increasing; NaN sorts below everything (never kept)
  if (x != x) return 0u;
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct Philox {
  // Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1)
  __device__ static void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  __device__ static void draw(uint64_t seed, uint64_t ctr, uint32_t row, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), row, 0x746f7070u /* "topp" */};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c[i];
  }
};

struct Scratch {
  unsigned long long mass[NBIN];
  uint32_t cnt[NBIN];
  unsigned long long wsum[16];
  float wf[16];
  uint32_t wu[16];
  unsigned long long bc_u64[4];
  uint32_t bc_u32[4];
};

__device__ __forceinline__ float block_max(float v, Scratch& sc) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sc.wf[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = sc.wf[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = fmaxf(m, sc.wf[i]);
  return m;
}
__device__ __forceinline__ float block_sum_f(float v, Scratch& sc) {  // fixed tree: the same total on every run
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sc.wf[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += sc.wf[i];
  return s;
}
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, Scratch& sc) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sc.wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += sc.wsum[i];
  return s;
}
__device__ __forceinline__ void block_minmax_u32(uint32_t& lo, uint32_t& hi, Scratch& sc) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    sc.wu[threadIdx.x >> 6] = lo;
    sc.wf[threadIdx.x >> 6] = __uint_as_float(hi);
  }
  __syncthreads();
  lo = sc.wu[0];
  hi = __float_as_uint(sc.wf[0]);
#pragma unroll
  for (int i = 1; i < 16; ++i) {
    lo = min(lo, sc.wu[i]);
    hi = max(hi, __float_as_uint(sc.wf[i]));
  }
}
// exclusive prefix sum over the block in thread order; `total` = sum over all threads
__device__ __forceinline__ unsigned long long block_exscan_u64(unsigned long long v, unsigned long long& total, Scratch& sc) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) sc.wsum[w] = inc;
  __syncthreads();
  unsigned long long base = 0;
  total = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < w) base += sc.wsum[i];
    total += sc.wsum[i];
  }
  return base + inc - v;
}

struct Row {
  const float* x;
  int V;
  float temperature, ms;  // ms = max(x / T)
  __device__ __forceinline__ unsigned long long q(float xv) const {  // fixed-point mass of one token
    if (xv != xv) return 0ull;
    const float e = expf(xv / temperature - ms);  // softmax(logits / temperature) numerator (generate.py:153)
    return (unsigned long long)(e * QSCALE);
  }
};

// The last sorted position (descending value, ties by ascending index) whose mass-before is <= target, as
// (key value, mass of everything above that value, number of tokens tied at it).  Block-uniform result.
struct Sel {
  uint32_t key, cnt;
  unsigned long long above;
};
// Every logit of the row once, in no particular order across threads: 16 values per trip from four independent 16-byte
// loads (round 6: each pass was a loop of one 4-byte load per trip - V / 1024 dependent L2 round trips; a draw at 32768
// logits 110 -> 79 us, at 131072 364 -> 240 us.  What remains are ~9 passes of ONE block over the row - at 131072 logits a
// CU's L2 read of 512 KB per pass - with ~8 block barriers each; timing ablations: the histogram's LDS atomics as plain adds, or
// spread over lane-dependent bins (no same-address contention), changed nothing, and fusing the total-mass pass into the
// first histogram pass did not either).
template <class F>
__device__ __forceinline__ void for_logits(const Row& r, F&& f) {
  if ((r.V & 3) == 0 && (reinterpret_cast<size_t>(r.x) & 15) == 0) {
    const int V4 = r.V >> 2;
    for (int q = threadIdx.x; q < V4; q += 4 * ST) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(r.x + (size_t)min(q + k * ST, V4 - 1) * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (q + k * ST < V4) {
          f(v[k].x);
          f(v[k].y);
          f(v[k].z);
          f(v[k].w);
        }
    }
  } else {
    for (int i = threadIdx.x; i < r.V; i += ST) f(r.x[i]);
  }
}

__device__ Sel select_pos(const Row& r, uint32_t lo, uint32_t hi, unsigned long long target, uint32_t cnt0, Scratch& sc) {
  unsigned long long above = 0;  // mass of all keys > hi
  uint32_t cnt = cnt0;           // tokens tied at lo when the range is a single key already
  while (lo != hi) {
    const uint32_t span = hi - lo;  // >= 1
    const int bits = 32 - __clz((int)span);
    const int shift = bits > 10 ? bits - 10 : 0;
    const int nb = (int)(span >> shift) + 1;  // <= 1024
    __syncthreads();
    for (int i = threadIdx.x; i < NBIN; i += ST) {
      sc.mass[i] = 0;
      sc.cnt[i] = 0;
    }
    __syncthreads();
    for_logits(r, [&](float xv) {
      const uint32_t k = key_of(xv);
      if (k >= lo && k <= hi) {
        const uint32_t b = (k - lo) >> shift;
        atomicAdd(&sc.mass[b], r.q(xv));
        atomicAdd(&sc.cnt[b], 1u);
      }
    });
    __syncthreads();
    // thread t looks at bin nb - 1 - t (descending values): above(bin) = above + sum of the masses of higher bins
    const int t = threadIdx.x;
    const int bin = nb - 1 - t;
    const unsigned long long m = (bin >= 0) ? sc.mass[bin] : 0ull;
    const uint32_t c = (bin >= 0) ? sc.cnt[bin] : 0u;
    unsigned long long total;
    const unsigned long long ab = above + block_exscan_u64(m, total, sc);
    // the LOWEST non-empty bin whose first token is still within the target = the largest t that qualifies
    int cand = (bin >= 0 && c > 0 && ab <= target) ? t : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = max(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sc.wu[threadIdx.x >> 6] = (uint32_t)(cand + 1);
    __syncthreads();
    int best = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) best = max(best, (int)sc.wu[i]);
    best -= 1;  // >= 0: the top non-empty bin always qualifies (above <= target by construction)
    if (best < 0) best = 0;
    if (t == best) {
      sc.bc_u64[0] = ab;
      sc.bc_u32[0] = c;
    }
    __syncthreads();
    above = sc.bc_u64[0];
    cnt = sc.bc_u32[0];
    const uint32_t b = (uint32_t)(nb - 1 - best);
    const uint32_t nlo = lo + (b << shift);
    const uint32_t width = shift ? ((1u << shift) - 1u) : 0u;
    hi = (hi - nlo > width) ? nlo + width : hi;
    lo = nlo;
  }
  return Sel{lo, cnt, above};
}

__global__ __launch_bounds__(ST) void sample_top_p_kernel(const float* logits, int ld, int B, int V, float temperature, float top_p,
                                                         unsigned long long seed, unsigned long long offset,
                                                         const float* uniforms, int64_t* tok, float* lp, int64_t* hist_tok,
                                                         float* hist_lp, int hist_len, const uint32_t* ctrl) {
  __shared__ Scratch sc;
  // Behind the persistent decode engine: a launch that failed its residency gate (status 0x700) wrote nothing - no logits, no
  // step count - and poisons the workspace until the host has re-run the step on the launch path (GreedySession._recover).
  // Drawing from whatever the logits buffer holds would overwrite `tok`, which is the NEXT step's input id and the id the
  // recovery re-runs from.  (The stand-alone mi_sample_top_p passes no control block.)
  if (ctrl && ctrl[1] != 0) return;
  const int b = blockIdx.x;
  Row r;
  r.x = logits + (size_t)b * ld;
  r.V = V;
  r.temperature = temperature;
  // ---- pass 1: maxima and key range
  float M = -INFINITY;
  uint32_t klo = 0xffffffffu, khi = 0u;
  for_logits(r, [&](float xv) {
    if (xv == xv) {
      M = fmaxf(M, xv);
      const uint32_t k = key_of(xv);
      klo = min(klo, k);
      khi = max(khi, k);
    }
  });
  M = block_max(M, sc);
  block_minmax_u32(klo, khi, sc);
  if (klo > khi) {  // a row of NaN: nothing to draw from (torch.multinomial raises); token 0, logprob NaN
    if (threadIdx.x == 0) {
      tok[b] = 0;
      lp[b] = __uint_as_float(0x7fc00000u);
    }
    return;
  }
  r.ms = M / temperature;
  // ---- pass 2: total mass (fixed point) and the log-sum-exp of the UNSCALED logits (generate.py:134)
  unsigned long long S = 0;
  float S1 = 0.f;
  for_logits(r, [&](float xv) {
    S += r.q(xv);
    if (xv == xv) S1 += expf(xv - M);
  });
  S = block_sum_u64(S, sc);
  S1 = block_sum_f(S1, sc);
  // ---- the nucleus: last position with mass-before <= p * total (generate.py:166 `probs_sum - probs_sort > p` is dropped)
  const double pt = (double)top_p * (double)S;
  const unsigned long long target = pt >= (double)S ? S : (unsigned long long)pt;
  const Sel cut = select_pos(r, klo, khi, target, (uint32_t)V, sc);
  const unsigned long long qcut = r.q(float_of(cut.key));
  unsigned long long n_keep = cut.cnt;
  if (qcut > 0) {
    const unsigned long long fit = (target - cut.above) / qcut + 1ull;  // tied tokens whose mass-before is still <= target
    n_keep = fit < n_keep ? fit : n_keep;
  }
  const unsigned long long kept = cut.above + n_keep * qcut;  // > 0: the mode alone has mass 2^40
  // ---- the draw (generate.py:168 torch.multinomial on the renormalised kept masses): r uniform in [0, kept)
  double u;
  if (uniforms) {
    u = (double)uniforms[b];
  } else {
    uint32_t rnd[4];
    const unsigned long long step = ctrl ? (unsigned long long)ctrl[5] : 0ull;
    Philox::draw(seed, offset + step, (uint32_t)b, rnd);
    u = ((double)rnd[0] * 4294967296.0 + (double)rnd[1]) * (1.0 / 18446744073709551616.0);
  }
  u = u < 0.0 ? 0.0 : u;
  unsigned long long rr = (unsigned long long)(u * (double)kept);
  if (rr >= kept) rr = kept - 1;
  const Sel pick = select_pos(r, cut.key, khi, rr, cut.cnt, sc);   // (never below the cut: rr < kept)
  const unsigned long long qp = r.q(float_of(pick.key));
  unsigned long long rank = qp > 0 ? (rr - pick.above) / qp : 0ull;
  const unsigned long long lim = (pick.key == cut.key ? n_keep : (unsigned long long)pick.cnt) - 1ull;
  rank = rank > lim ? lim : rank;
  // ---- index of the rank-th token (ascending index) whose key is pick.key: thread t owns indices [t * per, t * per + per)
  const int per = (V + ST - 1) / ST;
  const int i0 = threadIdx.x * per, i1 = min(V, i0 + per);
  unsigned long long mine = 0;
  if ((per & 15) == 0 && i1 - i0 == per && (reinterpret_cast<size_t>(r.x) & 15) == 0) {  // (whole, 16-byte aligned ranges)
    for (int i = i0; i < i1; i += 16) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(r.x + i + 4 * k);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        mine += (key_of(v[k].x) == pick.key) + (key_of(v[k].y) == pick.key) + (key_of(v[k].z) == pick.key) + (key_of(v[k].w) == pick.key);
    }
  } else {
    for (int i = i0; i < i1; ++i) mine += (key_of(r.x[i]) == pick.key);
  }
  unsigned long long total;
  const unsigned long long before = block_exscan_u64(mine, total, sc);
  rank = rank >= total ? total - 1ull : rank;  // (total >= 1: the key was found in the row)
  if (rank >= before && rank < before + mine) {
    unsigned long long seen = before;
    for (int i = i0; i < i1; ++i) {
      if (key_of(r.x[i]) == pick.key) {
        if (seen == rank) {
          const float l = r.x[i] - M - logf(S1);  // log_softmax(logits)[token] (generate.py:134-136)
          tok[b] = i;
          lp[b] = l;
          if (hist_tok && hist_len > 0 && ctrl) {
            const uint32_t step = (ctrl[5] - 1u) % (uint32_t)hist_len;
            hist_tok[(size_t)step * B + b] = i;
            hist_lp[(size_t)step * B + b] = l;
          }
          break;
        }
        ++seen;
      }
    }
  }
}

}  // namespace

hipError_t launch_sample_top_p(const float* logits, int ld, int B, int V, float temperature, float top_p, uint64_t seed,
                               uint64_t offset, const float* uniforms, int64_t* tok, float* lp, int64_t* hist_tok, float* hist_lp,
                               int hist_len, const uint32_t* ctrl, hipStream_t s) {
  hipLaunchKernelGGL(sample_top_p_kernel, dim3(B), dim3(ST), 0, s, logits, ld, B, V, temperature, top_p,
                     (unsigned long long)seed, (unsigned long long)offset, uniforms, tok, lp, hist_tok, hist_lp, hist_len, ctrl);
  return hipGetLastError();
}
