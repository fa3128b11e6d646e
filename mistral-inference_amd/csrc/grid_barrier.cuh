// Grid-wide barrier for persistent kernels whose blocks are ALL co-resident (grid <= what the chip holds at once).
//
// Placement-independent by construction (cdna_hip_programming.md section 6, Guideline 16; MI355X_MICROARCH.md "Workgroup
// dispatch, XCD placement & inter-workgroup visibility"): data crosses the barrier only through write-through (sc0 sc1)
// stores before arrive() and sc0 sc1 loads after wait() - never through where a block happens to run.  Two-level
// ("barrier-xcd" shape of the guide's price list): blocks arrive on one of 8 group counters (group = block id % 8,
// which is also the XCD a block is observed to run on - used for speed only), the last arriver of a group arrives on
// the top counter, the last group publishes the epoch to 8 per-group generation words that the blocks poll with relaxed
// loads + s_sleep (one lane per block; no acquire inside the poll loop).  Counters are monotonic within a launch
// (targets scale with the epoch) and must be zeroed before every launch (a memset node ahead of the kernel).
// Every spin is bounded: on timeout the abort word is set and the kernel runs to completion with garbage results
// instead of hanging the device.
//
// The barrier is split in arrive() / wait() so that a block can put independent memory traffic (the next operator's
// weight stream) between the two.
#pragma once
#include "common.cuh"

struct GridBarrierState {  // 64-byte separated words
  uint32_t group_cnt[8][16];
  uint32_t top_cnt[16];
  uint32_t gen[8][16];
  uint32_t abort_flag[16];
};

struct GridBarrier {
  GridBarrierState* st;
  int block_id, n_blocks;
  uint32_t epoch;  // completed barriers so far

  __device__ __forceinline__ int group_size(int g) const { return (n_blocks - g + 7) / 8; }

  // Count this block in.  Precondition: everything this block publishes was stored WRITE-THROUGH (st_*_wt in
  // common.cuh) - then no release fence is needed (a per-block `buffer_wbl2` costs 2-6 us and serialises when several
  // blocks share a CU; measured 4x slower than two separate launches).  Call with ALL threads of the block.
  __device__ __forceinline__ void arrive() {
    ++epoch;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its own write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
      const int g = block_id & 7;
      const int n_groups = n_blocks < 8 ? n_blocks : 8;
      const uint32_t old = __hip_atomic_fetch_add(&st->group_cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == (uint32_t)group_size(g) * epoch) {
        const uint32_t t = __hip_atomic_fetch_add(&st->top_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == (uint32_t)n_groups * epoch) {
          for (int i = 0; i < n_groups; ++i)
            __hip_atomic_store(&st->gen[i][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }

  // Leave once every block has arrived.  Afterwards the other blocks' write-through stores are visible to
  // agent-coherent loads (ld*_coherent in common.cuh); plain loads of such data are NOT safe (no acquire is done).
  __device__ __forceinline__ void wait() {
    if (threadIdx.x == 0) {
      const int g = block_id & 7;
      uint32_t spins = 0;
      while (__hip_atomic_load(&st->gen[g][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22) || __hip_atomic_load(&st->abort_flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&st->abort_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
  }
};
