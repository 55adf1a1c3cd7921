// RETIRED (round 3): not part of libecad_hip.so any more -- no call site has launched it since the end of round 1 (gemm3s.h overtook it on FF2 once its second
// fragment set was carried across the K-step barrier).  Kept for its hand-pipelined consumer loop (sched_group_barrier-pinned fragment ring), which DESIGN 9
// refers to.  To build it again: include it from engine.hip after gemm.h and restore the launch branch removed in the commit that moved this file.
// pgemm.h -- persistent producer/consumer MFMA GEMM for the PixArt block on gfx950 (production path).
//
//   C[M,N] = epilogue( A[M,K] @ W[N,K]^T + bias ),   N % 144 == 0, K % 64 == 0
//
// Why this shape (measured on MI355X):
//  * every N of the model is a multiple of 144 (D = 1152 = 8*144), M = 8192 at batch 16: 256x144 tiles give
//    256 / 768 / 1024 tiles = exactly 1 / 3 / 4 tiles per CU;
//  * one workgroup per CU, PERSISTENT over its tiles: 4 consumer waves (one per SIMD) own 64 rows x 144 columns
//    each (4 x 9 MFMA 16x16x32 fragments, 144 accumulators) -- half the LDS fragment traffic of 8 waves x 32 rows,
//    which measured LDS-bound -- and 4 producer waves (one per SIMD) do nothing but issue LDS-DMA
//    (global_load_lds_dwordx4): DMA issue inside an MFMA wave's instruction stream stalls the matrix pipe (the
//    4-wave, no-producer variant measured 12 % slower than 8 mixed waves);
//  * a 3-deep ring of 50 KiB LDS stages (A 256x64 + W 144x64), filled two K-steps ahead ACROSS tile boundaries, one
//    raw s_barrier per K-step with a counted vmcnt on the producers only; while the consumers run a tile's
//    epilogue the first two K-steps of the next tile are already landing;
//  * epilogue through the just-consumed stage (wave-private 32-row slabs, no workgroup barrier): 8-byte packed
//    fragment writes, 16-byte row reads, fused `cache <- out; hidden += gate * out`.
#pragma once
#include "common.h"
#include "gemm.h"
#include <type_traits>
#include <utility>

namespace ecad {

constexpr int PG_BM = 256, PG_BN = 144, PG_STAGES = 3;
constexpr int PG_A_BYTES = PG_BM * 128, PG_STAGE_BYTES = (PG_BM + PG_BN) * 128;  // 51200
constexpr int PG_LDS_BYTES = PG_STAGES * PG_STAGE_BYTES;                         // 153600
constexpr int PG_CS = PG_BN + 8;                                                 // C slab row stride (elements)


// tile id -> workgroup: XCD x (blocks b with b % 8 == x share an L2) owns a contiguous chunk of the row-major tile
// order; inside the chunk its workgroups take consecutive tiles round by round, so the tiles resident on one XCD at
// any time are neighbours (shared A rows / W rows in L2).
struct PgSched {
  int first, stride, end;
  __device__ PgSched(int ntiles, int bid, int nwg) {
    const int xcd = bid & 7, local = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int len = xcd < r ? q + 1 : q;
    const int wg_in_xcd = (nwg >> 3) + ((nwg & 7) > xcd ? 1 : 0);
    first = start + local;
    stride = wg_in_xcd > 0 ? wg_in_xcd : 1;
    end = start + len;
  }
};

template <typename T, int EPI, int TAG>
__global__ __launch_bounds__(512) void pgemm_kernel(const GemmArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / PG_BN;
  const int tiles_m = (p.M + PG_BM - 1) / PG_BM;
  const int nk = p.K / GEMM_BK;
  const PgSched sched(tiles_m * tiles_n, blockIdx.x, gridDim.x);
  int my_tiles = 0;
  if (sched.first < sched.end) my_tiles = (sched.end - sched.first + sched.stride - 1) / sched.stride;
  const int total_steps = my_tiles * nk;

  if (wave >= 4) {
    // =================================================================== producer waves
    const int pw = wave - 4;
    constexpr int A_INST = PG_BM / 8 / 4;  // 8
    constexpr int B_TOTAL = PG_BN / 8;     // 18 -> waves 0,1 issue 5, waves 2,3 issue 4
    const T* a_src[A_INST];
    const T* b_src[5];
    int it_tile = sched.first, it_kt = 0;
    auto set_tile = [&](int tile) {
      int tm, tn;
      tile_coords(tile, tiles_m, tiles_n, tm, tn);
      const int m0 = tm * PG_BM, n0 = tn * PG_BN;
#pragma unroll
      for (int i = 0; i < A_INST; ++i) {
        const int r = (pw * A_INST + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int row = m0 + r;
        row = row < p.M ? row : p.M - 1;
        a_src[i] = p.A + (size_t)row * p.lda + c * 8;
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        int inst = pw + i * 4;
        inst = inst < B_TOTAL ? inst : B_TOTAL - 1;
        const int r = inst * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        b_src[i] = p.W + (size_t)(n0 + r) * p.ldw + c * 8;
      }
    };
    auto issue = [&](int stage) {
      char* abase = smem + stage * PG_STAGE_BYTES;
      char* bbase = abase + PG_A_BYTES;
#pragma unroll
      for (int i = 0; i < A_INST; ++i) {
        glds16(a_src[i], abase + (pw * A_INST + i) * 1024);
        a_src[i] += GEMM_BK;
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int inst = pw + i * 4;
        if (inst < B_TOTAL) glds16(b_src[i], bbase + inst * 1024);
        b_src[i] += GEMM_BK;
      }
#ifdef PG_ABLATE_DMA_REUSE  // timing-only: re-read the same K-slice every step (keeps DMA count, L2-hot)
#pragma unroll
      for (int i = 0; i < A_INST; ++i) a_src[i] -= GEMM_BK;
#pragma unroll
      for (int i = 0; i < 5; ++i) b_src[i] -= GEMM_BK;
#endif
      if (++it_kt == nk) {
        it_kt = 0;
        it_tile += sched.stride;
        if (it_tile < sched.end) set_tile(it_tile);
      }
    };
    if (total_steps > 0) {
      set_tile(it_tile);
      issue(0);
      if (total_steps > 1) issue(1);
    }
    int stage = 0, kt_in_tile = 0;
    for (int s = 0; s < total_steps; ++s) {
      // this wave issues 12 or 13 DMAs per step; "<= 12 outstanding" retires all of step s while step s+1 flies
      if (s + 1 < total_steps) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // step s published; consumers are done with step s-1 (and its epilogue)
#ifndef PG_ABLATE_NO_DMA
      if (s + 2 < total_steps) {
        int ns = stage + 2;
        ns = ns >= PG_STAGES ? ns - PG_STAGES : ns;
        issue(ns);
      }
#endif
      stage = stage + 1 >= PG_STAGES ? 0 : stage + 1;
      if (++kt_in_tile == nk) {
        kt_in_tile = 0;
        __builtin_amdgcn_s_barrier();  // matches the consumers' end-of-tile barrier (all operand reads of the tile done)
      }
    }
    return;
  }

  // ===================================================================== consumer waves
  const int l15 = lane & 15, g = lane >> 4;
  const int swz = (l15 >> 1) & 7;
  const int a_row_off = (wave * 64 + l15) * 128;
  const int b_row_off = l15 * 128;
  constexpr int FM = 4, FN = 9;
  int stage = 0;
  for (int tile = sched.first; tile < sched.end; tile += sched.stride) {
    int tm, tn;
    tile_coords(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * PG_BM, n0 = tn * PG_BN;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- main loop, register-pipelined by hand: the 18 B fragments of a K-step (2 k-substeps x 9) stream through a
    // 3-slot register ring two fragments ahead of their MFMAs (18 %% 3 == 0 keeps slots aligned across steps); the A fragments of k-substep 1 are fetched during
    // substep 0.  The barrier that publishes step s+1 sits at q = 16 of step s (after the step's last LDS read has
    // returned), so the first fragment reads of step s+1 overlap the last 8 MFMAs of step s.
    auto lds_a = [&](const char* sbase, int kk, int i) { return lds_read16(sbase + a_row_off + i * 2048 + (((kk * 4 + g) ^ swz) * 16)); };
    auto lds_b = [&](const char* sbase, int q) {
      const int kk = q / FN, j = q - kk * FN;
      return lds_read16(sbase + PG_A_BYTES + b_row_off + j * 2048 + (((kk * 4 + g) ^ swz) * 16));
    };
    constexpr int RING = 6, DIST = 5;  // 18 % RING == 0 keeps ring slots aligned across steps
    Pack8 afA[FM], afB[FM], wf[RING];
    __builtin_amdgcn_s_barrier();  // first step of the tile published
    {
      const char* cur = smem + stage * PG_STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < FM; ++i) afA[i] = lds_a(cur, 0, i);
#pragma unroll
      for (int f = 0; f < DIST; ++f) wf[f] = lds_b(cur, f);
    }
    int last_stage = 0;
    // one K-step; LAST = last step of the tile (no early barrier, no next-step prefetch).  sched_group_barrier pins
    // the issue order "fragment reads, then the 4 MFMAs of q": left alone, hipcc sinks every ds_read to just before
    // its first use and the single consumer wave per SIMD then eats the LDS latency 13 times per step.
    auto do_step = [&](auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      constexpr int NQ = 2 * FN;        // 18 fragment groups per step
      constexpr int QBAR = NQ - 2;      // early barrier position
      const char* cur = smem + stage * PG_STAGE_BYTES;
      const int nstage = stage + 1 >= PG_STAGES ? 0 : stage + 1;
      const char* nxt = smem + nstage * PG_STAGE_BYTES;
      static_for<NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int n_cur = (q + DIST < NQ) ? 1 : 0;
        constexpr int n_afb = (q >= 2 && q < 2 + FM) ? 1 : 0;
        // next-step prefetch: nothing may be read from the next stage before the barrier at QBAR; catch up there
        constexpr int n_next = LAST ? 0 : (q == QBAR ? FM + (QBAR + DIST - NQ + 1) : (q > QBAR ? 1 : 0));
        if constexpr (n_cur) wf[(q + DIST) % RING] = lds_b(cur, q + DIST);
        if constexpr (n_afb) afB[q - 2] = lds_a(cur, 1, q - 2);
        if constexpr (!LAST && q == QBAR) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS read of this step has returned
          __builtin_amdgcn_s_barrier();                        // step s+1 published / this stage released
#pragma unroll
          for (int i = 0; i < FM; ++i) afA[i] = lds_a(nxt, 0, i);
#pragma unroll
          for (int f = 0; f <= QBAR + DIST - NQ; ++f) wf[f % RING] = lds_b(nxt, f);
        }
        if constexpr (!LAST && q > QBAR) wf[(q + DIST) % RING] = lds_b(nxt, q + DIST - NQ);
        constexpr int kk = q / FN, j = q - kk * FN;
#ifdef PG_ABLATE_MFMA  // timing-only: keep the fragment reads live, skip the matrix work
        asm volatile("" ::"v"(wf[q % RING].u), "v"(afA[0].u), "v"(afB[0].u));
#else
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<T>(wf[q % RING], kk == 0 ? afA[i] : afB[i], acc[i][j]);
#endif
        constexpr int nreads = n_cur + n_afb + n_next;
        if constexpr (nreads > 0) __builtin_amdgcn_sched_group_barrier(0x100, nreads, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, FM, 0);
      });
      last_stage = stage;
      stage = nstage;
    };
#ifdef PG_STAMP
    unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int kt = 0; kt + 1 < nk; ++kt) do_step(std::false_type{});
    do_step(std::true_type{});
    __builtin_amdgcn_sched_barrier(0);  // keep the epilogue's loads out of the register-tight main loop
#ifdef PG_STAMP
    if (p.gate_table == nullptr && p.resid == nullptr && lane == 0 && wave == 0 && tile == sched.first) {
      unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + 2 * blockIdx.x;
      dbg[0] = st_c1 - st_c0;
      dbg[1] = st_r1 - st_r0;
    }
#endif

    // every consumer finished reading the last stage's operands -> it may be reused as C staging
    __builtin_amdgcn_s_barrier();
    // ---- epilogue: two 32-row halves through this wave's private slab of the stage just consumed
    T* slab = reinterpret_cast<T*>(smem + last_stage * PG_STAGE_BYTES) + wave * 32 * PG_CS;
    // gate vector of a tile that lies inside one sample (always for 256 / 4096 tokens): fetched once per tile by lanes 0 .. 17 into this
    // wave's private strip behind the four slabs, instead of three L1 loads next to every one of the 1152 output chunks of the wave
    // (ablation: the per-chunk gate loads cost 0.8 ms of the 20.5 ms FF2 spends per alpha-256 batch)
    float* gwave = reinterpret_cast<float*>(smem + last_stage * PG_STAGE_BYTES + 4 * 32 * PG_CS * sizeof(T)) + wave * PG_BN;
    bool gate_lds = false;
    if (EPI == EPI_BIAS_CACHE_RESID && p.gate_table) {
      const int m_last = (m0 + PG_BM <= p.M ? m0 + PG_BM : p.M) - 1;
      gate_lds = m0 / p.rows_per_batch == m_last / p.rows_per_batch;
      if (gate_lds && lane < PG_BN / 8) {
        const int n = n0 + lane * 8;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.gate_table + n);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.gate_table + n + 4);
        float gs[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        if (p.gate_mod) {
          float gm[8];
          unpack8<T>(gload16(p.gate_mod + (size_t)(m0 / p.rows_per_batch) * p.gate_mod_stride + n), gm);
#pragma unroll
          for (int e = 0; e < 8; ++e) gs[e] += gm[e];
        }
        *reinterpret_cast<f32x4*>(gwave + lane * 8) = (f32x4){gs[0], gs[1], gs[2], gs[3]};
        *reinterpret_cast<f32x4*>(gwave + lane * 8 + 4) = (f32x4){gs[4], gs[5], gs[6], gs[7]};
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int nl = 16 * j + 4 * g;
        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + nl);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = half * 2 + ii;
          Pack4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][j][r] + bv[r];
            if (EPI == EPI_BIAS_GELU) v = gelu_tanh(v);
            if (EPI == EPI_BIAS_SILU) v = silu(v);
            float_to_elem4<T>(o, r, v);
          }
          *reinterpret_cast<u32x2*>(slab + (16 * ii + l15) * PG_CS + nl) = o.u;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private slab: in-order LDS, no barrier needed
      constexpr int CPR = PG_BN / 8;                      // 18 chunks per row, 32 rows = 576 chunks = 9 per lane
#pragma unroll
      for (int it = 0; it < 9; ++it) {
        const int id = lane + it * 64;
        const int rl = id / CPR, cchunk = id - rl * CPR;
        const int m = m0 + wave * 64 + half * 32 + rl;
        const int n = n0 + cchunk * 8;
        if (m < p.M) {
          Pack8 v = lds_read16(slab + rl * PG_CS + cchunk * 8);
          if (EPI == EPI_BIAS_CACHE_RESID) {
            if (p.C) gstore16_nt(p.C + (size_t)m * p.ldc + n, v);  // cache <- un-gated sub-layer output
            T* rp = p.resid + (size_t)m * p.ldr + n;
            Pack8 hv = gload16(rp);
            float o[8], h[8];
            unpack8<T>(v, o);
            unpack8<T>(hv, h);
            if (gate_lds) {
              const f32x4 g0 = *reinterpret_cast<const f32x4*>(gwave + cchunk * 8), g1 = *reinterpret_cast<const f32x4*>(gwave + cchunk * 8 + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) { h[e] += g0[e] * o[e]; h[4 + e] += g1[e] * o[4 + e]; }
            } else if (p.gate_table) {
              const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.gate_table + n);
              const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.gate_table + n + 4);
              float gt[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
              if (p.gate_mod) {
                float gm[8];
                const int b = m / p.rows_per_batch;
                unpack8<T>(gload16(p.gate_mod + (size_t)b * p.gate_mod_stride + n), gm);
#pragma unroll
                for (int e = 0; e < 8; ++e) gt[e] += gm[e];
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) h[e] += gt[e] * o[e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) h[e] += o[e];
            }
            gstore16(rp, pack8<T>(h));
          } else {
            gstore16(p.C + (size_t)m * p.ldc + n, v);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slab reads done before the next half overwrites it
    }
  }
}

}  // namespace ecad
