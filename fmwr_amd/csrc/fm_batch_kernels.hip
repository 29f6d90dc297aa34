// Mini-batch hot path for gfx950: two gather-reduce kernels over sparse lists.
//
//   phase 1  fm_rows_forward : per example (CSR row) gather the V rows of its nonzeros, reduce
//                              sum_f / sum_sqr_f, form y_hat and the gradient multiplier.
//                              Replaces Model::predict (core/Model.h:75-103) + calculate_grad_mult
//                              (solver/SGD_Learner.h:180-191) for a whole tile, and
//                              Model::predict_batch / predict_prob (core/Model.h:106-180) when !TRAIN.
//   phase 2  fm_cols_update  : per feature (row of the tile's CSC) gather the per-example factor sums,
//                              reduce the coordinate's gradient sums and finish the coordinate: add / publish through the
//                              exchange buffer and/or apply the update (SGD_Learner.h:111-138,
//                              FTRL_Learner.h:88-113,158-202) once.  Its workgroup 0 also reduces phase 1's partial sums
//                              and does the w0 step (SGD_Learner.h:106-109, FTRL_Learner.h:80-86,161).
//            fm_cols_long_*  : the same for heavy-hitter features whose lists are too long for one lane group.
//
// A step (batch) is one or more tiles; parameters are frozen across the tiles of a step and the sums accumulate in the
// exchange buffer (which is also what N > 1 GPUs all-reduce).
//
// Both gathers use one skeleton: a group of LPR lanes owns one list; each lane keeps a 16-byte slice of
// the table row (4 floats / 2 doubles), so one wave-instruction fetches 64/LPR whole rows, every row as
// one contiguous 16*LPR-byte segment (k=16 fp32: 64 B).  The (id, x) entries of all lists of a workgroup
// are contiguous in memory; they are staged through LDS with coalesced loads so the row gathers issue
// back to back.  Accumulation is fp64 in registers (the kernels are bound by the memory system's random-line rate;
// VALU is idle), in list order, which makes every result independent of launch geometry and bitwise reproducible.
// No MFMA: this is a sparse gather-reduce, not a dense contraction.  No atomics anywhere.
#include <utility>

#include "fmx_internal.h"
#include "fm_probit.h"

#ifndef FMX_U
#define FMX_U 4  // row gathers kept in flight per lane
#endif
// Phase 1 keeps fewer in flight when the chip is full: a step large enough for the 256-thread workgroups runs at the gather
// ceiling with ONE entry (its V row and its w) outstanding per lane group -- the waves on a CU supply the parallelism, and
// more per wave only lengthens the queues (measured at configs[1]: 1 -> 0.147 ms per tile, 4 -> 0.164).  A small step is the
// opposite: a handful of waves per CU, every round a bare memory round trip, so it keeps eight.
// waves per SIMD the lean sparse form of phase 2 is compiled for (register budget 512 / this)
#ifndef FMX_SPARSE_WAVES
#define FMX_SPARSE_WAVES 4
#endif
// ... and the fp32 dense-tile form: its FTRL / TDAP instances sit at 127-133 VGPRs, on the edge between 4 and 3 waves per SIMD, and which
// side they fall on changed with unrelated edits (round 3: FTRL k = 64 went 0.94 -> 1.08 ms per tile at 3).  Pinned at 4; the few
// registers over go to scratch outside the entry loop.
#ifndef FMX_DENSE_WAVES
#define FMX_DENSE_WAVES 4
#endif
#ifndef FMX_U_LARGE
#define FMX_U_LARGE 4
#endif
#ifndef FMX_U_SMALL
#define FMX_U_SMALL 8  // (4 / 8 / 16: 16.0 / 14.5 / 14.1 us per 1 024-row step, 21.9 / 21.4 / 21.2 at 4 096, equal at 16 384, 56.0 / 56.7 / 58.0 at 24 576)
#endif

namespace fmx {

// table-row gather; FMX_NT_GATHER=1 marks it non-temporal (experiment: does the fill granularity / L2 policy change?)
#ifndef FMX_NT_GATHER
#define FMX_NT_GATHER 0
#endif
typedef float fx4 __attribute__((ext_vector_type(4)));
typedef double dx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 gather_row(const float* p) {
#if FMX_NT_GATHER
  fx4 v = __builtin_nontemporal_load(reinterpret_cast<const fx4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ double2 gather_row(const double* p) {
#if FMX_NT_GATHER
  dx2 v = __builtin_nontemporal_load(reinterpret_cast<const dx2*>(p));
  return make_double2(v.x, v.y);
#else
  return *reinterpret_cast<const double2*>(p);
#endif
}

// Bounds-checked gathers through a buffer descriptor (raw_buffer_load): an offset at or beyond the descriptor's size
// returns zeros WITHOUT a memory request, so the padding slots of a short list cost nothing -- with flat loads every slot of
// the unrolled batch is a request, and lists average 8 entries against batches of 4 (measured: a quarter of phase 2's requests).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t BUF_SKIP = 0x80000000u;  // beyond any table the buffer path is used for (< 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_row(__amdgpu_buffer_rsrc_t r, uint32_t off, float) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ double2 buf_row(__amdgpu_buffer_rsrc_t r, uint32_t off, double) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}
__device__ __forceinline__ float buf_elem(__amdgpu_buffer_rsrc_t r, uint32_t off, float) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ double buf_elem(__amdgpu_buffer_rsrc_t r, uint32_t off, double) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}

template <typename T> struct Slice;
template <> struct Slice<float> { using vec = float4; static constexpr int N = 4; };
template <> struct Slice<double> { using vec = double2; static constexpr int N = 2; };

// Table strides, as shifts.  The w-in-row layout (fmx_internal.h: w_in_row) exists for fp32 rows of at most 16 padded factors (LPR <= 4)
// only: there the shifts are launch arguments.  Everywhere else -- fp64 tables, wider rows -- they are the compile-time ones (rows KP
// apart, w a table of its own), so that those kernels keep the registers they had before the layout existed: the wide-row kernels are
// register-bound, and two more VGPRs took the FTRL k = 64 kernel from 4 to 3 waves per SIMD (0.94 -> 1.08 ms per tile, round 3).
template <typename ST, int LPR> struct RowStride {
  static constexpr bool DYN = sizeof(ST) == 4 && LPR <= WIR_MAX_KP / 4;
  static constexpr int KSH = __builtin_ctz((unsigned)(LPR * Slice<ST>::N));
  static __device__ __forceinline__ int v(int vsh) { if constexpr (DYN) return vsh; else return KSH; }
  static __device__ __forceinline__ int w(int wsh) { if constexpr (DYN) return wsh; else return 0; }
};

__device__ __forceinline__ void slice_get(const float4& v, double* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void slice_get(const double2& v, double* o) { o[0] = v.x; o[1] = v.y; }
// the other direction; the second argument only selects the element type
__device__ __forceinline__ float4 slice_make(const double* v, float) { return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); }
__device__ __forceinline__ double2 slice_make(const double* v, double) { return make_double2(v[0], v[1]); }
__device__ __forceinline__ float4 slice_make(const float* v, float) { return make_float4(v[0], v[1], v[2], v[3]); }

// solver/SGD_Learner.h:180-191 (copy in FTRL_Learner.h:204-215)
__device__ __forceinline__ double grad_mult(const Hyper& h, double y_hat, float y) {
  if (h.task == FMX_TASK_REGRESSION) {
    y_hat = fmin(h.max_t, y_hat);
    y_hat = fmax(h.min_t, y_hat);
    return -((double)y - y_hat);
  }
  return -(double)y * (1.0 - 1.0 / (1.0 + exp(-(double)y * y_hat)));
}

__device__ __forceinline__ double link_apply(const Hyper& h, double y_hat, int link, const double* __restrict__ pn_y) {
  if (link == FMX_LINK_LOGISTIC) return 1.0 / (1.0 + exp(-y_hat));  // core/Model.h:173-178
  if (link == FMX_LINK_PROBIT) return fast_pnorm(pn_y, y_hat);       // core/Model.h:166-171 (MCMC / ALS models)
  if (link == FMX_LINK_CLAMP) {                                       // src/FM.cpp:204-210
    if (y_hat < h.min_t) return h.min_t;
    if (y_hat > h.max_t) return h.max_t;
  }
  return y_hat;
}

// coalesced copy of `cnt` (id, x) entries starting at absolute offset c0 into LDS
template <bool BATCHED, int WGT = WG_THREADS>
__device__ __forceinline__ void stage_entries(uint2* stage, const uint32_t* __restrict__ ids,
                                              const float* __restrict__ xs, int64_t c0, int cnt, int unit) {
  // (non-temporal loads for these read-once streams were tried: phase 1 1.7 % slower, the forward-only pass 2 % faster)
  // All of a thread's loads are issued before the first LDS store: a chunk is up to 8 entries per thread, and a plain
  // load -> store loop pays one memory round trip PER ENTRY when nothing else hides it (small steps: 16 workgroups on the chip).
  // Phase 2 keeps the plain loop: its state variants (FTRL) are register-bound and lose 19 % to the eight extra pairs.
  if constexpr (!BATCHED) {
    for (int i = threadIdx.x; i < cnt; i += WGT)
      stage[i] = make_uint2(ids[c0 + i], unit ? 0x3f800000u : __float_as_uint(xs[c0 + i]));
    return;
  }
  constexpr int PER = STAGE_ENTRIES / WG_THREADS  /* a chunk is PER * WGT entries */;
  uint32_t id[PER], xb[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = threadIdx.x + u * WGT;
    const bool in = i < cnt;
    id[u] = in ? ids[c0 + i] : 0u;
    xb[u] = unit ? 0x3f800000u : (in ? __float_as_uint(xs[c0 + i]) : 0u);  // one-hot data: half the stream
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = threadIdx.x + u * WGT;
    if (i < cnt) stage[i] = make_uint2(id[u], xb[u]);
  }
}

// ---- the gradient multiplier travels INSIDE the S row (fp32 tables) ------------------------------------------------------
// Phase 2 needs, per entry, the row's factor sums (an S row: one request) and the row's multiplier (4 bytes from a side table:
// a second request, measured at 11 % of phase 2).  The multiplier is therefore folded into the S row itself:
//   EMBED_PAD  (k < kp): the row has padding slots; slot kp - 1 carries the multiplier as is.
//   EMBED_BITS (k == kp >= 16): its 32 bits replace the lowest mantissa bit(s) of the row's floats -- 2 bits of each of 16
//              floats, 1 bit of each of 32 (for kp = 64: of the first 32) -- after rounding the float to that width; S is an
//              intermediate of the step, the factor sums keep 22 or 23 mantissa bits (relative 2.4e-7 / 1.2e-7, against the
//              1e-5 the fp32 state is held to), the multiplier keeps all of its bits.
// The side table is still written (long lists and the fp64 tables read it).
enum EmbedMode : int { EMBED_NONE = 0, EMBED_PAD = 1, EMBED_BITS = 2 };
// Measured (profiles/r02_embed_ab.txt): kp = 16 phase 2 0.1633 -> 0.1583 ms per tile; kp = 32 0.231 -> 0.281 and kp = 64 0.750 -> 0.895
// (SLOWER: a 128- or 256-byte row is two to four sectors, the side request a third to a fifth of the traffic instead of half, and the
// bits have to be collected over 8 or 16 lanes).  So: rows of 64 bytes only; FMX_EMBED_MAX_KP overrides for A/B runs.
inline int embed_max_kp() { static const int v = [] { const char* s = getenv("FMX_EMBED_MAX_KP"); return s ? atoi(s) : 16; }(); return v; }
inline int embed_mode(int k, int kp, bool fp32) {
  if (!fp32 || kp > embed_max_kp()) return EMBED_NONE;
  return k < kp ? EMBED_PAD : (kp >= 16 ? EMBED_BITS : EMBED_NONE);
}

template <int LPR>
__device__ __forceinline__ float4 embed_store(float4 v, int lig, float mult, int mode) {
  constexpr int KP = LPR * 4;
  if (mode == EMBED_PAD) {
    if (lig == LPR - 1) v.w = mult;
  } else if (mode == EMBED_BITS) {
    constexpr int B = KP == 16 ? 2 : 1;              // bits per float
    constexpr uint32_t MASK = (1u << B) - 1u, HALF = 1u << (B - 1);
    if (KP < 64 || lig < 8) {
      const uint32_t m = __float_as_uint(mult);
      const int p0 = B * (lig * 4);
      uint32_t u;
      u = ((__float_as_uint(v.x) + HALF) & ~MASK) | ((m >> (p0)) & MASK); v.x = __uint_as_float(u);
      u = ((__float_as_uint(v.y) + HALF) & ~MASK) | ((m >> (p0 + B)) & MASK); v.y = __uint_as_float(u);
      u = ((__float_as_uint(v.z) + HALF) & ~MASK) | ((m >> (p0 + 2 * B)) & MASK); v.z = __uint_as_float(u);
      u = ((__float_as_uint(v.w) + HALF) & ~MASK) | ((m >> (p0 + 3 * B)) & MASK); v.w = __uint_as_float(u);
    }
  }
  return v;
}

// the inverse, on a gathered slice: returns the multiplier to every lane of the group and leaves the pure factor sums in v
template <int LPR>
__device__ __forceinline__ float embed_take(float4& v, int lig, int mode) {
  constexpr int KP = LPR * 4;
  if (mode == EMBED_PAD) {
    const float m = __shfl(v.w, (int)(threadIdx.x & 63) - lig + (LPR - 1));
    if (lig == LPR - 1) v.w = 0.f;
    return m;
  }
  constexpr int B = KP == 16 ? 2 : 1;
  constexpr uint32_t MASK = (1u << B) - 1u;
  uint32_t part = 0;
  if (KP < 64 || lig < 8) {
    const int p0 = B * (lig * 4);
    const uint32_t ux = __float_as_uint(v.x), uy = __float_as_uint(v.y), uz = __float_as_uint(v.z), uw = __float_as_uint(v.w);
    part = ((ux & MASK) << p0) | ((uy & MASK) << (p0 + B)) | ((uz & MASK) << (p0 + 2 * B)) | ((uw & MASK) << (p0 + 3 * B));
    v.x = __uint_as_float(ux & ~MASK); v.y = __uint_as_float(uy & ~MASK); v.z = __uint_as_float(uz & ~MASK); v.w = __uint_as_float(uw & ~MASK);
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) part |= (uint32_t)__shfl_xor((int)part, off);
  return __uint_as_float(part);
}
__device__ __forceinline__ double embed_take_none(double2&, int, int) { return 0.0; }

// Diagnostic build only (-DFMX_TRACE, profiles/trace_rows_forward.py): 100 MHz stamps from one workgroup of phase 1.  The product
// build compiles none of it.
#ifdef FMX_TRACE
__device__ unsigned long long fmx_trace_buf[16];
#define FMX_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) fmx_trace_buf[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int fmx_debug_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fmx_trace_buf), sizeof(fmx_trace_buf)) == hipSuccess ? 0 : 1;
}
#else
#define FMX_STAMP(i)
#endif

// ------------------------------------------------------------------------------------------------ phase 1
// SPLIT > 1 (the smallest steps, one-wave workgroups): SPLIT lane groups share a row, group `sub` taking its entries sub, sub + SPLIT,
// ... -- a 30-entry row is then ONE round of eight gathers per group instead of four -- and the partial sums meet in a fixed
// butterfly ((0+1)+(2+3)).  Deterministic; the association of a row's sums then differs from the large-step form by design
// (both are inside the 1e-5 bar and the fp64-state 1e-11 one: the sums are fp64).
#ifndef FMX_ROWS_WAVES
#define FMX_ROWS_WAVES_ATTR
#else
#define FMX_ROWS_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(FMX_ROWS_WAVES, FMX_ROWS_WAVES)))
#endif
template <typename T, int LPR, bool TRAIN, int WGT, int SPLIT = 1>
__global__ __launch_bounds__(WGT) FMX_ROWS_WAVES_ATTR void fm_rows_forward_k(RowsArgs a, Hyper h) {
  using vec_t = typename Slice<T>::vec;
  constexpr int VEC = Slice<T>::N;
  constexpr int KP = LPR * VEC;
  constexpr int RPW = WGT / (LPR * SPLIT);  // rows (lists) per workgroup
  constexpr int CHUNK = STAGE_ENTRIES / WG_THREADS * WGT;  // entries staged at a time
  constexpr int RU = WGT == 64 ? FMX_U_SMALL : FMX_U_LARGE;  // entries whose gathers are in flight together, per lane group
  __shared__ uint2 stage[CHUNK + RU * SPLIT];  // + RU * SPLIT: the gather rounds read whole groups of RU entries
  __shared__ double red[TRAIN ? RPW : 1];

  FMX_STAMP(0);
  const int tid = threadIdx.x;
  const int gid = (tid / LPR) / SPLIT;  // row of the workgroup
  const int sub = (tid / LPR) % SPLIT;  // this lane group's part of the row
  const int lig = tid % LPR;
  const int64_t R0 = (int64_t)blockIdx.x * RPW;
  const int64_t R1 = (R0 + RPW < a.nrows) ? R0 + RPW : a.nrows;
  const int64_t lo = a.row_ptr[a.r0 + R0];
  const int64_t hi = a.row_ptr[a.r0 + R1];
  const int64_t row = R0 + gid;
  const bool have = row < a.nrows;
  int64_t ta = 0, tb = 0;
  if (have) {
    ta = a.row_ptr[a.r0 + row];
    tb = a.row_ptr[a.r0 + row + 1];
  }
  const T* __restrict__ Vt = reinterpret_cast<const T*>(a.V) + lig * VEC;
  const T* __restrict__ wt = a.w ? reinterpret_cast<const T*>(a.w) : reinterpret_cast<const T*>(a.V);  // always readable
  const bool k1 = h.k1 != 0;
  // a launch that only wants the per-row factor sums (the q table of the ALS / Gibbs V sweep: no y_hat) has no use for w: every lane then reads w of feature 0 --
  // one hot word instead of a second random request per entry (the load itself stays: unconditional, see below)
  const uint32_t wsel = a.no_w ? 0u : 0xFFFFFFFFu;

  double s[VEC], q[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s[i] = 0.0; q[i] = 0.0; }
  double lin = (h.k0 && sub == 0) ? a.scal[SC_W0] : 0.0;  // core/Model.h:77-78

  for (int64_t c0 = lo; c0 < hi; c0 += CHUNK) {
    const int cnt = (hi - c0 < CHUNK) ? (int)(hi - c0) : CHUNK;
    FMX_STAMP(1);
    stage_entries<true, WGT>(stage, a.col, a.val, c0, cnt, a.unit);
    __syncthreads();
    FMX_STAMP(2);
    const int64_t b = ta > c0 ? ta : c0;
    const int64_t e = tb < c0 + cnt ? tb : c0 + cnt;
    for (int64_t t = b + sub; t < e; t += RU * SPLIT) {
      const int o = (int)(t - c0);
      // Straight-line on purpose: every LDS read, then every gather, then the arithmetic.  A `cond ? load : constant` here
      // compiles to a branch around the load with a wait for ALL outstanding loads at the join -- the rounds of one wave
      // then cost one memory round trip per ENTRY instead of one per RU entries (seen in the ISA; 17 of a small step's
      // 20 us).  So: read past the row's end (the stage array is padded), select afterwards; w is read even when the
      // model has no linear term (wt then points at valid memory) and dropped by the select below.
      uint2 en[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) en[u] = stage[o + u * SPLIT];
#pragma unroll
      for (int u = 1; u < RU; ++u)
        if (t + u * SPLIT >= e) en[u] = make_uint2(en[0].x, 0u);  // x = +0.0f pads
      vec_t vv[RU];
      T wv[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        vv[u] = gather_row(Vt + ((size_t)en[u].x << RowStride<T, LPR>::v(a.vsh)));
        wv[u] = wt[(size_t)(en[u].x & wsel) << RowStride<T, LPR>::w(a.wsh)];  // (w-in-row layout: the same 128-byte line as the V row)
        if constexpr (WGT != 64) {
          // large steps over a cache-sized table: let these land before the next entry's requests go out (see FMX_U_LARGE above)
          if (a.serial) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {  // nonzeros in row order: same association as core/Model.h:83-97
        const double x = (double)__uint_as_float(en[u].y);
        if (k1) lin += (double)wv[u] * x;
        double vf[VEC];
        slice_get(vv[u], vf);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const double tmp = vf[i] * x;
          s[i] += tmp;
          q[i] += tmp * tmp;
        }
      }
    }
    FMX_STAMP(3);
    __syncthreads();
    FMX_STAMP(4);
  }

  if constexpr (SPLIT > 1) {  // the row's SPLIT parts, in a fixed order
#pragma unroll
    for (int off = LPR; off < LPR * SPLIT; off <<= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s[i] += __shfl_xor(s[i], off); q[i] += __shfl_xor(q[i], off); }
      lin += __shfl_xor(lin, off);
    }
  }
  double pair = 0.0;
#pragma unroll
  for (int i = 0; i < VEC; ++i) pair += 0.5 * (s[i] * s[i] - q[i]);  // core/Model.h:100
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) pair += __shfl_xor(pair, off);
  const double y_hat = lin + pair;

  if constexpr (TRAIN) {
    double mult = 0.0;
    if (have && sub == 0) {
      mult = grad_mult(h, y_hat, a.y[a.r0 + row]);
      vec_t srow = slice_make(s, T());
      if constexpr (sizeof(T) == 4) srow = embed_store<LPR>(srow, lig, (float)mult, a.embed);  // the multiplier rides in the S row
      *reinterpret_cast<vec_t*>(reinterpret_cast<T*>(a.S) + (size_t)row * KP + lig * VEC) = srow;
      if (lig == 0) reinterpret_cast<T*>(a.amul)[row] = (T)mult;
    }
    FMX_STAMP(5);
    if (lig == 0 && sub == 0) red[gid] = mult;
    __syncthreads();
    FMX_STAMP(6);
    if (tid == 0) {  // fixed-order partial sums for the w0 step
      double g0 = 0.0, q0 = 0.0;
      for (int i = 0; i < RPW; ++i) { g0 += red[i]; q0 += red[i] * red[i]; }
      a.partials[2 * (size_t)blockIdx.x] = g0;
      a.partials[2 * (size_t)blockIdx.x + 1] = q0;
    }
    FMX_STAMP(7);
  } else {
    if (have && lig == 0 && sub == 0 && a.yhat) a.yhat[row] = link_apply(h, y_hat, a.link, a.pn_y);
    if constexpr (sizeof(T) == 8) {  // fp64 tables: optionally the per-row factor sums q[row][f] = sum_j x_j v_jf (ALS sweeps)
      if (have && sub == 0 && a.qout) {
        if (a.qout_t > 0) { a.qout[(size_t)(lig * VEC) * a.qout_t + row] = s[0]; a.qout[(size_t)(lig * VEC + 1) * a.qout_t + row] = s[1]; }
        else *reinterpret_cast<double2*>(a.qout + (size_t)row * KP + lig * VEC) = make_double2(s[0], s[1]);
      }
    }
  }
}

// ---- phase 1 on rows of differing lengths (opt-in: FMX_ROWS_PULL=1) ---------------------------------------------------------------------
// fm_rows_forward_k gives every lane group ONE row of its workgroup; a group idles once its row is done, and the workgroup's slot is held until its
// longest row is.  On SURVEY 8(d)'s ragged law (row lengths Poisson(30) clipped to [1, 64]) phase 1 takes 18 % longer than on rows of exactly 30
// entries (bench `value_ragged_rows`; profiles/r04_ragged_probe.txt: 0.169 ms per tile for lengths 30..30, 0.178 for 25..35, 0.196 for the Poisson
// law).  Two balancing forms were built, both bit for bit the static kernel, and NEITHER pays (profiles/r04_ragged_probe2.txt):
//   * dealing a workgroup's rows to its lane groups by descending length (every wave then holds rows of similar length): 0.1967 against 0.1943 ms --
//     the kernel is bound by requests in flight, not by instruction issue, and a workgroup still lives as long as its longest row (removed again);
//   * THIS kernel: the lane groups of a workgroup PULL rows from a counter (256 rows for 64 groups at k = 16), so every group walks entries until
//     the workgroup's rows are exhausted.  Rows pulled out of order cannot use the coalesced LDS stage: lane l of a group loads entry t + l of a round
//     itself and the group shares the round by shuffles -- two more gather-shaped instructions per round, which cost more (rows of exactly 30 entries:
//     0.188 against 0.169 ms) than the balance returns (Poisson law: 0.221 against 0.194).  Kept opt-in with its bitwise test as the record.
// What would pay is the flat form (a workgroup's staged entries cut evenly over the lane groups, row boundaries by ballot, a segmented combine in a fixed
// order): not built -- it changes the association of a row's sums (the bits), for 8 % of a ragged step.
// A row is walked by ONE lane group, entries in row order, fp64 accumulators, every per-row result stored under the row's own index; the w0 partial
// sums keep the static kernel's granularity (GROUPS rows each, in row order).
template <typename T, int LPR, bool TRAIN>
__global__ __launch_bounds__(WG_THREADS) void fm_rows_forward_dyn_k(RowsArgs a, Hyper h) {
  using vec_t = typename Slice<T>::vec;
  constexpr int VEC = Slice<T>::N;
  constexpr int KP = LPR * VEC;
  constexpr int GROUPS = WG_THREADS / LPR;   // lane groups = rows of one static workgroup = rows of one w0 partial sum
  constexpr int RPW = 4 * GROUPS;            // rows one workgroup pulls from
  constexpr int RU = FMX_U_LARGE;            // entries per round
  constexpr int PER = (RU + LPR - 1) / LPR;  // entries a lane loads per round
  __shared__ int next_row;
  __shared__ double red[TRAIN ? RPW : 1];
  const int tid = threadIdx.x;
  const int grp = tid / LPR, lig = tid % LPR;
  const int lane0 = (tid & 63) - lig;        // first lane of this group inside its wave
  const int64_t R0 = (int64_t)blockIdx.x * RPW;
  const int rows_here = (int)((R0 + RPW < a.nrows ? R0 + RPW : a.nrows) - R0);
  if (tid == 0) next_row = GROUPS;
  if (TRAIN) for (int i = tid; i < RPW; i += WG_THREADS) red[i] = 0.0;
  __syncthreads();
  const T* __restrict__ Vt = reinterpret_cast<const T*>(a.V) + lig * VEC;
  const T* __restrict__ wt = a.w ? reinterpret_cast<const T*>(a.w) : reinterpret_cast<const T*>(a.V);
  const bool k1 = h.k1 != 0;
  const int64_t last = a.row_ptr[a.r0 + R0 + rows_here] - 1;   // last entry of the workgroup's rows (clamp for the loads past a row's end)
  int r = grp;
  while (r < rows_here) {
    const int64_t ta = a.row_ptr[a.r0 + R0 + r], tb = a.row_ptr[a.r0 + R0 + r + 1];
    double s[VEC], q[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s[i] = 0.0; q[i] = 0.0; }
    double lin = h.k0 ? a.scal[SC_W0] : 0.0;  // core/Model.h:77-78
    // this lane's share of a round: entries t + lig, t + lig + LPR, ... (index clamped, value selected afterwards: no load under a condition)
    auto load_round = [&](int64_t t, uint32_t* id, uint32_t* xb) {
#pragma unroll
      for (int v = 0; v < PER; ++v) {
        const int64_t at = t + lig + v * LPR;
        const int64_t ac = at <= last ? (at >= 0 ? at : 0) : (last >= 0 ? last : 0);
        id[v] = a.col[ac];
        xb[v] = a.unit ? 0x3f800000u : __float_as_uint(a.val[ac]);
      }
    };
    uint32_t cid[PER], cxb[PER], nid[PER], nxb[PER];
    load_round(ta, cid, cxb);
    for (int64_t t = ta; t < tb; t += RU) {
      load_round(t + RU, nid, nxb);            // the next round's entries are on their way while this round gathers
      uint2 en[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        en[u].x = (uint32_t)__shfl((int)cid[u / LPR], lane0 + (u % LPR));
        en[u].y = (uint32_t)__shfl((int)cxb[u / LPR], lane0 + (u % LPR));
      }
#pragma unroll
      for (int u = 1; u < RU; ++u)
        if (t + u >= tb) en[u] = make_uint2(en[0].x, 0u);  // x = +0.0f pads
      vec_t vv[RU];
      T wv[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        vv[u] = gather_row(Vt + ((size_t)en[u].x << RowStride<T, LPR>::v(a.vsh)));
        wv[u] = wt[(size_t)en[u].x << RowStride<T, LPR>::w(a.wsh)];
        if (a.serial) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {  // nonzeros in row order: same association as core/Model.h:83-97
        const double x = (double)__uint_as_float(en[u].y);
        if (k1) lin += (double)wv[u] * x;
        double vf[VEC];
        slice_get(vv[u], vf);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const double tmp = vf[i] * x;
          s[i] += tmp;
          q[i] += tmp * tmp;
        }
      }
#pragma unroll
      for (int v = 0; v < PER; ++v) { cid[v] = nid[v]; cxb[v] = nxb[v]; }
    }
    double pair = 0.0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) pair += 0.5 * (s[i] * s[i] - q[i]);  // core/Model.h:100
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) pair += __shfl_xor(pair, off);
    const double y_hat = lin + pair;
    const int64_t row = R0 + r;
    if constexpr (TRAIN) {
      const double mult = grad_mult(h, y_hat, a.y[a.r0 + row]);
      vec_t srow = slice_make(s, T());
      if constexpr (sizeof(T) == 4) srow = embed_store<LPR>(srow, lig, (float)mult, a.embed);
      *reinterpret_cast<vec_t*>(reinterpret_cast<T*>(a.S) + (size_t)row * KP + lig * VEC) = srow;
      if (lig == 0) { reinterpret_cast<T*>(a.amul)[row] = (T)mult; red[r] = mult; }
    } else {
      if (lig == 0 && a.yhat) a.yhat[row] = link_apply(h, y_hat, a.link, a.pn_y);
      if constexpr (sizeof(T) == 8) {
        if (a.qout) {
          if (a.qout_t > 0) { a.qout[(size_t)(lig * VEC) * a.qout_t + row] = s[0]; a.qout[(size_t)(lig * VEC + 1) * a.qout_t + row] = s[1]; }
          else *reinterpret_cast<double2*>(a.qout + (size_t)row * KP + lig * VEC) = make_double2(s[0], s[1]);
        }
      }
    }
    int nr = 0;
    if (lig == 0) nr = atomicAdd(&next_row, 1);
    r = __shfl(nr, lane0);
  }
  if constexpr (TRAIN) {
    __syncthreads();
    if (tid < RPW / GROUPS && tid * GROUPS < rows_here) {   // the static kernel's partial sums: GROUPS rows each, in row order
      double g0 = 0.0, q0 = 0.0;
      for (int i = tid * GROUPS; i < (tid + 1) * GROUPS; ++i) { g0 += red[i]; q0 += red[i] * red[i]; }
      const size_t at = (size_t)blockIdx.x * (RPW / GROUPS) + tid;
      a.partials[2 * at] = g0;
      a.partials[2 * at + 1] = q0;
    }
  }
}

// ---- phase 1, flat form (opt-in: FMX_ROWS_FLAT=1): the entries of a block of rows as ONE stream, cut evenly over the lane groups ------------------------
// (north_star: "entries of a workgroup's rows as one stream ... row boundaries ... segmented combine"; the reference's loop is core/Model.h:83-97.)
// The static kernel gives lane group g row g of its block: on SURVEY 8(d)'s ragged law (Poisson(30) clipped to [1, 64]) the groups idle for 30 % of the
// block's life and phase 1 takes 18 % longer than on rows of exactly 30 entries.  Here the block's staged entries [0, cnt) are cut into G equal ranges;
// group g walks range g whatever rows it crosses, in rounds of RU gathers like the static kernel, and leaves one PIECE (the factor sums of a run of one
// row's entries) per row it touches in LDS: the FIRST piece of its range under its own index (it may continue a row that an earlier group began), every
// later piece -- which starts its row -- under the row's index.  After a barrier group g adds up the pieces of row g in entry order (the piece that starts
// the row, then the first pieces of the following groups while they still belong to it) and finishes the row exactly as the static kernel does.  No
// atomics; the order is fixed by the matrix.
//   * A row's sums associate differently from the static kernel's one sequential sum (pieces are summed first): same 1e-5 / 1e-11 bars against the
//     oracle, other last bits -- so the form could only be chosen per MATRIX by a rule on its row lengths, never by timing.
//   * The blocks are those of the MATRIX (rows b * G .. b * G + G - 1, whatever row the launch starts on) and a block is always cut as a whole, so a
//     row's bits depend on the matrix alone -- not on the launch, tile, step, rank share or schedule that reaches it.  A launch that starts or ends
//     inside a block walks the whole block and stores only its own rows.
// MEASURED (profiles/r04_ragged_forms.txt) and therefore NOT the default: perfectly balanced, and slower -- 0.218 against 0.193 ms per 262 144-row tile on
// the Poisson law, 0.217 against 0.172 on lengths 25..35, and equal (0.162) only where every range is exactly one row.  With its boundary handling switched
// off entirely it is as slow: the cost is not the pieces.  What the static kernel has and this walk gives up is that the lane groups of a wave sit at the
// same position of their column-sorted rows, so one gather instruction's rows come from one narrow band of the table (one column per stratum 0.149 ms,
// i.i.d. sorted columns 0.162, ragged rows 0.193, no common position 0.217).  The same record holds five more forms that were tried and removed: one-wave
// workgroups on large steps, two entries outstanding per lane group, no LDS stage at all, 4 / 6 / 8 waves per SIMD, and phase 1 without its w requests
// (which changes nothing: the second request per nonzero is free under the serial schedule).
template <typename T, int LPR, bool TRAIN>
__global__ __launch_bounds__(WG_THREADS) void fm_rows_forward_flat_k(RowsArgs a, Hyper h) {
  using vec_t = typename Slice<T>::vec;
  constexpr int VEC = Slice<T>::N;
  constexpr int KP = LPR * VEC;
  constexpr int G = WG_THREADS / LPR;  // lane groups = rows of a block
  constexpr int CHUNK = STAGE_ENTRIES;
  constexpr int RU = FMX_U_LARGE;
  constexpr int NP = VEC + 1;          // a piece, per lane: VEC factor sums and the lane's part of sum (v x)^2
  __shared__ uint2 stage[CHUNK + RU];
  __shared__ double piece[2][NP][WG_THREADS];  // [0][.][g * LPR + lane]: group g's first piece of the chunk; [1][.][r * LPR + lane]: the piece that starts row r
  __shared__ double plin[2][G];                // the pieces' linear parts
  __shared__ int rp[G + 1];                    // the block's row offsets, relative to its first entry
  __shared__ double red[TRAIN ? G : 1];

  const int tid = threadIdx.x;
  const int grp = tid / LPR, lig = tid % LPR;
  const int64_t G0 = (a.r0 / G + (int64_t)blockIdx.x) * G;  // the block's first row, in the matrix
  const int64_t G1 = G0 + G < a.nmat ? G0 + G : a.nmat;
  const int rows_here = (int)(G1 - G0);
  const int64_t lo = a.row_ptr[G0], hi = a.row_ptr[G1];
  for (int i = tid; i <= G; i += WG_THREADS) rp[i] = (int)(a.row_ptr[G0 + (i < rows_here ? i : rows_here)] - lo);
  const int64_t row = G0 + grp;          // the row this lane group finishes
  const int64_t lr = row - a.r0;         // ... and its index in the launch
  const bool have = grp < rows_here && lr >= 0 && lr < a.nrows;
  const T* __restrict__ Vt = reinterpret_cast<const T*>(a.V) + lig * VEC;
  const T* __restrict__ wt = a.w ? reinterpret_cast<const T*>(a.w) : reinterpret_cast<const T*>(a.V);  // always readable
  const bool k1 = h.k1 != 0;

  double s[VEC], qs = 0.0;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.0;
  double lin = h.k0 ? a.scal[SC_W0] : 0.0;  // core/Model.h:77-78

  for (int64_t c0 = lo; c0 < hi; c0 += CHUNK) {
    const int cnt = (hi - c0 < CHUNK) ? (int)(hi - c0) : CHUNK;
    stage_entries<true>(stage, a.col, a.val, c0, cnt, a.unit);
    __syncthreads();  // (the first one also publishes rp)
    const int base = (int)(c0 - lo);
    const int per = (cnt + G - 1) / G;
    const int p0 = grp * per;
    const int p1 = p0 + per < cnt ? p0 + per : cnt;
    if (p0 < p1) {
      int r = 0;
      {  // the row holding entry p0: the last r with rp[r] <= base + p0 (rp[0] = 0 <= it < rp[rows_here])
        int hi_i = rows_here;
        const int x = base + p0;
        while (hi_i - r > 1) {
          const int mid = (r + hi_i) >> 1;
          if (rp[mid] <= x) r = mid; else hi_i = mid;
        }
      }
      int rend = rp[r + 1] - base;  // where row r ends, in chunk positions
      int slot = grp;               // where the running piece goes: this group's first piece under the group's index ...
      int kind = 0;
      double ps[VEC], pq = 0.0, pl = 0.0;
#pragma unroll
      for (int i = 0; i < VEC; ++i) ps[i] = 0.0;
      for (int t = p0; t < p1; t += RU) {
        // straight-line as in fm_rows_forward_k: every LDS read, every gather, then the arithmetic (no load under a condition)
        uint2 en[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) en[u] = stage[t + u];
#pragma unroll
        for (int u = 1; u < RU; ++u)
          if (t + u >= p1) en[u] = make_uint2(en[0].x, 0u);
        vec_t vv[RU];
        T wv[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          vv[u] = gather_row(Vt + ((size_t)en[u].x << RowStride<T, LPR>::v(a.vsh)));
          wv[u] = wt[(size_t)en[u].x << RowStride<T, LPR>::w(a.wsh)];
          if (a.serial) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          if (t + u < p1) {
            if (t + u >= rend) {  // row r ends before this entry: its piece is complete
#pragma unroll
              for (int i = 0; i < VEC; ++i) { piece[kind][i][slot * LPR + lig] = ps[i]; ps[i] = 0.0; }
              piece[kind][VEC][slot * LPR + lig] = pq;
              if (lig == 0) plin[kind][slot] = pl;
              pq = 0.0; pl = 0.0;
              do { ++r; rend = rp[r + 1] - base; } while (t + u >= rend);  // (rows without entries are passed over)
              slot = r; kind = 1;   // ... every later one starts its row: under the row's index
            }
            const double x = (double)__uint_as_float(en[u].y);
            if (k1) pl += (double)wv[u] * x;
            double vf[VEC];
            slice_get(vv[u], vf);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const double tmp = vf[i] * x;
              ps[i] += tmp;
              pq += tmp * tmp;
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) piece[kind][i][slot * LPR + lig] = ps[i];
      piece[kind][VEC][slot * LPR + lig] = pq;
      if (lig == 0) plin[kind][slot] = pl;
    }
    __syncthreads();
    if (grp < rows_here) {  // the pieces of row grp inside this chunk, in entry order
      const int b = rp[grp] - base > 0 ? rp[grp] - base : 0;
      const int e = rp[grp + 1] - base < cnt ? rp[grp + 1] - base : cnt;
      if (b < e) {
        int g = b / per;
        const int gB = (e - 1) / per;
        if (b > g * per) {  // the row starts inside group g's range: that piece lies under the row's index
#pragma unroll
          for (int i = 0; i < VEC; ++i) s[i] += piece[1][i][grp * LPR + lig];
          qs += piece[1][VEC][grp * LPR + lig];
          lin += plin[1][grp];
          ++g;
        }
        for (; g <= gB; ++g) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) s[i] += piece[0][i][g * LPR + lig];
          qs += piece[0][VEC][g * LPR + lig];
          lin += plin[0][g];
        }
      }
    }
    // (no barrier here: the next chunk's staging writes `stage` only, and its pieces are written after its own first barrier)
  }

  double pair = 0.0;
#pragma unroll
  for (int i = 0; i < VEC; ++i) pair += s[i] * s[i];
  pair = 0.5 * (pair - qs);  // core/Model.h:100
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) pair += __shfl_xor(pair, off);
  const double y_hat = lin + pair;

  if constexpr (TRAIN) {
    double mult = 0.0;
    if (have) {
      mult = grad_mult(h, y_hat, a.y[row]);
      vec_t srow = slice_make(s, T());
      if constexpr (sizeof(T) == 4) srow = embed_store<LPR>(srow, lig, (float)mult, a.embed);
      *reinterpret_cast<vec_t*>(reinterpret_cast<T*>(a.S) + (size_t)lr * KP + lig * VEC) = srow;
      if (lig == 0) reinterpret_cast<T*>(a.amul)[lr] = (T)mult;
    }
    if (lig == 0) red[grp] = mult;
    __syncthreads();
    if (tid == 0) {  // fixed-order partial sums for the w0 step (rows of the block that are not the launch's count as zero)
      double g0 = 0.0, q0 = 0.0;
      for (int i = 0; i < G; ++i) { g0 += red[i]; q0 += red[i] * red[i]; }
      a.partials[2 * (size_t)blockIdx.x] = g0;
      a.partials[2 * (size_t)blockIdx.x + 1] = q0;
    }
  } else {
    if (have && lig == 0 && a.yhat) a.yhat[lr] = link_apply(h, y_hat, a.link, a.pn_y);
    if constexpr (sizeof(T) == 8) {
      if (have && a.qout) {
        if (a.qout_t > 0) { a.qout[(size_t)(lig * VEC) * a.qout_t + lr] = s[0]; a.qout[(size_t)(lig * VEC + 1) * a.qout_t + lr] = s[1]; }
        else *reinterpret_cast<double2*>(a.qout + (size_t)lr * KP + lig * VEC) = make_double2(s[0], s[1]);
      }
    }
  }
}

template <typename T, bool TRAIN>
static int launch_rows_flat(fmx_engine* e, const RowsArgs& a, int kp) {
  constexpr int VEC = Slice<T>::N;
  const int lpr = kp / VEC;
  const int64_t grid = rows_flat_blocks(a.r0, a.nrows, WG_THREADS / lpr);
  if (grid == 0) return FMX_OK;
  FMX_CHECK(grid < (1LL << 31), FMX_ERR_INVALID, "rows_forward: grid too large (%lld)", (long long)grid);
  FMX_CHECK(a.nmat >= a.r0 + a.nrows, FMX_ERR_STATE, "rows_forward (flat): the matrix has %lld rows, the launch ends at %lld", (long long)a.nmat,
            (long long)(a.r0 + a.nrows));
  dim3 g((unsigned)grid), b(WG_THREADS);
#define FMX_FLAT_CASE(L) case L: hipLaunchKernelGGL((fm_rows_forward_flat_k<T, L, TRAIN>), g, b, 0, e->stream, a, e->hyper); break;
  switch (lpr) {
    FMX_FLAT_CASE(1) FMX_FLAT_CASE(2) FMX_FLAT_CASE(4) FMX_FLAT_CASE(8) FMX_FLAT_CASE(16) FMX_FLAT_CASE(32) FMX_FLAT_CASE(64)
    default: FMX_CHECK(false, FMX_ERR_INVALID, "unsupported padded factor count %d", kp);
  }
#undef FMX_FLAT_CASE
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

template <typename T, bool TRAIN>
static int launch_rows_dyn(fmx_engine* e, const RowsArgs& a, int kp) {
  constexpr int VEC = Slice<T>::N;
  const int lpr = kp / VEC;
  const int rpw = 4 * (WG_THREADS / lpr);
  const int64_t grid = (a.nrows + rpw - 1) / rpw;
  if (grid == 0) return FMX_OK;
  FMX_CHECK(grid < (1LL << 31), FMX_ERR_INVALID, "rows_forward: grid too large (%lld)", (long long)grid);
  dim3 g((unsigned)grid), b(WG_THREADS);
#define FMX_DYN_CASE(L) case L: hipLaunchKernelGGL((fm_rows_forward_dyn_k<T, L, TRAIN>), g, b, 0, e->stream, a, e->hyper); break;
  switch (lpr) {
    FMX_DYN_CASE(1) FMX_DYN_CASE(2) FMX_DYN_CASE(4) FMX_DYN_CASE(8) FMX_DYN_CASE(16) FMX_DYN_CASE(32) FMX_DYN_CASE(64)
    default: FMX_CHECK(false, FMX_ERR_INVALID, "unsupported padded factor count %d", kp);
  }
#undef FMX_DYN_CASE
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

template <typename T, bool TRAIN, int WGT, int SPLIT>
static int launch_rows_t(fmx_engine* e, const RowsArgs& a, int kp) {
  constexpr int VEC = Slice<T>::N;
  const int lpr = kp / VEC;
  const int rpw = WGT / (lpr * SPLIT);
  FMX_CHECK(rpw >= 1, FMX_ERR_INVALID, "rows_forward: %d lane groups of %d lanes do not fit %d threads", SPLIT, lpr, WGT);
  const int64_t grid = (a.nrows + rpw - 1) / rpw;
  if (grid == 0) return FMX_OK;
  FMX_CHECK(grid < (1LL << 31), FMX_ERR_INVALID, "rows_forward: grid too large (%lld)", (long long)grid);
  dim3 g((unsigned)grid), b(WGT);
#define FMX_ROWS_CASE(L)                                                                                          \
  case L:                                                                                                         \
    if constexpr (L * SPLIT <= WGT) hipLaunchKernelGGL((fm_rows_forward_k<T, L, TRAIN, WGT, SPLIT>), g, b, 0, e->stream, a, e->hyper); \
    break;
  switch (lpr) {
    FMX_ROWS_CASE(1) FMX_ROWS_CASE(2) FMX_ROWS_CASE(4) FMX_ROWS_CASE(8)
    FMX_ROWS_CASE(16) FMX_ROWS_CASE(32) FMX_ROWS_CASE(64)
    default: FMX_CHECK(false, FMX_ERR_INVALID, "unsupported padded factor count %d", kp);
  }
#undef FMX_ROWS_CASE
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}
template <typename T, bool TRAIN>
static int launch_rows_w(fmx_engine* e, const RowsArgs& a, int kp) {
  if (a.sort_rows && a.wg_threads != 64) return launch_rows_dyn<T, TRAIN>(e, a, kp);   // FMX_ROWS_PULL=1, rows of differing lengths, wide workgroups
  if (a.flat == 1 && a.wg_threads != 64) return launch_rows_flat<T, TRAIN>(e, a, kp);      // FMX_ROWS_FLAT=1, rows of differing lengths (rows_flat), wide workgroups
  if (a.wg_threads == 64) return a.split == 4 ? launch_rows_t<T, TRAIN, 64, 4>(e, a, kp) : launch_rows_t<T, TRAIN, 64, 1>(e, a, kp);
  return launch_rows_t<T, TRAIN, WG_THREADS, 1>(e, a, kp);
}

int launch_rows_forward(fmx_engine* e, const RowsArgs& a_in, bool train, bool fp64_tables) {
  RowsArgs a = a_in;
  FMX_CHECK(a.vs >= (fp64_tables ? e->kp64 : e->kp32) && a.ws >= 1 && (a.vs & (a.vs - 1)) == 0 && (a.ws & (a.ws - 1)) == 0, FMX_ERR_STATE,
            "rows_forward: table strides not set (%d, %d)", a.vs, a.ws);
  {  // kernels of fp64 tables and of rows wider than 16 factors have their strides compiled in (RowStride)
    const int kp = fp64_tables ? e->kp64 : e->kp32;
    FMX_CHECK((!fp64_tables && kp <= WIR_MAX_KP) || (a.vs == kp && a.ws == 1), FMX_ERR_STATE, "rows_forward: strides (%d, %d) on a table of compiled strides (%d, 1)",
              a.vs, a.ws, kp);
  }
  a.vsh = __builtin_ctz((unsigned)a.vs); a.wsh = __builtin_ctz((unsigned)a.ws);
  static const bool embed_ok = [] { const char* v = getenv("FMX_EMBED_MULT"); return !(v && v[0] == '0'); }();
  static const int force = [] { const char* v = getenv("FMX_ROWS_SERIAL"); return v ? atoi(v) : -1; }();
  RowsTune& tu = e->rows_tune;
  // The schedule of one wide launch of `nrows` rows (RowsTune): pinned, decided, or -- for a large launch while the engine is still
  // measuring -- this launch's turn in the alternation; *trial is then its index and the launch is bracketed by two events.
  auto pick = [&](const RowsArgs& r, int* serial, int* trial) -> int {
    *trial = -1;
    *serial = 1;
    if (force >= 0) { *serial = force; return FMX_OK; }
    const bool timed = r.wg_threads != 64 && r.nrows >= RowsTune::MIN_ROWS;
    const int64_t key = r.unit ? 1 : 0;
    if (timed && key != tu.key) {
      // the other kind of data measures again; a half-measured set of events is drained first so that no slot is re-recorded
      // while an earlier recording of it is still pending on the stream
      if (tu.events && tu.launches > 0 && tu.decided < 0) FMX_HIP(hipStreamSynchronize(e->stream));
      tu.key = key; tu.decided = -1; tu.launches = 0;
    }
    if (tu.decided >= 0) { *serial = tu.decided; return FMX_OK; }
    if (!timed) return FMX_OK;
    if (!tu.events) {
      bool ok = true;
      for (auto& ev : tu.ev) ok = ok && hipEventCreate(&ev) == hipSuccess;
      FMX_CHECK(ok, FMX_ERR_HIP, "rows_forward: could not create the tuning events");
      tu.events = true;
    }
    if (tu.launches >= RowsTune::TRIALS) {  // every trial slot is used up and nothing was decided (a timed launch failed on the way): stop measuring
      tu.decided = 1;
      return FMX_OK;
    }
    *trial = tu.launches++;
    *serial = *trial % 2 == 0 ? 1 : 0;
    FMX_HIP(hipEventRecord(tu.ev[2 * *trial], e->stream));
    return FMX_OK;
  };
  // closes trial `trial` (also when its launch failed: the slot's second event is recorded either way, so a later decision never
  // reads a half-recorded pair)
  auto done = [&](int trial) -> int {
    if (trial < 0) return FMX_OK;
    FMX_HIP(hipEventRecord(tu.ev[2 * trial + 1], e->stream));
    if (tu.launches >= RowsTune::TRIALS && tu.decided < 0) {  // the one wait of the measurement
      FMX_HIP(hipEventSynchronize(tu.ev[2 * trial + 1]));
      double ms[2] = {0.0, 0.0};
      for (int i = 2; i < RowsTune::TRIALS; ++i) {
        float t = 0.f;
        FMX_HIP(hipEventElapsedTime(&t, tu.ev[2 * i], tu.ev[2 * i + 1]));
        ms[i % 2 == 0 ? 1 : 0] += t;
      }
      tu.ms[0] = ms[0]; tu.ms[1] = ms[1];
      tu.decided = ms[1] <= ms[0] ? 1 : 0;
    }
    return FMX_OK;
  };
  a.embed = (train && embed_ok) ? embed_mode(e->k, fp64_tables ? e->kp64 : e->kp32, !fp64_tables) : EMBED_NONE;  // the same rule as launch_cols_update
  prof_begin(e, FMX_KERNEL_ROWS_FORWARD);
  int st = FMX_OK;
  if (train) {
    int trial;
    FMX_TRY(pick(a, &a.serial, &trial));
    st = fp64_tables ? launch_rows_w<double, true>(e, a, e->kp64) : launch_rows_w<float, true>(e, a, e->kp32);
    { const int st2 = done(trial); if (st == FMX_OK) st = st2; }
  } else {
    // Forward-only passes over many rows go out as launches of 262 144 rows: measured at configs[1]
    // (profiles/forward_probe.py) such launches run at 0.58 ns/row, 1 M-row launches at 0.69, 4 M-row launches at 0.75 --
    // the same optimum as the training tiles.  (A forward-only process measures its schedule on these launches.)
    const int64_t SLAB = 1 << 18;
    const int kp = fp64_tables ? e->kp64 : e->kp32;
    for (int64_t off = 0; off < a.nrows && st == FMX_OK; off += SLAB) {
      RowsArgs s = a;
      s.r0 = a.r0 + off;
      s.nrows = a.nrows - off < SLAB ? a.nrows - off : SLAB;
      if (a.yhat) s.yhat = a.yhat + off;
      if (a.qout) s.qout = a.qout_t > 0 ? a.qout + off : a.qout + (size_t)off * kp;
      s.wg_threads = rows_wg_threads(a.nrows, kp / (fp64_tables ? 2 : 4));
      s.split = rows_split(a.nrows, kp / (fp64_tables ? 2 : 4));
      int trial;
      FMX_TRY(pick(s, &s.serial, &trial));
      st = fp64_tables ? launch_rows_w<double, false>(e, s, kp) : launch_rows_w<float, false>(e, s, kp);
      { const int st2 = done(trial); if (st == FMX_OK) st = st2; }
    }
  }
  prof_end(e);
  return st;
}

// ------------------------------------------------------------------------------------------------ scalar
// Runs in workgroup 0 of fm_cols_update_k: deterministic reduction of phase 1's per-workgroup partial sums and the w0
// step (SGD_Learner.h:106-109; FTRL_Learner.h:80-86,161).  Scalars are double-buffered: every kernel of a step reads
// `sin` (the step's start state) and only this function writes `sout`; the host flips the two after the launch.
// mode: SCALAR_FUSED reduce + update, SCALAR_PUBLISH reduce only -> exchange-buffer tail, SCALAR_FROM_TAIL update from the
// (all-reduced) tail.
// The row count travels in the exchange tail as two parts, rows = hi * 4096 + lo: each part (and its sum over the ranks)
// stays far below 2^24, so an fp32 all-reduce(sum) carries global batches of up to 2^36 rows exactly.
template <typename ST>
__device__ __forceinline__ void tail_put_rows(ST* gtail, double rows) {
  const double hi = floor(rows / 4096.0);
  gtail[2] = (ST)hi;
  gtail[3] = (ST)(rows - hi * 4096.0);
}
template <typename ST>
__device__ __forceinline__ double tail_get_rows(const ST* gtail) { return (double)gtail[2] * 4096.0 + (double)gtail[3]; }

template <typename ST>
__device__ __forceinline__ void scalar_update(const double* __restrict__ partials, int64_t n_partials, const double* sin,
                                              double* sout, ST* gtail, const Hyper& h, double rows, int mode,
                                              double* sg, double* sq) {
  const int phase = (mode == SCALAR_PUBLISH) ? 1 : (mode == SCALAR_FROM_TAIL) ? 2 : 0;
  double g0 = 0.0, q0 = 0.0;
  if (phase != 2) {
    for (int64_t i = threadIdx.x; i < n_partials; i += WG_THREADS) { g0 += partials[2 * i]; q0 += partials[2 * i + 1]; }
  }
  sg[threadIdx.x] = g0; sq[threadIdx.x] = q0;
  __syncthreads();
  for (int off = WG_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { sg[threadIdx.x] += sg[threadIdx.x + off]; sq[threadIdx.x] += sq[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  g0 = sg[0]; q0 = sq[0];
  if (phase == 1) { gtail[0] = (ST)g0; gtail[1] = (ST)q0; tail_put_rows(gtail, rows); return; }
  if (phase == 2) { g0 = gtail[0]; q0 = gtail[1]; if (rows <= 0.0) rows = tail_get_rows(gtail); }
  for (int i = 0; i < SC_COUNT; ++i) sout[i] = sin[i];
  sout[SC_G0] = g0; sout[SC_Q0] = q0;
  if (h.mean && rows > 0.0) {  // FMX_REDUCE_MEAN: w0 occurs in every example -> one step with the batch-mean multiplier
    g0 /= rows;
    q0 = g0 * g0;
    rows = 1.0;
  }
  const double w0 = sin[SC_W0];
  if (h.kind == UPD_TDAP) {  // solver/TDAP_Learner.h:96-106, :192 with the batch sums
    double u = sin[SC_N0], nu = sin[SC_T_NU], dl = sin[SC_T_DELTA], hh = sin[SC_T_H], z0 = sin[SC_Z0];
    if (h.k0) {
      const double u_new = u + q0;
      nu += g0;
      const double sigma = (sqrt(u_new) - sqrt(u)) / h.alpha_w;
      const double age = rows == 1.0 ? h.egamma : exp(-h.gamma * rows);
      dl = age * (dl + sigma);
      hh = age * (hh + sigma * w0);
      u = u_new;
      z0 = nu - hh;
    }
    sout[SC_N0] = u; sout[SC_T_NU] = nu; sout[SC_T_DELTA] = dl; sout[SC_T_H] = hh; sout[SC_Z0] = z0;
    sout[SC_W0] = -z0 / dl;
  } else if (h.kind == UPD_FTRL) {
    double z0 = sin[SC_Z0], n0 = sin[SC_N0];
    if (h.k0) {  // solver/FTRL_Learner.h:80-86 with the batch sums G0, Q0
      const double n_new = n0 + q0;
      z0 += g0 - w0 * (sqrt(n_new) - sqrt(n0)) / h.alpha_w;
      n0 = n_new;
    }
    sout[SC_Z0] = z0; sout[SC_N0] = n0;
    sout[SC_W0] = -z0 * h.alpha_w / (h.beta_w + sqrt(n0));  // FTRL_Learner.h:161
  } else {
    if (h.kind == UPD_SGD_L1) {  // solver/SGD_Learner.h:92-97, once per example (SUM) or once per batch (MEAN)
      sout[SC_UW] = sin[SC_UW] + rows * (h.lr * h.regw);
      sout[SC_UV] = sin[SC_UV] + rows * (h.lr * h.regv);
    }
    if (h.k0) sout[SC_W0] = w0 - h.lr * (g0 + rows * h.reg0 * w0);  // SGD_Learner.h:106-109
  }
}

// ------------------------------------------------------------------------------------------------ phase 2
template <typename ST>
struct ColsTables {
  ST *V, *w, *sV, *sw, *nV, *nw;
  ST *t1V, *t1w, *t2V, *t2w, *t3V, *t3w;  // TDAP: nu, delta, h (u in nV / nw, z in sV / sw: the sequential learner's tables)
  const ST* S;
  const ST* amul;
  const double* scal;   // this step's start scalars (read-only during the step)
  double* scal_out;     // next step's scalars (written by workgroup 0)
  const double* partials;
  int64_t n_partials;
  ST* gbuf;
  uint32_t p;
  int has_q;
  uint32_t gb_feats;         // features per exchange-buffer block
  int64_t gb_block_elems;    // elements per block
  ST* gtail;                 // the 4-element tail {sum mult, sum mult^2, rows hi, rows lo}: end of gbuf, or the compact exchange's own
  int64_t s_rows;            // rows of S / amul reachable from the pointers above (sizes the buffer descriptors)
  ST* crec;                  // compact exchange: one record per occurring feature (see record layout below)
  int rec_elems;             // elements per record
  int w_full;                // w-in-row layout: store the row's whole second half with w (see store_w)
  int vsh, wsh;              // log2 of the two strides below
  int vs, ws;                // V rows lie vs elements apart, feature j's w is w[j * ws] (fmx_internal.h: w_in_row); the STATE tables keep stride KP
};

// Compact exchange record of one occurring feature (fmx_grad_compact): G[KP] | (has_q: Q[KP]) | Gw | Qw | cnt | id.
// The id travels as a bit pattern (a collective that only moves bytes keeps it; records are all-gathered, never summed).
__device__ __forceinline__ float id_to_elem(uint32_t id, float) { return __uint_as_float(id); }
__device__ __forceinline__ double id_to_elem(uint32_t id, double) { return __longlong_as_double((long long)id); }
__device__ __forceinline__ uint32_t elem_to_id(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint32_t elem_to_id(double v) { return (uint32_t)__double_as_longlong(v); }

// solver/SGD_Learner.h:195-204
__device__ __forceinline__ void apply_penalty(double& theta, double u, double& q) {
  const double old = theta;
  if (theta > 0) theta = fmax(0.0, old - (u + q));
  else if (theta < 0) theta = fmin(0.0, old + (u - q));
  q += theta - old;
}

// solver/FTRL_Learner.h:177-182 / :194-199
__device__ __forceinline__ double ftrl_prox(double z, double n, double l1, double l2, double alpha, double beta) {
  if (fabs(z) <= l1) return 0.0;
  const double sign = z < 0.0 ? -1.0 : 1.0;
  return -(z - sign * l1) / ((beta + sqrt(n)) / alpha + l2);
}

// One coordinate's update from its batch sums (G = sum g, Q = sum g^2, cnt occurrences); state in/out.
template <int KIND, typename ST>
__device__ __forceinline__ double coord_update(const Hyper& h, bool is_w, double theta, double G, double Q, double cnt,
                                               double decay, double u, ST& st_a, ST& st_b, bool keep) {
  if constexpr (KIND == UPD_SGD_L2) {
    (void)Q; (void)u; (void)st_a; (void)st_b; (void)cnt; (void)is_w;
    if (!keep) return theta;
    return (theta - h.lr * G) * decay;  // SGD_Learner.h:114-119 / :130-135 with c touches folded: (1 - lr*reg)^c
  } else if constexpr (KIND == UPD_SGD_L1) {
    (void)Q; (void)decay; (void)st_b; (void)cnt; (void)is_w;
    if (!keep) return theta;
    double t = theta - h.lr * G;
    double q = st_a;
    apply_penalty(t, u, q);
    st_a = (ST)q;
    return t;
  } else {
    (void)decay; (void)u; (void)cnt;
    const double alpha = is_w ? h.alpha_w : h.alpha_v, beta = is_w ? h.beta_w : h.beta_v;
    const double l1 = is_w ? h.l1w : h.l1v, l2 = is_w ? h.l2w : h.l2v;
    double z = st_a, n = st_b;
    if (keep) {  // FTRL_Learner.h:88-113 with the batch sums (the sigma terms telescope)
      const double n_new = n + Q;
      z += G - theta * (sqrt(n_new) - sqrt(n)) / alpha;
      n = n_new;
      st_a = (ST)z; st_b = (ST)n;
    }
    return ftrl_prox((double)st_a, (double)st_b, l1, l2, alpha, beta);
  }
}

// Mini-batch TDAP coordinate (oracle: fmo_tdap_apply_sums; at one occurrence it is solver/TDAP_Learner.h:97-105 / :115-126 /
// :134-141 followed by calculate_param :208-213 / :222-229): state u, nu, delta, h, z in/out, returns the new value.
template <typename ST>
__device__ __forceinline__ double tdap_update(const Hyper& h, bool is_w, double theta, double G, double Q, double cnt, ST& su, ST& snu, ST& sdl,
                                              ST& sh, ST& sz, bool keep) {
  const double alpha = is_w ? h.alpha_w : h.alpha_v;
  const double l1 = is_w ? h.l1w : h.l1v, l2 = is_w ? h.l2w : h.l2v;
  if (keep) {
    const double u_old = su;
    const double u = u_old + Q;
    const double nu = (double)snu + G;
    const double sigma = (sqrt(u) - sqrt(u_old)) / alpha;
    const double age = cnt == 1.0 ? h.egamma : exp(-h.gamma * cnt);  // one decay per occurrence
    const double dl = age * ((double)sdl + sigma);
    const double hh = age * ((double)sh + sigma * theta);
    su = (ST)u; snu = (ST)nu; sdl = (ST)dl; sh = (ST)hh;
    sz = (ST)((double)snu - (double)sh);  // z = nu - h from the STORED values: what a reload of the state gives
  }
  const double z = sz;
  if (fabs(z) <= l1) return 0.0;
  const double sign = z < 0.0 ? -1.0 : 1.0;
  return -(z - sign * l1) / ((double)sdl + l2);
}

// One (feature, lane-slice) worth of batch sums: the lane's factors (4 of an fp32 table, 2 of an fp64 one) plus the
// feature's linear term
struct CoordSums {
  double G[4], Q[4];
  double Gw, Qw, cnt;
};

__device__ __forceinline__ void sums_zero(CoordSums& s) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { s.G[i] = 0.0; s.Q[i] = 0.0; }
  s.Gw = 0.0; s.Qw = 0.0; s.cnt = 0.0;
}

// one occurrence (row r with value x) of the feature whose batch-start V slice is vf
template <bool NEED_Q, typename VT, typename ST>
__device__ __forceinline__ void sums_add(CoordSums& s, const double* vf, const VT& srow, ST amul, float xf) {
  constexpr int VEC = 16 / sizeof(ST);
  const double x = (double)xf;
  const double ax = (double)amul * x;  // mult * x: the w gradient, SGD_Learner.h:114
  s.Gw += ax;
  if (NEED_Q) s.Qw += ax * ax;
  s.cnt += 1.0;
  double sf[4];
  slice_get(srow, sf);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const double g = ax * (sf[i] - vf[i] * x);  // mult*(sum_f*x - v*x*x), SGD_Learner.h:129
    s.G[i] += g;
    if (NEED_Q) s.Q[i] += g * g;
  }
}

// Store of the updated linear weight.  Separate w table (ws == 1): lane 0 stores the word.  w-in-row layout (ws > 1): the row's
// second half -- slot KP = w, the rest padding -- is stored WHOLE, every lane of the group its 16-byte slice of it: a 4-byte store
// into an otherwise clean 64-byte sector is a partial write (the memory side has to read, merge and re-encode the sector: ECC), a full
// sector is a plain write.  Measured at p = 16 M, k = 16 (profiles/r03_wir_ab.txt).  full == 0: the 4-byte store (A/B runs).
// WIR_OK = RowStride::DYN: elsewhere only the word store is compiled.
template <typename ST, int VEC, bool WIR_OK>
__device__ __forceinline__ void store_w(ST* wbase, size_t at_w, int ws, int lig, double wn, int full) {
  using vec_t = typename Slice<ST>::vec;
  if constexpr (WIR_OK) {
    if (ws > 1 && full) {
      double sl[4] = {lig == 0 ? wn : 0.0, 0.0, 0.0, 0.0};
      *reinterpret_cast<vec_t*>(wbase + at_w + lig * VEC) = slice_make(sl, ST());
      return;
    }
  }
  if (lig == 0) wbase[at_w] = (ST)wn;
}

// What happens to a feature's sums: [+ the exchange buffer's] -> [publish] -> [apply the update].  Shared by the main
// kernel (short lists) and the long-list finisher.  Called by every lane of the feature's group; lig == 0 handles w.
template <typename ST, int LPR, int KIND, bool GBUF = true>
__device__ __forceinline__ void cols_finish(const ColsArgs& a, const Hyper& h, const ColsTables<ST>& T, int64_t j, int lig, const double* vf,
                                            CoordSums& s, double rows, int64_t ci = 0, const ST* w_pre = nullptr) {
  using vec_t = typename Slice<ST>::vec;
  constexpr int VEC = Slice<ST>::N;
  constexpr int KP = LPR * VEC;
  constexpr bool NEED_Q = (KIND == UPD_FTRL || KIND == UPD_TDAP);
  // exchange buffer: blocks of F features, each GV [F][KP] | GW [F] | CNT [F] | (has_q: QV [F][KP] | QW [F]); then tail[4]
  const size_t at = (size_t)j * KP + lig * VEC;    // in the optimizer-state tables
  const size_t at_v = ((size_t)j << RowStride<ST, LPR>::v(T.vsh)) + lig * VEC;  // in the V table
  const size_t at_w = (size_t)j << RowStride<ST, LPR>::w(T.wsh);                // in the w table (or the V row's w slot)
  if (GBUF && (a.load_gbuf || a.store_gbuf)) {
    const uint32_t F = T.gb_feats;
    const uint32_t blk = (uint32_t)j / F, r = (uint32_t)j - blk * F;
    ST* gGV = T.gbuf + (size_t)blk * T.gb_block_elems;
    ST* gGW = gGV + (size_t)F * KP;
    ST* gCN = gGW + F;
    ST* gQV = gCN + F;
    ST* gQW = gQV + (size_t)F * KP;
    const size_t gat = (size_t)r * KP + lig * VEC;
    if (a.load_gbuf) {  // sums of earlier tiles of this step, or the all-reduced sums of the whole global batch
      double g[VEC];
      slice_get(*reinterpret_cast<const vec_t*>(gGV + gat), g);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s.G[i] += g[i];
      s.Gw += gGW[r];
      s.cnt += gCN[r];
      if (NEED_Q && T.has_q) {
        slice_get(*reinterpret_cast<const vec_t*>(gQV + gat), g);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.Q[i] += g[i];
        s.Qw += gQW[r];
      }
    }
    if (a.store_gbuf) {  // publish the sums so far (every feature, zeros included)
      *reinterpret_cast<vec_t*>(gGV + gat) = slice_make(s.G, ST());
      if (NEED_Q && T.has_q) *reinterpret_cast<vec_t*>(gQV + gat) = slice_make(s.Q, ST());
      if (lig == 0) {
        gGW[r] = (ST)s.Gw;
        gCN[r] = (ST)s.cnt;
        if (NEED_Q && T.has_q) gQW[r] = (ST)s.Qw;
      }
    }
  }
  if (a.store_compact) {  // record ci of the compact exchange: the sums of one occurring feature
    // owner-sharded exchange: the record goes to the list's slot in owner-major order (the part for owner o is then one slice)
    const size_t slot = a.rec_pos ? (size_t)a.rec_pos[ci] : (size_t)ci;
    ST* rec = T.crec + slot * T.rec_elems;
    *reinterpret_cast<vec_t*>(rec + lig * VEC) = slice_make(s.G, ST());
    const int qo = (NEED_Q && T.has_q) ? KP : 0;
    if (qo) *reinterpret_cast<vec_t*>(rec + KP + lig * VEC) = slice_make(s.Q, ST());
    if (lig == 0) {
      rec[KP + qo] = (ST)s.Gw;
      rec[KP + qo + 1] = (ST)(qo ? s.Qw : 0.0);
      rec[KP + qo + 2] = (ST)s.cnt;
      rec[KP + qo + 3] = id_to_elem((uint32_t)j, ST());
    }
  }
  // untouched coordinates keep their value (lazy regularisation, SURVEY A-10)
  if (!a.apply || s.cnt == 0.0) return;

  double cnt = s.cnt;
  if (h.mean) {  // FMX_REDUCE_MEAN: one reference step with the mean gradient of the coordinate's occurrences
    const double inv = 1.0 / cnt;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s.G[i] *= inv; s.Q[i] = s.G[i] * s.G[i]; }
    s.Gw *= inv; s.Qw = s.Gw * s.Gw;
    cnt = 1.0;
  }
  double decay_v = 1.0, decay_w = 1.0, u_w = 0.0, u_v = 0.0;
  if constexpr (KIND == UPD_SGD_L2) {
    // (1 - lr*reg)^cnt; cnt == 1 always under FMX_REDUCE_MEAN, so the transcendental is off the common path
    if (cnt == 1.0) { decay_v = h.decay_v; decay_w = h.decay_w; }
    else {
      decay_v = h.decay_v > 0.0 ? exp(cnt * h.log_decay_v) : pow(h.decay_v, cnt);
      decay_w = h.decay_w > 0.0 ? exp(cnt * h.log_decay_w) : pow(h.decay_w, cnt);
    }
  }
  if constexpr (KIND == UPD_SGD_L1) {  // the penalty level after this batch (same expression as scalar_update)
    const double r = h.mean ? 1.0 : rows;
    u_w = T.scal[SC_UW] + r * (h.lr * h.regw);
    u_v = T.scal[SC_UV] + r * (h.lr * h.regv);
  }
  if constexpr (KIND == UPD_TDAP) {
    ST* const tabs[5] = {T.nV, T.t1V, T.t2V, T.t3V, T.sV};  // u, nu, delta, h, z
    ST st[5][VEC];
    double t5[VEC];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      slice_get(*reinterpret_cast<const vec_t*>(tabs[q] + at), t5);
#pragma unroll
      for (int i = 0; i < VEC; ++i) st[q][i] = (ST)t5[i];
    }
    double outv[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) outv[i] = tdap_update<ST>(h, false, vf[i], s.G[i], s.Q[i], cnt, st[0][i], st[1][i], st[2][i], st[3][i], st[4][i], true);
    *reinterpret_cast<vec_t*>(T.V + at_v) = slice_make(outv, ST());
#pragma unroll
    for (int q = 0; q < 5; ++q) *reinterpret_cast<vec_t*>(tabs[q] + at) = slice_make(st[q], ST());
    double wn = 0.0;
    if (lig == 0) {  // like FTRL, TDAP recomputes w on every touched column even with keep.w1 off (TDAP_Learner.h:203-214)
      ST u = T.nw[j], nu = T.t1w[j], dl = T.t2w[j], hh = T.t3w[j], z = T.sw[j];
      wn = tdap_update<ST>(h, true, (double)(w_pre ? *w_pre : T.w[at_w]), s.Gw, s.Qw, cnt, u, nu, dl, hh, z, h.k1 != 0);
      T.nw[j] = u; T.t1w[j] = nu; T.t2w[j] = dl; T.t3w[j] = hh; T.sw[j] = z;
    }
    store_w<ST, VEC, RowStride<ST, LPR>::DYN>(T.w, at_w, T.ws, lig, wn, T.w_full);
    return;
  }
  double tmp[VEC];
  ST sa_[VEC], sb_[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { sa_[i] = (ST)0; sb_[i] = (ST)0; }
  // every state element of the coordinate is requested in ONE round -- the V side's and, by every lane of the group (same address:
  // one request), the w side's, which used to be read behind `if (lig == 0)` after the V stores: a second dependent round trip
  ST wa = (ST)0, wb = (ST)0;
  if constexpr (KIND != UPD_SGD_L2) {
    const vec_t raw_a = *reinterpret_cast<const vec_t*>(T.sV + at);
    wa = T.sw[j];
    if constexpr (KIND == UPD_FTRL) {
      const vec_t raw_b = *reinterpret_cast<const vec_t*>(T.nV + at);
      wb = T.nw[j];
      slice_get(raw_b, tmp);
#pragma unroll
      for (int i = 0; i < VEC; ++i) sb_[i] = (ST)tmp[i];
    }
    slice_get(raw_a, tmp);
#pragma unroll
    for (int i = 0; i < VEC; ++i) sa_[i] = (ST)tmp[i];
  }
  double out[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    out[i] = coord_update<KIND, ST>(h, false, vf[i], s.G[i], s.Q[i], cnt, decay_v, u_v, sa_[i], sb_[i], true);
  *reinterpret_cast<vec_t*>(T.V + at_v) = slice_make(out, ST());
  if constexpr (KIND != UPD_SGD_L2) *reinterpret_cast<vec_t*>(T.sV + at) = slice_make(sa_, ST());
  if constexpr (KIND == UPD_FTRL) *reinterpret_cast<vec_t*>(T.nV + at) = slice_make(sb_, ST());
  // FTRL recomputes w on every touched column even when keep.w1 is off (FTRL_Learner.h:172-183); SGD skips (:111)
  const bool k1 = h.k1 != 0;
  if (k1 || KIND == UPD_FTRL) {
    double wn = 0.0;
    if (lig == 0) {
      wn = coord_update<KIND, ST>(h, true, (double)(w_pre ? *w_pre : T.w[at_w]), s.Gw, s.Qw, cnt, decay_w, u_w, wa, wb, k1);
      if constexpr (KIND != UPD_SGD_L2) T.sw[j] = wa;
      if constexpr (KIND == UPD_FTRL) T.nw[j] = wb;
    }
    store_w<ST, VEC, RowStride<ST, LPR>::DYN>(T.w, at_w, T.ws, lig, wn, T.w_full);
  }
}

template <typename ST, int LPR>
__device__ __forceinline__ ST* exchange_tail(const ColsTables<ST>& T) {
  return T.gtail;
}

// Main phase-2 kernel: one group of LPR lanes per feature list.  Lists longer than a.long_min entries (heavy hitters of a
// skewed feature distribution) are left to the long-list kernels below; walking them with one group would serialise the tile.
// (88-90 VGPRs: 5 workgroups per CU.  Asking the register allocator for 6 waves per SIMD spills to scratch -- phase 2 0.16 -> 0.31 ms per
// tile -- and 4 or 8 or 16 gathers in flight per lane (FMX_U) change nothing or lose: profiles/r02_small_batch.txt, r02_ab.txt; ONE or TWO S rows
// outstanding per lane group -- phase 1's winning schedule -- lose here: 0.200 / 0.171 against 0.153 ms per tile, profiles/r04_ragged_forms.txt section 9.)
// SPARSE: the lean form for a sparse tile walked list by list (a.direct) with no dense exchange buffer in play -- no staging
// array, no exchange-buffer code; fewer registers and 4 KB of LDS, so more workgroups per CU.  That walk is latency x occupancy
// bound (three dependent memory rounds per list, lists of one to four entries), not byte bound: DESIGN.md section 6.1.
template <typename ST, int LPR, int KIND, bool SPARSE = false>
__global__ __launch_bounds__(WG_THREADS, SPARSE ? FMX_SPARSE_WAVES : (sizeof(ST) == 4 ? FMX_DENSE_WAVES : 1)) void fm_cols_update_k(ColsArgs a, Hyper h, ColsTables<ST> T) {
  using vec_t = typename Slice<ST>::vec;
  constexpr int VEC = Slice<ST>::N;
  constexpr int KP = LPR * VEC;
  constexpr int FPW = WG_THREADS / LPR;  // features (lists) per workgroup
  constexpr bool NEED_Q = (KIND == UPD_FTRL || KIND == UPD_TDAP);
  __shared__ uint2 stage[SPARSE ? 1 : STAGE_ENTRIES + FMX_U];
  __shared__ double red_g[WG_THREADS], red_q[WG_THREADS];
  __shared__ unsigned long long wg_next;

  const int tid = threadIdx.x;
  const int gid = tid / LPR;
  const int lig = tid % LPR;
  // lists of this workgroup: features I0.. (dense walk) or the I0..-th occurring features (sparse tile)
  const int64_t n_lists = a.tfeat ? (int64_t)a.n_tfeat : (int64_t)a.f1;
  const int64_t I0 = (a.tfeat ? 0 : (int64_t)a.f0) + (int64_t)blockIdx.x * FPW;
  const int64_t I1 = (I0 + FPW < n_lists) ? I0 + FPW : n_lists;
  if (I0 >= n_lists) {  // a launch without lists: only workgroup 0 exists, for the w0 step (nothing below may be read)
    if (blockIdx.x == 0 && a.scalar != SCALAR_NONE)
      scalar_update(T.partials, T.n_partials, T.scal, T.scal_out, exchange_tail<ST, LPR>(T), h, a.global_rows, a.scalar, red_g, red_q);
    return;
  }
  bool have = I0 + gid < n_lists;
  // Every load below is unconditional, on an index clamped into the workgroup's own range, and its value is selected afterwards.
  // A load inside `if (have)` (or `cond ? load : constant`) compiles to a branch with a wait for everything outstanding at the
  // join, and the front of this kernel used to be a chain of such waits: feature id -> V row -> w -> list offsets -> entries
  // -> S rows, six dependent round trips where the data dependences ask for three (ids and offsets; V row, w and entries; S rows).
  const int64_t idx = have ? I0 + gid : I0;
  // list offsets: the dense per-feature array, or its compact copy for the occurring features (their entries are
  // contiguous: the features between them have none)
  const uint32_t* __restrict__ off = a.tfeat ? a.toff : a.bptr;
  uint32_t off_a = 0, off_b = 0;
  int64_t j = idx;
  uint32_t row0 = 0, x0 = 0x3f800000u;  // SPARSE: the list's first entry, inline in the directory
  if (a.tfeat) {  // one join, one wait: the id and both offsets travel together
    j = (int64_t)a.tfeat[idx];
    off_a = a.toff[idx]; off_b = a.toff[idx + 1];
    if constexpr (SPARSE) {
      if (a.inline0) { row0 = a.trow0[idx]; x0 = a.tval0[idx]; }
    }
  } else if (SPARSE || a.walk) {
    off_a = a.bptr[idx]; off_b = a.bptr[idx + 1];
  }
  const vec_t v_raw = *reinterpret_cast<const vec_t*>(T.V + ((size_t)j << RowStride<ST, LPR>::v(T.vsh)) + lig * VEC);
  // w_j is needed only after the walk: ask for it now, beside the V row, instead of paying its round trip at the end
  ST w_pre = T.w[(size_t)j << RowStride<ST, LPR>::w(T.wsh)];
  CoordSums s;
  sums_zero(s);
  double vf[VEC];

  if (SPARSE || a.walk) {
    const int64_t lo = SPARSE ? 0 : off[I0], hi = SPARSE ? 0 : off[I1];
    int64_t ta = have ? (int64_t)off_a : 0, tb = have ? (int64_t)off_b : 0;
    if (a.long_min > 0 && tb - ta > (int64_t)a.long_min) { have = false; ta = tb = 0; }  // a long list: not ours
    const ST* __restrict__ St = T.S + lig * VEC;
    const __amdgpu_buffer_rsrc_t s_rsrc = table_rsrc(T.S, a.buf_gather ? (uint32_t)(T.s_rows * (KP * sizeof(ST))) : 0u);
    const __amdgpu_buffer_rsrc_t a_rsrc = table_rsrc(T.amul, a.buf_gather ? (uint32_t)(T.s_rows * sizeof(ST)) : 0u);
    // one batch of FMX_U entries of this group's list: gather the S rows (and multipliers), add the occurrences in row order
    auto take = [&](uint2 (&en)[FMX_U]) {
      bool ok[FMX_U];
#pragma unroll
      for (int u = 0; u < FMX_U; ++u) {
        ok[u] = en[u].x < a.rows_active;  // truncated batch: rows beyond the limit do not take part (padding slots: 0xFFFFFFFF)
        if (!ok[u]) en[u].x = 0;
      }
      vec_t sv[FMX_U];
      ST av[FMX_U];
      if (a.buf_gather) {  // wave-uniform: padding slots issue no request
#pragma unroll
        for (int u = 0; u < FMX_U; ++u) sv[u] = buf_row(s_rsrc, ok[u] ? en[u].x * (uint32_t)(KP * sizeof(ST)) + (uint32_t)(lig * 16) : BUF_SKIP, ST());
        if (a.embed) {
          if constexpr (sizeof(ST) == 4) {
#pragma unroll
            for (int u = 0; u < FMX_U; ++u) av[u] = embed_take<LPR>(sv[u], lig, a.embed);
          }
        } else {
#pragma unroll
          for (int u = 0; u < FMX_U; ++u) av[u] = buf_elem(a_rsrc, ok[u] ? en[u].x * (uint32_t)sizeof(ST) : BUF_SKIP, ST());
        }
      } else {
#pragma unroll
        for (int u = 0; u < FMX_U; ++u) {
          sv[u] = gather_row(St + (size_t)en[u].x * KP);
          av[u] = T.amul[en[u].x];
        }
        if constexpr (sizeof(ST) == 4) {
          if (a.embed) {
#pragma unroll
            for (int u = 0; u < FMX_U; ++u) (void)embed_take<LPR>(sv[u], lig, a.embed);  // strip the embedded bits; the side table gave the multiplier
          }
        }
      }
      // the V row is first needed HERE (converted per batch: four conversions, and the wait for it falls after this batch's
      // gathers have gone out instead of before the first entry is read)
      slice_get(v_raw, vf);
#pragma unroll
      for (int u = 0; u < FMX_U; ++u)  // occurrences in row order
        if (ok[u]) sums_add<NEED_Q>(s, vf, sv[u], av[u], __uint_as_float(en[u].y));
    };
    // the same, one entry at a time (the lean form's first batch)
    auto issue1 = [&](uint32_t row, bool ok, vec_t& sv, ST& av) {
      if (a.buf_gather) {
        sv = buf_row(s_rsrc, ok ? row * (uint32_t)(KP * sizeof(ST)) + (uint32_t)(lig * 16) : BUF_SKIP, ST());
        av = buf_elem(a_rsrc, (ok && !a.embed) ? row * (uint32_t)sizeof(ST) : BUF_SKIP, ST());  // embedded: no request goes out
      } else {
        sv = gather_row(St + (size_t)(ok ? row : 0u) * KP);
        av = T.amul[ok ? row : 0u];
      }
    };
    auto fin1 = [&](vec_t& sv, ST& av) {
      if constexpr (sizeof(ST) == 4) {
        if (a.embed) {
          const float m = embed_take<LPR>(sv, lig, a.embed);
          if (a.buf_gather) av = m;
        }
      }
    };
    int64_t t_first = ta;  // where the list-by-list loops below start
    if constexpr (SPARSE) if (a.inline0) {
      // Round 2 of the list: next to its V row and w goes out the S row of its FIRST entry -- whose row number came inline with the
      // directory -- and the reads of entries 1..3: the first gathers overlap the V row's fetch.  Same additions, same order.
      // Measured (profiles/r02_direct_lists.txt): Criteo shape (six entries per list) 0.425 -> 0.385 ms per step; tiles of one-entry
      // lists (uniform 33 M features) LOSE 2-6 % -- they are bound by the rate of random requests, not by rounds, and the
      // earlier gather only deepens the queues (section 6.2's finding again) -- so the launcher asks for it from two entries
      // per list on average.
      const bool ok0 = ta < tb && row0 < a.rows_active;
      vec_t sv0;
      ST av0;
      issue1(row0, ok0, sv0, av0);
      uint32_t r[FMX_U], x[FMX_U];
#pragma unroll
      for (int u = 1; u < FMX_U; ++u) { r[u] = 0u; x[u] = 0x3f800000u; }
      // (loads under a branch: its join waits for this round's requests -- which is what the next line does anyway; a wave of
      // one-entry lists skips the block)
      if (tb - ta > 1) {
#pragma unroll
        for (int u = 1; u < FMX_U; ++u) r[u] = a.brow[ta + u < tb ? ta + u : ta];
        if (!a.unit) {
#pragma unroll
          for (int u = 1; u < FMX_U; ++u) x[u] = __float_as_uint(a.bval[ta + u < tb ? ta + u : ta]);
        }
      }
      fin1(sv0, av0);
      slice_get(v_raw, vf);
      if (ok0) sums_add<NEED_Q>(s, vf, sv0, av0, __uint_as_float(x0));
      if (tb - ta > 1) {
        vec_t sv[FMX_U];
        ST av[FMX_U];
        bool ok[FMX_U];
#pragma unroll
        for (int u = 1; u < FMX_U; ++u) {
          ok[u] = ta + u < tb && r[u] < a.rows_active;
          issue1(r[u], ok[u], sv[u], av[u]);
        }
#pragma unroll
        for (int u = 1; u < FMX_U; ++u) fin1(sv[u], av[u]);
#pragma unroll
        for (int u = 1; u < FMX_U; ++u)
          if (ok[u]) sums_add<NEED_Q>(s, vf, sv[u], av[u], __uint_as_float(x[u]));
      }
      t_first = ta + FMX_U;
    }
    if constexpr (SPARSE) {  // (the launcher picks this instantiation whenever a.direct is set)
      // Sparse tiles (lists of one or two entries): every group reads its own entries straight from memory -- neighbouring groups
      // read neighbouring addresses, so the loads coalesce by themselves -- and the workgroup never meets at a barrier: one
      // dependent round trip fewer per list, in a regime that is nothing but dependent round trips.
      if (a.unit) {
        for (int64_t t = t_first; t < tb; t += FMX_U) {
          uint32_t r[FMX_U];
#pragma unroll
          for (int u = 0; u < FMX_U; ++u) r[u] = a.brow[t + u < tb ? t + u : t];
          uint2 en[FMX_U];
#pragma unroll
          for (int u = 0; u < FMX_U; ++u) en[u] = make_uint2(t + u < tb ? r[u] : 0xFFFFFFFFu, 0x3f800000u);
          take(en);
        }
      } else {
        for (int64_t t = t_first; t < tb; t += FMX_U) {
          uint32_t r[FMX_U], x[FMX_U];
#pragma unroll
          for (int u = 0; u < FMX_U; ++u) {
            const int64_t tt = t + u < tb ? t + u : t;
            r[u] = a.brow[tt];
            x[u] = __float_as_uint(a.bval[tt]);
          }
          uint2 en[FMX_U];
#pragma unroll
          for (int u = 0; u < FMX_U; ++u) en[u] = make_uint2(t + u < tb ? r[u] : 0xFFFFFFFFu, x[u]);
          take(en);
        }
      }
    } else {
      int64_t c0 = lo;
      while (c0 < hi) {
        if (a.long_min > 0) {
          // The entries of long lists lie between those of this workgroup's short lists (one heavy hitter can hold 10^5 of
          // them): jump straight to the first entry that a short list still needs.
          if (tid == 0) wg_next = ~0ull;
          __syncthreads();
          if (tb > c0) atomicMin(&wg_next, (unsigned long long)(ta > c0 ? ta : c0));
          __syncthreads();
          const unsigned long long nx = wg_next;
          __syncthreads();
          if (nx == ~0ull) break;
          c0 = (int64_t)nx;
        }
        const int cn = (hi - c0 < STAGE_ENTRIES) ? (int)(hi - c0) : STAGE_ENTRIES;
        const int64_t b = ta > c0 ? ta : c0;
        const int64_t e = tb < c0 + cn ? tb : c0 + cn;
        stage_entries<false>(stage, a.brow, a.bval, c0, cn, a.unit);
        __syncthreads();
        for (int64_t t = b; t < e; t += FMX_U) {
          const int o = (int)(t - c0);
          uint2 en[FMX_U];
#pragma unroll
          for (int u = 0; u < FMX_U; ++u) {  // the array is padded: read first, select afterwards
            const uint2 v = stage[o + u];
            en[u] = make_uint2(t + u < e ? v.x : 0xFFFFFFFFu, v.y);
          }
          take(en);
        }
        __syncthreads();
        c0 += STAGE_ENTRIES;
      }
    }
  }
  slice_get(v_raw, vf);
  ST* gtail = exchange_tail<ST, LPR>(T);
  double rows = a.global_rows;
  if (a.apply && a.load_gbuf && rows <= 0.0) rows = tail_get_rows(gtail);  // the global row count travelled in the reduced buffer

  if (have) cols_finish<ST, LPR, KIND, !SPARSE>(a, h, T, j, lig, vf, s, rows, I0 + gid, &w_pre);

  if (blockIdx.x == 0 && a.scalar != SCALAR_NONE)
    scalar_update(T.partials, T.n_partials, T.scal, T.scal_out, gtail, h, a.global_rows, a.scalar, red_g, red_q);
}

// ---- long lists ---------------------------------------------------------------------------------------------------
// A list longer than long_min entries is cut into segments of <= LIST_SEG entries; one WAVE walks a segment: its 64/LPR lane
// groups take every (64/LPR)-th entry, partial sums are combined across the groups by a fixed butterfly and written out (fp64).
// A second kernel adds a feature's segment sums in segment order and finishes the feature exactly like the main kernel.
// Fixed geometry and order: results stay bitwise reproducible.
constexpr int LONG_STRIDE(int kp) { return 2 * kp + 4; }  // doubles per segment: G[kp] | Q[kp] | Gw, Qw, cnt, pad

template <typename ST, int LPR, bool NEED_Q>
__global__ __launch_bounds__(WG_THREADS) void fm_cols_long_partial_k(LongArgs la, ColsArgs a, ColsTables<ST> T) {
  using vec_t = typename Slice<ST>::vec;
  constexpr int VEC = Slice<ST>::N;
  constexpr int KP = LPR * VEC;
  constexpr int NSUB = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int64_t seg = (int64_t)blockIdx.x * (WG_THREADS / 64) + (threadIdx.x >> 6);
  if (seg >= la.n_seg) return;
  const int sub = lane / LPR, lig = lane % LPR;
  const int64_t j = la.lfeat[la.seg_feat[seg]];
  if (j < (int64_t)a.f0 || j >= (int64_t)a.f1) return;  // not in this launch's feature range (uniform over the wave)
  const int64_t ta = la.seg_begin[seg], tb = la.seg_end[seg];
  double vf[VEC];
  slice_get(*reinterpret_cast<const vec_t*>(T.V + ((size_t)j << RowStride<ST, LPR>::v(T.vsh)) + lig * VEC), vf);
  CoordSums s;
  sums_zero(s);
  const ST* __restrict__ St = T.S + lig * VEC;
  // unconditional loads on clamped indices, values selected afterwards (a `cond ? load : constant` costs a wait per entry: the
  // segment of a wave is ONE dependent chain, and this kernel's duration is that chain's -- 0.13 ms per step on the Criteo shape
  // with eight serialised reads per round, section 6.2)
  for (int64_t t = ta + sub; t < tb; t += (int64_t)NSUB * FMX_U) {
    uint32_t r[FMX_U], xb[FMX_U];
    float x[FMX_U];
    bool ok[FMX_U];
#pragma unroll
    for (int u = 0; u < FMX_U; ++u) {
      const int64_t tt = t + (int64_t)u * NSUB;
      r[u] = a.brow[tt < tb ? tt : t];
      xb[u] = 0x3f800000u;
    }
    if (!a.unit) {  // (the join of this branch waits for the round's reads: what the next lines need anyway)
#pragma unroll
      for (int u = 0; u < FMX_U; ++u) {
        const int64_t tt = t + (int64_t)u * NSUB;
        xb[u] = __float_as_uint(a.bval[tt < tb ? tt : t]);
      }
    }
#pragma unroll
    for (int u = 0; u < FMX_U; ++u) {
      const bool in = t + (int64_t)u * NSUB < tb;
      x[u] = in ? __uint_as_float(xb[u]) : 0.f;
      ok[u] = in && r[u] < a.rows_active;
      if (!ok[u]) r[u] = 0;
    }
    vec_t sv[FMX_U];
    ST av[FMX_U];
#pragma unroll
    for (int u = 0; u < FMX_U; ++u) {
      sv[u] = gather_row(St + (size_t)r[u] * KP);
      av[u] = T.amul[r[u]];
    }
    if constexpr (sizeof(ST) == 4) {
      if (a.embed) {
#pragma unroll
        for (int u = 0; u < FMX_U; ++u) (void)embed_take<LPR>(sv[u], lig, a.embed);
      }
    }
#pragma unroll
    for (int u = 0; u < FMX_U; ++u)
      if (ok[u]) sums_add<NEED_Q>(s, vf, sv[u], av[u], x[u]);
  }
  // combine the NSUB lane groups (fixed butterfly => deterministic)
#pragma unroll
  for (int o = 32; o >= LPR; o >>= 1) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      s.G[i] += __shfl_xor(s.G[i], o);
      if (NEED_Q) s.Q[i] += __shfl_xor(s.Q[i], o);
    }
    s.Gw += __shfl_xor(s.Gw, o);
    if (NEED_Q) s.Qw += __shfl_xor(s.Qw, o);
    s.cnt += __shfl_xor(s.cnt, o);
  }
  if (sub == 0) {
    double* out = la.partial + (size_t)seg * LONG_STRIDE(KP);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { out[lig * VEC + i] = s.G[i]; out[KP + lig * VEC + i] = s.Q[i]; }
    if (lig == 0) { out[2 * KP] = s.Gw; out[2 * KP + 1] = s.Qw; out[2 * KP + 2] = s.cnt; }
  }
}

// one WAVE per long feature: its lane groups add the feature's segment sums strided (a heavy hitter has hundreds of
// segments), a fixed butterfly combines them, group 0 finishes the feature
template <typename ST, int LPR, int KIND>
__global__ __launch_bounds__(WG_THREADS) void fm_cols_long_finish_k(LongArgs la, ColsArgs a, Hyper h, ColsTables<ST> T) {
  using vec_t = typename Slice<ST>::vec;
  constexpr int VEC = Slice<ST>::N;
  constexpr int KP = LPR * VEC;
  constexpr int NSUB = 64 / LPR;
  constexpr bool NEED_Q = (KIND == UPD_FTRL || KIND == UPD_TDAP);
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * (WG_THREADS / 64) + (threadIdx.x >> 6);
  if (i >= la.n_long) return;
  const int sub = lane / LPR, lig = lane % LPR;
  const int64_t j = la.lfeat[i];
  if (j < (int64_t)a.f0 || j >= (int64_t)a.f1) return;
  double vf[VEC];
  slice_get(*reinterpret_cast<const vec_t*>(T.V + ((size_t)j << RowStride<ST, LPR>::v(T.vsh)) + lig * VEC), vf);
  CoordSums s;
  sums_zero(s);
  for (uint32_t sg = la.lseg_ptr[i] + sub; sg < la.lseg_ptr[i + 1]; sg += NSUB) {
    const double* in = la.partial + (size_t)sg * LONG_STRIDE(KP);
#pragma unroll
    for (int q = 0; q < VEC; ++q) { s.G[q] += in[lig * VEC + q]; if (NEED_Q) s.Q[q] += in[KP + lig * VEC + q]; }
    s.Gw += in[2 * KP]; if (NEED_Q) s.Qw += in[2 * KP + 1]; s.cnt += in[2 * KP + 2];
  }
#pragma unroll
  for (int o = 32; o >= LPR; o >>= 1) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      s.G[q] += __shfl_xor(s.G[q], o);
      if (NEED_Q) s.Q[q] += __shfl_xor(s.Q[q], o);
    }
    s.Gw += __shfl_xor(s.Gw, o);
    if (NEED_Q) s.Qw += __shfl_xor(s.Qw, o);
    s.cnt += __shfl_xor(s.cnt, o);
  }
  if (sub != 0) return;
  ST* gtail = exchange_tail<ST, LPR>(T);
  double rows = a.global_rows;
  if (a.apply && a.load_gbuf && rows <= 0.0) rows = tail_get_rows(gtail);
  cols_finish<ST, LPR, KIND>(a, h, T, j, lig, vf, s, rows, la.lpos ? (int64_t)la.lpos[i] : j);
}

template <typename ST, int KIND>
static int launch_cols_kind(fmx_engine* e, const ColsArgs& a, const LongArgs& la, const ColsTables<ST>& T) {
  const int lpr = mb_lpr(e);
  const int fpw = WG_THREADS / lpr;
  const int64_t lists = a.tfeat ? (int64_t)a.n_tfeat : (int64_t)a.f1 - (int64_t)a.f0;
  const int64_t grid = lists > 0 ? (lists + fpw - 1) / fpw : 1;  // at least workgroup 0: it also does the w0 step
  FMX_CHECK(grid < (1LL << 31), FMX_ERR_INVALID, "cols_update: grid too large");
  dim3 g((unsigned)grid), b(WG_THREADS);
  const bool lng = a.walk && la.n_long > 0;
  constexpr bool NQ = (KIND == UPD_FTRL || KIND == UPD_TDAP);
  // (Tried and removed, profiles/r03_prefix_pass.txt: the d always-present features of Criteo-shaped rows summed by ONE pass over the S
  // rows instead of one long list each.  Their lists cost only 22 us of fm_cols_long_partial_k's 116 per 262 144-row step -- the S
  // table is cache resident and 13 x 256 one-wave segments fill the chip -- while the single pass has 256 segments' worth of waves
  // and 150 live accumulators per lane: 133 us.  The heads of the categorical fields are what the long-list kernels spend their time on.)
  dim3 g1((unsigned)((la.n_seg + (WG_THREADS / 64) - 1) / (WG_THREADS / 64))), g2((unsigned)((la.n_long + (WG_THREADS / 64) - 1) / (WG_THREADS / 64)));
  const bool sparse_form = a.direct && !a.load_gbuf && !a.store_gbuf;
  // The long lists' two kernels touch other features than the main kernel (which skips every long list) and read nothing it writes: on
  // a sparse tile they run BESIDE it, on a stream of their own.  The list-by-list walk is bound by dependent rounds x occupancy and
  // leaves the memory system idle; the segments of the long lists stream S rows at cache bandwidth (section 6.7).  FMX_LONG_SIDE=0: one stream.
  static const bool side_ok = [] { const char* v = getenv("FMX_LONG_SIDE"); return !(v && v[0] == '0'); }();
  // (only where the long lists are real work: a fork and a join between two streams cost about 10 us -- profiles/probes/stream_hop.hip: 13.5 us per
  // dependency hop against 2.5 us per kernel on one stream -- which a handful of long lists cannot earn back, and the side stream's one-off creation
  // took 5 ms out of a 40-step timed region when the first tile with a long list came late: uniform columns over 250 000 features, 1 087 -> 681 M)
  const bool side = lng && sparse_form && side_ok && la.n_seg >= 2048;
  hipStream_t ls = e->stream;
  if (side) {
    if (!e->side) {
      FMX_HIP(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
      FMX_HIP(hipEventCreateWithFlags(&e->side_fork, hipEventDisableTiming));
      FMX_HIP(hipEventCreateWithFlags(&e->side_join, hipEventDisableTiming));
    }
    FMX_HIP(hipEventRecord(e->side_fork, e->stream));      // everything enqueued so far (phase 1, the plan's arrival) comes first
    FMX_HIP(hipStreamWaitEvent(e->side, e->side_fork, 0));
    ls = e->side;
  }
#define FMX_COLS_CASE(L)                                                                                        \
  case L:                                                                                                       \
    if (side) {                                                                                                 \
      hipLaunchKernelGGL((fm_cols_long_partial_k<ST, L, NQ>), g1, b, 0, ls, la, a, T);                          \
      hipLaunchKernelGGL((fm_cols_long_finish_k<ST, L, KIND>), g2, b, 0, ls, la, a, e->hyper, T);               \
    }                                                                                                           \
    if (sparse_form) hipLaunchKernelGGL((fm_cols_update_k<ST, L, KIND, true>), g, b, 0, e->stream, a, e->hyper, T); \
    else hipLaunchKernelGGL((fm_cols_update_k<ST, L, KIND>), g, b, 0, e->stream, a, e->hyper, T);                \
    if (lng && !side) {                                                                                         \
      hipLaunchKernelGGL((fm_cols_long_partial_k<ST, L, NQ>), g1, b, 0, e->stream, la, a, T);                   \
      hipLaunchKernelGGL((fm_cols_long_finish_k<ST, L, KIND>), g2, b, 0, e->stream, la, a, e->hyper, T);        \
    }                                                                                                           \
    break;
  switch (lpr) {
    FMX_COLS_CASE(1) FMX_COLS_CASE(2) FMX_COLS_CASE(4) FMX_COLS_CASE(8)
    FMX_COLS_CASE(16) FMX_COLS_CASE(32) FMX_COLS_CASE(64)
    default: FMX_CHECK(false, FMX_ERR_INVALID, "unsupported padded factor count %d", mb_kp(e));
  }
#undef FMX_COLS_CASE
  if (side) {
    FMX_HIP(hipEventRecord(e->side_join, e->side));
    FMX_HIP(hipStreamWaitEvent(e->stream, e->side_join, 0));
  }
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

template <typename ST>
static int launch_cols_state(fmx_engine* e, const ColsArgs& a, const LongArgs& la, const ColsTables<ST>& T) {
  FMX_CHECK((sizeof(ST) == 4 && e->kp32 <= WIR_MAX_KP) || (T.vs == (sizeof(ST) == 8 ? e->kp64 : e->kp32) && T.ws == 1), FMX_ERR_STATE,
            "cols_update: strides (%d, %d) on a table of compiled strides (RowStride)", T.vs, T.ws);
  switch (e->hyper.kind) {
    case UPD_SGD_L2: return launch_cols_kind<ST, UPD_SGD_L2>(e, a, la, T);
    case UPD_SGD_L1: return launch_cols_kind<ST, UPD_SGD_L1>(e, a, la, T);
    case UPD_TDAP: return launch_cols_kind<ST, UPD_TDAP>(e, a, la, T);
    default: return launch_cols_kind<ST, UPD_FTRL>(e, a, la, T);
  }
}

template <typename ST>
static ColsTables<ST> cols_tables(fmx_engine* e, const ColsArgs& a, int has_q) {
  constexpr bool W = sizeof(ST) == 8;
  ColsTables<ST> T{};
  T.V = (ST*)mb_vbase(e); T.w = (ST*)mb_wbase(e); T.vs = mb_vstride(e); T.ws = mb_wstride(e);
  T.vsh = __builtin_ctz((unsigned)T.vs); T.wsh = __builtin_ctz((unsigned)T.ws);
  { static const bool full = [] { const char* v = getenv("FMX_WIR_FULL_STORE"); return !(v && v[0] == '0'); }(); T.w_full = full ? 1 : 0; }
  T.sV = (ST*)(W ? (void*)e->dsV : (void*)e->sV); T.sw = (ST*)(W ? (void*)e->dsw : (void*)e->sw);
  T.nV = (ST*)(W ? (void*)e->dnV : (void*)e->nV); T.nw = (ST*)(W ? (void*)e->dnw : (void*)e->nw);
  T.t1V = (ST*)(W ? (void*)e->dt1V : (void*)e->t1V); T.t1w = (ST*)(W ? (void*)e->dt1w : (void*)e->t1w);
  T.t2V = (ST*)(W ? (void*)e->dt2V : (void*)e->t2V); T.t2w = (ST*)(W ? (void*)e->dt2w : (void*)e->t2w);
  T.t3V = (ST*)(W ? (void*)e->dt3V : (void*)e->t3V); T.t3w = (ST*)(W ? (void*)e->dt3w : (void*)e->t3w);
  const int kp = W ? e->kp64 : e->kp32;
  T.S = (const ST*)e->S + (size_t)a.s_row0 * kp;
  T.amul = (const ST*)e->amul + a.s_row0;
  T.s_rows = e->ws_rows - a.s_row0;
  T.scal = e->scal; T.scal_out = e->scal_next; T.partials = e->partials; T.n_partials = a.n_partials;
  T.gbuf = (ST*)e->gbuf; T.p = (uint32_t)e->p; T.has_q = has_q;
  T.gb_feats = (uint32_t)e->gb_feats; T.gb_block_elems = e->gb_block_elems;
  // the tail lives at the end of the dense exchange buffer, or alone for the compact exchange (no p-sized buffer there)
  T.gtail = a.compact_tail ? (ST*)e->ctail : (e->gbuf ? (ST*)e->gbuf + e->gb_blocks * e->gb_block_elems : nullptr);
  T.crec = (ST*)e->crec;
  T.rec_elems = e->rec_elems;
  return T;
}

int launch_cols_update(fmx_engine* e, const ColsArgs& a_in, const LongArgs& la) {
  ColsArgs a = a_in;
  if (a.f1 == 0 && a.f0 == 0) a.f1 = (uint32_t)e->p;  // the whole feature range
  FMX_CHECK(a.f0 <= a.f1 && a.f1 <= e->p && !(a.tfeat && (a.f0 != 0 || a.f1 != e->p)), FMX_ERR_INVALID, "bad feature range [%u, %u)", a.f0, a.f1);
  FMX_CHECK(!(a.load_gbuf || a.store_gbuf || ((a.scalar == SCALAR_PUBLISH || a.scalar == SCALAR_FROM_TAIL) && !a.compact_tail)) || e->gbuf != nullptr,
            FMX_ERR_STATE, "exchange buffer not allocated");
  FMX_CHECK(!(a.store_compact || a.compact_tail) || (e->ctail != nullptr && (!a.store_compact || e->crec != nullptr)), FMX_ERR_STATE, "compact exchange buffers not allocated");
  const int has_q = exchange_has_q(e) ? 1 : 0;
  // gathers of S rows / multipliers go through bounds-checked buffer descriptors while the workspace stays below 2 GiB
  // (FMX_BUF_GATHER=0 in the environment switches back to flat loads: tuning only)
  static const bool buf_ok = [] { const char* v = getenv("FMX_BUF_GATHER"); return !(v && v[0] == '0'); }();
  static const bool embed_ok = [] { const char* v = getenv("FMX_EMBED_MULT"); return !(v && v[0] == '0'); }();
  a.embed = embed_ok ? embed_mode(e->k, mb_kp(e), !mb_wide(e)) : EMBED_NONE;
  static const bool direct_ok = [] { const char* v = getenv("FMX_DIRECT_LISTS"); return !(v && v[0] == '0'); }();
  // Measured on the Criteo shape (six entries per occurring feature, heads in the long-list kernels): walking the lists straight from
  // memory 0.42 ms per step against 0.59 through the LDS staging with its barriers and its skip-the-long-lists handshake
  // (profiles/r02_direct_lists.txt); FMX_DIRECT_MAX_AVG overrides the bound for A/B runs.
  static const int direct_avg = [] { const char* v = getenv("FMX_DIRECT_MAX_AVG"); return v && atoi(v) > 0 ? atoi(v) : 16; }();
  a.direct = (direct_ok && a.walk && a.tfeat && a.n_tfeat > 0 && a.list_entries < direct_avg * (int64_t)a.n_tfeat) ? 1 : 0;  // sparse tile, short lists on average
  {  // the list-by-list form over a DENSE directory too, for steps that touch no exchange buffer (no staging, no barriers): small
     // but consistent at configs[1]'s shapes (profiles/r02_direct_lists.txt: phase 2 -1 % ... -7 %); FMX_DIRECT_DENSE=0 switches it off
    static const bool dense_direct = [] { const char* v = getenv("FMX_DIRECT_DENSE"); return !(v && v[0] == '0'); }();
    if (dense_direct && direct_ok && a.walk && !a.tfeat && !a.load_gbuf && !a.store_gbuf && a.f0 == 0 && a.f1 == e->p) a.direct = 1;
  }
  a.inline0 = (a.direct && a.trow0 && a.list_entries >= 2 * (int64_t)a.n_tfeat) ? 1 : 0;  // (small launches of one-entry lists do not gain either)
  a.buf_gather = (buf_ok && a.walk && (int64_t)(e->ws_rows - a.s_row0) * mb_kp(e) * (int64_t)mb_elem(e) < (1LL << 31)) ? 1 : 0;
  prof_begin(e, FMX_KERNEL_COLS_UPDATE);
  int st;
  if (mb_wide(e)) {
    ColsTables<double> T = cols_tables<double>(e, a, has_q);
    st = launch_cols_state<double>(e, a, la, T);
  } else {
    ColsTables<float> T = cols_tables<float>(e, a, has_q);
    st = launch_cols_state<float>(e, a, la, T);
  }
  prof_end(e);
  if (st == FMX_OK && (a.scalar == SCALAR_FUSED || a.scalar == SCALAR_FROM_TAIL)) std::swap(e->scal, e->scal_next);  // the kernel wrote the next step's scalars
  return st;
}

// ------------------------------------------------------------------------------------------------ compact exchange
// The update half of a step whose gradient sums arrive as RECORDS (one per feature per rank that saw the feature) instead of
// the dense p-sized buffer: the records of all ranks, concatenated in rank order, have been stably sorted by feature id
// (fm_ingest.hip: merge_records), so list i holds the positions of feature rfeat[i]'s records in ascending rank order.  One
// lane group per feature adds them in that order -- the same sums, in the same order, as the dense all-reduce of two ranks --
// and finishes the feature through cols_finish() like every other phase-2 path.
struct RecArgs {
  const void* recs;        // all parts; record r at recs + r * rec_elems
  const uint32_t* pos;     // [total] record positions sorted by (feature, part)
  const uint32_t* roff;    // [n + 1] list offsets into pos
  const uint32_t* rfeat;   // [n] feature ids, ascending
  const uint32_t* d_n;     // device: number of lists
};

template <typename ST, int LPR, int KIND>
__global__ __launch_bounds__(WG_THREADS) void fm_apply_records_k(RecArgs r, ColsArgs a, Hyper h, ColsTables<ST> T) {
  using vec_t = typename Slice<ST>::vec;
  constexpr int VEC = Slice<ST>::N;
  constexpr int KP = LPR * VEC;
  constexpr int FPW = WG_THREADS / LPR;
  constexpr bool NEED_Q = (KIND == UPD_FTRL || KIND == UPD_TDAP);
  __shared__ double red_g[WG_THREADS], red_q[WG_THREADS];
  const int gid = threadIdx.x / LPR, lig = threadIdx.x % LPR;
  const int64_t n = (int64_t)*r.d_n;
  const int64_t i = (int64_t)blockIdx.x * FPW + gid;
  ST* gtail = exchange_tail<ST, LPR>(T);
  if (i < n) {
    // Three dependent rounds -- {feature id, list bounds} -> {V row, w, record positions} -> {records} -- with every load of a round
    // issued before the first one is used (the plain loop paid two rounds per PART on top of the V row's own: section 6.2).
    const int64_t j = r.rfeat[i];
    const uint32_t t0 = r.roff[i], t1 = r.roff[i + 1];
    const vec_t v_raw = *reinterpret_cast<const vec_t*>(T.V + ((size_t)j << RowStride<ST, LPR>::v(T.vsh)) + lig * VEC);
    const ST w_pre = T.w[(size_t)j << RowStride<ST, LPR>::w(T.wsh)];
    const int qo = (NEED_Q && T.has_q) ? KP : 0;
    const ST* recs = reinterpret_cast<const ST*>(r.recs);
    // The parts are added in the exchange's element type, in rank order: exactly what an all-reduce(sum) of the dense buffer
    // does with two ranks (one addition per element), so both forms of the exchange give the same bits.
    ST aG[VEC], aQ[VEC], aGw = (ST)0, aQw = (ST)0, aC = (ST)0;
#pragma unroll
    for (int q = 0; q < VEC; ++q) { aG[q] = (ST)0; aQ[q] = (ST)0; }
    constexpr int RU = 4;  // parts per round
    for (uint32_t t = t0; t < t1; t += RU) {
      uint32_t pp[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) pp[u] = r.pos[t + u < t1 ? t + u : t];
      vec_t gv[RU], qv[RU];
      ST gw[RU], qw[RU], gc[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const ST* rec = recs + (size_t)pp[u] * T.rec_elems;
        gv[u] = *reinterpret_cast<const vec_t*>(rec + lig * VEC);
        qv[u] = *reinterpret_cast<const vec_t*>(rec + qo + lig * VEC);  // qo == 0: the same slice again (not used)
        gw[u] = rec[KP + qo];
        qw[u] = rec[KP + qo + 1];
        gc[u] = rec[KP + qo + 2];
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        if (t + u < t1) {
          double g[VEC];
          slice_get(gv[u], g);
#pragma unroll
          for (int q = 0; q < VEC; ++q) aG[q] = aG[q] + (ST)g[q];
          if (qo) {
            slice_get(qv[u], g);
#pragma unroll
            for (int q = 0; q < VEC; ++q) aQ[q] = aQ[q] + (ST)g[q];
            aQw = aQw + qw[u];
          }
          aGw = aGw + gw[u];
          aC = aC + gc[u];
        }
      }
    }
    double vf[VEC];
    slice_get(v_raw, vf);
    CoordSums s;
    sums_zero(s);
#pragma unroll
    for (int q = 0; q < VEC; ++q) { s.G[q] = (double)aG[q]; s.Q[q] = (double)aQ[q]; }
    s.Gw = (double)aGw; s.Qw = (double)aQw; s.cnt = (double)aC;
    double rows = a.global_rows;
    if (rows <= 0.0) rows = tail_get_rows(gtail);
    cols_finish<ST, LPR, KIND, false>(a, h, T, j, lig, vf, s, rows, 0, &w_pre);
  }
  if (blockIdx.x == 0 && a.scalar != SCALAR_NONE)
    scalar_update(T.partials, T.n_partials, T.scal, T.scal_out, gtail, h, a.global_rows, a.scalar, red_g, red_q);
}

// sort keys of the concatenated parts: linear index i -> (part, local record) -> key = the record's feature id, value = its position.
// The parts' prefix sums (and where each part starts in the buffer) travel BY VALUE in the kernel arguments: no pageable host
// buffer behind an asynchronous copy, and no host-synchronous staging inside a step.
template <typename ST>
__global__ void record_keys_k(const ST* __restrict__ recs, int rec_elems, int id_at, RecParts parts, int64_t total, uint32_t* __restrict__ keys,
                              uint32_t* __restrict__ pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int part = 0;
  while (part + 1 < parts.n && parts.prefix[part + 1] <= i) ++part;
  const int64_t at = parts.start[part] + (i - parts.prefix[part]);
  keys[i] = elem_to_id(recs[(size_t)at * rec_elems + id_at]);
  pos[i] = (uint32_t)at;
}

int launch_record_keys(fmx_engine* e, const void* recs, const RecParts& parts, int64_t total, uint32_t* keys, uint32_t* pos) {
  if (total <= 0) return FMX_OK;
  const int has_q = exchange_has_q(e) ? 1 : 0;
  const int id_at = mb_kp(e) * (1 + has_q) + 3;
  const dim3 g((unsigned)((total + 255) / 256)), b(256);
  if (mb_wide(e)) hipLaunchKernelGGL((record_keys_k<double>), g, b, 0, e->stream, (const double*)recs, e->rec_elems, id_at, parts, total, keys, pos);
  else hipLaunchKernelGGL((record_keys_k<float>), g, b, 0, e->stream, (const float*)recs, e->rec_elems, id_at, parts, total, keys, pos);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

template <typename ST, int KIND>
static int launch_records_kind(fmx_engine* e, const RecArgs& r, const ColsArgs& a, const ColsTables<ST>& T, int64_t max_lists) {
  const int lpr = mb_lpr(e);
  const int fpw = WG_THREADS / lpr;
  const int64_t grid = max_lists > 0 ? (max_lists + fpw - 1) / fpw : 1;  // the list count is on the device: surplus workgroups leave at once
  FMX_CHECK(grid < (1LL << 31), FMX_ERR_INVALID, "apply_records: grid too large");
  dim3 g((unsigned)grid), b(WG_THREADS);
#define FMX_REC_CASE(L) case L: hipLaunchKernelGGL((fm_apply_records_k<ST, L, KIND>), g, b, 0, e->stream, r, a, e->hyper, T); break;
  switch (lpr) {
    FMX_REC_CASE(1) FMX_REC_CASE(2) FMX_REC_CASE(4) FMX_REC_CASE(8) FMX_REC_CASE(16) FMX_REC_CASE(32) FMX_REC_CASE(64)
    default: FMX_CHECK(false, FMX_ERR_INVALID, "unsupported padded factor count %d", mb_kp(e));
  }
#undef FMX_REC_CASE
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

template <typename ST>
static int launch_records_state(fmx_engine* e, const RecArgs& r, const ColsArgs& a, int64_t max_lists) {
  const int has_q = exchange_has_q(e) ? 1 : 0;
  const ColsTables<ST> T = cols_tables<ST>(e, a, has_q);
  switch (e->hyper.kind) {
    case UPD_SGD_L2: return launch_records_kind<ST, UPD_SGD_L2>(e, r, a, T, max_lists);
    case UPD_SGD_L1: return launch_records_kind<ST, UPD_SGD_L1>(e, r, a, T, max_lists);
    case UPD_TDAP: return launch_records_kind<ST, UPD_TDAP>(e, r, a, T, max_lists);
    default: return launch_records_kind<ST, UPD_FTRL>(e, r, a, T, max_lists);
  }
}

int launch_apply_records(fmx_engine* e, const void* recs, const uint32_t* pos, const uint32_t* roff, const uint32_t* rfeat, const uint32_t* d_n,
                         int64_t max_lists, int64_t global_rows) {
  RecArgs r{recs, pos, roff, rfeat, d_n};
  ColsArgs a{};
  a.apply = 1;
  a.scalar = SCALAR_FROM_TAIL;
  a.compact_tail = 1;
  a.global_rows = (double)global_rows;
  prof_begin(e, FMX_KERNEL_COLS_UPDATE);
  const int st = mb_wide(e) ? launch_records_state<double>(e, r, a, max_lists) : launch_records_state<float>(e, r, a, max_lists);
  prof_end(e);
  if (st == FMX_OK) std::swap(e->scal, e->scal_next);
  return st;
}

}  // namespace fmx
