// mlp_grad_reduce.hip -- the parameter-gradient reduction of the two-layer field's adjoint (K3m), hand-written.
//
// K3m streams, per RK stage and series, the factors of the parameter gradients to HBM (rk4_mlp_adjoint.hip):
//     G [row][M]   dL/d(pre-activation) of a layer, already weighted by the quadrature weight
//     X [row][XC]  the layer's input, a "1" in column N, zeros behind it              row = (stage, series)
// and the gradients are   dW | db = G^T [X | 1]   -- an (M x N) result over a K dimension of up to 10^6..10^7 rows
// (M x N = 256 x 128 for the output layer, 128 x 32 for the hidden layer).  Round 1 left this to torch.bmm
// (hipBLASLt / Tensile kernels: 15 of the 52 ms of a two-layer backward).  Here: split-K over workgroups on
// v_mfma_f32_16x16x4_f32, every operand fetched with 16-byte loads straight from the row-major factors:
//   * a wave owns MT*16 rows of M; its M tiles are INTERLEAVED (tile t holds rows base + MT*i + t), so that ONE
//     vector load of MT consecutive floats of a G row is the A operand of all MT tiles for one K index;
//   * the N tiles are interleaved the same way in groups of 4 (tile 4g + t holds columns 64 g + 4 j + t): one float4
//     of an X row is the B operand of 4 tiles;  K step s, lane quarter kq  <->  row 4 s + kq of the slab;
//   * the bias gradient is the column sum of G: the lanes add up the A operands they load anyway;
//   * every workgroup reduces its slab of rows into registers (MT x NT tiles per wave) and writes one partial; a second
//     kernel adds the partials in slab order into the caller's accumulator -- deterministic, no atomics.
#include "cde_mfma.h"
#include "cde_dopri_adj.h"

namespace cde {

template <int MT> struct VecOf;
template <> struct VecOf<4> { using type = f32x4; };
template <> struct VecOf<2> { using type = f32x2; };

// M = 4 waves * MT * 16 rows per workgroup (blockIdx.y selects the block of M rows when a G row is wider: `gc` floats);
// N = NG * (NV * 16) columns, NV = floats per B operand read (4 or 2), NG groups of NV tiles.
// The slab is walked in tiles of GRAD_TILE_ROWS rows that go from HBM straight into a ring of three LDS buffers
// (global_load_lds_dwordx4: one wave-wide load lands 1 KB of consecutive LDS, no staging registers): while the MFMAs
// of tile t run from LDS, the loads of tiles t+1 and t+2 are in flight.  These loads return into LDS, not into
// registers, so the wait for tile t can be written by hand -- s_waitcnt vmcnt(loads of two tiles), then a barrier that
// does NOT wait on vmcnt -- which is exactly what the compiler's own s_waitcnt placement would not do across the
// loop's back edge.  (History, DESIGN.md: straight from global loads with one group of 16 rows in flight, 57 % of the
// MFMA peak; register rings defeated by the compiler's waits or unsafe; register-staged LDS double buffering, 67 %.)
constexpr int GRAD_TILE_ROWS = 16, GRAD_RING = 3;

template <int MT, int NV, int NG>
constexpr size_t grad_partial_lds_bytes() {
  return (size_t)(GRAD_RING * GRAD_TILE_ROWS * (4 * MT * 16 + NG * NV * 16) + 256) * sizeof(float);
}

__device__ __forceinline__ void ring_barrier() {           // LDS traffic of this wave done, then the workgroup barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int MT, int NV, int NG>
__device__ __forceinline__ void mlp_grad_partial_body(const float* __restrict__ G, const float* __restrict__ X,
                                                      int64_t rows, int64_t rows_per_slab, int xc,
                                                      float* __restrict__ partial, int gc,
                                                      const unsigned char* __restrict__ gate, int gate_parity,
                                                      int slabs_per_stage, const int block_x, const int block_y) {
  constexpr int M = 4 * MT * 16, NT = NG * NV, N = NT * 16, R = GRAD_TILE_ROWS;
  int64_t lo = (int64_t)block_x * rows_per_slab;
  if (gate) {
    // K4am (dopri5_mlp_adjoint.hip): one block of rows per stored stage, `slabs_per_stage` slabs each; the attempt launch
    // just before this one left, in the controller block, what it computed -- nothing once the interval is finished,
    // one stage (initial-step norms), two, or the six weighted stages of a step
    const AdjCtrl* k = reinterpret_cast<const AdjCtrl*>(gate + (gate_parity ^ 1) * ADJ_CTRL_STRIDE);
    if (k->c.phase == 4 && k->commit == 0) return;
    const int n_slots = k->mode == 0 ? 1 : k->mode == 1 ? 2 : 6;
    if (block_x >= n_slots * slabs_per_stage) return;
    // the rows of an attempt's first and last stage live in the block the controller names (AdjCtrl::src0 / six: the last
    // stage of an accepted step is the first stage of the next one and is not evaluated again)
    const int slot = block_x / slabs_per_stage;
    if (k->mode == 2 && slot == 0 && !(k->six & ADJ_FRESH0)) return;        // its image was formed when that stage was evaluated (R kernel: kst)
    const int blk = k->mode <= 1 ? slot : slot == 0 ? k->src0 : slot == 5 ? (k->six & 15) : slot;
    lo = ((int64_t)blk * slabs_per_stage + (block_x - slot * slabs_per_stage)) * rows_per_slab;
  }
  constexpr int TILE = R * (M + N);                          // floats per ring buffer: [G rows | X rows]
  // wave-wide loads (256 floats each) per tile and wave; every wave issues the same number (idle slots load into a
  // dummy area) so that one vmcnt value is right for all of them
  constexpr int GROWS = 256 / M, XROWS = 256 / N;            // rows covered by one wave-wide load
  constexpr int LG = (R / GROWS + 3) / 4, LX = (R / XROWS + 3) / 4, LOADS = LG + LX;
  using AV = typename VecOf<MT>::type;
  using BV = typename VecOf<NV>::type;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dummy = lds + GRAD_RING * TILE;
  const int m0 = block_y * M;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  int64_t hi = lo + rows_per_slab;
  hi = hi < rows ? hi : rows;
  f32x4 acc[MT][NT];
  float bsum[MT];
#pragma unroll
  for (int tm = 0; tm < MT; ++tm) {
    bsum[tm] = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto fetch = [&](int64_t r0, int buf) {                    // rows past the factors: a valid address, masked at use
    float* g_lds = lds + buf * TILE;
    float* x_lds = g_lds + R * M;
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int ld = w + 4 * j;                              // wave-wide load `ld` covers rows ld*GROWS ..
      const bool real = ld * GROWS < R;
      const int row = ld * GROWS + lane / (M / 4), c4 = lane % (M / 4);
      int64_t r = r0 + (real ? row : 0);
      r = r < rows ? r : rows - 1;
      __builtin_amdgcn_global_load_lds(G + r * gc + m0 + 4 * c4, real ? g_lds + ld * 256 : dummy, 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const int ld = w + 4 * j;
      const bool real = ld * XROWS < R;
      const int row = ld * XROWS + lane / (N / 4), c4 = lane % (N / 4);
      int64_t r = r0 + (real ? row : 0);
      r = r < rows ? r : rows - 1;
      __builtin_amdgcn_global_load_lds(X + r * xc + 4 * c4, real ? x_lds + ld * 256 : dummy, 16, 0, 0);
    }
  };
  fetch(lo, 0);
  fetch(lo + R, 1);
  int buf = 0;
  for (int64_t r0 = lo; r0 < hi; r0 += R) {
    const int nxt = buf >= 1 ? buf - 1 : GRAD_RING - 1;      // (buf + 2) % 3
    fetch(r0 + 2 * R, nxt);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");     // this wave's part of tile r0 has landed
    ring_barrier();                                                      // ... and everybody else's
    const float* g_lds = lds + buf * TILE + (w * MT * 16 + MT * i);
    const float* x_lds = lds + buf * TILE + R * M + NV * i;
#pragma unroll
    for (int s = 0; s < R / 4; ++s) {                        // K step s, lane quarter kq <-> row 4 s + kq of the tile
      const int row = 4 * s + kq;
      const bool on = r0 + row < hi;
      AV a = *reinterpret_cast<const AV*>(g_lds + row * M);
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) a[tm] = on ? a[tm] : 0.f;
      BV b[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) b[g] = *reinterpret_cast<const BV*>(x_lds + row * N + g * NV * 16);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int t = 0; t < NV; ++t) {
#pragma unroll
          for (int tm = 0; tm < MT; ++tm) acc[tm][g * NV + t] = mfma16(a[tm], b[g][t], acc[tm][g * NV + t]);
        }
      }
#pragma unroll
      for (int tm = 0; tm < MT; ++tm) bsum[tm] += a[tm];
    }
    ring_barrier();                                          // the buffer may be refilled two iterations from now
    buf = buf == GRAD_RING - 1 ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing may still be landing in LDS when the workgroup ends
  // partial[slab][m][N + 1]: D fragment of tile (tm, tn): lane (j = i, q = kq), register r = row 4q + r of the tile
  float* out = partial + ((int64_t)block_x * gc + m0) * (N + 1);
#pragma unroll
  for (int tm = 0; tm < MT; ++tm) {
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      const int col = (tn / NV) * NV * 16 + NV * i + (tn % NV);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = w * MT * 16 + MT * (4 * kq + r) + tm;
        out[m * (N + 1) + col] = acc[tm][tn][r];
      }
    }
    float s = bsum[tm];                                       // column sums: this lane saw rows kq, kq+4, ...
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (kq == 0) out[(w * MT * 16 + MT * i + tm) * (N + 1) + N] = s;
  }
}

// acc[m][0..N] += sum over slabs: acc has row stride `acc_stride`, the bias gradient lands in column N.  A block takes 64
// consecutive elements; its four waves each add up every fourth slab (coalesced 256-byte reads), the four sums meet
// in LDS and are combined in a fixed order -- deterministic.
__global__ __launch_bounds__(256) void mlp_grad_finish_kernel(const float* __restrict__ partial, int n_slabs, int M, int N,
                                                              float* __restrict__ acc, int acc_stride) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int id = blockIdx.x * 64 + lane;
  const int total = M * (N + 1);
  const int64_t stride = (int64_t)M * (N + 1);
  float s0 = 0.f, s1 = 0.f;
  if (id < total) {
    int b = w;
    for (; b + 4 < n_slabs; b += 8) { s0 += partial[b * stride + id]; s1 += partial[(b + 4) * stride + id]; }
    for (; b < n_slabs; b += 4) s0 += partial[b * stride + id];
  }
  part[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && id < total) {
    const int m = id / (N + 1), col = id - m * (N + 1);
    acc[m * acc_stride + col] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  }
}

constexpr int GRAD_SLABS = 512;

template <int MT, int NV, int NG>
__global__ __launch_bounds__(256, 2) void mlp_grad_partial_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                                  int64_t rows, int64_t rows_per_slab, int xc,
                                                                  float* __restrict__ partial, int gc,
                                                                  const unsigned char* __restrict__ gate = nullptr,
                                                                  int gate_parity = 0, int slabs_per_stage = 0) {
  mlp_grad_partial_body<MT, NV, NG>(G, X, rows, rows_per_slab, xc, partial, gc, gate, gate_parity, slabs_per_stage,
                                    (int)blockIdx.x, (int)blockIdx.y);
}

// K4am: both layers' factor rows of one attempt in ONE launch (blockIdx.y = 0, 1: the two row halves of [dW2 | db2] = G2^T U,
// 2: [dW1 | db1] = G1^T Z on the 128 x 36 tiles) -- a launch boundary less per attempted step
__global__ __launch_bounds__(256, 2) void mlp_grad_partial_pair_kernel(const float* __restrict__ G2, const float* __restrict__ U,
                                                                       const float* __restrict__ G1, const float* __restrict__ Z,
                                                                       int64_t rows, int64_t rows_per_slab,
                                                                       float* __restrict__ part2, float* __restrict__ part1,
                                                                       const unsigned char* __restrict__ gate, int gate_parity,
                                                                       int slabs_per_stage, const float* __restrict__ G2hi,
                                                                       float* __restrict__ part2hi) {
  // 32 hidden units x 16 channels: blockIdx.y = 3, 4 are the row halves of the upper hidden units' [dW2 | db2] = G2hi^T U
  if (blockIdx.y >= 3) {
    mlp_grad_partial_body<2, 4, 2>(G2hi, U, rows, rows_per_slab, 132, part2hi, 256, gate, gate_parity, slabs_per_stage,
                                   (int)blockIdx.x, (int)blockIdx.y - 3);
    return;
  }
  // (layer 2 in two halves of 128 rows: two workgroups per CU instead of one -- a single wave per SIMD overlaps neither its
  //  LDS reads nor its partial stores with its own MFMAs, the second workgroup's waves run in exactly those gaps)
  if (blockIdx.y < 2)
    mlp_grad_partial_body<2, 4, 2>(G2, U, rows, rows_per_slab, 132, part2, 256, gate, gate_parity, slabs_per_stage,
                                   (int)blockIdx.x, (int)blockIdx.y);
  else
    mlp_grad_partial_body<2, 2, 1>(G1, Z, rows, rows_per_slab, 36, part1, 128, gate, gate_parity, slabs_per_stage,
                                   (int)blockIdx.x, 0);
}

#define CDE_GRAD_PARTIAL(MTV, NVV, NGV, GRID, ...)                                                                 \
  do {                                                                                                             \
    const size_t lds_bytes = cde::grad_partial_lds_bytes<MTV, NVV, NGV>();                                         \
    (void)hipFuncSetAttribute((const void*)cde::mlp_grad_partial_kernel<MTV, NVV, NGV>,                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                         \
    cde::mlp_grad_partial_kernel<MTV, NVV, NGV><<<GRID, 256, lds_bytes, s>>>(__VA_ARGS__);                         \
  } while (0)

// Kw (rk4_wide.hip): acc (M, N + 1) += G^T [Z | 1], G (rows, M = 512), Z (rows, N = 64 or 32)
size_t wide_grad_reduce_partial_bytes(int M, int N) { return (size_t)GRAD_SLABS * M * (N + 1) * sizeof(float); }

int launch_wide_grad_reduce(const float* G, const float* Z, int64_t rows, int M, int N, float* acc, float* partial,
                            hipStream_t s) {
  if (rows <= 0) return CDE_OK;
  if (M % 256 != 0 || (N != 64 && N != 32)) return CDE_ERR_UNSUPPORTED;
  int64_t per = (rows + GRAD_SLABS - 1) / GRAD_SLABS;
  per = (per + 15) / 16 * 16;
  const int slabs = (int)((rows + per - 1) / per);
  const dim3 grid((unsigned)slabs, (unsigned)(M / 256));
  if (N == 64) CDE_GRAD_PARTIAL(4, 4, 1, grid, G, Z, rows, per, N, partial, M);
  else CDE_GRAD_PARTIAL(4, 2, 1, grid, G, Z, rows, per, N, partial, M);
  mlp_grad_finish_kernel<<<(M * (N + 1) + 63) / 64, 256, 0, s>>>(partial, slabs, M, N, acc, N + 1);
  return check_launch();
}

// K4am: the factor rows of one attempt launch -> per-slab partials of both layers (no second pass: the R kernel of
// dopri5_mlp_adjoint.hip adds the slabs of each stage in order)
int launch_mlp_adjoint_factor_reduce(const float* G2, const float* U, const float* G1, const float* Z, int64_t rows_per_stage,
                                     int sps, int64_t rows_per_slab, float* part2, float* part1, const unsigned char* ctrl,
                                     int parity, hipStream_t s, const float* G2hi, float* part2hi) {
  const int64_t rows = 7 * rows_per_stage;                     // (MADJ_FSLOTS blocks of rows; 6 of them are read per attempt)
  const dim3 grid((unsigned)(6 * sps), G2hi ? 5 : 3);
  const size_t a = grad_partial_lds_bytes<2, 4, 2>(), b = grad_partial_lds_bytes<2, 2, 1>();
  const size_t lds_bytes = a > b ? a : b;
  (void)hipFuncSetAttribute((const void*)mlp_grad_partial_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes);
  mlp_grad_partial_pair_kernel<<<grid, 256, lds_bytes, s>>>(G2, U, G1, Z, rows, rows_per_slab, part2, part1, ctrl, parity, sps, G2hi,
                                                               part2hi);
  return check_launch();
}

}  // namespace cde

// workspace for one reduction call: GRAD_SLABS partials of the larger layer (256 x 129 floats each)
extern "C" size_t cde_mlp_grad_reduce_workspace_bytes(void) { return (size_t)cde::GRAD_SLABS * 256 * 129 * sizeof(float); }

// layer = 2: G (rows, 256), X (rows, 132) = [u (128) | 1 | 0 0 0]  ->  acc (256, 132) += G^T X   (columns 0..128)
// layer = 1: G (rows, 128), X (rows, 36)  = [z (32)  | 1 | 0 0 0]  ->  acc (128, 36)  += G^T X   (columns 0..32)
extern "C" int cde_mlp_grad_reduce(const void* G, const void* X, int64_t rows, int layer, void* acc, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (rows < 0 || (layer != 1 && layer != 2)) return CDE_ERR_SHAPE;
  if (rows == 0) return CDE_OK;
  if (!G || !X || !acc || !workspace) return CDE_ERR_NULL;
  if (workspace_bytes < cde_mlp_grad_reduce_workspace_bytes()) return CDE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int64_t per = (rows + cde::GRAD_SLABS - 1) / cde::GRAD_SLABS;
  per = (per + 15) / 16 * 16;
  const int slabs = (int)((rows + per - 1) / per);
  if (layer == 2) {
    CDE_GRAD_PARTIAL(4, 4, 2, dim3((unsigned)slabs), (const float*)G, (const float*)X, rows, per, 132, (float*)workspace, 256);
    cde::mlp_grad_finish_kernel<<<(256 * 129 + 63) / 64, 256, 0, s>>>((const float*)workspace, slabs, 256, 128, (float*)acc, 132);
  } else {
    CDE_GRAD_PARTIAL(2, 2, 1, dim3((unsigned)slabs), (const float*)G, (const float*)X, rows, per, 36, (float*)workspace, 128);
    cde::mlp_grad_finish_kernel<<<(128 * 33 + 63) / 64, 256, 0, s>>>((const float*)workspace, slabs, 128, 32, (float*)acc, 36);
  }
  return cde::check_launch();
}
