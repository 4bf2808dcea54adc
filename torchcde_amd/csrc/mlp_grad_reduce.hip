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

namespace cde {

template <int MT> struct VecOf;
template <> struct VecOf<4> { using type = f32x4; };
template <> struct VecOf<2> { using type = f32x2; };

// M = 4 waves * MT * 16 rows per workgroup (blockIdx.y selects the block of M rows when a G row is wider: `gc` floats);
// N = NG * (NV * 16) columns, NV = floats per B operand read (4 or 2), NG groups of NV tiles.
// The slab is walked in tiles of GRAD_TILE_ROWS rows staged through LDS, double buffered: while the MFMAs of tile k run
// from LDS, the global loads of tile k+1 are in flight into registers, and are written to the other LDS buffer after
// the MFMAs.  (The first version fed the MFMAs straight from global loads with one group of 16 rows in flight: an HBM
// round trip under load is longer than the 128 MFMAs of such a group -- 57 % of the MFMA peak at 2.1 TB/s.  Deeper
// register rings do not survive the compiler's s_waitcnt placement across the loop's back edge; see DESIGN.md.)
constexpr int GRAD_TILE_ROWS = 16;

template <int MT, int NV, int NG>
constexpr size_t grad_partial_lds_bytes() {
  return (size_t)2 * GRAD_TILE_ROWS * ((4 * MT * 16 + 4) + (NG * NV * 16 + 4)) * sizeof(float);
}

template <int MT, int NV, int NG>
__global__ __launch_bounds__(256, 2) void mlp_grad_partial_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                                  int64_t rows, int64_t rows_per_slab, int xc,
                                                                  float* __restrict__ partial, int gc) {
  constexpr int M = 4 * MT * 16, NT = NG * NV, N = NT * 16, R = GRAD_TILE_ROWS;
  constexpr int GS = M + 4, XS = N + 4;                      // LDS row strides (floats)
  constexpr int GV = R * M / 4 / 256, XV = (R * N / 4 + 255) / 256;     // float4 per thread and tile
  using AV = typename VecOf<MT>::type;
  using BV = typename VecOf<NV>::type;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int m0 = blockIdx.y * M;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int64_t lo = (int64_t)blockIdx.x * rows_per_slab;
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
  float4 gst[GV], xst[XV];                                   // the tile in flight
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int f = tid + 256 * j, row = f / (M / 4), c4 = f % (M / 4);
      const int64_t r = r0 + row;
      gst[j] = r < hi ? *reinterpret_cast<const float4*>(G + r * gc + m0 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int f = tid + 256 * j, row = f / (N / 4), c4 = f % (N / 4);
      const int64_t r = r0 + row;
      xst[j] = (r < hi && row < R) ? *reinterpret_cast<const float4*>(X + r * xc + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&](int buf) {
    float* g_lds = lds + buf * R * (GS + XS);
    float* x_lds = g_lds + R * GS;
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int f = tid + 256 * j, row = f / (M / 4), c4 = f % (M / 4);
      *reinterpret_cast<float4*>(g_lds + row * GS + 4 * c4) = gst[j];
    }
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int f = tid + 256 * j, row = f / (N / 4), c4 = f % (N / 4);
      if (row < R) *reinterpret_cast<float4*>(x_lds + row * XS + 4 * c4) = xst[j];
    }
  };
  fetch(lo);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int64_t r0 = lo; r0 < hi; r0 += R) {
    const bool more = r0 + R < hi;
    if (more) fetch(r0 + R);
    const float* g_lds = lds + buf * R * (GS + XS) + (w * MT * 16 + MT * i);
    const float* x_lds = lds + buf * R * (GS + XS) + R * GS + NV * i;
#pragma unroll
    for (int s = 0; s < R / 4; ++s) {                        // K step s, lane quarter kq <-> row 4 s + kq of the tile
      const int row = 4 * s + kq;
      const AV a = *reinterpret_cast<const AV*>(g_lds + row * GS);
      BV b[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) b[g] = *reinterpret_cast<const BV*>(x_lds + row * XS + g * NV * 16);
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
    if (more) stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial[slab][m][N + 1]: D fragment of tile (tm, tn): lane (j = i, q = kq), register r = row 4q + r of the tile
  float* out = partial + ((int64_t)blockIdx.x * gc + m0) * (N + 1);
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
