// encoder.hip -- XLM-RoBERTa-large-shaped batch encoder (BGE-M3 / bge-reranker-v2-m3 architecture) on MFMA.
//
// Replaces the transformer forward the reference reaches through
//   rag/providers/hf/embeddings.py:54,77  (LangChain HuggingFaceEmbeddings -> sentence-transformers encode:
//   CLS pooling + L2 normalisation, normalize_embeddings=True at :34)
// and supplies the cross-encoder scoring the reference only configures (rag/config.py:25-27).
//
// Architecture (post-LN BERT family): embeddings (word + learned position + type) -> LayerNorm -> L x [ QKV projection,
// 16-head scaled-dot-product attention with key padding mask, output projection + residual + LayerNorm,
// FFN (H->F, exact-erf GELU, F->H) + residual + LayerNorm ] -> CLS row -> L2 normalise (embedding) or
// dense+tanh+out_proj (reranker logit).  Weights stay in torch-owned HBM tensors; this file only receives pointers.
//
// Kernels (all bf16 operands, fp32 accumulate on v_mfma_f32_16x16x32_bf16):
//   K5  embed_ln_kernel      gather + LayerNorm -> bf16
//   K6/K8/K9 gemm_bf16_kernel 128x128x64 LDS-tiled GEMM, C = A[M,K] * W[N,K]^T, fused epilogues:
//            QKV (bias, 1/8 scale on q, head-major q/k and TRANSPOSED v), GELU (bias + erf-GELU), RESID (bias + residual
//            -> fp32 pre-LayerNorm buffer)
//            gemm_skinny_kernel: the same products for <= 384 token rows (one query, a few short texts) as
//            weight-streaming workgroups of 16/32 columns; split-K partials are summed by the LayerNorm
//   K7  attention2_kernel    flash-style, computed transposed: 128 queries x 32-key tiles, online softmax in fp32, P stays in registers
//       layernorm_kernel     fp32 pre-LN rows -> bf16
//   K10 pool_kernel / rerank_head_kernel
// LDS tiles are stored k-chunk-major ([k/8][row][8 bf16]) with row ^= (chunk & 7): fragment reads (ds_read_b128) and
// the staging writes are both bank-conflict free (checked by brute force over the ds_read_b128 lane groups).
// Bound: MFMA (bf16 dense peak ~2.5 PFLOP/s); flops/token ~ L*(8H^2 + 4HF) + attention.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace hiprag {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __bf16 bf16;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kGemmThreads = 256;

__device__ __forceinline__ int swz_unit(int kc, int row, int rows) { return kc * rows + (row ^ (kc & 7)); }

enum { EPI_QKV = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PART = 3, EPI_RESID16 = 4 };   // RESID16: bf16 pre-LayerNorm rows (gemm256.h)

struct GemmArgs {
    const bf16* A;      // [M, K] row-major
    const bf16* W;      // [N, K] row-major (torch Linear weight)
    const float* bias;  // [N]
    int M, N, K;
    // EPI_QKV
    bf16* q;            // [nseq, heads, S, 64]
    bf16* k;            // [nseq, heads, S, 64]
    bf16* vt;           // [nseq, heads, 64, S]
    int S, heads, H;
    // EPI_GELU
    bf16* out_bf16;     // [M, N]
    // EPI_RESID
    const bf16* resid;  // [M, N]
    float* out_f32;     // [M, N]
    // EPI_PART: raw fp32 partial products [ksplit][M][N] (K split over workgroups; the LayerNorm that follows adds them)
    int ksplit;
};

// Exact-erf GELU, x * Phi(x), with ONE transcendental per element.  Phi(-z) = erfc(z / sqrt 2) / 2 for z = |x| is written
// as exp2(p(z)), p = degree-7 polynomial fit of log2 Phi(-z) on [0, 6] (Chebyshev least squares in fp64, monomial
// coefficients below): relative error of Phi(-z) <= 9.1e-6, absolute error of the GELU value <= 5.1e-7 over all x
// (checked against fp64 erf on 2M points; bf16, which the value is rounded to next, resolves 3.9e-3 relative).  Then
// GELU(x) = x * (x > 0 ? 1 - Phi(-z) : Phi(-z)); beyond |x| = 6 the tail is clamped at Phi(-6) = 1e-9.
// The Abramowitz-Stegun form used before took two transcendentals (v_rcp, v_exp: 8 issue cycles each) and ~15 other
// instructions per element; the H -> F epilogue runs 128 of these per lane and tile with nothing to hide behind.
__device__ __forceinline__ float gelu_exact(float x)
{
    const float z = fminf(fabsf(x), 6.f);
    float p = -1.9448814327915898e-06f;
    p = __builtin_fmaf(p, z, 6.385969027178362e-05f);
    p = __builtin_fmaf(p, z, -0.0009488638024777174f);
    p = __builtin_fmaf(p, z, 0.008582341484725475f);
    p = __builtin_fmaf(p, z, -0.05411824584007263f);
    p = __builtin_fmaf(p, z, -0.4582975208759308f);
    p = __builtin_fmaf(p, z, -1.1513246297836304f);
    p = __builtin_fmaf(p, z, -0.9999869465827942f);
    const float t = __builtin_amdgcn_exp2f(p);
    return x * (x > 0.f ? 1.f - t : t);
}

// LDS tile image: 128 rows x 64 bf16 = 128-B rows of eight 16-B chunks, chunk kc of row r stored at slot kc ^ (r & 7).
// One global_load_lds_dwordx4 wave-instruction fills eight whole rows (lane = row*8 + slot fetches chunk slot ^ (row&7)
// of its row: LDS side lane-linear as the DMA requires, global side 128 B contiguous per row), and the MFMA fragment
// reads (16 rows x one chunk per ds_read_b128 lane group) hit 16 distinct 16-B bank slots: conflict-free both ways.
__device__ __forceinline__ int tile_unit(int row, int kc) { return row * 8 + (kc ^ (row & 7)); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int EPI>
__global__ __launch_bounds__(kGemmThreads) void gemm_bf16_kernel(GemmArgs g)
{
    __shared__ bf16x8 lds[2][2][BM * BK / 8];  // [buffer][A|B][unit]; 64 KiB, reused as the fp32 C tile in the epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;    // 2 x 2 waves, 64 x 64 each
    const int r16 = lane & 15, kq = lane >> 4;

    // XCD-aware block mapping: consecutive block ids go round-robin over the 8 XCDs, so give each XCD a contiguous range
    // of tiles; inside it the N index runs fastest, i.e. the blocks sharing an A row-panel share one L2.
    const int nbn = g.N / BN;
    const int ksplit = EPI == EPI_PART ? g.ksplit : 1;
    const int nblk = gridDim.x / ksplit;
    const int split = blockIdx.x / nblk, bid = blockIdx.x - split * nblk;
    const int per = nblk >> 3, rem = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int pid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
    // tile order inside an XCD's range: groups of 8 column tiles, row tiles fastest inside a group -- the group's weight
    // tiles (1024 rows of W = 2 MB at K = 1024) stay L2-resident while the activations stream past them (3 % over
    // row-major tile order on the 256 x 512 batch)
    constexpr int GN = 8;
    const int nbm = nblk / nbn;
    const int grp = pid / (GN * nbm), rest = pid - grp * GN * nbm;
    const int gw = min(GN, nbn - grp * GN);          // width of this (possibly last, narrower) group
    const int m0 = (rest / gw) * BM, n0 = (grp * GN + rest % gw) * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging by LDS-DMA: 16 wave-instructions of 1 KiB per operand tile, 4 per wave
    const int srow = lane >> 3, sslot = lane & 7;
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int inst = wave * 4 + i;
            const int row = inst * 8 + srow;
            const int kc = sslot ^ (row & 7);
            const bf16* ga = g.A + (size_t)(m0 + row) * g.K + k0 + kc * 8;
            const bf16* gb = g.W + (size_t)(n0 + row) * g.K + k0 + kc * 8;
            __builtin_amdgcn_global_load_lds((const void*)ga, (lds_ptr_t)&lds[buf][0][inst * 64], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)gb, (lds_ptr_t)&lds[buf][1][inst * 64], 16, 0, 0);
        }
    };

    const int nk = g.K / ksplit / BK;
    const int kbeg = split * (g.K / ksplit);
    stage(0, kbeg);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of tile kt has landed
        __syncthreads();                                   // ... everyone's has, and buffer buf^1 is no longer being read
        if (kt + 1 < nk) stage(buf ^ 1, kbeg + (kt + 1) * BK);    // DMA of the next tile flies under this tile's MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
            const int kc = ks * 4 + kq;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = lds[buf][0][tile_unit(wr * 64 + i * 16 + r16, kc)];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = lds[buf][1][tile_unit(wc * 64 + j * 16 + r16, kc)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();  // all fragment reads done: the staging buffers become the C tile

    // ---- epilogue: C tile through LDS so that every global store is a 16-byte vector ---------------------------------
    float* ct = reinterpret_cast<float*>(&lds[0][0][0]);  // [128][128] fp32 = 64 KiB
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ct[(wr * 64 + i * 16 + 4 * kq + r) * BN + wc * 64 + j * 16 + r16] = acc[i][j][r];
    __syncthreads();

    if (EPI == EPI_QKV && n0 >= 2 * g.H) {
        // V third: stored transposed [seq, head, d, S]: thread = one column (d) x 8 consecutive tokens -> one 16-B store
        const int col = tid & 127, half = tid >> 7;
        const int n = n0 + col;
        const float bias = g.bias[n];
        const int hn = n - 2 * g.H;
        const int head = hn >> 6, dd = hn & 63;
        for (int rc = half; rc < 16; rc += 2) {
            const int m = m0 + rc * 8;
            if (m >= g.M) break;
            const int seq = m / g.S, s0 = m - seq * g.S;
            bf16x8 v;
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = (bf16)(ct[(rc * 8 + t) * BN + col] + bias);
            *reinterpret_cast<bf16x8*>(g.vt + (((size_t)seq * g.heads + head) * 64 + dd) * g.S + s0) = v;
        }
        return;
    }
    // row-major outputs: thread = 8 consecutive columns of one row; 16 column groups x 128 rows = 2048 vectors
    for (int v = tid; v < BM * (BN / 8); v += kGemmThreads) {
        const int row = v >> 4, cg = v & 15;
        const int m = m0 + row, n = n0 + cg * 8;
        if (m >= g.M) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(ct + row * BN + cg * 8);
        const float4 c1 = *reinterpret_cast<const float4*>(ct + row * BN + cg * 8 + 4);
        if (EPI == EPI_PART) {
            float* dst = g.out_f32 + ((size_t)split * g.M + m) * g.N + n;
            *reinterpret_cast<float4*>(dst) = c0;
            *reinterpret_cast<float4*>(dst + 4) = c1;
            continue;
        }
        const float4 b0 = *reinterpret_cast<const float4*>(g.bias + n);
        const float4 b1 = *reinterpret_cast<const float4*>(g.bias + n + 4);
        float x[8] = {c0.x + b0.x, c0.y + b0.y, c0.z + b0.z, c0.w + b0.w, c1.x + b1.x, c1.y + b1.y, c1.z + b1.z, c1.w + b1.w};
        if (EPI == EPI_QKV) {
            const int which = n / g.H, hn = n - which * g.H;   // which in {0 (q), 1 (k)}: the V third returned above
            const int head = hn >> 6, dd = hn & 63;
            const int seq = m / g.S, s = m - seq * g.S;
            const float scale = which == 0 ? 0.125f : 1.f;     // 1/sqrt(64) folded into q, exact
            bf16x8 o;
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] = (bf16)(x[t] * scale);
            bf16* dst = (which == 0 ? g.q : g.k) + ((((size_t)seq * g.heads + head) * g.S + s) * 64 + dd);
            *reinterpret_cast<bf16x8*>(dst) = o;
        } else if (EPI == EPI_GELU) {
            bf16x8 o;
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] = (bf16)gelu_exact(x[t]);
            *reinterpret_cast<bf16x8*>(g.out_bf16 + (size_t)m * g.N + n) = o;
        } else {
            const bf16x8 rs = *reinterpret_cast<const bf16x8*>(g.resid + (size_t)m * g.N + n);
            float4 o0, o1;
            o0.x = x[0] + (float)rs[0]; o0.y = x[1] + (float)rs[1]; o0.z = x[2] + (float)rs[2]; o0.w = x[3] + (float)rs[3];
            o1.x = x[4] + (float)rs[4]; o1.y = x[5] + (float)rs[5]; o1.z = x[6] + (float)rs[6]; o1.w = x[7] + (float)rs[7];
            float* dst = g.out_f32 + (size_t)m * g.N + n;
            *reinterpret_cast<float4*>(dst) = o0;
            *reinterpret_cast<float4*>(dst + 4) = o1;
        }
    }
}

#include "gemm256.h"

// ---------------------------------------------------------------------------------------------------------
// Small-batch GEMM (the single-query latency path: embed_single / a handful of short texts, rows <= ~1024).
// With 64..1024 activation rows the 128x128 tiles above leave 8..32 workgroups walking K serially (40 us for the F->H
// GEMM of one query); here the work is what it really is -- streaming the weight matrix once: a workgroup owns 16 output
// columns over a K range of <= 1024 (its four waves a quarter each, W fragments loaded straight from HBM in MFMA layout,
// 16 B per lane, and kept in registers), multiplies them with up to four 64-row blocks of A (fragments from L2), sums the
// four waves' partial tiles through LDS in a fixed order and runs the epilogue.  K > 1024 splits over workgroups
// (ksplit = K / Kc) which write raw fp32 partials [ksplit][M][N]; the LayerNorm that follows adds them up in split order,
// so there are no atomics and results do not depend on scheduling.
// ---------------------------------------------------------------------------------------------------------
constexpr int SK_STEPS = 8;   // 32-deep MFMA steps per wave (K range per workgroup <= 4 * 8 * 32 = 1024)

// NT: 16-column tiles per wave (the workgroup owns 16 * NT output columns; NT = 2 halves the L2 reads of A, which every
// column tile repeats, and is used from 128 rows up).  FULL: the K range per workgroup is exactly 1024 (the real model:
// H = 1024, F = 4096) -- every load is issued before the first MFMA; otherwise a plain loop over `steps`.
template <int EPI, int NT, bool FULL>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g, int kc, int steps, int mb_per_wg)
{
    constexpr int WN = 16 * NT;
    __shared__ float red[4][64][WN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int ntile = g.N / WN;
    const int split = blockIdx.x / ntile, n0 = (blockIdx.x - split * ntile) * WN;
    const int kw0 = split * kc + wave * (kc >> 2);
    const size_t k16 = (size_t)16 * g.K;

    const bf16* wp = g.W + (size_t)(n0 + r16) * g.K + kw0 + kq * 8;
    bf16x8 bfr[NT][SK_STEPS];
    if (FULL) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int s = 0; s < SK_STEPS; ++s) bfr[j][s] = *reinterpret_cast<const bf16x8*>(wp + j * k16 + s * 32);
    }

    const int mb0 = blockIdx.y * mb_per_wg, mb1 = min(mb0 + mb_per_wg, g.M >> 6);
    for (int mb = mb0; mb < mb1; ++mb) {
        f32x4 acc[4][NT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16* ap = g.A + (size_t)(mb * 64 + r16) * g.K + kw0 + kq * 8;
        if (FULL) {
            bf16x8 af[4][SK_STEPS];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < SK_STEPS; ++s) af[i][s] = *reinterpret_cast<const bf16x8*>(ap + i * k16 + s * 32);
            __builtin_amdgcn_sched_barrier(0);  // all 32 (+ 8 NT) loads in flight before the first MFMA waits: one round trip
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < SK_STEPS; ++s)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], bfr[j][s], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll 1
            for (int s = 0; s < steps; ++s) {
                bf16x8 b[NT], a[4];
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(wp + j * k16 + s * 32);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ap + i * k16 + s * 32);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][i * 16 + 4 * kq + r][j * 16 + r16] = acc[i][j][r];
        __syncthreads();

        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        if (EPI == EPI_QKV && n0 >= 2 * g.H) {
            // V third, stored transposed [seq, head, d, S]: one column x 4 consecutive tokens -> one 8-B store
#pragma unroll
            for (int e = 0; e < NT; ++e) {
                const int ch = tid + 256 * e;
                const int col = ch % WN, rg = (ch / WN) * 4;
                const int n = n0 + col, hn = n - 2 * g.H;
                const int head = hn >> 6, dd = hn & 63;
                const float bias = g.bias[n];
                const int m = mb * 64 + rg;
                const int seq = m / g.S, s0 = m - seq * g.S;
                bf16x4 v;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = (bf16)((((red[0][rg + u][col] + red[1][rg + u][col]) + red[2][rg + u][col]) + red[3][rg + u][col]) + bias);
                *reinterpret_cast<bf16x4*>(g.vt + (((size_t)seq * g.heads + head) * 64 + dd) * g.S + s0) = v;
            }
        } else {
            // row-major outputs: 4 consecutive columns of one row per item
#pragma unroll
            for (int e = 0; e < NT; ++e) {
                const int ch = tid + 256 * e;
                const int row = ch / (4 * NT), c4 = (ch % (4 * NT)) * 4;
                const int m = mb * 64 + row, n = n0 + c4;
                const float4 p0 = *reinterpret_cast<const float4*>(&red[0][row][c4]);
                const float4 p1 = *reinterpret_cast<const float4*>(&red[1][row][c4]);
                const float4 p2 = *reinterpret_cast<const float4*>(&red[2][row][c4]);
                const float4 p3 = *reinterpret_cast<const float4*>(&red[3][row][c4]);
                float x[4] = {((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                              ((p0.w + p1.w) + p2.w) + p3.w};
                if (EPI == EPI_PART) {
                    *reinterpret_cast<float4*>(g.out_f32 + ((size_t)split * g.M + m) * g.N + n) = make_float4(x[0], x[1], x[2], x[3]);
                } else {
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
                    x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
                    bf16x4 o;
                    if (EPI == EPI_QKV) {
                        const int which = n / g.H, hn = n - which * g.H;   // 0 (q) or 1 (k)
                        const int head = hn >> 6, dd = hn & 63;
                        const int seq = m / g.S, s = m - seq * g.S;
                        const float scale = which == 0 ? 0.125f : 1.f;
#pragma unroll
                        for (int u = 0; u < 4; ++u) o[u] = (bf16)(x[u] * scale);
                        bf16* dst = (which == 0 ? g.q : g.k) + ((((size_t)seq * g.heads + head) * g.S + s) * 64 + dd);
                        *reinterpret_cast<bf16x4*>(dst) = o;
                    } else {  // EPI_GELU
#pragma unroll
                        for (int u = 0; u < 4; ++u) o[u] = (bf16)gelu_exact(x[u]);
                        *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.N + n) = o;
                    }
                }
            }
        }
        __syncthreads();  // the partial tiles are free for the next row block
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over rows of an fp32 [M,H] buffer -> bf16; one wave per row.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// PARTS: the row is the sum of `nsplit` fp32 partial products (gemm_skinny_kernel<EPI_PART>, split order) + bias + the
// bf16 residual row -- the same (c + bias) + residual order as the EPI_RESID epilogue of the tiled GEMM.
template <bool PARTS>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16* __restrict__ y, int M, int H,
                                                       float eps, int nsplit, const float* __restrict__ bias,
                                                       const bf16* __restrict__ resid)
{
    // one wave per row; the row is read ONCE (16 B per lane per load, kept in registers: H <= 2048 -> <= 8 float4)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * H);
    const int nv = H >> 8;  // float4 per lane (H is a multiple of 128; a 128-wide remainder is handled by half the lanes)
    float4 v[8], gm[8], bt[8];
    float s = 0.f;
    const int nvec = H >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    // every load of the row's working set is issued up front and UNCONDITIONALLY (clamped column): a load under a lane
    // mask is compiled into branch + load + wait, one memory round trip each
    if (PARTS) {   // latency path only: the batch path is bandwidth-bound and better off with fewer live registers
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cc = min(i * 64 + lane, nvec - 1);
            gm[i] = g4[cc];
            bt[i] = b4[cc];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 64 + lane;
        if (PARTS) {
            v[i] = xr[min(c, nvec - 1)];
            if (c >= nvec) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            v[i] = c < nvec ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (PARTS) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4r;
            const int cc = min(c, nvec - 1);
            float4 p[3];
#pragma unroll
            for (int sp = 1; sp < 4; ++sp) p[sp - 1] = xr[(size_t)min(sp, nsplit - 1) * M * (H >> 2) + cc];
            const float4 b = reinterpret_cast<const float4*>(bias)[cc];
            const bf16x4r rs = reinterpret_cast<const bf16x4r*>(resid + (size_t)row * H)[cc];
            if (c < nvec) {
#pragma unroll
                for (int sp = 1; sp < 4; ++sp)
                    if (sp < nsplit) { v[i].x += p[sp - 1].x; v[i].y += p[sp - 1].y; v[i].z += p[sp - 1].z; v[i].w += p[sp - 1].w; }
                v[i].x = (v[i].x + b.x) + (float)rs[0]; v[i].y = (v[i].y + b.y) + (float)rs[1];
                v[i].z = (v[i].z + b.z) + (float)rs[2]; v[i].w = (v[i].w + b.w) + (float)rs[3];
            }
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    (void)nv;
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4* yr = reinterpret_cast<bf16x4*>(y + (size_t)row * H);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            const float4 g = PARTS ? gm[i] : g4[c], b = PARTS ? bt[i] : b4[c];
            bf16x4 o;
            o[0] = (bf16)((v[i].x - mean) * rstd * g.x + b.x);
            o[1] = (bf16)((v[i].y - mean) * rstd * g.y + b.y);
            o[2] = (bf16)((v[i].z - mean) * rstd * g.z + b.z);
            o[3] = (bf16)((v[i].w - mean) * rstd * g.w + b.w);
            yr[c] = o;
        }
    }
}

// LayerNorm over bf16 pre-LN rows (the big-batch path: gemm256's RESID16 epilogue) -> bf16.  One wave per FOUR consecutive
// rows: all of their loads (16 B per lane each, H <= 1024 here: 2 per row) are in flight before the first reduction, gamma
// and beta are loaded once per wave; statistics in fp32.  HBM-bound: 2 + 2 bytes per element.
constexpr int kLn16Rows = 4;
__global__ __launch_bounds__(256) void layernorm16_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16* __restrict__ y, int M, int H,
                                                         float eps)
{
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kLn16Rows;
    if (row0 >= M) return;
    const int nvec = H >> 3;                                   // 16-B vectors per row, <= 128
    const int c0 = min(lane, nvec - 1), c1 = min(64 + lane, nvec - 1);
    const bool a0 = lane < nvec, a1 = 64 + lane < nvec;
    bf16x8 v[kLn16Rows][2];
#pragma unroll
    for (int r = 0; r < kLn16Rows; ++r) {
        const bf16x8* xr = reinterpret_cast<const bf16x8*>(x + (size_t)min(row0 + r, M - 1) * H);
        v[r][0] = xr[c0];
        v[r][1] = xr[c1];
    }
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float gg[2][8], bb[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = i == 0 ? c0 : c1;
        const float4 g0 = g4[2 * c], g1 = g4[2 * c + 1], b0 = b4[2 * c], b1 = b4[2 * c + 1];
        gg[i][0] = g0.x; gg[i][1] = g0.y; gg[i][2] = g0.z; gg[i][3] = g0.w; gg[i][4] = g1.x; gg[i][5] = g1.y; gg[i][6] = g1.z; gg[i][7] = g1.w;
        bb[i][0] = b0.x; bb[i][1] = b0.y; bb[i][2] = b0.z; bb[i][3] = b0.w; bb[i][4] = b1.x; bb[i][5] = b1.y; bb[i][6] = b1.z; bb[i][7] = b1.w;
    }
#pragma unroll
    for (int r = 0; r < kLn16Rows; ++r) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (a0 ? (float)v[r][0][e] : 0.f) + (a1 ? (float)v[r][1][e] : 0.f);
        const float mean = wave_sum(s) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d0 = (float)v[r][0][e] - mean, d1 = (float)v[r][1][e] - mean;
            q += (a0 ? d0 * d0 : 0.f) + (a1 ? d1 * d1 : 0.f);
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
        if (row0 + r < M) {
            bf16x8* yr = reinterpret_cast<bf16x8*>(y + (size_t)(row0 + r) * H);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)(((float)v[r][i][e] - mean) * rstd * gg[i][e] + bb[i][e]);
                if (i == 0 ? a0 : a1) yr[i == 0 ? c0 : c1] = o;
            }
        }
    }
}

// K5: embeddings.  tokens [nseq, S] (pad beyond len), position id = s + pad_id + 1 for real tokens, pad_id for pads
// (transformers' create_position_ids_from_input_ids with no interior pads).  One wave per token.
__global__ __launch_bounds__(256) void embed_ln_kernel(const int* __restrict__ tokens, const int* __restrict__ lens, int S,
                                                      int nseq, const bf16* __restrict__ word, const bf16* __restrict__ pos,
                                                      const bf16* __restrict__ type0, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16* __restrict__ y, int H, int pad_id,
                                                      float eps, int M)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    bf16* yr = y + (size_t)row * H;
    const int seq = row / S, s = row - seq * S;
    if (seq >= nseq || s >= lens[seq]) {  // padding rows are zeros: they never reach a real token (keys are masked)
        for (int c = lane; c < H; c += 64) yr[c] = (bf16)0.f;
        return;
    }
    const int tok = tokens[(size_t)seq * S + s];
    const bf16* w = word + (size_t)tok * H;
    const bf16* p = pos + (size_t)(s + pad_id + 1) * H;
    float vals[32];  // H <= 2048
    float sum = 0.f;
    int n = 0;
    for (int c = lane; c < H; c += 64, ++n) {
        const float t = (float)w[c] + (float)p[c] + (float)type0[c];
        vals[n] = t;
        sum += t;
    }
    const float mean = wave_sum(sum) / (float)H;
    float var = 0.f;
    for (int i = 0; i < n; ++i) { const float t = vals[i] - mean; var += t * t; }
    const float rstd = rsqrtf(wave_sum(var) / (float)H + eps);
    n = 0;
    for (int c = lane; c < H; c += 64, ++n) yr[c] = (bf16)((vals[n] - mean) * rstd * gamma[c] + beta[c]);
}

// ---------------------------------------------------------------------------------------------------------
// K7: attention -- transposed flash attention on v_mfma_f32_32x32x16_bf16 with LDS-DMA staging.
// q,k: [nseq, heads, S, 64] (q pre-scaled by 1/8), vt: [nseq, heads, 64, S]; ctx out: [M, H] row-major.  Everything is computed
// TRANSPOSED (S^T = K Q^T, O^T = V^T P^T) so that the probabilities never leave registers.
// What round 1's kernel (16x16x32 MFMAs, register staging; removed in round 3) spent its time on was not the matrix pipe
// (0.17 busy, r01) but ~550 vector instructions per 64-key tile and wave: the key mask, the rescale of the output accumulators and their trips to and from the accumulator file on
// EVERY tile, three operations per exponential, four cross-lane shuffles per tile, K / V tiles hauled global -> registers
// -> LDS between two barriers.  Here:
//   * a wave owns ONE tile of 32 queries; S^T = K Q^T puts a query on lanes q and q + 32, each holding 16 of a 32-key
//     tile's scores: the row maximum needs one v_permlane32_swap per 64 keys, the row sum none at all until the end
//     (every lane keeps its own partial sum; the rescale factor is the same for both halves);
//   * the accumulators are rescaled only in tiles where some row maximum of the wave actually rose (wave-uniform branch;
//     exact -- not a thresholded skip), the key mask only runs in the one tile that crosses the sequence length;
//   * p = exp2(s * log2e - m * log2e): one fma + one v_exp per score;
//   * K and V^T tiles (8 KiB each) arrive by LDS-DMA into a double buffer, swizzled on the source side
//     (chunk ^ ((row >> 1) & 7): conflict-free for the b128 K-fragment reads AND the 8-byte V^T-fragment reads): one
//     barrier per tile, the next tile's DMA in flight under this tile's MFMAs;
//   * P^T never leaves registers: element j of lane half h of k-step s is key 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the order
//     the S^T accumulators already have -- and the V^T fragments are read in that order (two 8-byte runs);
//   * the output tile goes through the (by then free) LDS buffers into 16-byte row-major stores.
// grid (ceil(S/128), heads, nseq), 4 waves x 32 queries.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256, 2) void attention2_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                        const bf16* __restrict__ vt, const int* __restrict__ lens,
                                                        bf16* __restrict__ ctx, int S, int heads, int H)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * 16384];   // [buffer][K tile 8 KiB | V^T tile 8 KiB]
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int seq = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 128;
    const int len = lens[seq];
    if (q0 >= len) return;   // whole query tile is padding (ctx rows stay zero: memset by the caller)
    const size_t hb = (size_t)seq * heads + head;
    const bf16* qh = q + hb * S * 64;
    const bf16* kh = k + hb * S * 64;
    const bf16* vh = vt + hb * 64 * S;
    const int nt = (len + 63) >> 6;

    // staging: 16 wave-instructions per tile (8 rows of 128 B each), 4 per wave: waves 0,1 the K tile, waves 2,3 V^T
    const int srow = lane >> 3, sslot = lane & 7;
    auto issue = [&](int t, int buf) {
        const int kt0 = t * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int inst = (wave & 1) * 4 + u;             // 0..7 inside the operand tile
            const int row = inst * 8 + srow;                 // key (K) or d (V^T)
            const int chunk = sslot ^ ((row >> 1) & 7);
            const bf16* src = wave < 2 ? kh + (size_t)(kt0 + row) * 64 + chunk * 8 : vh + (size_t)row * S + kt0 + chunk * 8;
            char* dst = lds + buf * 16384 + (wave < 2 ? 0 : 8192) + inst * 1024;
            __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr_t)dst, 16, 0, 0);
        }
    };
    issue(0, 0);

    // Q fragments (B operand of S^T = K Q^T): lane (query r32, half hh) holds Q[query][16 ks + 8 hh .. + 7], ks = 0..3
    bf16x8 qf[4];
    {
        const int qi = min(q0 + wave * 32 + r32, S - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qh + (size_t)qi * 64 + ks * 16 + hh * 8);
    }
    f32x16 o[2];   // O^T tiles: d = 32 dt + (r & 3) + 8 (r >> 2) + 4 hh, query r32
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float mrow = -INFINITY, lpart = 0.f;       // running max of the query's row; THIS lane's share of the row sum
    const float LOG2E = 1.4426950408889634f;
    const int swz = (r32 >> 1) & 7;            // swizzle of this lane's LDS row (row = r32 in every fragment read)
    // 16 registers of zeros that stay zeros: the first MFMA of every score tile takes them as its C operand (D != C), which
    // saves the 32 v_mov per tile that zeroing two accumulator tiles costs; opaque, or hipcc re-materialises them
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    asm volatile("" : "+v"(zero16));

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1, kt0 = t * 64;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile t has landed
        __builtin_amdgcn_s_barrier();                       // ... everyone's has; buffer buf ^ 1 is no longer being read
        if (t + 1 < nt) issue(t + 1, buf ^ 1);
        const char* kb = lds + buf * 16384;
        const char* vb = kb + 8192;
        // S^T tiles: sc[kt][r] = score(key kt0 + 32 kt + (r & 3) + 8 (r >> 2) + 4 hh, query r32)
        f32x16 sc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + (kt * 32 + r32) * 128 + (((2 * ks + hh) ^ swz) << 4));
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero16 : sc[kt], 0, 0, 0);
            }
        }
        if (kt0 + 64 > len) {   // the one tile that crosses the sequence length: padded keys never win
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * hh >= len) sc[kt][r] = -INFINITY;
        }
        float tmax = sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[1][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));           // the other half of the query's scores
        if (__any(tmax > mrow)) {                            // some row maximum of this wave rose: rescale (exact)
            const float mn = fmaxf(mrow, tmax);              // finite: key 0 of the first tile is always valid
            const float alpha = __builtin_amdgcn_exp2f((mrow - mn) * LOG2E);
            mrow = mn;
            lpart *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        const float mL = mrow * LOG2E;
        // P^T fragments: k-step s of key tile kt takes accumulator registers 8 s .. 8 s + 7
        bf16x8 pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][8 * s + j], LOG2E, -mL));
                    lpart += p;
                    pf[kt][s][j] = (bf16)p;
                }
        // O^T += V^T P^T: A fragment of d tile dt, key tile kt, k-step s = V^T[d = 32 dt + r32][keys 32 kt + 16 s + 4 hh + 0..3
        // and 32 kt + 16 s + 8 + 4 hh + 0..3]: two 8-byte runs, 16-B chunk (4 kt + 2 s) resp. (4 kt + 2 s + 1), half hh
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vr = vb + (dt * 32 + r32) * 128 + hh * 8;
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vr + (((4 * kt + 2 * s) ^ swz) << 4));
                    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vr + (((4 * kt + 2 * s + 1) ^ swz) << 4));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][s], o[dt], 0, 0, 0);
                }
    }
    // row sum: the two halves of a query hold disjoint keys
    const float inv = 1.f / (lpart + __shfl_xor(lpart, 32));
    __builtin_amdgcn_s_barrier();    // every wave is done with the K / V^T buffers: they become the output staging area
    char* ob = lds + wave * 4096;    // 32 queries x 128 B, 16-B chunk c of row q at c ^ (q & 7)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (bf16)(o[dt][4 * g4 + r] * inv);
            const int d = 32 * dt + 8 * g4 + 4 * hh;       // 4 consecutive d
            *reinterpret_cast<bf16x4*>(ob + r32 * 128 + (((d >> 3) ^ (r32 & 7)) << 4) + (d & 4) * 2) = v;
        }
    // read back: 8 lanes per query row (16 B each), 8 rows per pass, 4 passes
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = p * 8 + (lane >> 3), c = lane & 7;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ob + row * 128 + ((c ^ (row & 7)) << 4));
        const int qi = q0 + wave * 32 + row;
        if (qi < S) *reinterpret_cast<bf16x8*>(ctx + ((size_t)seq * S + qi) * H + head * 64 + c * 8) = v;
    }
}

// K10a: CLS row (token 0 of each sequence) -> L2 normalise -> fp32 [nseq, H]; zero for empty sequences.
__global__ __launch_bounds__(64) void pool_kernel(const bf16* __restrict__ x, const int* __restrict__ lens, int S, int H,
                                                 float* __restrict__ out, int normalize)
{
    const int seq = blockIdx.x, lane = threadIdx.x;
    const bf16* xr = x + (size_t)seq * S * H;
    float ss = 0.f;
    for (int c = lane; c < H; c += 64) { const float v = (float)xr[c]; ss += v * v; }
    ss = wave_sum(ss);
    const float inv = (lens[seq] > 0) ? (normalize ? 1.f / fmaxf(sqrtf(ss), 1e-12f) : 1.f) : 0.f;
    for (int c = lane; c < H; c += 64) out[(size_t)seq * H + c] = (float)xr[c] * inv;
}

// K10b: XLM-R classification head on the CLS row: logit = w_out . tanh(W_dense cls + b_dense) + b_out.
__global__ __launch_bounds__(256) void rerank_head_kernel(const bf16* __restrict__ x, int S, int H,
                                                         const bf16* __restrict__ wd, const float* __restrict__ bd,
                                                         const bf16* __restrict__ wo, const float* __restrict__ bo,
                                                         float* __restrict__ logits)
{
    __shared__ float cls[2048];
    __shared__ float part[4];
    const int seq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16* xr = x + (size_t)seq * S * H;
    for (int c = tid; c < H; c += 256) cls[c] = (float)xr[c];
    __syncthreads();
    float acc = 0.f;
    for (int n = wave; n < H; n += 4) {  // one wave per output feature
        const bf16* w = wd + (size_t)n * H;
        float s = 0.f;
        for (int c = lane; c < H; c += 64) s += (float)w[c] * cls[c];
        s = wave_sum(s);
        if (lane == 0) acc += (float)wo[n] * tanhf(s + bd[n]);
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (tid == 0) logits[seq] = part[0] + part[1] + part[2] + part[3] + bo[0];
}

// ---------------------------------------------------------------------------------------------------------
struct Encoder {
    std::mutex mu;
    int device = 0;
    hipenc_config cfg{};
    hipenc_weights w{};
    std::vector<hipenc_layer_weights> layers;
    DevBuf tokens, lens, x, q, k, vt, ctx, pre, ffn, pooled;
    double flops_last = 0.0;
    int small_rows = 384;    // batches of at most this many (padded) token rows take the small-batch GEMM; 0 = never
                             // (HIPENC_SMALL_ROWS; measured crossover with the split-K tiled GEMM: 256 rows 1.72 vs 2.11 ms,
                             // 512 rows 2.37 vs 2.17 ms)

    int n_cu = 256;
    int use256 = 1;          // HIPENC_GEMM256=0 keeps every batch on the 128 x 128 kernel (A/B runs)

    // the 256 x 256 persistent kernel: whole 256-tiles in M and N, an even number of k-tiles, and enough tiles to fill the CUs
    bool big_ok(int M, int N, int K) const
    {
        return use256 && M % G2_T == 0 && N % G2_T == 0 && K % (2 * G2_BK) == 0 && (M / G2_T) * (N / G2_T) >= n_cu / 2;
    }
    template <int EPI>
    void launch256(const GemmArgs& a, int Mpad, hipStream_t st) const
    {
        const int nbm = Mpad / G2_T, nbn = a.N / G2_T;
        const int grid = std::min(nbm * nbn, n_cu);
        hipLaunchKernelGGL(gemm256_kernel<EPI>, dim3(grid), dim3(G2_THREADS), 0, st, a, nbm, nbn);
    }

    // K range per workgroup of the small-batch GEMM: K / ksplit, a multiple of 128 and at most 1024
    static int skinny_split(int K) { return (K + 1023) / 1024; }
    static bool skinny_ok(int K) { const int sp = skinny_split(K); return K % (sp * 128) == 0; }

    template <int EPI>
    static void launch_skinny(GemmArgs a, hipStream_t st)
    {
        const int sp = skinny_split(a.K), kc = a.K / sp, steps = kc / 128;
        const bool wide = a.M >= 128 && a.N % 32 == 0;     // two column tiles per wave: half the L2 reads of A
        // row blocks run one after the other inside a workgroup (a round trip to L2 each): spread them over workgroups
        // until the launch has ~1024 of them, W comes out of L2 for all but the first
        const int gx = (a.N / (wide ? 32 : 16)) * sp, mblocks = a.M >> 6;
        const int mb = std::max(1, (gx * mblocks + 1023) / 1024);
        const dim3 grid(gx, (mblocks + mb - 1) / mb);
        if (steps == SK_STEPS) {
            if (wide) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 2, true>), grid, dim3(256), 0, st, a, kc, steps, mb);
            else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 1, true>), grid, dim3(256), 0, st, a, kc, steps, mb);
        } else {
            if (wide) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 2, false>), grid, dim3(256), 0, st, a, kc, steps, mb);
            else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 1, false>), grid, dim3(256), 0, st, a, kc, steps, mb);
        }
    }

    int32_t forward(const int32_t* tok_host, const int32_t* lens_host, int nseq, int max_len, float* out_dev, int mode,
                    hipStream_t st)
    {
        const int H = cfg.hidden, F = cfg.ffn, heads = cfg.heads;
        const int S = ((max_len + 63) / 64) * 64;
        const int T = nseq * S;
        // big batches: 256 x 256 persistent tiles (gemm256.h) once the narrowest GEMM (N = H) fills every CU
        const int M256 = ((T + G2_T - 1) / G2_T) * G2_T;
        const bool big = H <= 1024 && big_ok(M256, H, H) && big_ok(M256, F, H) && big_ok(M256, H, F) && (M256 / G2_T) * (H / G2_T) >= n_cu;
        const int M = big ? M256 : ((T + BM - 1) / BM) * BM;
        int32_t rc;
        if ((rc = tokens.reserve((size_t)nseq * S * 4))) return rc;
        if ((rc = lens.reserve((size_t)nseq * 4))) return rc;
        if ((rc = x.reserve((size_t)M * H * 2))) return rc;
        if ((rc = q.reserve((size_t)M * H * 2))) return rc;
        if ((rc = k.reserve((size_t)M * H * 2))) return rc;
        if ((rc = vt.reserve((size_t)M * H * 2))) return rc;
        if ((rc = ctx.reserve((size_t)M * H * 2))) return rc;
        // one query / a few short texts: weight-streaming GEMMs instead of 128 x 128 tiles (S is a multiple of 64, so is T)
        const bool small = small_rows > 0 && T <= small_rows && H <= 1024 && skinny_ok(H) && skinny_ok(F) && skinny_split(F) <= 4;
        const int fsplit = skinny_split(F);
        // tiled path, N = H products (out-proj, F -> H): with few row tiles their (H/128) x (M/128) workgroups leave most
        // CUs idle while each walks K serially (F -> H at 2560 rows: 160 workgroups x 64 k-tiles = 64 us of a 165 us
        // layer) -- split K until the launch has ~512 workgroups; the LayerNorm sums the partials
        auto tile_split = [&](int K) {
            int ks = 1;
            while (ks < 4 && (H / BN) * (M / BM) * ks < 512 && K % (ks * 2 * BK) == 0 && K / (ks * 2) >= 256) ks *= 2;
            return ks;
        };
        const int osplit = (small || big) ? 1 : tile_split(H), dsplit = (small || big) ? 1 : tile_split(F);
        if ((rc = pre.reserve((size_t)M * H * 4 * (small ? fsplit : std::max(osplit, dsplit))))) return rc;
        if ((rc = ffn.reserve((size_t)M * F * 2))) return rc;
        // host staging: pad token rows to S with pad_id
        std::vector<int32_t> tp((size_t)nseq * S, cfg.pad_id);
        for (int s = 0; s < nseq; ++s) {
            if (lens_host[s] < 0 || lens_host[s] > max_len) { set_error("seq_lens[%d]=%d outside [0,%d]", s, lens_host[s], max_len); return HIPRAG_E_INVALID; }
            for (int t = 0; t < lens_host[s]; ++t) {
                const int32_t id = tok_host[(size_t)s * max_len + t];
                if (id < 0 || id >= cfg.vocab) { set_error("token id %d outside the vocabulary", id); return HIPRAG_E_INVALID; }
                tp[(size_t)s * S + t] = id;
            }
        }
        HR_CHECK_HIP(hipMemcpyAsync(tokens.p, tp.data(), tp.size() * 4, hipMemcpyHostToDevice, st));
        HR_CHECK_HIP(hipMemcpyAsync(lens.p, lens_host, (size_t)nseq * 4, hipMemcpyHostToDevice, st));
        HR_CHECK_HIP(hipStreamSynchronize(st));  // tp is a local vector
        HR_CHECK_HIP(hipMemsetAsync(ctx.p, 0, (size_t)M * H * 2, st));  // rows of skipped (all-padding) query tiles

        const bf16* X = x.as<bf16>();
        hipLaunchKernelGGL(embed_ln_kernel, dim3((M + 3) / 4), dim3(256), 0, st, tokens.as<int>(), lens.as<int>(), S, nseq,
                           (const bf16*)w.word_emb, (const bf16*)w.pos_emb, (const bf16*)w.type_emb, (const float*)w.emb_ln_g,
                           (const float*)w.emb_ln_b, x.as<bf16>(), H, cfg.pad_id, cfg.ln_eps, M);
        for (int l = 0; l < cfg.layers; ++l) {
            const hipenc_layer_weights& L = layers[l];
            GemmArgs g{};
            g.A = X; g.W = (const bf16*)L.wqkv; g.bias = (const float*)L.bqkv; g.M = M; g.N = 3 * H; g.K = H;
            g.q = q.as<bf16>(); g.k = k.as<bf16>(); g.vt = vt.as<bf16>(); g.S = S; g.heads = heads; g.H = H;
            // rows >= T exist only as GEMM padding; the QKV scatter must not write them
            g.M = T;
            if (small) launch_skinny<EPI_QKV>(g, st);
            else if (big) launch256<EPI_QKV>(g, M, st);
            else hipLaunchKernelGGL(gemm_bf16_kernel<EPI_QKV>, dim3((3 * H / BN) * (M / BM)), dim3(kGemmThreads), 0, st, g);
            hipLaunchKernelGGL(attention2_kernel, dim3((S + 127) / 128, heads, nseq), dim3(256), 0, st, (const bf16*)q.as<bf16>(),
                                   (const bf16*)k.as<bf16>(), (const bf16*)vt.as<bf16>(), (const int*)lens.as<int>(), ctx.as<bf16>(), S,
                                   heads, H);
            GemmArgs o{};
            o.A = ctx.as<bf16>(); o.W = (const bf16*)L.wo; o.bias = (const float*)L.bo; o.M = M; o.N = H; o.K = H;
            o.resid = X; o.out_f32 = pre.as<float>();
            if (small) {
                o.M = T;
                launch_skinny<EPI_PART>(o, st);
                hipLaunchKernelGGL(layernorm_kernel<true>, dim3((T + 3) / 4), dim3(256), 0, st, (const float*)pre.as<float>(),
                                   (const float*)L.ln1_g, (const float*)L.ln1_b, x.as<bf16>(), T, H, cfg.ln_eps, skinny_split(H),
                                   (const float*)L.bo, X);
            } else if (big) {
                o.out_bf16 = pre.as<bf16>();      // bf16 pre-LayerNorm rows: half the store and LayerNorm-read bytes
                launch256<EPI_RESID16>(o, M, st);
                hipLaunchKernelGGL(layernorm16_kernel, dim3((M + 4 * kLn16Rows - 1) / (4 * kLn16Rows)), dim3(256), 0, st, (const bf16*)pre.as<bf16>(),
                                   (const float*)L.ln1_g, (const float*)L.ln1_b, x.as<bf16>(), M, H, cfg.ln_eps);
            } else if (osplit > 1) {
                o.ksplit = osplit;
                hipLaunchKernelGGL(gemm_bf16_kernel<EPI_PART>, dim3((H / BN) * (M / BM) * osplit), dim3(kGemmThreads), 0, st, o);
                hipLaunchKernelGGL(layernorm_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, (const float*)pre.as<float>(),
                                   (const float*)L.ln1_g, (const float*)L.ln1_b, x.as<bf16>(), M, H, cfg.ln_eps, osplit,
                                   (const float*)L.bo, X);
            } else {
                hipLaunchKernelGGL(gemm_bf16_kernel<EPI_RESID>, dim3((H / BN) * (M / BM)), dim3(kGemmThreads), 0, st, o);
                hipLaunchKernelGGL(layernorm_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, (const float*)pre.as<float>(),
                                   (const float*)L.ln1_g, (const float*)L.ln1_b, x.as<bf16>(), M, H, cfg.ln_eps, 1,
                                   (const float*)nullptr, (const bf16*)nullptr);
            }
            GemmArgs f1{};
            f1.A = X; f1.W = (const bf16*)L.w1; f1.bias = (const float*)L.b1; f1.M = M; f1.N = F; f1.K = H;
            f1.out_bf16 = ffn.as<bf16>();
            if (small) { f1.M = T; launch_skinny<EPI_GELU>(f1, st); }
            else if (big) launch256<EPI_GELU>(f1, M, st);
            else hipLaunchKernelGGL(gemm_bf16_kernel<EPI_GELU>, dim3((F / BN) * (M / BM)), dim3(kGemmThreads), 0, st, f1);
            GemmArgs f2{};
            f2.A = ffn.as<bf16>(); f2.W = (const bf16*)L.w2; f2.bias = (const float*)L.b2; f2.M = M; f2.N = H; f2.K = F;
            f2.resid = X; f2.out_f32 = pre.as<float>();
            if (small) {
                f2.M = T;
                launch_skinny<EPI_PART>(f2, st);
                hipLaunchKernelGGL(layernorm_kernel<true>, dim3((T + 3) / 4), dim3(256), 0, st, (const float*)pre.as<float>(),
                                   (const float*)L.ln2_g, (const float*)L.ln2_b, x.as<bf16>(), T, H, cfg.ln_eps, fsplit,
                                   (const float*)L.b2, X);
            } else if (big) {
                f2.out_bf16 = pre.as<bf16>();
                launch256<EPI_RESID16>(f2, M, st);
                hipLaunchKernelGGL(layernorm16_kernel, dim3((M + 4 * kLn16Rows - 1) / (4 * kLn16Rows)), dim3(256), 0, st, (const bf16*)pre.as<bf16>(),
                                   (const float*)L.ln2_g, (const float*)L.ln2_b, x.as<bf16>(), M, H, cfg.ln_eps);
            } else if (dsplit > 1) {
                f2.ksplit = dsplit;
                hipLaunchKernelGGL(gemm_bf16_kernel<EPI_PART>, dim3((H / BN) * (M / BM) * dsplit), dim3(kGemmThreads), 0, st, f2);
                hipLaunchKernelGGL(layernorm_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, (const float*)pre.as<float>(),
                                   (const float*)L.ln2_g, (const float*)L.ln2_b, x.as<bf16>(), M, H, cfg.ln_eps, dsplit,
                                   (const float*)L.b2, X);
            } else {
                hipLaunchKernelGGL(gemm_bf16_kernel<EPI_RESID>, dim3((H / BN) * (M / BM)), dim3(kGemmThreads), 0, st, f2);
                hipLaunchKernelGGL(layernorm_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, (const float*)pre.as<float>(),
                                   (const float*)L.ln2_g, (const float*)L.ln2_b, x.as<bf16>(), M, H, cfg.ln_eps, 1,
                                   (const float*)nullptr, (const bf16*)nullptr);
            }
        }
        if (mode == 0 || mode == 2) {
            hipLaunchKernelGGL(pool_kernel, dim3(nseq), dim3(64), 0, st, X, (const int*)lens.as<int>(), S, H, out_dev,
                               mode == 0 ? 1 : 0);
        } else {
            if (!w.cls_dense_w || !w.cls_out_w) { set_error("encoder was created without a classification head"); return HIPRAG_E_INVALID; }
            hipLaunchKernelGGL(rerank_head_kernel, dim3(nseq), dim3(256), 0, st, X, S, H, (const bf16*)w.cls_dense_w,
                               (const float*)w.cls_dense_b, (const bf16*)w.cls_out_w, (const float*)w.cls_out_b, out_dev);
        }
        HR_CHECK_HIP(hipGetLastError());
        double tok = (double)T;
        flops_last = tok * cfg.layers * (8.0 * H * H + 4.0 * H * F) + (double)nseq * cfg.layers * 4.0 * S * (double)S * H;
        return HIPRAG_OK;
    }
};

Registry<Encoder>& reg()
{
    static Registry<Encoder> r;
    return r;
}

}  // namespace

size_t clear_encoder_registry() { return reg().clear(); }
}  // namespace hiprag

using namespace hiprag;

extern "C" {

int32_t hipenc_create(const hipenc_config* cfg, const hipenc_weights* weights, int32_t device, uint64_t* out_handle)
{
    HR_REQUIRE(cfg && weights && out_handle, "null argument");
    HR_REQUIRE(cfg->hidden > 0 && cfg->hidden % 128 == 0 && cfg->hidden <= 2048, "hidden must be a multiple of 128, <= 2048");
    HR_REQUIRE(cfg->heads * 64 == cfg->hidden, "head_dim must be 64 (heads * 64 == hidden)");
    HR_REQUIRE(cfg->ffn > 0 && cfg->ffn % 128 == 0, "ffn must be a multiple of 128");
    HR_REQUIRE(cfg->layers > 0 && cfg->vocab > 0 && cfg->max_pos > cfg->pad_id + 1, "bad encoder config");
    HR_REQUIRE(weights->layers && weights->word_emb && weights->pos_emb && weights->type_emb && weights->emb_ln_g &&
                   weights->emb_ln_b, "null weight pointer");
    auto e = std::make_shared<Encoder>();
    e->device = device;
    e->cfg = *cfg;
    e->w = *weights;
    e->layers.assign(weights->layers, weights->layers + cfg->layers);
    if (const char* sr = std::getenv("HIPENC_SMALL_ROWS")) e->small_rows = std::atoi(sr);
    if (const char* g2 = std::getenv("HIPENC_GEMM256")) e->use256 = std::atoi(g2);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->n_cu = cus;
    }
    for (const auto& L : e->layers)
        HR_REQUIRE(L.wqkv && L.bqkv && L.wo && L.bo && L.ln1_g && L.ln1_b && L.w1 && L.b1 && L.w2 && L.b2 && L.ln2_g && L.ln2_b,
                   "null layer weight pointer");
    e->w.layers = nullptr;
    HR_CHECK_HIP(hipSetDevice(device));
    *out_handle = reg().put(e);
    return HIPRAG_OK;
}

int32_t hipenc_destroy(uint64_t h)
{
    auto e = reg().get(h);
    if (!e) { set_error("unknown encoder handle"); return HIPRAG_E_HANDLE; }
    {
        std::lock_guard<std::mutex> guard(e->mu);
        (void)hipSetDevice(e->device);
        (void)hipDeviceSynchronize();
    }
    reg().erase(h);
    return HIPRAG_OK;
}

static int32_t enc_run(uint64_t h, const int32_t* token_ids, const int32_t* seq_lens, int32_t nseq, int32_t max_len,
                       float* out_dev, int mode, void* stream)
{
    auto e = reg().get(h);
    if (!e) { set_error("unknown encoder handle"); return HIPRAG_E_HANDLE; }
    std::lock_guard<std::mutex> guard(e->mu);
    HR_CHECK_HIP(hipSetDevice(e->device));
    HR_REQUIRE(nseq >= 0 && max_len > 0, "bad batch shape");
    if (nseq == 0) return HIPRAG_OK;
    HR_REQUIRE(token_ids && seq_lens && out_dev, "null argument");
    HR_REQUIRE(max_len + e->cfg.pad_id + 1 < e->cfg.max_pos, "max_len %d exceeds the position table", max_len);
    return e->forward(token_ids, seq_lens, nseq, max_len, out_dev, mode, (hipStream_t)stream);
}

int32_t hipenc_forward(uint64_t h, const int32_t* token_ids_host, const int32_t* seq_lens_host, int32_t nseq, int32_t max_len,
                       float* out_dev, void* stream)
{
    return enc_run(h, token_ids_host, seq_lens_host, nseq, max_len, out_dev, 0, stream);
}

int32_t hipenc_score_pairs(uint64_t h, const int32_t* token_ids_host, const int32_t* seq_lens_host, int32_t nseq,
                           int32_t max_len, float* out_logits_dev, void* stream)
{
    return enc_run(h, token_ids_host, seq_lens_host, nseq, max_len, out_logits_dev, 1, stream);
}

int32_t hipenc_linear(const void* a_dev, const void* w_dev, const float* bias_dev, int32_t M, int32_t N, int32_t K,
                      int32_t epilogue, const void* resid_dev, void* out_dev, void* out_k_dev, void* out_vt_dev, int32_t S,
                      int32_t heads, int32_t impl, void* stream)
{
    HR_REQUIRE(a_dev && w_dev && bias_dev && out_dev, "null argument");
    HR_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0, "M, N multiples of 128 and K of 64");
    HR_REQUIRE(epilogue >= 0 && epilogue <= 3, "epilogue: 0 = qkv, 1 = gelu, 2 = bias + residual -> f32, 3 = bias + residual -> bf16");
    HR_REQUIRE(epilogue != 3 || impl != 1, "the bf16 residual epilogue exists in the 256-tile kernel only");
    HR_REQUIRE(impl >= 0 && impl <= 2, "impl: 0 = auto, 1 = 128 x 128 tiles, 2 = 256 x 256 persistent tiles");
    GemmArgs g{};
    g.A = (const bf16*)a_dev; g.W = (const bf16*)w_dev; g.bias = bias_dev; g.M = M; g.N = N; g.K = K;
    if (epilogue == EPI_QKV) {
        HR_REQUIRE(out_k_dev && out_vt_dev && S > 0 && heads > 0 && N == 3 * heads * 64 && M % S == 0 && S % 64 == 0,
                   "qkv epilogue: N = 3 * heads * 64, M a multiple of S, S a multiple of 64");
        g.q = (bf16*)out_dev; g.k = (bf16*)out_k_dev; g.vt = (bf16*)out_vt_dev; g.S = S; g.heads = heads; g.H = heads * 64;
    } else if (epilogue == EPI_GELU) {
        g.out_bf16 = (bf16*)out_dev;
    } else {
        HR_REQUIRE(resid_dev, "null residual");
        g.resid = (const bf16*)resid_dev; g.out_f32 = (float*)out_dev; g.out_bf16 = (bf16*)out_dev;
    }
    int dev = 0, cus = 256;
    HR_CHECK_HIP(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool can256 = M % G2_T == 0 && N % G2_T == 0 && K % (2 * G2_BK) == 0 && (epilogue != EPI_QKV || g.H % G2_T == 0);
    HR_REQUIRE(impl != 2 || can256, "the 256-tile kernel needs M, N multiples of 256 and K of 128");
    hipStream_t st = (hipStream_t)stream;
    HR_REQUIRE(epilogue != 3 || can256, "the bf16 residual epilogue needs M, N multiples of 256 and K of 128");
    if (impl == 2 || epilogue == 3 || (impl == 0 && can256 && (M / G2_T) * (N / G2_T) >= cus)) {
        const int nbm = M / G2_T, nbn = N / G2_T, grid = std::min(nbm * nbn, cus);
        if (epilogue == EPI_QKV) hipLaunchKernelGGL(gemm256_kernel<EPI_QKV>, dim3(grid), dim3(G2_THREADS), 0, st, g, nbm, nbn);
        else if (epilogue == EPI_GELU) hipLaunchKernelGGL(gemm256_kernel<EPI_GELU>, dim3(grid), dim3(G2_THREADS), 0, st, g, nbm, nbn);
        else if (epilogue == 3) hipLaunchKernelGGL(gemm256_kernel<EPI_RESID16>, dim3(grid), dim3(G2_THREADS), 0, st, g, nbm, nbn);
        else hipLaunchKernelGGL(gemm256_kernel<EPI_RESID>, dim3(grid), dim3(G2_THREADS), 0, st, g, nbm, nbn);
    } else {
        const dim3 grid((N / BN) * (M / BM));
        if (epilogue == EPI_QKV) hipLaunchKernelGGL(gemm_bf16_kernel<EPI_QKV>, grid, dim3(kGemmThreads), 0, st, g);
        else if (epilogue == EPI_GELU) hipLaunchKernelGGL(gemm_bf16_kernel<EPI_GELU>, grid, dim3(kGemmThreads), 0, st, g);
        else hipLaunchKernelGGL(gemm_bf16_kernel<EPI_RESID>, grid, dim3(kGemmThreads), 0, st, g);
    }
    HR_CHECK_HIP(hipGetLastError());
    return HIPRAG_OK;
}

int32_t hipenc_last_flops(uint64_t h, double* out_flops)
{
    auto e = reg().get(h);
    if (!e) { set_error("unknown encoder handle"); return HIPRAG_E_HANDLE; }
    HR_REQUIRE(out_flops, "null out");
    *out_flops = e->flops_last;
    return HIPRAG_OK;
}

}  // extern "C"
