// dense_index.hip -- flat (brute-force) vector index for MI355X / gfx950.
//
// Replaces faiss.IndexFlatL2 / IndexFlatIP as used by the reference:
//   build   rag/storage/faiss_index.py:121-124   (np.array(float32) -> IndexFlatL2(d).add)
//   search  rag/storage/faiss_index.py:81-83     (index.search(float32[1,d], k))
//
// Data layout in HBM (ours, not FAISS's): rows are stored in 32-row BLOCKS, each block as P = d_pad/8 PIECES of
// 1 KiB.  Piece p of block B holds, for lane = h*32 + r (h in {0,1}, r in 0..31), the four floats
// X[32B + r][8p + 4h + 0..3].  That is exactly the A-operand fragment of four consecutive
// v_mfma_f32_32x32x2_f32 instructions, so the scan reads the index with fully coalesced 16-B-per-lane loads
// straight into MFMA operand registers: no LDS staging, no transposes, and one wave's work is one contiguous
// 1 KiB * P run per block.  d_pad is d rounded up to 128 floats (zero filled).
//
// One search pass answers up to 32 queries (the N dimension of the 32x32 MFMA tile):
//   K1  scan_kernel     every wave streams a contiguous range of blocks; queries sit in LDS in B-fragment order;
//                       exact-fp32 MFMA (a k-ordered fmaf chain) gives a 32 rows x 32 queries score tile; the
//                       epilogue keeps only max-over-16-rows ("group maxima", one per lane and block) staged in LDS and
//                       written as whole lines -> gmax[query][group]  (N/16 floats/query)
//   K2a select_kernel   per (query, 4096-group slice): exact top-(K'+1) of the group maxima
//   K2b finish_kernel   per query: merge slice winners -> K' best groups, re-score their 16*K' rows in fp64 from the
//                       fp32 data, exact top-k under (score, id); then a CERTIFICATE: every row outside the K' groups
//                       has fp32 score <= m (the (K'+1)-th group maximum), hence exact score <= m + eps, eps a
//                       worst-case bound of the fp32 chain error.  If the k-th exact score is not > m + eps the
//                       query is flagged and
//   K2c/K2d exhaustive  (launched always, exit at once unless flagged) re-score EVERY row in fp64 and select.
// So results are exact for any input (ties, duplicates, zero vectors), and the common case reads the index once.
//
// Bound: HBM.  Algorithmic bytes per pass = nblocks * P * 1024 (+ norms in L2 mode) -- hipidx_stats.bytes_per_pass.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "topk_device.h"

namespace hiprag {
namespace {

constexpr int kRowsPerBlock = 32;
constexpr int kPieceFloats = 256;
constexpr int kPieceVec4 = 64;
// Position (in float4 units) inside a 1 KiB piece of the four k-values [8p + 4h, 8p + 4h + 4) of row r of the block:
// quad-major, so that the 4 rows x 2 halves of a row QUAD are one contiguous 128-byte line of every piece -- the unit the
// fp64 re-score reads (fin_rescore: 128 whole lines per quad instead of 256 half lines 512 bytes apart).  The fp32 scans
// read whole pieces and only permute which lane takes which 16 bytes.
__host__ __device__ __forceinline__ int piece_slot(int h, int r) { return ((r >> 2) << 3) | (h << 2) | (r & 3); }
// a PASS = the queries that share one read of the index: 32 (full hi/lo or fp32 operands) or 64 (hi-only query tiles)
constexpr int kMaxQ = 1024;        // most queries per LAUNCH (16 passes of 64): see DenseIndex::update_launch_q
constexpr int kMaxScanWaves = 12;  // stamp slots per scan workgroup
constexpr int kMaxDPad = 1024;    // d_pad limit (the 128 KiB query tile of the scan)
constexpr int kSelChunk = kTile;   // entries per select tile (topk_device.h)
constexpr int kSelThreads = 256;
constexpr int kExRows = 1024;      // rows per workgroup in the exhaustive path (16 KiB of LDS: it must fit BESIDE a
                                   // resident scan workgroup, or its launch would serialise behind the next scan)
constexpr int kMaxK = 1000;
constexpr int kSlackGroups = 6;
// 64-query tiles: eps ~ |x| |q - bf16(q)| ~ 1.2e-3 |x||q| is ~25x wider, and the number of rows within eps of the k-th score
// grows with k (1M unit vectors, d = 1024: ~2 at k = 10, ~8 at k = 50), so the slack does too
__host__ __device__ constexpr int slack_groups64(int k) { return k > 16 ? k : 16; }
constexpr int kMaxK64 = 57;   // deepest k on the 64-query tiles: K' is capped at 63 (wave lists), round B takes what that misses

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------
// build: row-major -> blocked layout, squared norms
// ------------------------------------------------------------------------------------------------------
__global__ void retile_kernel(const float* __restrict__ src, int64_t row0, int64_t n, int d, int P,
                              float4* __restrict__ xb)
{
    const int64_t blk0 = row0 / kRowsPerBlock;
    const int64_t nblk = (row0 + n - 1) / kRowsPerBlock - blk0 + 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblk * P * kPieceVec4) return;
    const int lane = (int)(t & 63);
    const int64_t pp = t >> 6;
    const int p = (int)(pp % P);
    const int64_t blk = blk0 + pp / P;
    const int r = lane & 31, h = lane >> 5;
    const int64_t row = blk * kRowsPerBlock + r;
    if (row < row0 || row >= row0 + n) return;
    const int col = 8 * p + 4 * h;
    const float* s = src + (row - row0) * (int64_t)d + col;
    float4 v;
    v.x = col + 0 < d ? s[0] : 0.f;
    v.y = col + 1 < d ? s[1] : 0.f;
    v.z = col + 2 < d ? s[2] : 0.f;
    v.w = col + 3 < d ? s[3] : 0.f;
    xb[(blk * P + p) * kPieceVec4 + piece_slot(h, r)] = v;
}

__global__ void untile_kernel(const float4* __restrict__ xb, int64_t row0, int64_t n, int d, int P,
                              float* __restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_row = (int64_t)P * 2;
    if (t >= n * per_row) return;
    const int64_t row = row0 + t / per_row;
    const int ph = (int)(t % per_row);
    const int p = ph >> 1, h = ph & 1;
    const int64_t blk = row / kRowsPerBlock;
    const int r = (int)(row % kRowsPerBlock);
    float4 v = xb[(blk * P + p) * kPieceVec4 + piece_slot(h, r)];
    const int col = 8 * p + 4 * h;
    float* o = dst + (row - row0) * (int64_t)d + col;
    if (col + 0 < d) o[0] = v.x;
    if (col + 1 < d) o[1] = v.y;
    if (col + 2 < d) o[2] = v.z;
    if (col + 3 < d) o[3] = v.w;
}

// Row statistics of one add: |x|^2 (fp64 sum -> float, stored per row; its running maximum rounded UP) and
// |x - bf16(x)|^2 (running maximum, rounded up: the certificate of the bf16 scan bounds |<x - x^, q^>| by |x - x^| |q^|).
// One wave per row at a time, rows grid-strided, each row read ONCE for both sums; a wave keeps its maxima in registers
// and issues two atomics when it is done (one atomicMax per ROW on a single address serialised 125 k of them per
// 125 k-row add: 1.4 ms for a kernel that reads 512 MB).
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ src, int64_t row0, int64_t n, int d,
                                                        float* __restrict__ norms, unsigned* __restrict__ max_norm2_bits,
                                                        unsigned* __restrict__ max_dx2_bits)
{
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    float mx_n = 0.f, mx_d = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < n; row += nw) {
        const float* s = src + row * (int64_t)d;
        double acc = 0.0, dcc = 0.0;
#pragma unroll 8
        for (int c = lane; c < d; c += 64) {
            const float f = s[c];
            const double v = (double)f, dv = v - (double)(float)(__bf16)f;
            acc += v * v;
            dcc += dv * dv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { acc += __shfl_xor(acc, off); dcc += __shfl_xor(dcc, off); }
        float f = (float)acc;
        if (lane == 0) norms[row0 + row] = f;
        // round up so the stored maxima are upper bounds of the exact values
        if ((double)f < acc) f = nextafterf(f, INFINITY);
        float g = (float)dcc;
        if ((double)g < dcc) g = nextafterf(g, INFINITY);
        mx_n = fmaxf(mx_n, f);
        mx_d = fmaxf(mx_d, g);
    }
    if (lane == 0) {
        atomicMax(max_norm2_bits, __float_as_uint(mx_n));
        atomicMax(max_dx2_bits, __float_as_uint(mx_d));
    }
}

// bf16 FILTER copy of the rows (scan operand of the default mode): block of 32 rows = P/2 pieces of 1 KiB, piece p holds
// for lane l = h*32 + r the eight values bf16(X[32B + r][16p + 8h + 0..7]) -- the A fragment of one
// v_mfma_f32_32x32x16_bf16, 16 bytes per lane, so the scan streams 2 bytes per element straight into MFMA registers.
// The fp32 blocked copy above stays the source of every exact (fp64) re-score.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__global__ void retile_bf16_kernel(const float* __restrict__ src, int64_t row0, int64_t n, int d, int P2,
                                   bf16x8_t* __restrict__ xh)
{
    const int64_t blk0 = row0 / kRowsPerBlock;
    const int64_t nblk = (row0 + n - 1) / kRowsPerBlock - blk0 + 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblk * P2 * 64) return;
    const int lane = (int)(t & 63);
    const int64_t pp = t >> 6;
    const int p = (int)(pp % P2);
    const int64_t blk = blk0 + pp / P2;
    const int r = lane & 31, h = lane >> 5;
    const int64_t row = blk * kRowsPerBlock + r;
    if (row < row0 || row >= row0 + n) return;
    const int col = 16 * p + 8 * h;
    const float* s = src + (row - row0) * (int64_t)d + col;
    bf16x8_t v;
    if ((d & 3) == 0 && col + 8 <= d) {   // the usual case: two 16-B loads (eight predicated scalar loads cost a round trip each)
        const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
        v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
        v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)(col + j < d ? s[j] : 0.f);
    }
    xh[(blk * P2 + p) * 64 + lane] = v;
}

// ------------------------------------------------------------------------------------------------------
// K1: the scan
// ------------------------------------------------------------------------------------------------------
// Query fragments: Qf[p][lane = h*32 + b] = Q[b][8p + 4h + 0..3] (zero for b >= nq or columns >= d) -- the LDS image
// of the scan's B operand, written once per pass so every workgroup's prologue is one coalesced 16-B-per-lane copy.
__global__ __launch_bounds__(256) void qprep_kernel(const float* __restrict__ q, int nq, int d, int P, float4* __restrict__ qf)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * kPieceVec4) return;
    const int p = idx >> 6, l = idx & 63;
    const int b = l & 31, h = l >> 5;
    const int col = 8 * p + 4 * h;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < nq) {
        const float* s = q + (int64_t)b * d + col;
        if (col + 0 < d) v.x = s[0];
        if (col + 1 < d) v.y = s[1];
        if (col + 2 < d) v.z = s[2];
        if (col + 3 < d) v.w = s[3];
    }
    qf[idx] = v;
}

struct ScanArgs {
    const float4* xb;     // blocked index
    const void* xh;       // bf16 filter copy (scan_bf16_kernel)
    const float4* qf;     // [P*64] query fragments (qprep_kernel; fp32 path)
    const float* q;       // [nq, d] row-major queries (split path builds its fragments in the scan prologue)
    const float* norms;   // [rows] squared norms (L2 only)
    float* gmax;          // [kMaxQ, gstride] group maxima: one per (block, lane half) = 16 rows, chunk-permuted
    float* gmax2;         // same layout: the best quad maximum of the group's OTHER three quads (see block_lane_top2)
    int64_t gstride;
    int64_t nblocks;
    int64_t ntotal;
    int nq, d, P;
    unsigned long long* stamps;  // timing only (else null): [waves][2] wall-clock ticks at wave entry / exit
};

// ---- group maxima -----------------------------------------------------------------------------------------------
// A GROUP is the 16 rows one lane holds of a block's score tile: rows 32*blk + 8g + 4h + j (g, j in 0..3) for lane half
// h.  The scan keeps only max-over-group (N/16 floats per query, 8 MB per pass at 1M x 32 queries): written per block
// as 16 B per lane the same data costs ~15 % of the scan (32 MB of partial-line writes turning the HBM bus around), as
// one float per lane it is nearly free.  Each wave parks its lane maxima in LDS ([16 blocks][64 lanes]) and every 16
// blocks each lane writes its 16 values as one 64-B run, so a query's (h = 0, h = 1) pair fills a whole 128-B line:
//   slot of (block cb + j of a chunk of cnt blocks starting at cb, half h) = 2*cb + h*cnt + j      (see group_decode)
constexpr int kChunk = 16;        // blocks per flush of the fp32 scan and the default of the split scans (template CH)

__host__ __device__ __forceinline__ int64_t scan_blocks_per_wave(int64_t nblocks, int64_t nwaves)
{
    int64_t bpw = (nblocks + nwaves - 1) / nwaves;
    bpw = (bpw + 1) & ~(int64_t)1;   // even, so every chunk base is even and 16-value runs are 16-byte aligned
    // whole 16-block chunks (= whole, aligned 128-byte lines of output per query and flush) where that idles few waves
    const int64_t r16 = (bpw + 15) & ~(int64_t)15;
    if (bpw > 16 && r16 * 100 <= bpw * 106) bpw = r16;
    return bpw;
}

// slot -> (block, lane half); inverse of the permutation above.  bpw = scan_blocks_per_wave(nblocks, waves of the scan).
__device__ __forceinline__ void group_decode(int64_t slot, int64_t bpw, int64_t nblocks, int chunk, int64_t& blk, int& h)
{
    const int64_t bq = slot >> 1;                      // lies inside the same chunk as the group's block
    const int64_t b0 = (bq / bpw) * bpw;
    const int64_t cb = b0 + ((bq - b0) / chunk) * chunk;
    int64_t end = b0 + bpw;
    if (end > nblocks) end = nblocks;
    const int64_t cnt = end - cb < chunk ? end - cb : chunk;
    const int64_t off = slot - 2 * cb;
    h = off >= cnt ? 1 : 0;
    blk = cb + off - (h ? cnt : 0);
}

// A group's 16 rows are four QUADS of 4 consecutive rows (quad g = rows 8g + 4h + 0..3: 64 contiguous bytes of every
// piece, the unit the fp64 re-score reads).  The scan keeps per group
//   first  = the largest quad maximum, with the quad number g in its two low mantissa bits, and
//   second = the largest quad maximum among the other three quads (tagged the same way, the tag unused),
// so the finish re-scores ONE quad per selected group (16 KiB instead of 64 KiB) and `second` bounds the 12 rows it did
// not read; a group whose `second` can still reach the top k gets its other quads re-scored as well (fin_final_kernel).
// Replacing two mantissa bits moves a value by < 2^-21 |v|: the certificate's eps carries that term (kTagSlack).
// L2: scores are 2<x,q> - |x|^2.  Padded tail rows get -FLT_MAX (finite, so the tag cannot turn it into a NaN).
__device__ __forceinline__ float tag_quad(float v, unsigned g) { return __uint_as_float((__float_as_uint(v) & ~3u) | g); }

template <int METRIC>
__device__ __forceinline__ float block_lane_top2(const f32x16& acc, const f32x4 (&nrm)[4], int64_t blk, int h, const ScanArgs& a,
                                                 float& second)
{
    float sc[16];
    if (METRIC == HIPRAG_METRIC_L2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sc[4 * g + 0] = 2.f * acc[4 * g + 0] - nrm[g][0];
            sc[4 * g + 1] = 2.f * acc[4 * g + 1] - nrm[g][1];
            sc[4 * g + 2] = 2.f * acc[4 * g + 2] - nrm[g][2];
            sc[4 * g + 3] = 2.f * acc[4 * g + 3] - nrm[g][3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = acc[i];
    }
    if ((blk + 1) * kRowsPerBlock > a.ntotal) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = blk * kRowsPerBlock + 8 * (i >> 2) + 4 * h + (i & 3);
            if (row >= a.ntotal) sc[i] = -FLT_MAX;
        }
    }
    float qm[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        qm[g] = tag_quad(fmaxf(fmaxf(sc[4 * g], sc[4 * g + 1]), fmaxf(sc[4 * g + 2], sc[4 * g + 3])), (unsigned)g);
    float m1 = fmaxf(qm[0], qm[1]), m2 = fminf(qm[0], qm[1]);
    m2 = __builtin_amdgcn_fmed3f(m1, m2, qm[2]);   // new runner-up = median(best, runner-up, newcomer)
    m1 = fmaxf(m1, qm[2]);
    m2 = __builtin_amdgcn_fmed3f(m1, m2, qm[3]);
    m1 = fmaxf(m1, qm[3]);
    second = m2;
    return m1;
}

// cnt <= kChunk newest values of a shift chain (block cb + t of the run sits in mh[cnt - 1 - t]) as one run at dst:
// 16-byte stores when cnt is a multiple of four (small shards: 4 blocks per wave and pass), else value by value.
// cnt is wave-uniform; dst is 16-byte aligned then (even bpw, chunks of 8 or 16 blocks, h * cnt a multiple of 4).
template <int kChunk>
__device__ __forceinline__ void store_run(const float (&mh)[kChunk], int cnt, float* __restrict__ dst)
{
    if ((cnt & 3) == 0) {
#pragma unroll
        for (int g = 1; g <= kChunk / 4; ++g) {
            if (cnt == 4 * g) {
#pragma unroll
                for (int v = 0; v < g; ++v)
                    reinterpret_cast<float4*>(dst)[v] = make_float4(mh[4 * g - 1 - 4 * v], mh[4 * g - 2 - 4 * v], mh[4 * g - 3 - 4 * v],
                                                                    mh[4 * g - 4 - 4 * v]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < kChunk; ++t)
            if (t < cnt) dst[cnt - 1 - t] = mh[t];
    }
}

// Park one block's lane maximum in a 16-register shift chain (mh[0] = newest); on the last block of a chunk (or of the
// wave's range) write the chunk out as one run per lane.  Registers, not LDS: the query tile already takes 128 KiB.
// Plain stores: with the ring loads hidden in asm they are the only VMEM ops hipcc sees here, so they never
// make it drain the queue; in the hand-counted vmcnt they are extra YOUNGER ops.
template <int kChunk>
__device__ __forceinline__ void park_and_flush(float (&mh)[kChunk], float m, int64_t blk, int64_t b0, int64_t b1, int lane,
                                               float* __restrict__ gm, int64_t gstride, int qoff)
{
#pragma unroll
    for (int t = kChunk - 1; t > 0; --t) mh[t] = mh[t - 1];
    mh[0] = m;
    const int j = (int)((blk - b0) % kChunk);
    if (j != kChunk - 1 && blk != b1 - 1) return;
    const int cnt = j + 1;
    const int64_t cb = blk - j;
    const int h = lane >> 5, qb = (lane & 31) + qoff;
    float* dst = gm + (int64_t)qb * gstride + 2 * cb + (int64_t)h * cnt;
    if (cnt == kChunk) {
#pragma unroll
        for (int v = 0; v < kChunk / 4; ++v)   // block cb + t sits in mh[kChunk - 1 - t]
            reinterpret_cast<float4*>(dst)[v] = make_float4(mh[kChunk - 1 - 4 * v], mh[kChunk - 2 - 4 * v], mh[kChunk - 3 - 4 * v],
                                                            mh[kChunk - 4 - 4 * v]);
    } else {
        store_run(mh, cnt, dst);
    }
}

// The bf16 scan's forms of the flush.  park() keeps the shift chain; flush_full() writes a complete chunk (kChunk / 4
// 16-byte stores per lane, back to back), flush_partial() the short last chunk of a range.
// What the output costs (1M x 1024, 512 queries per launch: 256 MB of group maxima beside 16.4 GB of rows; scan alone
// 2.40 ms with the stores compiled out): 8-block chunks, i.e. a 64-byte half line per query and flush, +0.17 to +0.5 ms
// depending on where the allocation landed; 16-block chunks = whole 128-byte lines (rows are line aligned: gmax_stride),
// +0.03 to +0.3 ms.  Whole lines must reach L2 TOGETHER: the same 16-block chunk written four stores at a time over the
// next twelve blocks (so that no burst of 16 stores sits in the wave's vmcnt) was 10 % slower than the burst, `nt`
// stores 25 % slower, a [chunk][query] layout with 8 KiB contiguous per flush slower than the per-query rows, and
// staging through LDS so that each store instruction writes eight whole lines changed nothing.
// The stores are plain C++ stores: hipcc sees no other vector-memory operation in the loop, so they never make it drain
// the queue, and it keeps the wait state a 16-byte store needs before its data registers are written again (an inline-asm
// store followed by the next flush's register moves corrupted values).  In the ring's counted waits they are extra
// operations in flight: vmcnt(RING - 1) then waits for more than it needs, never for less (loads return in order among
// themselves; allowing for the stores with a larger count is NOT safe: acknowledgements of stores overtake older loads).
template <int kChunk>
__device__ __forceinline__ void park(float (&mh)[kChunk], float m)
{
#pragma unroll
    for (int t = kChunk - 1; t > 0; --t) mh[t] = mh[t - 1];
    mh[0] = m;
}

template <int kChunk>
__device__ __forceinline__ void flush_full(const float (&mh)[kChunk], int lane, float* __restrict__ gm, int64_t gstride, int qoff,
                                           int64_t cb)
{
    const int h = lane >> 5, qb = (lane & 31) + qoff;
    float* dst = gm + (int64_t)qb * gstride + 2 * cb + (int64_t)h * kChunk;
#pragma unroll
    for (int v = 0; v < kChunk / 4; ++v) {   // block cb + t sits in mh[kChunk - 1 - t]
        reinterpret_cast<float4*>(dst)[v] = make_float4(mh[kChunk - 1 - 4 * v], mh[kChunk - 2 - 4 * v], mh[kChunk - 3 - 4 * v],
                                                        mh[kChunk - 4 - 4 * v]);
    }
}

template <int kChunk>
__device__ __forceinline__ void flush_partial(const float (&mh)[kChunk], int cnt, int lane, float* __restrict__ gm, int64_t gstride,
                                              int qoff, int64_t cb)
{
    const int h = lane >> 5, qb = (lane & 31) + qoff;
    store_run(mh, cnt, gm + (int64_t)qb * gstride + 2 * cb + (int64_t)h * cnt);
}

template <int METRIC, int NWAVES, int RING = 16>
__global__ __launch_bounds__(NWAVES * 64) void scan_kernel(ScanArgs a)
{
    extern __shared__ float4 qs[];  // [P][64] query fragments: lane = h*32 + b holds Q[b][8p + 4h + 0..3]
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> block ranges live in SGPRs
    const int P = a.P;

    // contiguous block range of this wave: equal shares of ceil(nblocks / W) blocks, so every active wave ends at the
    // same time (a ragged last round would leave a quarter of the waves streaming alone at latency-bound rates)
    const int64_t gw = (int64_t)blockIdx.x * NWAVES + wave;
    const int64_t W = (int64_t)gridDim.x * NWAVES;
    const int64_t bpw = scan_blocks_per_wave(a.nblocks, W);
    const int64_t b0 = min(gw * bpw, a.nblocks);
    const int64_t b1 = min(b0 + bpw, a.nblocks);
    const int S = (int)((b1 - b0) * P);  // pieces in this wave's stream
    if (a.stamps && lane == 0) a.stamps[2 * gw] = wall_clock64();
    const float4* base = a.xb + b0 * P * kPieceVec4;  // wave-uniform; lanes add 16 B each through the VGPR offset
    float mh[kChunk], ms[kChunk];  // lane first / second values of the current chunk (shift chains)
#pragma unroll
    for (int t = 0; t < kChunk; ++t) { mh[t] = 0.f; ms[t] = 0.f; }
    const unsigned lane16 = (unsigned)piece_slot(lane >> 5, lane & 31) * 16u;   // this lane's 16 bytes of a piece (A fragment of row lane & 31)
    const int h = lane >> 5;

    // The X stream is driven by hand: loads are inline asm (invisible to hipcc's waitcnt pass, which otherwise drains
    // the queue with vmcnt(0) at the loop back-edge) and every use is fenced by a counted s_waitcnt that takes the
    // ring slot as an in/out operand, so no consumer can be scheduled above its wait.  vmcnt(RING-1) before slot i is
    // exact when only the ring is in flight and merely conservative when the epilogue's store / norm loads are queued too.
    // The ring is armed BEFORE the query tile is staged, so HBM is streaming while the prologue runs.
    f32x4 ring[RING];
    if (S > 0) {
#pragma unroll
        for (int i = 0; i < RING; ++i) {
            const unsigned voff = lane16 + (unsigned)min(i, S - 1) * 1024u;
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(ring[i]) : "v"(voff), "s"(base) : "memory");
        }
    }

    for (int idx = tid; idx < P * kPieceVec4; idx += NT) qs[idx] = a.qf[idx];
    __syncthreads();
    if (S <= 0) {
        if (a.stamps && lane == 0) a.stamps[2 * gw + 1] = wall_clock64();
        return;
    }

    int s = 0;
    float4 bnext = qs[lane];
    for (int64_t blk = b0; blk < b1; ++blk) {
        f32x4 nrm[4];
        if (METRIC == HIPRAG_METRIC_L2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* np = a.norms + blk * kRowsPerBlock + 8 * g + 4 * h;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nrm[g]) : "v"(np) : "memory");
            }
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int pp = 0; pp < P; pp += RING) {
#pragma unroll
            for (int i = 0; i < RING; ++i) {
                // one step = one 1 KiB piece: 4 MFMAs on the piece loaded RING steps ago, then re-arm its ring slot
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ring[i]) : "n"(RING - 1) : "memory");
                const f32x4 av = ring[i];
                const float4 bv = bnext;
                int nx = pp + i + 1;
                nx = nx == P ? 0 : nx;
                bnext = qs[nx * kPieceVec4 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv.w, acc, 0, 0, 0);
                const unsigned voff = lane16 + (unsigned)min(s + RING + i, S - 1) * 1024u;
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(ring[i]) : "v"(voff), "s"(base) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            s += RING;
        }
        if (METRIC == HIPRAG_METRIC_L2)  // the 4 norm loads were issued before this block's P >= RING ring re-arms
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(nrm[0]), "+v"(nrm[1]), "+v"(nrm[2]), "+v"(nrm[3]) : "n"(RING) : "memory");
        float sec;
        const float fst = block_lane_top2<METRIC>(acc, nrm, blk, h, a, sec);
        park_and_flush(mh, fst, blk, b0, b1, lane, a.gmax, a.gstride, 0);
        park_and_flush(ms, sec, blk, b0, b1, lane, a.gmax2, a.gstride, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's clamped re-arms are still in flight
    if (a.stamps && lane == 0) a.stamps[2 * gw + 1] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------------
// K1 (bf16 hi/lo split operands).  The fp32 MFMA above keeps the matrix pipe ~75 % busy at HBM rate, so the scan is
// co-limited.  Here every fp32 value v is split on the fly into hi = bf16(v), lo = bf16(v - hi) and a pair of pieces
// (16 k-values) costs three v_mfma_f32_32x32x16_bf16 (hi*hi into one accumulator, hi*lo and lo*hi into a second)
// instead of eight fp32 MFMAs: ~5x less matrix time for ~12 VALU per piece, which leaves HBM as the only limit.
// The dropped lo*lo term and the split residues are bounded by 3.02 * 2^-18 |x_i q_i| per element; the certificate's
// eps carries that term (finish kernels, `split` flag), so results stay exact: rows are still re-scored in fp64.
// Query tile in LDS: for piece pair pp and lane (h, b): 8 bf16 hi parts then (second half of the tile) 8 bf16 lo parts
// of Q[b][8(2pp) + 4h + 0..3], Q[b][8(2pp+1) + 4h + 0..3] -- the same k order the A fragment gets from two pieces.
// ------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo)
{
    const bf16x2 h = __builtin_convertvector(f32x2{v0, v1}, bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
    hi = __builtin_bit_cast(unsigned, h);
    const float r0 = v0 - __uint_as_float(hi << 16);
    const float r1 = v1 - __uint_as_float(hi & 0xFFFF0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void qprep_split_kernel(const float* __restrict__ q, int nq, int d, int P,
                                                         u32x4* __restrict__ qf)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;  // one (pair, lane)
    const int npairs = P / 2;
    if (idx >= npairs * 64) return;
    const int pp = idx >> 6, l = idx & 63;
    const int b = l & 31, h = l >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = 8 * (2 * pp + (j >> 2)) + 4 * h + (j & 3);
        v[j] = (b < nq && col < d) ? q[(int64_t)b * d + col] : 0.f;
    }
    u32x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hh, ll;
        split_pair(v[2 * j], v[2 * j + 1], hh, ll);
        hi[j] = hh;
        lo[j] = ll;
    }
    qf[idx] = hi;
    qf[npairs * 64 + idx] = lo;
}

template <int METRIC, int NWAVES, int RING = 16, int QT = 1, int CH = kChunk, bool MULTI = true>
__global__ __launch_bounds__(NWAVES * 64) void scan_split_kernel(ScanArgs a)
{
    extern __shared__ float4 qs[];  // [P/2][64] hi fragments, then [P/2][64] lo fragments (16 B each)
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = a.P;
    const int npairs = P / 2;
    // The scan is statically partitioned, so a CU that also hosts a tail workgroup of an earlier pass becomes the
    // straggler of the whole launch: let the scan's waves win issue arbitration against co-resident tail waves.
    __builtin_amdgcn_s_setprio(3);

    const int64_t gw = (int64_t)blockIdx.x * NWAVES + wave;
    const int64_t W = (int64_t)gridDim.x * NWAVES;
    const int64_t bpw = scan_blocks_per_wave(a.nblocks, W);
    const int64_t b0 = min(gw * bpw, a.nblocks);
    const int64_t b1 = min(b0 + bpw, a.nblocks);
    const int S = (int)((b1 - b0) * P);
    const float4* base = a.xb + b0 * P * kPieceVec4;
    if (a.stamps && lane == 0) a.stamps[2 * gw] = wall_clock64();
    // lane first / second values of the current chunk (shift chains); index = query tile
    float mh[QT][CH], ms[QT][CH];
#pragma unroll
    for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int u = 0; u < QT; ++u) { mh[u][t] = 0.f; ms[u][t] = 0.f; }
    const unsigned lane16 = (unsigned)piece_slot(lane >> 5, lane & 31) * 16u;   // this lane's 16 bytes of a piece (A fragment of row lane & 31)
    const int h = lane >> 5;

    f32x4 ring[RING];
    if (S > 0) {
#pragma unroll
        for (int i = 0; i < RING; ++i) {
            const unsigned voff = lane16 + (unsigned)min(i, S - 1) * 1024u;
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(ring[i]) : "v"(voff), "s"(base) : "memory");
        }
    }
    // One launch runs ceil(nq / QPP) PASSES back to back: every pass stages its own query tile and streams the wave's
    // block range again.  The piece stream is cyclic -- the last re-arms of pass p already fetch the first pieces of pass
    // p + 1 -- so HBM keeps streaming across the pass boundary, and there is no kernel boundary (a dependent launch
    // costs 35-45 us of idle GPU, a third of a pass at 8-way shard sizes).
    constexpr int QPP = 32 * QT;
    // MULTI = false is the single-pass form (nq <= QPP) that latency-bound callers -- one query at a time -- get: no
    // pass loop, and its own kernel name in profiles
    const int npass = MULTI ? (a.nq + QPP - 1) / QPP : 1;
    const bf16x8* qhi = reinterpret_cast<const bf16x8*>(qs);
    const bf16x8* qlo = qhi + npairs * 64;
    for (int pass = 0; pass < npass; ++pass) {
    const int qbase = pass * QPP;
    const bool wrap = pass + 1 < npass;
    if (pass) __syncthreads();  // every wave is done with the previous tile
    // Query tile: every workgroup splits the (L2-resident) row-major queries into hi/lo bf16 fragments itself -- 8 units
    // of (pair, lane) per thread, two 16-byte reads each -- instead of a separate preparation launch.  The ring above
    // is already streaming while this runs.  Waves without blocks (tiny indexes) still help staging.
    {
        u32x4* qhi_w = reinterpret_cast<u32x4*>(qs);
        u32x4* qlo_w = qhi_w + npairs * 64;
        const bool vec_ok = (a.d & 3) == 0;
        int tid_p = tid;
        asm volatile("" : "+v"(tid_p));   // (see scan_bf16_kernel: keeps per-thread address arithmetic out of the pass loop's live set)
        for (int idx = tid_p; idx < npairs * 64; idx += NT) {
            const int pp = idx >> 6, l = idx & 63;
            const int hh = l >> 5;
            unsigned hA[4], lA[4], hB[4];
#pragma unroll
            for (int tile = 0; tile < QT; ++tile) {
            const int b = qbase + (l & 31) + 32 * tile;
            float v[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int col = 8 * (2 * pp + half) + 4 * hh;
                if (b < a.nq && vec_ok && col + 3 < a.d) {
                    const float4 t = *reinterpret_cast<const float4*>(a.q + (int64_t)b * a.d + col);
                    v[4 * half + 0] = t.x; v[4 * half + 1] = t.y; v[4 * half + 2] = t.z; v[4 * half + 3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[4 * half + j] = (b < a.nq && col + j < a.d) ? a.q[(int64_t)b * a.d + col + j] : 0.f;
                }
            }
            unsigned h0, h1, h2, h3, l0, l1, l2, l3;
            split_pair(v[0], v[1], h0, l0);
            split_pair(v[2], v[3], h1, l1);
            split_pair(v[4], v[5], h2, l2);
            split_pair(v[6], v[7], h3, l3);
            if (tile == 0) { hA[0] = h0; hA[1] = h1; hA[2] = h2; hA[3] = h3; lA[0] = l0; lA[1] = l1; lA[2] = l2; lA[3] = l3; }
            else { hB[0] = h0; hB[1] = h1; hB[2] = h2; hB[3] = h3; }
            }
            qhi_w[idx] = u32x4{hA[0], hA[1], hA[2], hA[3]};
            // second half of the tile: lo parts of queries 0..31 (QT == 1) or hi parts of queries 32..63 (QT == 2)
            qlo_w[idx] = QT == 1 ? u32x4{lA[0], lA[1], lA[2], lA[3]} : u32x4{hB[0], hB[1], hB[2], hB[3]};
        }
    }
    __syncthreads();
    if (S <= 0) continue;

    int s = 0;
    bf16x8 bh_next = qhi[lane], bl_next = qlo[lane];
    for (int64_t blk = b0; blk < b1; ++blk) {
        f32x4 nrm[4];
        if (METRIC == HIPRAG_METRIC_L2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* np = a.norms + blk * kRowsPerBlock + 8 * g + 4 * h;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nrm[g]) : "v"(np) : "memory");
            }
        }
        f32x16 acc_hi, acc_lo;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc_hi[i] = 0.f; acc_lo[i] = 0.f; }
        for (int pp = 0; pp < npairs; pp += RING / 2) {
#pragma unroll
            for (int i = 0; i < RING / 2; ++i) {
                // one step = two 1 KiB pieces (16 k-values per lane half): split, 3 MFMAs, re-arm both ring slots
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ring[2 * i]), "+v"(ring[2 * i + 1]) : "n"(RING - 2) : "memory");
                const f32x4 v0 = ring[2 * i], v1 = ring[2 * i + 1];
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                split_pair(v0[0], v0[1], h0, l0);
                split_pair(v0[2], v0[3], h1, l1);
                split_pair(v1[0], v1[1], h2, l2);
                split_pair(v1[2], v1[3], h3, l3);
                const u32x4 ahi = {h0, h1, h2, h3}, alo = {l0, l1, l2, l3};
                const bf16x8 ah = __builtin_bit_cast(bf16x8, ahi), al = __builtin_bit_cast(bf16x8, alo);
                const bf16x8 bh = bh_next, bl = bl_next;
                int nx = pp + i + 1;
                nx = nx == npairs ? 0 : nx;
                bh_next = qhi[nx * 64 + lane];
                bl_next = qlo[nx * 64 + lane];
                if (QT == 1) {  // 32 queries: hi*hi | hi*lo + lo*hi
                    acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc_hi, 0, 0, 0);
                    acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc_lo, 0, 0, 0);
                    acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc_lo, 0, 0, 0);
                } else {        // 64 queries, hi-only query fragments: (hi + lo of x) * hi of q, one accumulator per tile
                    acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc_hi, 0, 0, 0);
                    acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc_hi, 0, 0, 0);
                    acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc_lo, 0, 0, 0);
                    acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc_lo, 0, 0, 0);
                }
                // re-arm: past the end of the range the stream wraps to the first pieces of the next pass (S >= RING
                // whenever S > 0, so one subtraction is enough); the last pass repeats its final piece instead
                int n0 = s + RING + 2 * i, n1 = n0 + 1;
                n0 = n0 < S ? n0 : (wrap ? n0 - S : S - 1);
                n1 = n1 < S ? n1 : (wrap ? n1 - S : S - 1);
                const unsigned voff0 = lane16 + (unsigned)n0 * 1024u;
                const unsigned voff1 = lane16 + (unsigned)n1 * 1024u;
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(ring[2 * i]) : "v"(voff0), "s"(base) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(ring[2 * i + 1]) : "v"(voff1), "s"(base) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            s += RING;
        }
        if (METRIC == HIPRAG_METRIC_L2)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(nrm[0]), "+v"(nrm[1]), "+v"(nrm[2]), "+v"(nrm[3]) : "n"(RING) : "memory");
        int lane_b = lane;
        asm volatile("" : "+v"(lane_b));
        const int h_b = lane_b >> 5;
        if (QT == 1) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = acc_hi[i] + acc_lo[i];
            float sec;
            const float fst = block_lane_top2<METRIC>(acc, nrm, blk, h_b, a, sec);
            park_and_flush(mh[0], fst, blk, b0, b1, lane_b, a.gmax, a.gstride, qbase);
            park_and_flush(ms[0], sec, blk, b0, b1, lane_b, a.gmax2, a.gstride, qbase);
        } else {
            float sec;
            float fst = block_lane_top2<METRIC>(acc_hi, nrm, blk, h_b, a, sec);
            park_and_flush(mh[0], fst, blk, b0, b1, lane_b, a.gmax, a.gstride, qbase);
            park_and_flush(ms[0], sec, blk, b0, b1, lane_b, a.gmax2, a.gstride, qbase);
            fst = block_lane_top2<METRIC>(acc_lo, nrm, blk, h_b, a, sec);
            park_and_flush(mh[QT - 1], fst, blk, b0, b1, lane_b, a.gmax, a.gstride, qbase + 32);
            park_and_flush(ms[QT - 1], sec, blk, b0, b1, lane_b, a.gmax2, a.gstride, qbase + 32);
        }
    }
    }  // pass
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.stamps && lane == 0) a.stamps[2 * gw + 1] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------------
// K1 (bf16 filter, the default).  The candidate scan does not need the rows' low bits: it streams the bf16 copy of the
// index -- HALF the bytes of the fp32 rows -- against bf16 query tiles, two tiles (64 queries) per pass, one
// v_mfma_f32_32x32x16_bf16 per tile and 1 KiB piece, no conversion work at all.  What the truncation costs is carried by
// the certificate (scan_eps, mode 3): |<x, q> - <x^, q^>| <= |x| |q - q^| + |x - x^| |q^|, both deviations computed
// exactly (per query at search time, maximum over rows at add time), ~2.3e-3 |x| |q| for ordinary data.  The fp64
// re-score, round B and the exhaustive path read the fp32 rows, so results are the same exact ones as in every mode.
// Structure as scan_split_kernel: equal contiguous block ranges, hand-counted ring of `nt` loads armed before the
// prologue, passes back to back with a cyclic piece stream, first / second quad values per group.
// Query tile in LDS: piece p, lane (h, b): Q^[b][16p + 8h + 0..7] for queries 0..31, then the same for queries 32..63.
// ------------------------------------------------------------------------------------------------------
template <int METRIC, int NWAVES, int RING, bool MULTI, int CH = 8>
__global__ __launch_bounds__(NWAVES * 64) void scan_bf16_kernel(ScanArgs a)
{
    extern __shared__ float4 qs[];
    constexpr int NT = NWAVES * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P2 = a.P / 2;   // 1 KiB pieces per block of the bf16 copy
    __builtin_amdgcn_s_setprio(3);

    const int64_t gw = (int64_t)blockIdx.x * NWAVES + wave;
    const int64_t W = (int64_t)gridDim.x * NWAVES;
    const int64_t bpw = scan_blocks_per_wave(a.nblocks, W);
    const int64_t b0 = min(gw * bpw, a.nblocks);
    const int64_t b1 = min(b0 + bpw, a.nblocks);
    const int S = (int)((b1 - b0) * P2);
    const float4* base = reinterpret_cast<const float4*>(a.xh) + b0 * P2 * kPieceVec4;
    if (a.stamps && lane == 0) a.stamps[2 * gw] = wall_clock64();
    float mh[2][CH], ms[2][CH];
#pragma unroll
    for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) { mh[u][t] = 0.f; ms[u][t] = 0.f; }
    const unsigned lane16 = (unsigned)lane * 16u;
    const int h = lane >> 5;
    f32x4 ring[RING];
    if (S > 0) {
#pragma unroll
        for (int i = 0; i < RING; ++i) {
            const unsigned voff = lane16 + (unsigned)min(i, S - 1) * 1024u;
            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(ring[i]) : "v"(voff), "s"(base) : "memory");
        }
    }
    const int npass = MULTI ? (a.nq + 63) / 64 : 1;
    const bf16x8* q0 = reinterpret_cast<const bf16x8*>(qs);
    const bf16x8* q1 = q0 + P2 * 64;
    for (int pass = 0; pass < npass; ++pass) {
        const int qbase = pass * 64;
        const bool wrap = pass + 1 < npass;
        if (pass) __syncthreads();  // every wave is done with the previous tile
        {
            bf16x8* qw = reinterpret_cast<bf16x8*>(qs);
            const bool vec_ok = (a.d & 3) == 0;
            // opaque copies of the thread / lane index at the top of every pass and block epilogue: left visible, hipcc hoists
            // their address arithmetic out of the pass loop and keeps ~60 registers of it alive through the ring loop
            int tid_p = tid;
            asm volatile("" : "+v"(tid_p));
            int idx0 = tid_p;
            if (vec_ok && a.d >= 4) {
                // four fragments per step, their eight 16-B loads issued together and UNCONDITIONALLY (clamped query and
                // column, masked afterwards): predicated, each load was a branch + load + s_waitcnt vmcnt(0) -- 32
                // dependent L2 round trips per thread and pass
                constexpr int UQ = 4;
                for (; idx0 + (UQ - 1) * NT < 2 * P2 * 64; idx0 += UQ * NT) {
                    float4 t0[UQ], t1[UQ];
#pragma unroll
                    for (int u4 = 0; u4 < UQ; ++u4) {
                        const int idx = idx0 + u4 * NT;
                        const int tile = idx >= P2 * 64;
                        const int u = idx - tile * P2 * 64;
                        const int p = u >> 6, l = u & 63;
                        const int b = min(qbase + (l & 31) + 32 * tile, a.nq - 1);
                        const int col = 16 * p + 8 * (l >> 5);
                        const float* qp = a.q + (int64_t)b * a.d;
                        t0[u4] = *reinterpret_cast<const float4*>(qp + min(col, a.d - 4));
                        t1[u4] = *reinterpret_cast<const float4*>(qp + min(col + 4, a.d - 4));
                    }
#pragma unroll
                    for (int u4 = 0; u4 < UQ; ++u4) {
                        const int idx = idx0 + u4 * NT;
                        const int tile = idx >= P2 * 64;
                        const int u = idx - tile * P2 * 64;
                        const int p = u >> 6, l = u & 63;
                        const int b = qbase + (l & 31) + 32 * tile;
                        const int col = 16 * p + 8 * (l >> 5);
                        const bool ok0 = b < a.nq && col + 3 < a.d, ok1 = b < a.nq && col + 7 < a.d;
                        bf16x8 o;
                        o[0] = (__bf16)(ok0 ? t0[u4].x : 0.f); o[1] = (__bf16)(ok0 ? t0[u4].y : 0.f);
                        o[2] = (__bf16)(ok0 ? t0[u4].z : 0.f); o[3] = (__bf16)(ok0 ? t0[u4].w : 0.f);
                        o[4] = (__bf16)(ok1 ? t1[u4].x : 0.f); o[5] = (__bf16)(ok1 ? t1[u4].y : 0.f);
                        o[6] = (__bf16)(ok1 ? t1[u4].z : 0.f); o[7] = (__bf16)(ok1 ? t1[u4].w : 0.f);
                        qw[idx] = o;
                    }
                }
            }
            for (int idx = idx0; idx < 2 * P2 * 64; idx += NT) {
                const int tile = idx >= P2 * 64;
                const int u = idx - tile * P2 * 64;
                const int p = u >> 6, l = u & 63;
                const int b = qbase + (l & 31) + 32 * tile;
                const int col = 16 * p + 8 * (l >> 5);
                float v[8];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int c = col + 4 * half;
                    if (b < a.nq && vec_ok && c + 3 < a.d) {
                        const float4 t = *reinterpret_cast<const float4*>(a.q + (int64_t)b * a.d + c);
                        v[4 * half + 0] = t.x; v[4 * half + 1] = t.y; v[4 * half + 2] = t.z; v[4 * half + 3] = t.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[4 * half + j] = (b < a.nq && c + j < a.d) ? a.q[(int64_t)b * a.d + c + j] : 0.f;
                    }
                }
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
                qw[idx] = o;
            }
        }
        __syncthreads();
        if (S <= 0) continue;

        int s = 0;
        bf16x8 n0v = q0[lane], n1v = q1[lane];   // query fragments are read one piece ahead
        for (int64_t blk = b0; blk < b1; ++blk) {
            f32x4 nrm[4];
            if (METRIC == HIPRAG_METRIC_L2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* np = a.norms + blk * kRowsPerBlock + 8 * g + 4 * h;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nrm[g]) : "v"(np) : "memory");
                }
            }
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
            for (int pp = 0; pp < P2; pp += RING) {
#pragma unroll
                for (int i = 0; i < RING; ++i) {
                    // one step = one 1 KiB piece (16 k-values): two MFMAs (one per query tile), re-arm the ring slot.
                    // vmcnt(RING - 1) is exact with only the ring in flight; stores in flight (and the L2 metric's norm
                    // loads) make it wait for more than it needs, never for less -- loads return in order among themselves
                    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ring[i]) : "n"(RING - 1) : "memory");
                    const bf16x8 av = __builtin_bit_cast(bf16x8, ring[i]);
                    const bf16x8 bv0 = n0v, bv1 = n1v;
                    int nx = pp + i + 1;
                    nx = nx == P2 ? 0 : nx;
                    n0v = q0[nx * 64 + lane];
                    n1v = q1[nx * 64 + lane];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv1, acc1, 0, 0, 0);
                    int n0 = s + RING + i;
                    n0 = n0 < S ? n0 : (wrap ? n0 - S : S - 1);
                    const unsigned voff = lane16 + (unsigned)n0 * 1024u;
                    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(ring[i]) : "v"(voff), "s"(base) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                s += RING;
            }
            if (METRIC == HIPRAG_METRIC_L2)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(nrm[0]), "+v"(nrm[1]), "+v"(nrm[2]), "+v"(nrm[3]) : "n"(RING) : "memory");
            float sec, sec1;
            int lane_b = lane;
            asm volatile("" : "+v"(lane_b));
            const int h_b = lane_b >> 5;
            const float fst = block_lane_top2<METRIC>(acc0, nrm, blk, h_b, a, sec);
            const float fst1 = block_lane_top2<METRIC>(acc1, nrm, blk, h_b, a, sec1);
            const int j = (int)((blk - b0) % CH);
            park(mh[0], fst); park(ms[0], sec); park(mh[1], fst1); park(ms[1], sec1);
            if (j == CH - 1) {
                flush_full(mh[0], lane_b, a.gmax, a.gstride, qbase, blk - j);
                flush_full(ms[0], lane_b, a.gmax2, a.gstride, qbase, blk - j);
                flush_full(mh[1], lane_b, a.gmax, a.gstride, qbase + 32, blk - j);
                flush_full(ms[1], lane_b, a.gmax2, a.gstride, qbase + 32, blk - j);
            } else if (blk == b1 - 1) {   // the short last chunk of the range
                flush_partial(mh[0], j + 1, lane_b, a.gmax, a.gstride, qbase, blk - j);
                flush_partial(ms[0], j + 1, lane_b, a.gmax2, a.gstride, qbase, blk - j);
                flush_partial(mh[1], j + 1, lane_b, a.gmax, a.gstride, qbase + 32, blk - j);
                flush_partial(ms[1], j + 1, lane_b, a.gmax2, a.gstride, qbase + 32, blk - j);
            }
        }
    }  // pass
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.stamps && lane == 0) a.stamps[2 * gw + 1] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------------
// fp64 re-scoring of one 4-row group straight from the blocked layout (wave-wide; result for row r0 + (lane&3)
// is returned in every lane with that low index).  The summation order depends only on the row's contents.
// ------------------------------------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ double rescore4(const float4* __restrict__ xb, int P, int64_t blk, int r0,
                                           const float* __restrict__ qv /* LDS, d_pad floats, zero padded */)
{
    const int lane = threadIdx.x & 63;
    const int rr = lane & 3, hh = (lane >> 2) & 1, pq = lane >> 3;
    const float4* src = xb + blk * P * kPieceVec4 + piece_slot(hh, r0 + rr);
    // P <= 128 (LDS limit of the scan), so a lane touches at most 16 pieces.  Loads go out in batches of 8 before their
    // first use: a dependent-latency loop here costs an HBM round trip per piece and used to dominate the finish kernel;
    // all 16 at once spills at the 128-VGPR budget of the 16-wave finish workgroup.  (P >= 1: d >= 1.)
    double acc = 0.0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // UNCONDITIONAL load (pieces past P re-read the last one and are skipped below): `p < P ? src[..] : 0` is
            // compiled into branch + load + s_waitcnt vmcnt(0), i.e. sixteen serialized memory round trips per quad
            const int p = min(pq + 8 * (8 * half + i), P - 1);
            x[i] = src[p * kPieceVec4];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = pq + 8 * (8 * half + i);
            if (p < P) {
                const float* qq = qv + 8 * p + 4 * hh;
                if (METRIC == HIPRAG_METRIC_IP) {
                    acc += (double)x[i].x * (double)qq[0];
                    acc += (double)x[i].y * (double)qq[1];
                    acc += (double)x[i].z * (double)qq[2];
                    acc += (double)x[i].w * (double)qq[3];
                } else {
                    double t;
                    t = (double)x[i].x - (double)qq[0]; acc += t * t;
                    t = (double)x[i].y - (double)qq[1]; acc += t * t;
                    t = (double)x[i].z - (double)qq[2]; acc += t * t;
                    t = (double)x[i].w - (double)qq[3]; acc += t * t;
                }
            }
        }
    }
#pragma unroll
    for (int off = 4; off <= 32; off <<= 1) acc += __shfl_xor(acc, off);
    return acc;
}

struct FinishArgs {
    const float4* xb;
    const float* q;            // [nq, d]
    const u64* ck;             // [nq, nchunks*K1]
    const i64* ci;
    const unsigned* max_norm2_bits;  // [0] max |x|^2, [1] max |x - bf16(x)|^2 (float bits)
    double* out64;             // [nq, k]
    float* out32;              // [nq, k] or null
    int64_t* out_ids;          // [nq, k]
    int* flags;                // [nq]
    int* arrivals;             // [nq] exhaustive-path arrival counters, zeroed here
    unsigned long long* fallback_counter;
    int64_t ntotal, id_base, ncand;  // ncand = nchunks*K1
    int64_t bpw, nblocks;            // to decode group slots (group_decode)
    u64* sel;                        // [nq, 64] selected groups (fast path)
    u64* cand_k;                     // [nq, 1024] re-scored candidates
    i64* cand_i;
    double* qn2;                     // [nq][2]: exact |q|^2 and |q - bf16(q)|^2
    float* sec;                      // [nq, 64] `second` of every selected group (fin_rescore -> fin_final)
    const float* gmax2;              // the scan's second-value array, gstride floats per query
    int64_t gstride;
    int d, P, k, Kp;           // Kp = K' groups re-scored; K1 = Kp + 1
    unsigned long long* dbg;   // HIPRAG_DEBUG_GAPS only: [8] wall-clock stamps of this launch's tail kernels
    int chunk;                 // blocks per flush of the scan that filled gmax
    float* tau;                // [nq] round-B threshold of a flagged query: every row that can reach its top k sits in a group
                               //      whose `first` is >= tau (second round of the finish, below)
    int* rb_count;             // [nq] groups collected by round B (zeroed by fin_final)
    int split;                 // scan operand mode: 0 exact fp32, 1 bf16 hi/lo split, 2 split x + hi-only queries (64/pass)
};

// Certificate slack: |scan value - exact score| <= eps for every row, on the scale the scan selects by (IP: <x,q>;
// L2: 2<x,q> - |x|^2 = |q|^2 - dist).  Terms: fp32 accumulation ((d_pad + 2) u, u = 2^-24; + 80 u for the bf16 split's
// extra roundings), the dropped lo*lo products of the split (2^-16), the hi-only query tiles of the 64-query mode
// (1.97e-3 ~ 2^-9), and the quad tag in the two low mantissa bits (kTagSlack of the value's magnitude).
constexpr double kTagSlack = 4.76837158203125e-07;  // 2^-21
template <int METRIC>
__device__ __forceinline__ double scan_eps(int dpad, int split, double qn2, double xn2, double dq2, double dx2 = 0.0)
{
    const double xn = sqrt(xn2), qn = sqrt(qn2);
    const double u = 5.9604644775390625e-08;  // 2^-24
    double eps = 1.05 * (double)(dpad + (split ? 80 : 2)) * u * qn * xn;
    if (split == 1) eps += 1.52587890625e-05 * qn * xn;
    // 64-query tiles: the scan sees q^ = bf16(q) and x^ = hi + lo.  |<x, q - q^>| <= |x| |q - q^| (Cauchy-Schwarz) with
    // |q - q^| computed exactly per query (dq2, same RNE conversion as the scan prologue) -- about 0.3 * 2^-9 |q| for
    // ordinary data instead of the element-wise worst case 2^-9 |q|, which is what lets K' = k + 12 certify;
    // |<x - x^, q^>| <= 2^-17 |x| |q^| for the two-term split of the rows.
    if (split == 2) eps += 7.62939453125e-06 * 1.01 * qn * xn + 1.0001 * sqrt(dq2) * xn;
    // bf16 filter copy: |<x, q> - <x^, q^>| <= |x| |q - q^| + |x - x^| |q^|, |q^| <= |q| + |q - q^|; dx2 = max over rows
    if (split == 3) eps += 1.0001 * (sqrt(dq2) * xn + sqrt(dx2) * (qn + sqrt(dq2)));
    if (METRIC == HIPRAG_METRIC_IP) return eps + kTagSlack * (qn * xn + eps);
    eps = 2.0 * eps + 4.0 * u * (xn * xn + qn * xn) + 4.0 * u * qn2;  // 2 * acc - norm, and the rounding of |q|^2 - dist
    return eps + kTagSlack * (2.0 * qn * xn + xn * xn + eps);
}
// k-th exact score (ordered key) on the scan's scale; -inf while fewer than k rows are known
template <int METRIC>
__device__ __forceinline__ double kth_on_scan_scale(u64 kth_key, double qn2)
{
    if (kth_key == 0) return -INFINITY;
    return METRIC == HIPRAG_METRIC_IP ? unord64(kth_key) : qn2 - (-unord64(kth_key));
}

template <int METRIC>
__device__ __forceinline__ void write_result(double* out64, float* out32, int64_t* out_ids, int64_t o, u64 key, i64 id,
                                             int64_t id_base)
{
    double s;
    if (key == 0) {
        s = METRIC == HIPRAG_METRIC_IP ? -DBL_MAX : DBL_MAX;
        out_ids[o] = -1;
        if (out32) out32[o] = METRIC == HIPRAG_METRIC_IP ? -FLT_MAX : FLT_MAX;
    } else {
        s = METRIC == HIPRAG_METRIC_IP ? unord64(key) : -unord64(key);
        out_ids[o] = id + id_base;
        if (out32) out32[o] = (float)s;
    }
    out64[o] = s;
}

// ------------------------------------------------------------------------------------------------------
// K2b: per query -- merge chunk winners, re-score K' groups in fp64, final top-k, certificate
// ------------------------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(kSelThreads) void finish_kernel(FinishArgs a)
{
    extern __shared__ unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    i64* ids = reinterpret_cast<i64*>(keys + kSelChunk);
    u64* selk = reinterpret_cast<u64*>(ids + kSelChunk);
    const int K1 = a.Kp + 1;
    i64* seli = reinterpret_cast<i64*>(selk + K1);
    KeyId* red = reinterpret_cast<KeyId*>(seli + K1);
    double* dred = reinterpret_cast<double*>(red + 2 * (kSelThreads / 64));
    u64& kth_key = *reinterpret_cast<u64*>(dred + kSelThreads / 64);  // all LDS in the one dynamic array
    float* qv = reinterpret_cast<float*>(dred + 2 * (kSelThreads / 64) + 1);   // dred: |q|^2 parts, kth_key, |q - q^|^2 parts

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int dpad = a.P * 8;

    // query into LDS (zero padded) and its exact squared norm
    double qpart = 0.0, dpart = 0.0;
    for (int c = tid; c < dpad; c += kSelThreads) {
        float v = c < a.d ? a.q[(int64_t)q * a.d + c] : 0.f;
        qv[c] = v;
        qpart += (double)v * (double)v;
        const double dv = (double)v - (double)(float)(__bf16)v;
        dpart += dv * dv;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { qpart += __shfl_xor(qpart, off); dpart += __shfl_xor(dpart, off); }
    if (lane == 0) { dred[wave] = qpart; dred[kSelThreads / 64 + 1 + wave] = dpart; }
    if (tid == 0) kth_key = 0;
    __syncthreads();
    double qn2 = 0.0, dq2 = 0.0;
    for (int w = 0; w < kSelThreads / 64; ++w) { qn2 += dred[w]; dq2 += dred[kSelThreads / 64 + 1 + w]; }

    {
        const u64* sk = a.ck + (int64_t)q * a.ncand;
        const i64* si = a.ci + (int64_t)q * a.ncand;
        wg_stream_topk<kSelThreads>([&](i64 i, u64& k, i64& id) { k = sk[i]; id = si[i]; }, a.ncand, K1, keys, ids, red,
                                    selk, seli);
    }

    // re-score the K' selected groups (16 rows each)
    for (int j = wave; j < a.Kp; j += kSelThreads / 64) {
        const u64 gk = selk[j];
        const i64 gi = seli[j];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u64 key = 0;
            i64 row = -1;
            if (gk != 0) {
                int64_t blk;
                int gh;
                group_decode(gi, a.bpw, a.nblocks, a.chunk, blk, gh);
                const int r0 = 8 * g + 4 * gh;
                const double s = rescore4<METRIC>(a.xb, a.P, blk, r0, qv);
                row = blk * kRowsPerBlock + r0 + (lane & 3);
                if (row < a.ntotal) key = ord64(METRIC == HIPRAG_METRIC_IP ? s : -s);
            }
            if (lane < 4) { keys[j * 16 + g * 4 + lane] = key; ids[j * 16 + g * 4 + lane] = row; }
        }
    }
    __syncthreads();

    const int64_t ob = (int64_t)q * a.k;
    wg_topk_rounds<kSelThreads>(keys, ids, a.Kp * 16, a.k, red, [&](int r, u64 k, i64 id) {
        write_result<METRIC>(a.out64, a.out32, a.out_ids, ob + r, k, id, a.id_base);
        if (r == a.k - 1) kth_key = k;
    });

    if (tid == 0) {
        int flag = 0;
        const u64 bk = selk[a.Kp];  // best group NOT re-scored
        if (bk != 0) {
            const float m = unord32((u32)(bk >> 32));
            if (m > -1.0e38f) {  // below that: padding only, every real row was re-scored
                const double eps = scan_eps<METRIC>(dpad, a.split, qn2, (double)__uint_as_float(a.max_norm2_bits[0]), dq2, (double)__uint_as_float(a.max_norm2_bits[1]));
                if (!(kth_on_scan_scale<METRIC>(kth_key, qn2) > (double)m + eps)) flag = 1;
            }
        }
        a.flags[q] = flag;
        a.arrivals[q] = 0;
        if (flag) atomicAdd(a.fallback_counter, 1ull);
    }
}

// ------------------------------------------------------------------------------------------------------
// K2b (fast form, K' + 1 <= 64, i.e. k <= 57): the same stages as finish_kernel on the wave-resident sorted lists of
// topk_device.h instead of barrier-per-round argmax, split into three small launches so that every stage gets the
// parallelism and register budget it wants (one monolithic 16-wave kernel spilled and serialised its HBM round trips):
//   fin_merge_kernel    one 16-wave workgroup per query: each wave reduces 1/16 of select_wave_kernel's winners, wave 0
//                       merges the 16 lists -> sel[q][0..K'] (packed value|slot), plus the query's exact |q|^2
//   fin_rescore_kernel  one wave per (query, group, row quad): fp64 re-score straight from the blocked layout, all of a
//                       lane's loads in flight at once -> cand[q][16 * K']
//   fin_final_kernel    one wave per query: exact top-k of the candidates under (score, id), certificate, flags
// All three fit beside a resident scan workgroup (<= 10 KiB LDS).
// ------------------------------------------------------------------------------------------------------
constexpr int kFinWaves = 16;
constexpr int kCandPerQuery = 256;   // 64 group slots x the 4 rows of the tagged quad

template <int NPL, int NW>
__global__ __launch_bounds__(NW * 64) void fin_merge_kernel(FinishArgs a)
{
    __shared__ u64 lists[NW > 1 ? NW * 64 : 1];
    __shared__ double dred[2 * NW];
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x;
    const int K1 = a.Kp + 1;
    if (a.dbg && tid == 0) atomicMin(a.dbg + 1, (unsigned long long)wall_clock64());

    double qpart = 0.0, dpart = 0.0;
#pragma unroll 4
    for (int c = tid; c < a.d; c += NT) {
        const float vf = a.q[(int64_t)q * a.d + c];
        const double v = (double)vf, dv = v - (double)(float)(__bf16)vf;   // the scan's query tile holds bf16(q), RNE
        qpart += v * v;
        dpart += dv * dv;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { qpart += __shfl_xor(qpart, off); dpart += __shfl_xor(dpart, off); }
    const u64* sk = a.ck + (int64_t)q * a.ncand;
    const i64* si = a.ci + (int64_t)q * a.ncand;
    // unconditional loads of keys AND slots (clamped index), combined afterwards: behind `if (i < ncand)` / `if (kk)`
    // every candidate cost two dependent, serialized memory round trips
    u64 c[NPL];
    i64 cs[NPL];
#pragma unroll
    for (int n = 0; n < NPL; ++n) {
        const int64_t i = min(((int64_t)wave * NPL + n) * 64 + lane, a.ncand - 1);
        c[n] = sk[i];
        cs[n] = si[i];
    }
#pragma unroll
    for (int n = 0; n < NPL; ++n) {
        const int64_t i = ((int64_t)wave * NPL + n) * 64 + lane;
        c[n] = (i < a.ncand && c[n] != 0) ? (c[n] | (u64)(0xFFFFFFFFu - (u32)cs[n])) : 0;
    }
    WaveListPacked L;
    wave_topk_packed<NPL>(c, K1, L);
    if (NW == 1) {  // the usual case (<= 512 candidates): one wave, no LDS, runs beside a resident scan workgroup
        a.sel[(int64_t)q * 64 + lane] = lane < K1 ? L.e : 0;
        if (lane == 0) { a.qn2[2 * q] = qpart; a.qn2[2 * q + 1] = dpart; }
        return;
    }
    if (lane == 0) { dred[wave] = qpart; dred[NW + wave] = dpart; }
    lists[wave * 64 + lane] = lane < K1 ? L.e : 0;
    __syncthreads();
    if (wave == 0) {
        u64 c2[NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) c2[n] = lists[n * 64 + lane];
        WaveListPacked L2;
        wave_topk_packed<NW>(c2, K1, L2);
        a.sel[(int64_t)q * 64 + lane] = lane < K1 ? L2.e : 0;
        if (lane == 0) {
            double qn2 = 0.0, dq2 = 0.0;
            for (int w = 0; w < NW; ++w) { qn2 += dred[w]; dq2 += dred[NW + w]; }
            a.qn2[2 * q] = qn2;
            a.qn2[2 * q + 1] = dq2;
        }
    }
}

// fin_select_kernel: select_wave_kernel + fin_merge_kernel in one launch, one 16-wave workgroup per query: the K' + 1 best
// group maxima by one threshold pass over the query's `first` array (select_threshold_topk, topk_device.h) -> sel[q][0..K'],
// and the query's exact |q|^2 and |q - bf16(q)|^2.
__global__ __launch_bounds__(1024) void fin_select_kernel(FinishArgs a, const float* __restrict__ gmax, i64 gstride, i64 ngroups)
{
    __shared__ SelectScratch S;
    __shared__ double dred[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x;
    double qpart = 0.0, dpart = 0.0;
    for (int c = tid; c < a.d; c += 1024) {
        const float vf = a.q[(int64_t)q * a.d + c];
        const double v = (double)vf, dv = v - (double)(float)(__bf16)vf;   // the scan's query tile holds bf16(q), RNE
        qpart += v * v;
        dpart += dv * dv;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { qpart += __shfl_xor(qpart, off); dpart += __shfl_xor(dpart, off); }
    if (lane == 0) { dred[wave] = qpart; dred[16 + wave] = dpart; }
    select_threshold_topk(gmax + (i64)q * gstride, ngroups, a.Kp + 1, S, a.sel + (int64_t)q * 64);   // has barriers
    if (tid == 0) {
        double qn2 = 0.0, dq2 = 0.0;
        for (int w = 0; w < 16; ++w) { qn2 += dred[w]; dq2 += dred[16 + w]; }
        a.qn2[2 * q] = qn2;
        a.qn2[2 * q + 1] = dq2;
    }
}

// grid (ceil(K'/4), nq), 256 threads: wave w re-scores the tagged quad (4 rows) of selected group j = 4 * blockIdx.x + w
template <int METRIC>
__global__ __launch_bounds__(256) void fin_rescore_kernel(FinishArgs a)
{
    extern __shared__ float qv[];  // d_pad floats (4 KiB: fits beside a resident scan workgroup)
    const int tid = threadIdx.x, lane = tid & 63;
    const int j = blockIdx.x * 4 + (tid >> 6), q = blockIdx.y;
    const int dpad = a.P * 8;
    if (a.dbg && tid == 0) atomicMin(a.dbg + 2, (unsigned long long)wall_clock64());
    for (int c = tid; c < dpad; c += 256) qv[c] = c < a.d ? a.q[(int64_t)q * a.d + c] : 0.f;
    __syncthreads();
    if (j >= a.Kp) return;
    const u64 e = a.sel[(int64_t)q * 64 + j];
    u64 key = 0;
    i64 row = -1;
    float sec = -FLT_MAX;
    if (e != 0) {
        const i64 slot = (i64)packed_index(e);
        const int g = (int)(__float_as_uint(packed_value(e)) & 3u);
        int64_t blk;
        int gh;
        group_decode(slot, a.bpw, a.nblocks, a.chunk, blk, gh);
        const int r0 = 8 * g + 4 * gh;
        const double s = rescore4<METRIC>(a.xb, a.P, blk, r0, qv);
        row = blk * kRowsPerBlock + r0 + (lane & 3);
        if (row < a.ntotal) key = ord64(METRIC == HIPRAG_METRIC_IP ? s : -s);
        sec = a.gmax2[(int64_t)q * a.gstride + slot];
    }
    if (lane < 4) {
        const int64_t o = (int64_t)q * kCandPerQuery + j * 4 + lane;
        a.cand_k[o] = key;
        a.cand_i[o] = row;
    }
    if (lane == 0) a.sec[(int64_t)q * 64 + j] = sec;
}

// one wave per query: exact top-k of the 4 K' re-scored rows under (score, id); then every selected group whose `second`
// could still reach the k-th score gets its other three quads re-scored here (about one query in a thousand has one);
// then the certificate against the best group that was not selected.
template <int METRIC>
__global__ __launch_bounds__(64) void fin_final_kernel(FinishArgs a)
{
    __shared__ float qv[kMaxDPad];  // staged only when a group needs its other quads
    const int q = blockIdx.x, lane = threadIdx.x;
    const int ncand = a.Kp * 4;
    if (a.dbg && lane == 0) atomicMin(a.dbg + 3, (unsigned long long)wall_clock64());
    const u64* ck = a.cand_k + (int64_t)q * kCandPerQuery;
    const i64* ci = a.cand_i + (int64_t)q * kCandPerQuery;
    u64 ckc[4];
    i64 cic[4];
    u64 m = 0;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int i = n * 64 + lane;
        ckc[n] = i < ncand ? ck[i] : 0ull;
        cic[n] = i < ncand ? ci[i] : -1;
        m = ckc[n] > m ? ckc[n] : m;
    }
    const u64 e_l = a.sel[(int64_t)q * 64 + lane];                       // lanes 0..K' hold the selected groups
    const float m2_l = lane < a.Kp ? a.sec[(int64_t)q * 64 + lane] : -FLT_MAX;
    const double qn2 = a.qn2[2 * q], dq2 = a.qn2[2 * q + 1];
    const int dpad = a.P * 8;
    const double eps = scan_eps<METRIC>(dpad, a.split, qn2, (double)__uint_as_float(a.max_norm2_bits[0]), dq2, (double)__uint_as_float(a.max_norm2_bits[1]));

    const u64 t0 = wave_kth_of_lanes(m, a.k);   // k lanes hold a key >= t0: nothing below t0 can reach the top k
    WaveListPair F;
    F.init();
#pragma unroll
    for (int n = 0; n < 4; ++n) F.offer(ckc[n] >= t0 ? ckc[n] : 0ull, cic[n], a.k);

    unsigned long long done = 0;
    bool staged = false;
    for (;;) {
        const double kth = kth_on_scan_scale<METRIC>(readlane_u64(F.k, a.k - 1), qn2);
        const bool need = lane < a.Kp && e_l != 0 && m2_l > -1.0e38f && !(kth > (double)m2_l + eps);
        const unsigned long long mask = __ballot(need) & ~done;
        if (!mask) break;
        const int j = __builtin_ctzll(mask);
        done |= 1ull << j;
        if (!staged) {
            for (int c = lane; c < dpad; c += 64) qv[c] = c < a.d ? a.q[(int64_t)q * a.d + c] : 0.f;
            __syncthreads();
            staged = true;
        }
        const u64 e = readlane_u64(e_l, j);
        const int tagged = (int)(__float_as_uint(packed_value(e)) & 3u);
        int64_t blk;
        int gh;
        group_decode((i64)packed_index(e), a.bpw, a.nblocks, a.chunk, blk, gh);
        for (int g = 0; g < 4; ++g) {
            if (g == tagged) continue;
            const int r0 = 8 * g + 4 * gh;
            const double s = rescore4<METRIC>(a.xb, a.P, blk, r0, qv);
            const i64 row = blk * kRowsPerBlock + r0 + (lane & 3);
            const u64 key = (lane < 4 && row < a.ntotal) ? ord64(METRIC == HIPRAG_METRIC_IP ? s : -s) : 0ull;
            F.offer(key, row, a.k);
        }
    }

    if (lane < a.k) write_result<METRIC>(a.out64, a.out32, a.out_ids, (int64_t)q * a.k + lane, F.k, F.id, a.id_base);
    const u64 kth_key = readlane_u64(F.k, a.k - 1);
    const u64 bk = readlane_u64(e_l, a.Kp);  // best group NOT selected
    if (lane == 0) {
        int flag = 0;
        if (bk != 0) {
            const float m32 = packed_value(bk);
            if (m32 > -1.0e38f && !(kth_on_scan_scale<METRIC>(kth_key, qn2) > (double)m32 + eps)) flag = 1;
        }
        a.flags[q] = flag;
        a.arrivals[q] = 0;
        a.rb_count[q] = 0;
        if (flag) {
            // tau = (k-th exact score among the rows seen so far) - eps, rounded DOWN to float: a row that beats that k-th
            // score has scan value >= tau, so round B re-scores every group whose `first` reaches tau
            const double t = kth_on_scan_scale<METRIC>(kth_key, qn2) - eps;
            float tf = t > -3.0e38 ? (float)t : -FLT_MAX;
            if ((double)tf > t) tf = nextafterf(tf, -INFINITY);
            a.tau[q] = tf;
        }
        if (a.dbg) atomicMin(a.dbg + 4, ~(unsigned long long)wall_clock64());
    }
}

// ------------------------------------------------------------------------------------------------------
// Round B of the finish: a query whose certificate failed (more than K' groups within eps of its k-th score -- half of
// the queries at k = 50 on the 64-query tiles, where K' is capped at 63 by the wave lists) is NOT sent to the exhaustive
// path straight away.  fin_final left tau = (k-th exact score so far) - eps; every row that can still enter the top k
// has a scan value >= tau, so it suffices to re-score the groups whose `first` reaches tau:
//   roundb_collect_kernel   flagged queries only: compact those group slots (atomic append, cap kRoundBGroups), leaving out
//                           the K' groups the first round already re-scored
//   roundb_rescore_kernel   the tagged quad of every collected group in fp64
//   roundb_final_kernel     exact top-k under (score, id) of those rows and the first round's top-k (other quads where a
//                           `second` demands it); clears the flag
// Costs one more read of the query's N/16 group values and ~1 MB of rows -- microseconds, against a full fp64 pass over
// the index for the exhaustive path, which now only sees queries with more than kRoundBGroups such groups (ties).
// ------------------------------------------------------------------------------------------------------
constexpr int kRoundBGroups = 256;

struct RoundBArgs {
    const float4* xb;
    const float* q;
    const float* gmax;         // `first` values, gstride per query
    int64_t gstride, ngroups;
    int* flags;
    const float* tau;
    int* count;                // [nq]
    u32* slots;                // [nq][kRoundBGroups]
    u64* bk;                   // [nq][kRoundBGroups * 4]: the tagged quad of every collected group
    i64* bi;
    float* bsec;               // [nq][kRoundBGroups] `second` of every collected group
    const float* gmax2;
    const double* qn2;         // [nq][2] from the first round
    const unsigned* max_norm2_bits;
    int split;
    double* out64;
    float* out32;
    int64_t* out_ids;
    unsigned long long* fallback_counter;   // queries that go on to the exhaustive path
    unsigned long long* roundb_counter;     // queries settled by round B
    int64_t ntotal, id_base, bpw, nblocks;
    int d, P, k, chunk;
    const u64* sel;            // [nq][64] the first round's selected groups (packed first | slot), sorted; lanes >= K' + 1 zero
    int Kp;                    // groups the first round re-scored: round B skips them and starts from its top-k
};

// grid (ceil(ngroups / 4096), nq), 256 threads, 16 values per thread
__global__ __launch_bounds__(256) void roundb_collect_kernel(RoundBArgs a)
{
    const int q = blockIdx.y;
    if (!a.flags[q]) return;
    const float tau = a.tau[q];
    // groups the first round already re-scored are exactly those whose packed (first | slot) key is >= the K'-th selected
    // one (keys are unique); their rows are represented by the first round's top-k, which roundb_final starts from
    const u64 e_last = a.sel[(int64_t)q * 64 + a.Kp - 1];
    const float* src = a.gmax + (int64_t)q * a.gstride;
    const int64_t base = (int64_t)blockIdx.x * 4096 + threadIdx.x * 4;
    // all four loads first (unconditional; the row of gmax is padded to gstride >= ngroups rounded up to 4 -- past
    // ngroups the values are masked below), then the rare appends
    float4 xs[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int64_t i = min(base + it * 1024, (int64_t)(max((int64_t)0, a.ngroups - 1) & ~3ll));
        xs[it] = *reinterpret_cast<const float4*>(src + i);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int64_t i = base + it * 1024;
        float v[4] = {xs[it].x, xs[it].y, xs[it].z, xs[it].w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (i + t >= a.ngroups) v[t] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (v[t] >= tau && v[t] > -1.0e38f && e_last != 0 && pack_key(v[t], (u32)(i + t)) < e_last) {
                const int pos = atomicAdd(a.count + q, 1);
                if (pos < kRoundBGroups) a.slots[(int64_t)q * kRoundBGroups + pos] = (u32)(i + t);
            }
    }
}

// grid (8, nq), 256 threads: wave w re-scores the TAGGED quad of collected groups j = 4 * blockIdx.x + w, + 32, ...
template <int METRIC>
__global__ __launch_bounds__(256) void roundb_rescore_kernel(RoundBArgs a)
{
    extern __shared__ float qv[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = blockIdx.y;
    if (!a.flags[q]) return;
    const int n = a.count[q];
    if (n > kRoundBGroups || (int)blockIdx.x * 4 >= n) return;
    const int dpad = a.P * 8;
    for (int c = tid; c < dpad; c += 256) qv[c] = c < a.d ? a.q[(int64_t)q * a.d + c] : 0.f;
    __syncthreads();
    for (int j = blockIdx.x * 4 + w; j < n; j += gridDim.x * 4) {
        const i64 slot = (i64)a.slots[(int64_t)q * kRoundBGroups + j];
        const int g = (int)(__float_as_uint(a.gmax[(int64_t)q * a.gstride + slot]) & 3u);
        int64_t blk;
        int gh;
        group_decode(slot, a.bpw, a.nblocks, a.chunk, blk, gh);
        const int r0 = 8 * g + 4 * gh;
        const double s = rescore4<METRIC>(a.xb, a.P, blk, r0, qv);
        const i64 row = blk * kRowsPerBlock + r0 + (lane & 3);
        if (lane < 4) {
            const int64_t o = ((int64_t)q * kRoundBGroups + j) * 4 + lane;
            a.bk[o] = row < a.ntotal ? ord64(METRIC == HIPRAG_METRIC_IP ? s : -s) : 0ull;
            a.bi[o] = row;
        }
        if (lane == 0) a.bsec[(int64_t)q * kRoundBGroups + j] = a.gmax2[(int64_t)q * a.gstride + slot];
    }
}

// one wave per query: exact top-k of the <= 1024 re-scored rows; groups whose `second` could still reach the k-th score
// get their other three quads re-scored here (as in fin_final_kernel); every group with `first` >= tau is then covered
template <int METRIC>
__global__ __launch_bounds__(64) void roundb_final_kernel(RoundBArgs a)
{
    __shared__ float qv[kMaxDPad];
    const int q = blockIdx.x, lane = threadIdx.x;
    if (!a.flags[q]) return;
    const int n = a.count[q];
    if (n > kRoundBGroups) {   // too many groups within eps (massive ties): the exhaustive path settles it
        if (lane == 0) atomicAdd(a.fallback_counter, 1ull);
        return;
    }
    const int ncand = n * 4;
    const u64* ck = a.bk + (int64_t)q * kRoundBGroups * 4;
    const i64* ci = a.bi + (int64_t)q * kRoundBGroups * 4;
    u64 m = 0;
    for (int i = lane; i < ncand; i += 64) { const u64 kk = ck[i]; m = kk > m ? kk : m; }
    const u64 t0 = wave_kth_of_lanes(m, a.k);
    WaveListPair F;
    F.init();
    {   // start from the first round's exact top-k (fin_final wrote it): every row it re-scored is either in there or beaten
        u64 sk = 0;
        i64 si = -1;
        if (lane < a.k) {
            const i64 id = a.out_ids[(int64_t)q * a.k + lane];
            if (id >= 0) {
                const double s = a.out64[(int64_t)q * a.k + lane];
                sk = ord64(METRIC == HIPRAG_METRIC_IP ? s : -s);
                si = id - a.id_base;
            }
        }
        F.offer(sk, si, a.k);
    }
    for (int i0 = 0; i0 < ncand; i0 += 64) {
        const int i = i0 + lane;
        const u64 kk = i < ncand ? ck[i] : 0ull;
        F.offer(kk >= t0 ? kk : 0ull, i < ncand ? ci[i] : -1, a.k);
    }
    const double qn2 = a.qn2[2 * q], dq2 = a.qn2[2 * q + 1];
    const int dpad = a.P * 8;
    const double eps = scan_eps<METRIC>(dpad, a.split, qn2, (double)__uint_as_float(a.max_norm2_bits[0]), dq2,
                                        (double)__uint_as_float(a.max_norm2_bits[1]));
    bool staged = false;
    for (int j0 = 0; j0 < n; j0 += 64) {   // 64 groups at a time, one per lane
        const int j = j0 + lane;
        const float m2 = j < n ? a.bsec[(int64_t)q * kRoundBGroups + j] : -FLT_MAX;
        unsigned long long done = 0;
        for (;;) {
            const double kth = kth_on_scan_scale<METRIC>(readlane_u64(F.k, a.k - 1), qn2);
            const bool need = j < n && m2 > -1.0e38f && !(kth > (double)m2 + eps);
            const unsigned long long mask = __ballot(need) & ~done;
            if (!mask) break;
            const int l = __builtin_ctzll(mask);
            done |= 1ull << l;
            if (!staged) {
                for (int c = lane; c < dpad; c += 64) qv[c] = c < a.d ? a.q[(int64_t)q * a.d + c] : 0.f;
                __syncthreads();
                staged = true;
            }
            const i64 slot = (i64)a.slots[(int64_t)q * kRoundBGroups + j0 + l];
            const int tagged = (int)(__float_as_uint(a.gmax[(int64_t)q * a.gstride + slot]) & 3u);
            int64_t blk;
            int gh;
            group_decode(slot, a.bpw, a.nblocks, a.chunk, blk, gh);
            for (int g = 0; g < 4; ++g) {
                if (g == tagged) continue;
                const int r0 = 8 * g + 4 * gh;
                const double s = rescore4<METRIC>(a.xb, a.P, blk, r0, qv);
                const i64 row = blk * kRowsPerBlock + r0 + (lane & 3);
                const u64 key = (lane < 4 && row < a.ntotal) ? ord64(METRIC == HIPRAG_METRIC_IP ? s : -s) : 0ull;
                F.offer(key, row, a.k);
            }
        }
    }
    if (lane < a.k) write_result<METRIC>(a.out64, a.out32, a.out_ids, (int64_t)q * a.k + lane, F.k, F.id, a.id_base);
    if (lane == 0) {
        a.flags[q] = 0;
        atomicAdd(a.roundb_counter, 1ull);
    }
}

// For k beyond what the selection kernels hold (K' * 16 re-scored rows > 4096) every query goes straight to the exhaustive
// path: exact, just not fast -- k in the hundreds is outside anything the reference asks for (top_chunks = 50).
__global__ void flag_all_kernel(int* flags, int* arrivals, unsigned long long* fallback_counter, int nq)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) {
        flags[q] = 1;
        arrivals[q] = 0;
        atomicAdd(fallback_counter, 1ull);
    }
}

// ------------------------------------------------------------------------------------------------------
// K2c / K2d: exhaustive exact path for flagged queries (exit immediately otherwise)
// ------------------------------------------------------------------------------------------------------
struct ExArgs {
    const float4* xb;
    const float* q;
    const int* flags;
    int* arrivals;  // [nq] zeroed by the finish kernel of the same pass
    u64* ek;        // [nq, nslices*kk]
    i64* ei;
    double* out64;
    float* out32;
    int64_t* out_ids;
    int64_t ntotal, id_base;
    int d, P, k, kk, nslices, nq;
    unsigned long long* dbg;
};

// One launch: workgroup s re-scores rows [s*kExRows, +kExRows) of every FLAGGED query in fp64 and publishes its best
// kk; the last workgroup to arrive for a query (device-scope counter behind __threadfence) merges the slices and
// overwrites the query's results.  Unflagged passes cost one tiny launch: every workgroup reads nq flags and exits.
template <int METRIC>
__global__ __launch_bounds__(kSelThreads) void exhaustive_kernel(ExArgs a)
{
    extern __shared__ unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    i64* ids = reinterpret_cast<i64*>(keys + kExRows);
    u64* selk = reinterpret_cast<u64*>(ids + kExRows);
    i64* seli = reinterpret_cast<i64*>(selk + a.k);
    KeyId* red = reinterpret_cast<KeyId*>(seli + a.k);
    float* qv = reinterpret_cast<float*>(red + 2 * (kSelThreads / 64));
    int* ticket = reinterpret_cast<int*>(qv + a.P * 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dpad = a.P * 8;
    if (a.dbg && tid == 0) atomicMin(a.dbg + 5, (unsigned long long)wall_clock64());
    int mine = 0;
    for (int q = tid; q < a.nq; q += kSelThreads) mine |= a.flags[q];
    if (!__syncthreads_or(mine)) {  // the common case: a <= n_cu-workgroup launch that reads nq flags and leaves
        if (a.dbg && tid == 0) atomicMin(a.dbg + 6, ~(unsigned long long)wall_clock64());
        return;
    }
    for (int slice = blockIdx.x; slice < a.nslices; slice += gridDim.x) {
        const int64_t row_base = (int64_t)slice * kExRows;
        const int nrows = (int)min((int64_t)kExRows, a.ntotal - row_base);
        for (int q = 0; q < a.nq; ++q) {
            if (!a.flags[q]) continue;  // uniform across the workgroup
            __syncthreads();
            for (int c = tid; c < dpad; c += kSelThreads) qv[c] = c < a.d ? a.q[(int64_t)q * a.d + c] : 0.f;
            __syncthreads();
            for (int g = wave; g * 4 < kExRows; g += kSelThreads / 64) {
                const int64_t row0 = row_base + (int64_t)g * 4;
                u64 key = 0;
                const int64_t row = row0 + (lane & 3);
                if (row0 < a.ntotal) {
                    const double s = rescore4<METRIC>(a.xb, a.P, row0 / kRowsPerBlock, (int)(row0 % kRowsPerBlock), qv);
                    if (row < a.ntotal) key = ord64(METRIC == HIPRAG_METRIC_IP ? s : -s);
                }
                if (lane < 4) { keys[g * 4 + lane] = key; ids[g * 4 + lane] = row; }
            }
            __syncthreads();
            const int64_t M = (int64_t)a.nslices * a.kk;
            u64* ok = a.ek + (int64_t)q * M + (int64_t)slice * a.kk;
            i64* oi = a.ei + (int64_t)q * M + (int64_t)slice * a.kk;
            wg_topk_rounds<kSelThreads>(keys, ids, max(nrows, 0), a.kk, red, [&](int r, u64 k, i64 id) { ok[r] = k; oi[r] = id; });
            // Publish this slice, then take a ticket (cdna_hip_programming.md Guideline 16, counter form): stores drained
            // by their wave -> workgroup barrier -> one lane: agent-scope release, explicit drain, relaxed agent atomic.
            // The last arriver acquires once, and the barrier after it holds every wave's loads behind the invalidate.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int t = __hip_atomic_fetch_add(a.arrivals + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == a.nslices - 1) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *ticket = t;
            }
            __syncthreads();
            if (*ticket == a.nslices - 1) {  // last arriver merges
                const u64* sk = a.ek + (int64_t)q * M;
                const i64* si = a.ei + (int64_t)q * M;
                wg_stream_topk<kSelThreads, kExRows>([&](i64 i, u64& k, i64& id) { k = sk[i]; id = si[i]; }, M, a.k, keys, ids,
                                                     red, selk, seli);
                for (int r = tid; r < a.k; r += kSelThreads)
                    write_result<METRIC>(a.out64, a.out32, a.out_ids, (int64_t)q * a.k + r, selk[r], seli[r], a.id_base);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// host object
// ------------------------------------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (function, size) instead of on every launch
// Row stride of the group-maxima arrays, in floats: a multiple of 32 so that every query's row starts on a 128-byte line
// and a wave's 16-block chunk (32 floats per query) is exactly one line (with rows at 16-byte granularity every flush
// straddled two lines: partial-line writes all the way to HBM).  HIPRAG_GMAX_ALIGN overrides (experiments).
static int64_t gmax_stride(int64_t nblocks)
{
    static const int64_t al = [] { const char* e = getenv("HIPRAG_GMAX_ALIGN"); return e ? std::max<int64_t>(4, atoll(e)) : (int64_t)32; }();
    return ((2 * nblocks + al - 1) / al) * al;
}

static int32_t ensure_lds(const void* fn, size_t bytes)
{
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> done;
    std::lock_guard<std::mutex> g(mu);
    auto it = done.find(fn);
    if (it != done.end() && it->second >= bytes) return HIPRAG_OK;
    HR_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[fn] = bytes;
    return HIPRAG_OK;
}

struct DenseIndex {
    std::mutex mu;
    int device = 0;
    int d = 0, P = 0, metric = 0;
    int64_t ntotal = 0, cap_blocks = 0, id_base = 0;
    int n_cu = 256;
    int scan_cus = 256;       // workgroups of a scan launch (one per CU); HIPRAG_SCAN_SPARE_CUS leaves some CUs to the tails
    int scan_mode = 3;        // HIPRAG_SCAN_MODE: f32 = 0 (exact fp32 MFMA), split = 1 (bf16 hi/lo of the fp32 rows, 32 q/pass),
                              // q64 = 2 (hi/lo rows x hi-only queries, 64 q/pass), bf16 = 3 (bf16 filter copy, 64 q/pass; default)
    DevBuf xb, xh, norms, scalars;  // xh: bf16 filter copy; scalars: [0] max |x|^2 bits (u32), [1] max |x - bf16(x)|^2 bits,
                                // [2..3] fallback counter (u64), [4..5] round-B counter
    // search workspace
    struct Workspace { DevBuf gmax, qf, ck, ci, flags, ek, ei, fin, rb; int split = 0, chunk = kChunk, waves = 8; int k = 0, q = 0; int64_t blocks = 0; int ev_idx = -1; };
    static constexpr int kSlots = 8;   // passes in flight: the scan of pass i+1 runs beside the tails of passes i, i-1, ...
    Workspace ws[kSlots];
    DevBuf qbuf, o64, o32, oid;
    int launch_q = 256;       // queries one begin/finish pair takes (a multiple of the pass size): update_launch_q
    int launch_env = 0;       // HIPRAG_LAUNCH_QUERIES (0 = size launches by the index)
    // stats
    int64_t passes = 0, queries = 0, launches = 0;
    // timing: a ring of event pairs around the scan kernel, averaged by get_stats (no sync inside the search path)
    static constexpr int kEvRing = 512;
    bool timing = false;
    std::vector<hipEvent_t> evs;   // 2*kEvRing once timing was enabled
    DevBuf dbg_stamps;             // HIPRAG_DEBUG_GAPS: [kEvRing][8] tail-kernel stamps
    bool dbg_on = false;
    DevBuf stamps;                 // [kEvRing][n_cu * 8 waves][2] in-kernel wall-clock ticks of the same launches
    int wall_khz = 100000;
    int64_t ev_count = 0;          // launches since timing was (re)enabled
    int ev_every = 1;
    std::vector<char> ev_set;      // [kEvRing] whether the launch in that ring slot was bracketed by events

    int64_t nblocks() const { return (ntotal + kRowsPerBlock - 1) / kRowsPerBlock; }
    unsigned* max_norm2_bits() { return scalars.as<unsigned>(); }
    unsigned* max_dx2_bits() { return scalars.as<unsigned>() + 1; }
    unsigned long long* fallback_counter() { return reinterpret_cast<unsigned long long*>(scalars.as<unsigned>() + 2); }
    unsigned long long* roundb_counter() { return reinterpret_cast<unsigned long long*>(scalars.as<unsigned>() + 4); }

    // Ordering of `add` against everything else: add_dev enqueues its re-tiling kernels on the CALLER's stream, which may
    // be a non-blocking stream the null stream does not wait for.  `add_ev` marks the last add; grow / save / reconstruct
    // (null-stream copies) wait for it on the host, a search on another stream waits for it on the device.
    hipEvent_t add_ev = nullptr;
    bool add_pending = false;

    int32_t wait_adds_host()
    {
        if (add_pending) { HR_CHECK_HIP(hipEventSynchronize(add_ev)); add_pending = false; }
        return HIPRAG_OK;
    }
    int32_t wait_adds_stream(hipStream_t st)
    {
        if (add_pending) HR_CHECK_HIP(hipStreamWaitEvent(st, add_ev, 0));
        return HIPRAG_OK;
    }

    ~DenseIndex()
    {
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        if (add_ev) (void)hipEventDestroy(add_ev);
    }

    int32_t init()
    {
        HR_CHECK_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        HR_CHECK_HIP(hipGetDeviceProperties(&prop, device));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        const char* sp = getenv("HIPRAG_SCAN_SPARE_CUS");
        scan_cus = std::max(1, n_cu - (sp ? atoi(sp) : 0));
        const char* ms = getenv("HIPRAG_SCAN_MODE");
        if (ms) scan_mode = ms[0] == 'f' ? 0 : ms[0] == 's' ? 1 : ms[0] == 'q' ? 2 : 3;
        const char* lq = getenv("HIPRAG_LAUNCH_QUERIES");
        launch_env = lq ? atoi(lq) : 0;
        if (const char* sw = getenv("HIPRAG_SCAN_WAVES_SMALL")) scan_waves_small = atoi(sw);
        if (const char* sb = getenv("HIPRAG_SCAN_WAVES_SMALL_BLOCKS")) scan_waves_small_blocks = std::max(1, atoi(sb));
        if (const char* sc = getenv("HIPRAG_SCAN_CHUNK")) scan_chunk = atoi(sc) == 8 ? 8 : 16;
        if (const char* se = getenv("HIPRAG_SELECT")) select_by_threshold = se[0] != 'w';
        update_launch_q();
        int32_t rc = scalars.reserve(64);
        if (rc) return rc;
        HR_CHECK_HIP(hipMemset(scalars.p, 0, 64));
        return HIPRAG_OK;
    }

    int32_t grow(int64_t need_blocks)
    {
        if (need_blocks <= cap_blocks) return HIPRAG_OK;
        {   // the copies below run on the null stream: rows a previous add is still writing must have landed
            const int32_t wrc = wait_adds_host();
            if (wrc) return wrc;
        }
        int64_t nc = cap_blocks == 0 ? need_blocks : std::max(need_blocks, cap_blocks + cap_blocks / 2);
        size_t xbytes = (size_t)nc * P * kPieceFloats * sizeof(float);
        size_t nbytes = (size_t)nc * kRowsPerBlock * sizeof(float);
        size_t hbytes = xbytes / 2;   // bf16 filter copy
        void* nx = nullptr;
        void* nn = nullptr;
        void* nh = nullptr;
        HR_CHECK_HIP(hipMalloc(&nx, xbytes));
        hipError_t e = hipMalloc(&nn, nbytes);
        if (e == hipSuccess) e = hipMalloc(&nh, hbytes);
        if (e != hipSuccess) { (void)hipFree(nx); if (nn) (void)hipFree(nn); HR_CHECK_HIP(e); }
        HR_CHECK_HIP(hipMemset(nx, 0, xbytes));
        HR_CHECK_HIP(hipMemset(nn, 0, nbytes));
        HR_CHECK_HIP(hipMemset(nh, 0, hbytes));
        if (xb.p) {
            HR_CHECK_HIP(hipMemcpy(nx, xb.p, (size_t)cap_blocks * P * kPieceFloats * sizeof(float), hipMemcpyDeviceToDevice));
            HR_CHECK_HIP(hipMemcpy(nn, norms.p, (size_t)cap_blocks * kRowsPerBlock * sizeof(float), hipMemcpyDeviceToDevice));
            HR_CHECK_HIP(hipMemcpy(nh, xh.p, (size_t)cap_blocks * P * kPieceFloats * sizeof(float) / 2, hipMemcpyDeviceToDevice));
        }
        xb.release();
        norms.release();
        xh.release();
        xb.p = nx; xb.bytes = xbytes;
        norms.p = nn; norms.bytes = nbytes;
        xh.p = nh; xh.bytes = hbytes;
        cap_blocks = nc;
        return HIPRAG_OK;
    }

    // x_dev: [n,d] row-major on this device
    int32_t add_dev(const float* x_dev, int64_t n, hipStream_t st)
    {
        if (n == 0) return HIPRAG_OK;
        int32_t rc = grow((ntotal + n + kRowsPerBlock - 1) / kRowsPerBlock);
        if (rc) return rc;
        const int64_t blk0 = ntotal / kRowsPerBlock;
        const int64_t nblk = (ntotal + n - 1) / kRowsPerBlock - blk0 + 1;
        const int64_t threads = nblk * P * kPieceVec4;
        hipLaunchKernelGGL(retile_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, x_dev, ntotal, n, d, P,
                           xb.as<float4>());
        hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 4096)), dim3(256), 0, st, x_dev,
                           ntotal, n, d, norms.as<float>(), max_norm2_bits(), max_dx2_bits());
        hipLaunchKernelGGL(retile_bf16_kernel, dim3((unsigned)((nblk * (P / 2) * 64 + 255) / 256)), dim3(256), 0, st, x_dev, ntotal, n,
                           d, P / 2, xh.as<bf16x8_t>());
        HR_CHECK_HIP(hipGetLastError());
        if (!add_ev) HR_CHECK_HIP(hipEventCreateWithFlags(&add_ev, hipEventDisableTiming));
        HR_CHECK_HIP(hipEventRecord(add_ev, st));
        add_pending = true;
        ntotal += n;
        update_launch_q();
        return HIPRAG_OK;
    }

    int32_t add_host(const float* x, int64_t n)
    {
        if (n == 0) return HIPRAG_OK;
        const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(256ll << 20) / ((int64_t)d * 4));
        DevBuf stage;
        int32_t rc = stage.reserve((size_t)std::min(chunk_rows, n) * d * sizeof(float));
        if (rc) return rc;
        rc = grow((ntotal + n + kRowsPerBlock - 1) / kRowsPerBlock);
        if (rc) return rc;
        for (int64_t o = 0; o < n; o += chunk_rows) {
            const int64_t m = std::min(chunk_rows, n - o);
            HR_CHECK_HIP(hipMemcpy(stage.p, x + o * d, (size_t)m * d * sizeof(float), hipMemcpyHostToDevice));
            rc = add_dev(stage.as<float>(), m, nullptr);
            if (rc) return rc;
            HR_CHECK_HIP(hipStreamSynchronize(nullptr));
        }
        return HIPRAG_OK;
    }

    int pass_queries() const { return scan_mode >= 2 ? 64 : 32; }
    // Operand mode of a launch for top-k: the 64-query tiles re-score K' = min(63, k + max(16, k)) groups (measured on 1M
    // unit vectors: k + 12 certifies every query at k = 10 but only 99 % at k = 20; the 64-entry wave lists cap K' at 63,
    // where about half of the k = 50 queries -- the reference's retrieval depth, page_retriever.py:92 -- fail the
    // certificate and are settled by round B of the finish instead).  k > 57 uses the 32-query split scan.
    // Small shards (an 8-GPU row split of 1M rows leaves 125 k per GPU): with 8 waves per workgroup a wave streams two or
    // three 32-row blocks per pass; 4-wave workgroups stream twice as many each and leave half of every SIMD's registers to
    // the tail kernels of earlier steps.  Measured (1024 queries per launch, pipelined): 125 k rows 945 -> 902 us per step,
    // 250 k 1640 -> 1590, 500 k 2758 -> 2857 (slower), 1M equal -- so below 5 blocks per wave of the 8-wave partition in
    // round 1.  With 16-block flushes the 4-wave form writes whole lines at 500 k rows (16 blocks per wave and pass) where
    // the 8-wave form writes half lines: 345 -> 353 k queries/s, so the rule now reaches 9 blocks per wave (~590 k rows).
    bool select_by_threshold = true;   // HIPRAG_SELECT=wave: the per-wave selectors (select_wave_kernel) instead
    int scan_chunk = 16;               // HIPRAG_SCAN_CHUNK: blocks per flush of the multi-pass bf16 scan (16 = whole 128-byte lines)
    int scan_waves_small = 1;          // HIPRAG_SCAN_WAVES_SMALL=0 keeps 8 waves everywhere
    int scan_waves_small_blocks = 9;   // HIPRAG_SCAN_WAVES_SMALL_BLOCKS
    int mode_for(int k) const { return (scan_mode >= 2 && k > kMaxK64) ? 1 : scan_mode; }
    int pass_queries_for(int k) const { return mode_for(k) >= 2 ? 64 : 32; }
    // groups re-scored per query: the hi-only query tiles of the 64-query mode widen eps to 2^-9 |q||x|, so keep more
    int kprime(int k) const { return mode_for(k) >= 2 ? std::min(63, k + slack_groups64(k)) : k + kSlackGroups; }

    // Passes per launch.  A launch chained behind its predecessor pays ~45-60 us of dispatch bubble and the tails of a
    // launch are a fixed cost too, so launches are sized to last about as long as four passes over a 1M x 1024 index
    // (2.6 ms) whatever the index size: 4 passes there, 8 at half a million rows, 16 (the cap) at an 8-way shard of it --
    // where a 4-pass launch would spend a quarter of its time outside the scan.  HIPRAG_LAUNCH_QUERIES fixes the size.
    void update_launch_q()
    {
        const int pq = pass_queries();
        if (scan_mode == 0) { launch_q = pq; return; }   // the exact-fp32 scan (verification mode) runs one pass per launch
        if (launch_env > 0) { launch_q = std::max(pq, std::min(kMaxQ, launch_env / pq * pq)); return; }
        const double pass_bytes = (double)std::max<int64_t>(nblocks(), 1) * P * (scan_mode == 3 ? 512.0 : 1024.0);
        const int passes = (int)std::lround(4.0 * 4.096e9 / pass_bytes);
        launch_q = std::max(4, std::min(16, passes)) * pq;
    }

    // Workspace of one slot for (up to launch_q queries, k), allocated on first use: an unused slot costs nothing.
    int32_t reserve_slot(int slot, int k)
    {
        Workspace& w = ws[slot];
        const int64_t nb = std::max<int64_t>(nblocks(), 1);
        if (k <= w.k && nb <= w.blocks && launch_q <= w.q) return HIPRAG_OK;
        const int kk = std::max(k, w.k);
        const int64_t nbb = std::max(nb, w.blocks);
        const int64_t gstride = gmax_stride(nbb);
        const int64_t nchunks = (gstride + kSelChunk - 1) / kSelChunk;
        const int K1 = kprime(kk) + 1;
        const int64_t nslices = (nbb * kRowsPerBlock + kExRows - 1) / kExRows;
        const int ekk = std::min(kk, kExRows);
        const int64_t nlists = std::max(nchunks, ((gstride + kSelPerWave - 1) / kSelPerWave + 3) / 4 * 4);
        const size_t Q = (size_t)std::max(launch_q, w.q);
        int32_t rc;
        if ((rc = w.gmax.reserve(2 * Q * gstride * sizeof(float)))) return rc;  // first | second
        if ((rc = w.qf.reserve((size_t)P * kPieceVec4 * sizeof(float4)))) return rc;
        if ((rc = w.ck.reserve(Q * nlists * K1 * sizeof(u64)))) return rc;
        if ((rc = w.ci.reserve(Q * nlists * K1 * sizeof(i64)))) return rc;
        if ((rc = w.flags.reserve(2 * Q * sizeof(int)))) return rc;  // flags[Q] + arrivals[Q]
        // sel[Q][64] u64 | cand_k[Q][256] u64 | cand_i[Q][256] i64 | qn2[Q][2] f64 | sec[Q][64] f32
        if ((rc = w.fin.reserve(Q * (64 + 2 * kCandPerQuery + 2 + 32) * 8))) return rc;
        // round B: tau[Q] f32 | count[Q] i32 | slots[Q][256] u32 | sec[Q][256] f32 | keys[Q][1024] u64 | ids[Q][1024] i64
        if ((rc = w.rb.reserve(Q * (8 + kRoundBGroups * 8 + (size_t)kRoundBGroups * 4 * 16)))) return rc;
        if ((rc = w.ek.reserve(Q * nslices * ekk * sizeof(u64)))) return rc;
        if ((rc = w.ei.reserve(Q * nslices * ekk * sizeof(i64)))) return rc;
        w.k = kk;
        w.blocks = nbb;
        w.q = (int)Q;
        return HIPRAG_OK;
    }

    // phase 1 of a pass (<= kMaxQ queries): query fragments + the scan, into workspace `slot`
    template <int METRIC>
    int32_t scan_pass(const float* q_dev, int nq, int k, int slot, hipStream_t st)
    {
        const int mode = mode_for(k);
        Workspace& w = ws[slot];
        const int64_t nb = nblocks();
        ScanArgs sa;
        sa.xb = xb.as<float4>(); sa.qf = w.qf.as<float4>(); sa.q = q_dev; sa.norms = norms.as<float>(); sa.gmax = w.gmax.as<float>(); sa.xh = xh.p;
        sa.gmax2 = sa.gmax + (size_t)w.q * gmax_stride(w.blocks);
        sa.gstride = gmax_stride(w.blocks); sa.nblocks = nb; sa.ntotal = ntotal; sa.nq = nq; sa.d = d; sa.P = P;
        // operand path: "split" = bf16 hi/lo split MFMAs (default), "f32" = exact-fp32 MFMAs
        const bool split = mode != 0;
        w.split = mode;
        w.chunk = kChunk;
        w.waves = 8;
        const size_t scan_lds = (size_t)P * 1024;  // the query tile; 32 KiB of the CU's LDS stay free for tail kernels
        const int ev = (int)(ev_count % kEvRing);
        // HIP events cost two barrier packets per launch on the scan's stream; hipidx_enable_timing(h, n) brackets every n-th
        // launch only (the in-kernel stamps cover every launch either way)
        const bool use_ev = timing && ev_count % ev_every == 0;
        if (timing) HR_CHECK_HIP(hipMemsetAsync(stamps.as<unsigned long long>() + (size_t)ev * scan_cus * kMaxScanWaves * 2, 0, (size_t)scan_cus * kMaxScanWaves * 16, st));
        sa.stamps = timing ? stamps.as<unsigned long long>() + (size_t)ev * scan_cus * kMaxScanWaves * 2 : nullptr;
        if (mode == 3) {
            // ring depth: 16 pieces where that divides the pieces of a block (d_pad / 16), else 8 (32 spills in the multi-pass form)
            const int P2 = P / 2;
            const bool one_pass = nq <= 64;
            void (*scan)(ScanArgs);
            int nw = 8;   // a third wave per SIMD (12 per workgroup) does not fit 168 registers: 83 spills
            if (P2 % 16 == 0) scan = one_pass ? scan_bf16_kernel<METRIC, 8, 16, false> : scan_bf16_kernel<METRIC, 8, 16, true>;
            else scan = one_pass ? scan_bf16_kernel<METRIC, 8, 8, false> : scan_bf16_kernel<METRIC, 8, 8, true>;
            int ch = 8;
            if (P2 % 16 == 0 && scan_chunk == 16) { scan = one_pass ? scan_bf16_kernel<METRIC, 8, 16, false, 16> : scan_bf16_kernel<METRIC, 8, 16, true, 16>; ch = 16; }
            if (scan_waves_small > 0 && P2 % 16 == 0 && nb < (int64_t)scan_cus * 8 * scan_waves_small_blocks) {
                nw = 4;   // small shard (see scan_waves_small)
                if (scan_chunk == 16) scan = one_pass ? scan_bf16_kernel<METRIC, 4, 16, false, 16> : scan_bf16_kernel<METRIC, 4, 16, true, 16>;
                else { scan = one_pass ? scan_bf16_kernel<METRIC, 4, 16, false> : scan_bf16_kernel<METRIC, 4, 16, true>; ch = 8; }
            }
            w.waves = nw;
            w.chunk = ch;
            { int32_t lrc = ensure_lds(reinterpret_cast<const void*>(scan), scan_lds); if (lrc) return lrc; }
            if (use_ev) HR_CHECK_HIP(hipEventRecord(evs[2 * ev], st));
            if (nb > 0) hipLaunchKernelGGL(scan, dim3(scan_cus), dim3(nw * 64), scan_lds, st, sa);
        } else if (split) {
            void (*scan)(ScanArgs) = scan_split_kernel<METRIC, 8>;
            const bool one_pass = nq <= pass_queries_for(k);
            if (one_pass) scan = scan_split_kernel<METRIC, 8, 16, 1, kChunk, false>;
            if (mode == 2) {                                                             // 64 queries, hi-only query tiles
                if (scan_chunk == 16) {   // whole 128-byte lines per flush (see flush_full)
                    scan = one_pass ? scan_split_kernel<METRIC, 8, 16, 2, 16, false> : scan_split_kernel<METRIC, 8, 16, 2, 16>; w.chunk = 16;
                } else {
                    scan = one_pass ? scan_split_kernel<METRIC, 8, 16, 2, 8, false> : scan_split_kernel<METRIC, 8, 16, 2, 8>; w.chunk = 8;
                }
            }
            { int32_t lrc = ensure_lds(reinterpret_cast<const void*>(scan), scan_lds); if (lrc) return lrc; }
            if (use_ev) HR_CHECK_HIP(hipEventRecord(evs[2 * ev], st));
            if (nb > 0) hipLaunchKernelGGL(scan, dim3(scan_cus), dim3(8 * 64), scan_lds, st, sa);
        } else {
            int NW = 8;
            void (*scan)(ScanArgs) = scan_kernel<METRIC, 8>;
            { int32_t lrc = ensure_lds(reinterpret_cast<const void*>(scan), scan_lds); if (lrc) return lrc; }
            hipLaunchKernelGGL(qprep_kernel, dim3((unsigned)((P * kPieceVec4 + 255) / 256)), dim3(256), 0, st, q_dev, nq, d, P,
                               w.qf.as<float4>());
            if (use_ev) HR_CHECK_HIP(hipEventRecord(evs[2 * ev], st));
            if (nb > 0) hipLaunchKernelGGL(scan, dim3(scan_cus), dim3(NW * 64), scan_lds, st, sa);
        }
        if (use_ev) HR_CHECK_HIP(hipEventRecord(evs[2 * ev + 1], st));
        w.ev_idx = timing ? ev : -1;
        if (timing) { ev_set[ev] = use_ev; ++ev_count; }
        HR_CHECK_HIP(hipGetLastError());
        passes += (nq + pass_queries_for(k) - 1) / pass_queries_for(k);
        ++launches;
        queries += nq;
        return HIPRAG_OK;
    }

    // phase 2: group selection, fp64 re-score, final top-k + certificate, exhaustive fallback; reads workspace `slot`
    template <int METRIC>
    int32_t finish_pass(const float* q_dev, int nq, int k, int slot, double* o64p, float* o32p, int64_t* oidp, hipStream_t st)
    {
        Workspace& w = ws[slot];
        const int64_t nb = nblocks();
        const int64_t gstride = gmax_stride(w.blocks);
        const int64_t ngroups = nb * 2;
        const int Kp = kprime(k), K1 = Kp + 1;
        const int64_t nchunks = std::max<int64_t>(1, (ngroups + kSelChunk - 1) / kSelChunk);
        int* flags = w.flags.as<int>();
        int* arrivals = flags + w.q;

        FinishArgs fa;
        fa.xb = xb.as<float4>(); fa.q = q_dev; fa.ck = w.ck.as<u64>(); fa.ci = w.ci.as<i64>();
        fa.max_norm2_bits = max_norm2_bits(); fa.out64 = o64p; fa.out32 = o32p; fa.out_ids = oidp;
        fa.flags = flags; fa.arrivals = arrivals; fa.fallback_counter = fallback_counter();
        fa.dbg = (dbg_on && w.ev_idx >= 0) ? dbg_stamps.as<unsigned long long>() + (size_t)w.ev_idx * 8 : nullptr;
        if (fa.dbg) HR_CHECK_HIP(hipMemsetAsync(fa.dbg, 0xFF, 64, st));
        fa.ntotal = ntotal; fa.id_base = id_base; fa.d = d; fa.P = P; fa.k = k; fa.Kp = Kp; fa.split = w.split; fa.chunk = w.chunk;
        fa.nblocks = nb; fa.bpw = scan_blocks_per_wave(nb, (int64_t)scan_cus * w.waves);
        const int64_t sel_waves = std::max<int64_t>(1, (ngroups + kSelPerWave - 1) / kSelPerWave);
        const int64_t sel_slices = (sel_waves + 3) / 4;
        const int64_t wave_cand = sel_slices * 4 * K1;
        if (K1 <= 64 && wave_cand <= (int64_t)kFinWaves * 64 * 16) {
            if (!select_by_threshold)   // one wave filters kSelPerWave group maxima against its running K1-th best
                hipLaunchKernelGGL(select_wave_kernel<false>, dim3((unsigned)sel_slices, nq), dim3(256), 0, st,
                                   (const float*)w.gmax.as<float>(), (i64)gstride, (i64)ngroups, K1, w.ck.as<u64>(), w.ci.as<i64>(), fa.dbg);
            fa.ncand = wave_cand;
            u64* fin_base = w.fin.as<u64>();
            fa.sel = fin_base;
            fa.cand_k = fin_base + (size_t)w.q * 64;
            fa.cand_i = reinterpret_cast<i64*>(fa.cand_k + (size_t)w.q * kCandPerQuery);
            fa.qn2 = reinterpret_cast<double*>(fa.cand_i + (size_t)w.q * kCandPerQuery);
            fa.sec = reinterpret_cast<float*>(fa.qn2 + 2 * (size_t)w.q);
            fa.gmax2 = w.gmax.as<float>() + (size_t)w.q * gstride;
            fa.gstride = gstride;
            float* rb_tau = w.rb.as<float>();
            int* rb_count = reinterpret_cast<int*>(rb_tau + w.q);
            u32* rb_slots = reinterpret_cast<u32*>(rb_count + w.q);
            float* rb_sec = reinterpret_cast<float*>(rb_slots + (size_t)w.q * kRoundBGroups);
            u64* rb_k = reinterpret_cast<u64*>(rb_sec + (size_t)w.q * kRoundBGroups);
            i64* rb_i = reinterpret_cast<i64*>(rb_k + (size_t)w.q * kRoundBGroups * 4);
            fa.tau = rb_tau; fa.rb_count = rb_count;
            if (select_by_threshold) hipLaunchKernelGGL(fin_select_kernel, dim3(nq), dim3(1024), 0, st, fa, (const float*)w.gmax.as<float>(), (i64)gstride, (i64)ngroups);
            else if (wave_cand <= 64) hipLaunchKernelGGL((fin_merge_kernel<1, 1>), dim3(nq), dim3(64), 0, st, fa);
            else if (wave_cand <= 128) hipLaunchKernelGGL((fin_merge_kernel<2, 1>), dim3(nq), dim3(64), 0, st, fa);
            else if (wave_cand <= 256) hipLaunchKernelGGL((fin_merge_kernel<4, 1>), dim3(nq), dim3(64), 0, st, fa);
            else if (wave_cand <= 512) hipLaunchKernelGGL((fin_merge_kernel<8, 1>), dim3(nq), dim3(64), 0, st, fa);
            else if (wave_cand <= 2048) hipLaunchKernelGGL((fin_merge_kernel<8, 4>), dim3(nq), dim3(256), 0, st, fa);
            else hipLaunchKernelGGL((fin_merge_kernel<16, 16>), dim3(nq), dim3(1024), 0, st, fa);
            hipLaunchKernelGGL(fin_rescore_kernel<METRIC>, dim3((Kp + 3) / 4, nq), dim3(256), (size_t)P * 8 * sizeof(float), st, fa);
            hipLaunchKernelGGL(fin_final_kernel<METRIC>, dim3(nq), dim3(64), 0, st, fa);
            RoundBArgs rb;
            rb.xb = fa.xb; rb.q = q_dev; rb.gmax = w.gmax.as<float>(); rb.gstride = gstride; rb.ngroups = ngroups;
            rb.flags = flags; rb.tau = rb_tau; rb.count = rb_count; rb.slots = rb_slots; rb.bk = rb_k; rb.bi = rb_i;
            rb.out64 = o64p; rb.out32 = o32p; rb.out_ids = oidp; rb.fallback_counter = fallback_counter();
            rb.roundb_counter = roundb_counter(); rb.ntotal = ntotal; rb.id_base = id_base; rb.bpw = fa.bpw; rb.nblocks = nb;
            rb.d = d; rb.P = P; rb.k = k; rb.chunk = w.chunk; rb.sel = fa.sel; rb.Kp = Kp;
            rb.bsec = rb_sec; rb.gmax2 = fa.gmax2; rb.qn2 = fa.qn2; rb.max_norm2_bits = fa.max_norm2_bits; rb.split = fa.split;
            hipLaunchKernelGGL(roundb_collect_kernel, dim3((unsigned)std::max<int64_t>(1, (ngroups + 4095) / 4096), nq), dim3(256), 0, st, rb);
            hipLaunchKernelGGL(roundb_rescore_kernel<METRIC>, dim3(8, nq), dim3(256), (size_t)P * 8 * sizeof(float), st, rb);
            hipLaunchKernelGGL(roundb_final_kernel<METRIC>, dim3(nq), dim3(64), 0, st, rb);
        } else if ((int64_t)Kp * 16 > kSelChunk) {
            hipLaunchKernelGGL(flag_all_kernel, dim3((kMaxQ + 1023) / 1024), dim3(1024), 0, st, flags, arrivals, fallback_counter(), nq);
        } else {
            hipLaunchKernelGGL(select_f32_kernel<false>, dim3((unsigned)nchunks, nq), dim3(kSelThreads), 0, st,
                               (const float*)w.gmax.as<float>(), (i64)gstride, (i64)ngroups, K1, w.ck.as<u64>(), w.ci.as<i64>());
            fa.ncand = nchunks * K1;
            const size_t fin_lds = (size_t)kSelChunk * 16 + (size_t)K1 * 16 + 2 * (kSelThreads / 64) * sizeof(KeyId) +
                                   (2 * (kSelThreads / 64) + 1) * sizeof(double) + (size_t)P * 8 * sizeof(float);
            auto fin = finish_kernel<METRIC>;
            { int32_t lrc = ensure_lds(reinterpret_cast<const void*>(fin), fin_lds); if (lrc) return lrc; }
            hipLaunchKernelGGL(fin, dim3(nq), dim3(kSelThreads), fin_lds, st, fa);
        }

        ExArgs ea;
        ea.xb = xb.as<float4>(); ea.q = q_dev; ea.flags = flags; ea.arrivals = arrivals;
        ea.ek = w.ek.as<u64>(); ea.ei = w.ei.as<i64>();
        ea.out64 = o64p; ea.out32 = o32p; ea.out_ids = oidp; ea.ntotal = ntotal; ea.id_base = id_base;
        ea.d = d; ea.P = P; ea.k = k; ea.kk = std::min(k, kExRows); ea.nq = nq;
        ea.nslices = (int)std::max<int64_t>(1, (ntotal + kExRows - 1) / kExRows);
        ea.dbg = fa.dbg;
        const size_t ex_lds = (size_t)kExRows * 16 + (size_t)k * 16 + 2 * (kSelThreads / 64) * sizeof(KeyId) +
                              (size_t)P * 8 * sizeof(float) + 16;
        auto exk = exhaustive_kernel<METRIC>;
        { int32_t lrc = ensure_lds(reinterpret_cast<const void*>(exk), ex_lds); if (lrc) return lrc; }
        hipLaunchKernelGGL(exk, dim3(std::min(ea.nslices, n_cu)), dim3(kSelThreads), ex_lds, st, ea);
        HR_CHECK_HIP(hipGetLastError());
        return HIPRAG_OK;
    }

    int32_t prepare(int k, int slot)
    {
        int32_t rc = reserve_slot(slot, k);
        if (rc) return rc;
        if (timing && evs.empty()) {
            evs.resize(2 * kEvRing);
            ev_set.assign(kEvRing, 0);
            for (auto& e : evs) HR_CHECK_HIP(hipEventCreate(&e));
            int32_t src = stamps.reserve((size_t)kEvRing * scan_cus * kMaxScanWaves * 2 * sizeof(unsigned long long));
            if (src) return src;
            dbg_on = getenv("HIPRAG_DEBUG_GAPS") != nullptr;
            if (dbg_on) { int32_t drc = dbg_stamps.reserve((size_t)kEvRing * 8 * 8); if (drc) return drc; }
            (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, device);
            if (wall_khz <= 0) wall_khz = 100000;
        }
        return HIPRAG_OK;
    }

    int32_t begin_dev(const float* q_dev, int nq, int k, int slot, hipStream_t st)
    {
        if (add_pending) {   // rows of the last add may still be in flight on another stream
            if (hipEventQuery(add_ev) == hipSuccess) add_pending = false;
            else { const int32_t wrc = wait_adds_stream(st); if (wrc) return wrc; }
        }
        return metric == HIPRAG_METRIC_IP ? scan_pass<HIPRAG_METRIC_IP>(q_dev, nq, k, slot, st)
                                          : scan_pass<HIPRAG_METRIC_L2>(q_dev, nq, k, slot, st);
    }

    int32_t finish_dev(const float* q_dev, int nq, int k, int slot, double* o64p, float* o32p, int64_t* oidp, hipStream_t st)
    {
        return metric == HIPRAG_METRIC_IP ? finish_pass<HIPRAG_METRIC_IP>(q_dev, nq, k, slot, o64p, o32p, oidp, st)
                                          : finish_pass<HIPRAG_METRIC_L2>(q_dev, nq, k, slot, o64p, o32p, oidp, st);
    }

    int32_t search_dev(const float* q_dev, int nq, int k, double* o64p, float* o32p, int64_t* oidp, hipStream_t st)
    {
        int32_t rc = prepare(k, 0);
        if (rc) return rc;
        for (int o = 0; o < nq; o += launch_q) {
            const int m = std::min(launch_q, nq - o);
            const float* qo = q_dev + (int64_t)o * d;
            if ((rc = begin_dev(qo, m, k, 0, st))) return rc;
            if ((rc = finish_dev(qo, m, k, 0, o64p + (int64_t)o * k, o32p ? o32p + (int64_t)o * k : nullptr,
                                 oidp + (int64_t)o * k, st)))
                return rc;
        }
        return HIPRAG_OK;
    }
};

Registry<DenseIndex>& reg()
{
    static Registry<DenseIndex> r;
    return r;
}

#define GET_INDEX(h)                                                       \
    std::shared_ptr<DenseIndex> ix = reg().get(h);                         \
    if (!ix) { set_error("unknown dense index handle %llu", (unsigned long long)(h)); return HIPRAG_E_HANDLE; } \
    std::lock_guard<std::mutex> guard(ix->mu);                             \
    HR_CHECK_HIP(hipSetDevice(ix->device))

}  // namespace

size_t clear_dense_registry() { return reg().clear(); }
}  // namespace hiprag

using namespace hiprag;

extern "C" {

int32_t hipidx_create(int32_t d, int32_t metric, int32_t device, uint64_t* out_handle)
{
    HR_REQUIRE(out_handle, "out_handle is null");
    HR_REQUIRE(d > 0, "d must be positive (got %d)", d);
    HR_REQUIRE(metric == HIPRAG_METRIC_IP || metric == HIPRAG_METRIC_L2, "unknown metric %d", metric);
    const int P = ((d + 127) / 128) * 16;
    if ((size_t)P * 1024 > 128 * 1024) {
        set_error("d=%d needs %d KiB of LDS for the query tile; the scan supports d <= 1024 (128 KiB, leaving 32 KiB per CU "
                  "for the tail kernels of earlier passes)", d, P);
        return HIPRAG_E_UNSUPPORTED;
    }
    auto ix = std::make_shared<DenseIndex>();
    ix->device = device;
    ix->d = d;
    ix->P = P;
    ix->metric = metric;
    int32_t rc = ix->init();
    if (rc) return rc;
    *out_handle = reg().put(ix);
    return HIPRAG_OK;
}

int32_t hipidx_destroy(uint64_t h)
{
    std::shared_ptr<DenseIndex> ix = reg().get(h);
    if (!ix) { set_error("unknown dense index handle"); return HIPRAG_E_HANDLE; }
    {
        std::lock_guard<std::mutex> guard(ix->mu);
        (void)hipSetDevice(ix->device);
        (void)hipDeviceSynchronize();
    }
    reg().erase(h);
    return HIPRAG_OK;
}

int32_t hipidx_add(uint64_t h, const float* x_host, int64_t n)
{
    GET_INDEX(h);
    HR_REQUIRE(n >= 0 && (x_host || n == 0), "bad add arguments");
    return ix->add_host(x_host, n);
}

int32_t hipidx_add_dev(uint64_t h, const float* x_dev, int64_t n, void* stream)
{
    GET_INDEX(h);
    HR_REQUIRE(n >= 0 && (x_dev || n == 0), "bad add arguments");
    return ix->add_dev(x_dev, n, (hipStream_t)stream);
}

int32_t hipidx_ntotal(uint64_t h, int64_t* out_n)
{
    GET_INDEX(h);
    HR_REQUIRE(out_n, "null out");
    *out_n = ix->ntotal;
    return HIPRAG_OK;
}

int32_t hipidx_dim(uint64_t h, int32_t* out_d)
{
    GET_INDEX(h);
    HR_REQUIRE(out_d, "null out");
    *out_d = ix->d;
    return HIPRAG_OK;
}

int32_t hipidx_metric(uint64_t h, int32_t* out_metric)
{
    GET_INDEX(h);
    HR_REQUIRE(out_metric, "null out");
    *out_metric = ix->metric;
    return HIPRAG_OK;
}

int32_t hipidx_set_id_base(uint64_t h, int64_t id_base)
{
    GET_INDEX(h);
    ix->id_base = id_base;
    return HIPRAG_OK;
}

int32_t hipidx_pass_queries(uint64_t h, int32_t* out_n)
{
    GET_INDEX(h);
    HR_REQUIRE(out_n, "null out");
    *out_n = ix->pass_queries();
    return HIPRAG_OK;
}

int32_t hipidx_launch_queries(uint64_t h, int32_t* out_n)
{
    GET_INDEX(h);
    HR_REQUIRE(out_n, "null out");
    *out_n = ix->launch_q;
    return HIPRAG_OK;
}

int32_t hipidx_set_spare_cus(uint64_t h, int32_t n)
{
    GET_INDEX(h);
    HR_REQUIRE(n >= 0 && n < ix->n_cu, "spare CUs must be in 0..%d", ix->n_cu - 1);
    HR_CHECK_HIP(hipDeviceSynchronize());   // the finish kernels of launches in flight decode slots with the old partition
    ix->scan_cus = ix->n_cu - n;
    ix->stamps.release();                   // sized by the scan grid
    for (hipEvent_t e : ix->evs) (void)hipEventDestroy(e);
    ix->evs.clear();                        // re-created (with the stamp buffer) by the next enable_timing + search
    ix->timing = false;
    return HIPRAG_OK;
}

int32_t hipidx_reserve_rows(uint64_t h, int64_t n_rows)
{
    GET_INDEX(h);
    HR_REQUIRE(n_rows >= 0, "n_rows < 0");
    return ix->grow((n_rows + kRowsPerBlock - 1) / kRowsPerBlock);
}

int32_t hipidx_reserve_search(uint64_t h, int32_t k)
{
    GET_INDEX(h);
    HR_REQUIRE(k > 0 && k <= kMaxK, "k must be in 1..%d (got %d)", kMaxK, k);
    return ix->reserve_slot(0, k);
}

int32_t hipidx_search_dev(uint64_t h, const float* q_dev, int32_t nq, int32_t k, double* out_scores64_dev,
                          float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    GET_INDEX(h);
    HR_REQUIRE(nq >= 0, "nq < 0");
    HR_REQUIRE(k > 0 && k <= kMaxK, "k must be in 1..%d (got %d)", kMaxK, k);
    if (nq == 0) return HIPRAG_OK;
    HR_REQUIRE(q_dev && out_scores64_dev && out_ids_dev, "null device pointer");
    return ix->search_dev(q_dev, nq, k, out_scores64_dev, out_scores_dev, out_ids_dev, (hipStream_t)stream);
}

int32_t hipidx_search_begin_dev(uint64_t h, const float* q_dev, int32_t nq, int32_t k, int32_t slot, void* stream)
{
    GET_INDEX(h);
    HR_REQUIRE(nq > 0 && nq <= ix->launch_q, "search_begin takes 1..%d queries (got %d)", ix->launch_q, nq);
    HR_REQUIRE(k > 0 && k <= kMaxK, "k must be in 1..%d (got %d)", kMaxK, k);
    HR_REQUIRE(slot >= 0 && slot < DenseIndex::kSlots, "slot must be in 0..7");
    HR_REQUIRE(q_dev, "null device pointer");
    int32_t rc = ix->prepare(k, slot);
    if (rc) return rc;
    return ix->begin_dev(q_dev, nq, k, slot, (hipStream_t)stream);
}

int32_t hipidx_search_finish_dev(uint64_t h, const float* q_dev, int32_t nq, int32_t k, int32_t slot,
                                 double* out_scores64_dev, float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    GET_INDEX(h);
    HR_REQUIRE(nq > 0 && nq <= ix->launch_q, "search_finish takes 1..%d queries (got %d)", ix->launch_q, nq);
    HR_REQUIRE(slot >= 0 && slot < DenseIndex::kSlots, "slot must be in 0..7");
    HR_REQUIRE(k > 0 && k <= ix->ws[slot].k, "k=%d was not prepared by search_begin on slot %d", k, slot);
    HR_REQUIRE(q_dev && out_scores64_dev && out_ids_dev, "null device pointer");
    return ix->finish_dev(q_dev, nq, k, slot, out_scores64_dev, out_scores_dev, out_ids_dev, (hipStream_t)stream);
}

int32_t hipidx_search(uint64_t h, const float* q_host, int32_t nq, int32_t k, float* out_scores, int64_t* out_ids)
{
    GET_INDEX(h);
    HR_REQUIRE(nq >= 0, "nq < 0");
    HR_REQUIRE(k > 0 && k <= kMaxK, "k must be in 1..%d (got %d)", kMaxK, k);
    if (nq == 0) return HIPRAG_OK;
    HR_REQUIRE(q_host && out_scores && out_ids, "null pointer");
    int32_t rc;
    if ((rc = ix->qbuf.reserve((size_t)nq * ix->d * sizeof(float)))) return rc;
    if ((rc = ix->o64.reserve((size_t)nq * k * sizeof(double)))) return rc;
    if ((rc = ix->o32.reserve((size_t)nq * k * sizeof(float)))) return rc;
    if ((rc = ix->oid.reserve((size_t)nq * k * sizeof(int64_t)))) return rc;
    HR_CHECK_HIP(hipMemcpy(ix->qbuf.p, q_host, (size_t)nq * ix->d * sizeof(float), hipMemcpyHostToDevice));
    rc = ix->search_dev(ix->qbuf.as<float>(), nq, k, ix->o64.as<double>(), ix->o32.as<float>(), ix->oid.as<int64_t>(),
                        nullptr);
    if (rc) return rc;
    HR_CHECK_HIP(hipMemcpy(out_scores, ix->o32.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost));
    HR_CHECK_HIP(hipMemcpy(out_ids, ix->oid.p, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost));
    return HIPRAG_OK;
}

int32_t hipidx_reconstruct(uint64_t h, int64_t row, float* out_host)
{
    GET_INDEX(h);
    HR_REQUIRE(out_host, "null out");
    HR_REQUIRE(row >= 0 && row < ix->ntotal, "row %lld out of range [0,%lld)", (long long)row, (long long)ix->ntotal);
    DevBuf tmp;
    int32_t rc = tmp.reserve((size_t)ix->d * sizeof(float));
    if (rc) return rc;
    if ((rc = ix->wait_adds_host())) return rc;
    const int64_t threads = (int64_t)ix->P * 2;
    hipLaunchKernelGGL(untile_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr, ix->xb.as<float4>(), row,
                       (int64_t)1, ix->d, ix->P, tmp.as<float>());
    HR_CHECK_HIP(hipGetLastError());
    HR_CHECK_HIP(hipMemcpy(out_host, tmp.p, (size_t)ix->d * sizeof(float), hipMemcpyDeviceToHost));
    return HIPRAG_OK;
}

// File format "HIPIDX01": magic[8], int32 d, int32 metric, int64 ntotal, then ntotal*d fp32 row-major.
int32_t hipidx_save(uint64_t h, const char* path)
{
    GET_INDEX(h);
    HR_REQUIRE(path, "null path");
    { const int32_t wrc = ix->wait_adds_host(); if (wrc) return wrc; }
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("cannot open %s for writing", path); return HIPRAG_E_IO; }
    const char magic[8] = {'H', 'I', 'P', 'I', 'D', 'X', '0', '1'};
    int32_t hd[2] = {ix->d, ix->metric};
    int64_t nt = ix->ntotal;
    bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(hd, 4, 2, f) == 2 && fwrite(&nt, 8, 1, f) == 1;
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(64ll << 20) / ((int64_t)ix->d * 4));
    DevBuf tmp;
    std::vector<float> host;
    if (ok && nt > 0) {
        int32_t rc = tmp.reserve((size_t)std::min(chunk, nt) * ix->d * sizeof(float));
        if (rc) { fclose(f); return rc; }
        host.resize((size_t)std::min(chunk, nt) * ix->d);
    }
    for (int64_t o = 0; ok && o < nt; o += chunk) {
        const int64_t m = std::min(chunk, nt - o);
        const int64_t threads = m * ix->P * 2;
        hipLaunchKernelGGL(untile_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr, ix->xb.as<float4>(), o, m,
                           ix->d, ix->P, tmp.as<float>());
        if (hipMemcpy(host.data(), tmp.p, (size_t)m * ix->d * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            fclose(f);
            set_error("device read-back failed while saving");
            return HIPRAG_E_HIP;
        }
        ok = fwrite(host.data(), sizeof(float), (size_t)m * ix->d, f) == (size_t)m * ix->d;
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("write to %s failed", path); return HIPRAG_E_IO; }
    return HIPRAG_OK;
}

int32_t hipidx_load(const char* path, int32_t device, uint64_t* out_handle)
{
    HR_REQUIRE(path && out_handle, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return HIPRAG_E_IO; }
    char magic[8];
    int32_t hd[2];
    int64_t nt = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "HIPIDX01", 8) != 0 || fread(hd, 4, 2, f) != 2 ||
        fread(&nt, 8, 1, f) != 1 || nt < 0) {
        fclose(f);
        set_error("%s is not a HIPIDX01 file", path);
        return HIPRAG_E_IO;
    }
    uint64_t h = 0;
    int32_t rc = hipidx_create(hd[0], hd[1], device, &h);
    if (rc) { fclose(f); return rc; }
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(64ll << 20) / ((int64_t)hd[0] * 4));
    std::vector<float> host((size_t)std::min(chunk, std::max<int64_t>(nt, 1)) * hd[0]);
    for (int64_t o = 0; o < nt; o += chunk) {
        const int64_t m = std::min(chunk, nt - o);
        if (fread(host.data(), sizeof(float), (size_t)m * hd[0], f) != (size_t)m * hd[0]) {
            fclose(f);
            hipidx_destroy(h);
            set_error("%s is truncated", path);
            return HIPRAG_E_IO;
        }
        rc = hipidx_add(h, host.data(), m);
        if (rc) { fclose(f); hipidx_destroy(h); return rc; }
    }
    fclose(f);
    *out_handle = h;
    return HIPRAG_OK;
}

int32_t hipidx_enable_timing(uint64_t h, int32_t on)
{
    GET_INDEX(h);
    ix->timing = on != 0;
    ix->ev_every = std::max(1, on);
    ix->ev_count = 0;
    return HIPRAG_OK;
}

int32_t hipidx_get_stats(uint64_t h, hipidx_stats* out)
{
    GET_INDEX(h);
    HR_REQUIRE(out, "null out");
    HR_CHECK_HIP(hipDeviceSynchronize());
    unsigned long long fb = 0;
    HR_CHECK_HIP(hipMemcpy(&fb, ix->fallback_counter(), sizeof(fb), hipMemcpyDeviceToHost));
    out->passes = ix->passes;
    out->launches = ix->launches;
    out->queries = ix->queries;
    out->fallback_queries = (int64_t)fb;
    {
        unsigned long long rbq = 0;
        HR_CHECK_HIP(hipMemcpy(&rbq, ix->roundb_counter(), sizeof(rbq), hipMemcpyDeviceToHost));
        out->roundb_queries = (int64_t)rbq;
    }
    out->bytes_per_pass = ix->nblocks() * ix->P * (ix->scan_mode == 3 ? 512 : 1024) +
                          (ix->metric == HIPRAG_METRIC_L2 ? ix->nblocks() * kRowsPerBlock * 4 : 0);
    out->avg_scan_ms = -1.f;
    out->avg_scan_wall_ms = -1.f;
    out->avg_scan_gap_ms = 0.f;
    out->timed_passes = 0;
    if (!ix->evs.empty() && ix->ev_count > 0) {
        const int64_t n = std::min<int64_t>(ix->ev_count, DenseIndex::kEvRing);
        double sum = 0.0;
        int64_t ok = 0;
        for (int64_t i = 0; i < n; ++i) {
            float ms = 0.f;
            if (ix->ev_set[(size_t)i] && hipEventElapsedTime(&ms, ix->evs[2 * i], ix->evs[2 * i + 1]) == hipSuccess) { sum += ms; ++ok; }
        }
        if (ok) { out->avg_scan_ms = (float)(sum / ok); out->timed_passes = ok; }
        // the same launches on the GPU's own wall clock: first wave in -> last wave out, and the idle time between the
        // last wave of one scan and the first wave of the next (negative = the next scan started on CUs already free)
        if (ix->stamps.p && ix->nblocks() > 0) {
            const size_t per = (size_t)ix->scan_cus * kMaxScanWaves * 2;
            std::vector<unsigned long long> hst((size_t)n * per);
            HR_CHECK_HIP(hipMemcpy(hst.data(), ix->stamps.p, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<std::pair<unsigned long long, unsigned long long>> se((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                unsigned long long lo = ~0ull, hi = 0;
                for (size_t w = 0; w < per / 2; ++w) {
                    if (hst[(size_t)i * per + 2 * w] == 0) continue;   // slot of a wave this launch did not have (zeroed before the launch)
                    lo = std::min(lo, hst[(size_t)i * per + 2 * w]);
                    hi = std::max(hi, hst[(size_t)i * per + 2 * w + 1]);
                }
                se[(size_t)i] = {lo, hi};
            }
            double dsum = 0.0, gsum = 0.0;
            for (int64_t i = 0; i < n; ++i) dsum += (double)(se[(size_t)i].second - se[(size_t)i].first);
            const bool ordered = ix->ev_count <= DenseIndex::kEvRing;
            for (int64_t i = 0; ordered && i + 1 < n; ++i) gsum += (double)((long long)se[(size_t)i + 1].first - (long long)se[(size_t)i].second);
            if (ix->dbg_on && n > 45) {   // spread of the wave exit times inside one launch (how ragged the static partition ends)
                const size_t i = 44;
                std::vector<unsigned long long> ends;
                for (size_t w = 0; w < per / 2; ++w)
                    if (hst[i * per + 2 * w] != 0) ends.push_back(hst[i * per + 2 * w + 1]);
                std::sort(ends.begin(), ends.end());
                const double tk = 1e3 / ix->wall_khz, t0 = (double)se[i].first;
                fprintf(stderr, "[hiprag] launch %zu: wave exits at us after the first wave in: p1 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f\n", i,
                        ((double)ends[ends.size() / 100] - t0) * tk, ((double)ends[ends.size() / 10] - t0) * tk, ((double)ends[ends.size() / 2] - t0) * tk,
                        ((double)ends[ends.size() * 9 / 10] - t0) * tk, ((double)ends[ends.size() * 99 / 100] - t0) * tk, ((double)ends.back() - t0) * tk);
            }
            if (ix->dbg_on && n > 45) {   // the same exits grouped by XCD (workgroup id mod 8) and by wave slot inside the workgroup
                const size_t i = 44;
                const double tk = 1e3 / ix->wall_khz, t0 = (double)se[i].first;
                double bx[8] = {0}, bw[8] = {0}; int cx[8] = {0}, cw[8] = {0};
                for (size_t w = 0; w < per / 2; ++w) {
                    const double e = ((double)hst[i * per + 2 * w + 1] - t0) * tk;
                    if (e < 100) continue;   // waves without blocks
                    bx[(w / 8) % 8] += e; ++cx[(w / 8) % 8];
                    bw[w % 8] += e; ++cw[w % 8];
                }
                fprintf(stderr, "[hiprag] mean exit by XCD:");
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f", bx[x] / std::max(cx[x], 1));
                fprintf(stderr, " | by wave slot:");
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f", bw[x] / std::max(cw[x], 1));
                fprintf(stderr, "\n");
            }
            if (ix->dbg_on && ordered && ix->dbg_stamps.p) {
                std::vector<unsigned long long> ds((size_t)n * 8);
                HR_CHECK_HIP(hipMemcpy(ds.data(), ix->dbg_stamps.p, ds.size() * 8, hipMemcpyDeviceToHost));
                const double tk = 1e3 / ix->wall_khz;
                fprintf(stderr, "[hiprag] per launch, us after its scan's first wave: scan_end | select_in merge_in rescore_in final_in final_out exh_in exh_out | next scan in\n");
                for (int64_t i = 40; i < std::min<int64_t>(n - 1, 52); ++i) {
                    const unsigned long long t0 = se[(size_t)i].first;
                    auto rel = [&](unsigned long long t) { return (double)((long long)t - (long long)t0) * tk; };
                    const unsigned long long* d8 = &ds[(size_t)i * 8];
                    fprintf(stderr, "  %7.0f | %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f | %7.0f\n", rel(se[(size_t)i].second), rel(d8[0]),
                            rel(d8[1]), rel(d8[2]), rel(d8[3]), rel(~d8[4]), rel(d8[5]), rel(~d8[6]), rel(se[(size_t)i + 1].first));
                }
            }
            out->avg_scan_wall_ms = (float)(dsum / n / ix->wall_khz);
            out->avg_scan_gap_ms = ordered && n > 1 ? (float)(gsum / (n - 1) / ix->wall_khz) : 0.f;
        }
    }
    return HIPRAG_OK;
}

}  // extern "C"
