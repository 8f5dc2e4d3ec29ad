// dense_index.hip -- flat (brute-force) vector index for MI355X / gfx950.
//
// Replaces faiss.IndexFlatL2 / IndexFlatIP as used by the reference:
//   build   rag/storage/faiss_index.py:121-124   (np.array(float32) -> IndexFlatL2(d).add)
//   search  rag/storage/faiss_index.py:81-83     (index.search(float32[1,d], k))
//
// Data layout in HBM (ours, not FAISS's): rows are stored in 32-row BLOCKS, each block as P = d_pad/8 PIECES of
// 1 KiB.  Piece p of block B holds, for lane = h*32 + r (h in {0,1}, r in 0..31), the four floats
// X[32B + r][8p + 4h + 0..3] at float4 position piece_slot(h, r).  d_pad is d rounded up to 128 floats (zero filled).
// Beside the fp32 rows lives a bf16 FILTER COPY (retile_bf16_kernel): the same blocks with P/2 pieces of 1 KiB, piece p
// holding for lane (h, r) the eight values bf16(X[32B + r][16p + 8h + 0..7]) -- the A fragment of one
// v_mfma_f32_32x32x16_bf16, read with fully coalesced 16-B-per-lane loads straight into MFMA operand registers: no LDS
// staging, no transposes, one wave's work is one contiguous run per block.
//
// One search launch answers up to 1024 queries, 64 per PASS over the index:
//   K1  scan_bf16_kernel (default) / scan_split_kernel   every wave streams a contiguous range of blocks once per pass;
//       queries sit in LDS as bf16 B fragments; two 32 x 32 score tiles per block; per 16-row GROUP only the best and the
//       second-best quad maximum survive, and only for groups that reach a per-query bound the scan itself maintains
//       (class-slot maxima, see "the scan" below) -> a CANDIDATE LIST per query, a few hundred entries
//   K2  fin_kernel   per query: rank the list, re-score the best groups in fp64 from the fp32 rows, extend the re-scored
//       prefix until nothing listed can still reach the top k, exact top-k under (score, id); then a CERTIFICATE: every row
//       that was not re-scored has scan value <= m, hence exact score <= m + eps, eps a worst-case bound of the scan's
//       arithmetic.  If the k-th exact score is not > m + eps the query is flagged and
//   K3  exhaustive_kernel (launched always, exits at once unless a query is flagged) re-scores EVERY row in fp64.
// So results are exact for any input (ties, duplicates, zero vectors), and the common case reads the index once per pass.
//
// Bound: HBM.  Algorithmic bytes per pass = nblocks * P * 512 (bf16 copy) or * 1024 (q64: fp32 rows), + norms in L2 mode
// -- hipidx_stats.bytes_per_pass.
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
// fp64 re-score reads (128 whole lines per quad instead of 256 half lines 512 bytes apart).  The q64 scan reads whole
// pieces and only permutes which lane takes which 16 bytes.
__host__ __device__ __forceinline__ int piece_slot(int h, int r) { return ((r >> 2) << 3) | (h << 2) | (r & 3); }
constexpr int kMaxQ = 1024;        // most queries per LAUNCH (16 passes of 64): see DenseIndex::update_launch_q
constexpr int kMaxScanWaves = 12;  // stamp slots per scan workgroup
constexpr int kMaxDPad = 1024;     // d_pad limit (the 128 KiB query tile of the scan)
constexpr int kSelThreads = 256;
constexpr int kExRows = 1024;      // rows per workgroup in the exhaustive path (16 KiB of LDS)
constexpr int kMaxK = 1000;

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
// What a scan leaves behind (round 3).  A GROUP is the 16 rows one lane holds of a block's 32 x 32 score tile: rows
// 32*blk + 8g + 4h + j (g, j in 0..3) for lane half h; group id = 2*blk + h.  Per group the scan knows
//   first  = the largest maximum of its four row QUADS (quad g = rows 8g + 4h + 0..3), the quad number in its two low
//            mantissa bits, and
//   second = the largest quad maximum of the other three quads.
// Rounds 1-2 wrote both for EVERY group (N/16 floats x 2 per query: 256 MB per 512-query launch at 1M rows) and a
// selector read them back to keep a few dozen -- 1.6 % of the launch's bytes and 7-20 % of its time, depending on where the
// buffers had landed.  Now the scan keeps the few that matter itself:
//   * PUBLISH.  Every wave publishes the maximum `first` of each chunk it finishes (after the first block of its range, then
//     every kPublish blocks), per query, with one atomicMax into one of 64 class slots of that query (class = a mix of wave
//     and workgroup index, so that every class has members on every XCD and in every wave slot; one coalesced 256-byte
//     memory-side atomic per wave and chunk).
//   * BOUND.  theta(query) = min over the class slots: 64 distinct groups reach it, and on average ~4.5 x 64 groups of the
//     whole index do.  The waves of a workgroup take turns recomputing it, kRecompute blocks behind a publish; the result
//     goes into thetac[query] (global, atomicMax: it only ever rises) and into the workgroup's LDS copy.
//   * TEST.  A group's values stay in registers (shift chains) for kAge more blocks and are then tested against the LDS
//     copy of their query's bound: first >= bound -> staged in LDS (LDS atomics return on lgkmcnt: the hand-counted vmcnt
//     ring never sees them), else dropped.  A wave that gets there before the bound of the pass exists waits for it (bounded).
//   * FLUSH.  In the pass prologues (and at the end) the staged entries are tested again, against bounds refreshed
//     memory-side, and appended to the per-query CANDIDATE LISTS in global memory, one global atomicAdd per entry: ~350
//     entries per query at 1M rows, 3 MB per launch instead of 256.
// Exactness never depends on timing: every value that was dropped was below a bound that had been folded into thetac, so
// the FINAL thetac[q] bounds everything that is not on q's list -- a late publisher makes lists longer, never wrong.  The
// finish (fin_kernel) ranks a query's list, re-scores its best groups in fp64 and extends the re-scored prefix until nothing
// on the list can still reach the top k; the certificate takes the final thetac as the bound of the unseen rows.
constexpr int kClasses = 64;      // theta slots per query
constexpr int kCandCap = 4096;    // candidate-list entries per query in global memory (overflow -> exhaustive path)
constexpr int kPublish = 4;       // blocks per published chunk
constexpr int kDelay = 4;         // blocks between the end of a value's chunk and its test (publishers of other workgroups catch up)
constexpr int kAge = kPublish - 1 + kDelay;   // chain position at which a value is tested
constexpr int kRecompute = 2;     // blocks between a chunk's publish and the recomputation of theta by the wave whose turn it is
constexpr int kStageHalf = 896;   // staged appends per workgroup and pass (two alternating halves of 14 KiB of LDS); beyond that: direct global appends
constexpr int kNoFilterGroups = 1024;   // indexes with at most this many groups list every group (= the finish's LDS list)

struct Cand {       // one candidate group of one query, 16 bytes: staging entry and global list entry
    u64 key;        // pack_key(first, group id): ord32(first) << 32 | ~group
    float sec;      // second
    u32 q;          // query index inside the launch (staging only)
};

struct ScanArgs {
    const float4* xb;     // blocked fp32 rows (q64 mode)
    const void* xh;       // bf16 filter copy (bf16 mode)
    const float* q;       // [nq, d] row-major queries
    const void* qtile;    // bf16 mode, multi-pass launches: the passes' LDS tile images, made by qtile_kernel (else null)
    const float* norms;   // [rows] squared norms (L2 only)
    u32* slots;           // [passes][kClasses][64]: published chunk maxima (ord32 images), query = 64 * pass + lane
    u32* thetac;          // [Q] current bound per query (ord32 image; 0 = none yet)
    u32* count;           // [Q] appended candidates per query
    Cand* list;           // [Q][kCandCap]
    int64_t nblocks;
    int64_t ntotal;
    int nq, d, P;
    int filter;           // 0: list every group (small indexes)
    int ncls;             // theta classes in use: min(kClasses, waves that own blocks) -- every class needs a publisher, or theta never forms
    unsigned long long* stamps;  // timing only (else null): [waves][2] wall-clock ticks at wave entry / exit
    // start gate (hipidx_gate_tail_dev): every workgroup counts itself in when it starts; the one that completes the launch
    // (started == target: every workgroup of the grid holds its CU) writes the launch's sequence number into the signal word
    // the tail streams of the PREVIOUS launch wait on
    unsigned long long* started;
    unsigned long long target, seq;
    unsigned long long* gate;     // the gate word (device memory)
};

__host__ __device__ __forceinline__ int64_t scan_blocks_per_wave(int64_t nblocks, int64_t nwaves)
{
    return (nblocks + nwaves - 1) / nwaves;   // equal contiguous ranges: every active wave ends at the same time
}

// A group's 16 rows are four QUADS of 4 consecutive rows (quad g = rows 8g + 4h + 0..3: 64 contiguous bytes of every
// fp32 piece, the unit the fp64 re-score reads).  Replacing two mantissa bits by the quad number moves a value by
// < 2^-21 |v|: the certificate's eps carries that term (kTagSlack).
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

// ---- the in-register / in-LDS tail of both scan kernels --------------------------------------------------------------
// LDS behind the query tile: Cand stage[2][kStageHalf] (the passes alternate between the halves: the appends of pass p go to
// half p & 1 and are flushed in the prologue of pass p + 1, beside the staging of that pass's query tile) | u32 ctl[16] (ctl[h] =
// appends into half h since the launch began -- never reset, every thread remembers what has been flushed; ctl[4 + s] = tag of
// bound slot s)
// | u32 bound[kThetaBack][64] -- the workgroup's copy of the bounds of the last kThetaBack passes (slot = pass & 7, tag =
// pass + 1), which is what the in-loop test reads: NO global load sits in the block loop.  (A per-block global read of
// thetac did, at first: those lines are rewritten memory-side all the time, so the read misses L2, and because loads return
// in order every ring re-arm issued behind it waited for it -- 0.43 ms of a 2.6 ms launch.)
constexpr int kThetaBack = 8;    // passes a value can lag behind the pass its wave is in (kAge blocks / >= 1 block per pass)
struct TailLds {
    Cand* stage;
    u32* ctl;
    u32* bound;
};
struct ScanTail {
    float cf[2][kAge + 1], cs[2][kAge + 1];   // [query tile][age]: first / second of the last kAge + 1 blocks (shift chains)
    float pm0, pm1;                           // running maximum of `first` over the current chunk, per tile
    int t;                                    // blocks this wave has finished, over all passes
    int pa, ja;                               // pass and block offset of the value that is tested next
    int ev;                                   // publish events so far (the waves of a workgroup take turns recomputing theta)
    int rc_due, rc_pass;                      // pending recomputation of theta: when (in blocks of this wave) and for which pass
    u32 flushed0, flushed1;                   // appends of each stage half that have been flushed (the same in every thread;
                                              // two scalars, not an array: a dynamically indexed array would live in scratch memory,
                                              // whose accesses count in vmcnt and drained the load ring -- 2.6 -> 3.5 ms per launch)

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i <= kAge; ++i) { cf[0][i] = cf[1][i] = -FLT_MAX; cs[0][i] = cs[1][i] = -FLT_MAX; }
        pm0 = pm1 = -INFINITY;
        t = 0; pa = 0; ja = 0; ev = 0; rc_due = -1; rc_pass = 0; flushed0 = flushed1 = 0;
    }
};

__device__ __forceinline__ u32 load_agent_u32(const u32* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// memory-side read (a returning atomic): never served from an XCD's L2, hence never stale; ~88 requests per microsecond
// and 64-byte line chip-wide, so only for rare paths
__device__ __forceinline__ u32 load_memside_u32(u32* p)
{
    return __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the workgroup's LDS copy of a pass's bound: lane = query; raises, never lowers, the copy of the same pass
__device__ __forceinline__ void lds_put_bound(const TailLds& L, int pass, int lane, u32 v)
{
    const int slot = pass & (kThetaBack - 1);
    if (L.ctl[4 + slot] == (u32)pass + 1u) {
        if (v) atomicMax(L.bound + slot * 64 + lane, v);
    } else {
        L.bound[slot * 64 + lane] = v;
        if (lane == 0) L.ctl[4 + slot] = (u32)pass + 1u;
    }
}
__device__ __forceinline__ u32 lds_get_bound(const TailLds& L, int pass, int q_in_pass)
{
    const int slot = pass & (kThetaBack - 1);
    return L.ctl[4 + slot] == (u32)pass + 1u ? L.bound[slot * 64 + q_in_pass] : 0u;
}

// theta of the 64 queries of `pass` from their class slots (lane = query), folded into thetac and into the LDS copy.
// FRESH = memory-side reads (never stale, slow); otherwise sc1 loads, which the XCD's L2 may serve from a copy it fetched
// earlier -- an older, LOWER maximum: still a valid bound, only less tight; an empty class means no update.
template <bool FRESH>
__device__ __forceinline__ u32 recompute_theta(const ScanArgs& a, const TailLds& L, int pass, int lane)
{
    u32* s = a.slots + (size_t)pass * kClasses * 64 + lane;
    u32 m = 0xFFFFFFFFu;
#pragma unroll 16
    for (int c = 0; c < a.ncls; ++c) m = min(m, FRESH ? load_memside_u32(s + c * 64) : load_agent_u32(s + c * 64));
    if (pass * 64 + lane >= a.nq) m = 0;
    if (m != 0) atomicMax(a.thetac + pass * 64 + lane, m);   // BEFORE anything is dropped against m: the final thetac bounds every drop
    lds_put_bound(L, pass, lane, m);
    return m;
}

// A wave that reaches the test of a pass's values before that pass's bound exists -- waves are not in lockstep, the front
// runners of a launch are several blocks ahead of the rest -- would stage everything it holds (measured: a fifth of all
// tests in multi-pass launches, 13 k entries per query).  It waits instead, bounded (60 us): it is ahead of the publishers
// it waits for.  Polls are memory-side reads; the waiting wave recomputes theta itself, so the bound appears as soon as
// every class has a publisher.
__device__ __forceinline__ void wait_for_theta(const ScanArgs& a, const TailLds& L, int pass, int lane)
{
    const bool live = pass * 64 + lane < a.nq;
    const unsigned long long t0 = wall_clock64();
    for (int it = 0;; ++it) {
        u32 v = load_memside_u32(a.thetac + pass * 64 + lane);    // another workgroup may have formed it already
        if (__ballot(live && v == 0) && it) v = recompute_theta<true>(a, L, pass, lane);
        if (!__ballot(live && v == 0)) { lds_put_bound(L, pass, lane, v); break; }
        if (wall_clock64() - t0 > 6000ull) break;
        if (it) __builtin_amdgcn_s_sleep(127);
    }
}

// publish this wave's chunk maxima of `pass` (lane l ends up with the maximum of query 64 * pass + l)
__device__ __forceinline__ void publish_chunk(const ScanArgs& a, ScanTail& T, int pass, int lane, int cls, bool my_turn)
{
    const float x0 = fmaxf(T.pm0, __shfl_xor(T.pm0, 32)), x1 = fmaxf(T.pm1, __shfl_xor(T.pm1, 32));
    const float v = lane < 32 ? x0 : x1;
    if (pass * 64 + lane < a.nq) atomicMax(a.slots + ((size_t)pass * kClasses + cls) * 64 + lane, ord32(v));
    T.pm0 = T.pm1 = -INFINITY;
    // theta is recomputed kRecompute blocks LATER: every workgroup publishes a chunk at about the same moment, and a
    // recomputation right behind one's own publish finds classes whose publishers' atomics have not landed yet
    if (my_turn) { T.rc_due = T.t + kRecompute; T.rc_pass = pass; }
    ++T.ev;
}

__device__ __forceinline__ void append_direct(const ScanArgs& a, u64 key, float sec, u32 q)
{
    const u32 p = atomicAdd(a.count + q, 1u);
    if (p < (u32)kCandCap) a.list[(size_t)q * kCandCap + p] = Cand{key, sec, q};
}

__device__ __forceinline__ void stage_append(const ScanArgs& a, const TailLds& L, int half, u32 flushed, u64 key, float sec, u32 q)
{
    const u32 p = atomicAdd(L.ctl + half, 1u) - flushed;      // LDS atomic: returns on lgkmcnt, the hand-counted vmcnt ring never sees it
    if (p < (u32)kStageHalf) L.stage[half * kStageHalf + p] = Cand{key, sec, q};
    else append_direct(a, key, sec, q);
}

// test one parked value pair (block b0 + T.ja of pass T.pa) against the bounds th0 / th1 of its two queries; advance (pa, ja)
__device__ __forceinline__ void test_aged(const ScanArgs& a, ScanTail& T, const TailLds& L, int half, float f0, float s0, float f1, float s1,
                                          u32 th0, u32 th1, int lane, int64_t b0, int nbw)
{
    const int b = lane & 31, h = lane >> 5;
    const int q0 = T.pa * 64 + b, q1 = q0 + 32;
    const bool p0 = q0 < a.nq && ord32(f0) >= th0;
    const bool p1 = q1 < a.nq && ord32(f1) >= th1;
    if (__ballot(p0 || p1)) {   // wave-uniform: most blocks append nothing
        const u32 gid = (u32)(2 * (b0 + T.ja) + h);
        const u32 fl = half ? T.flushed1 : T.flushed0;
        if (p0) stage_append(a, L, half, fl, pack_key(f0, gid), s0, (u32)q0);
        if (p1) stage_append(a, L, half, fl, pack_key(f1, gid), s1, (u32)q1);
    }
    if (++T.ja == nbw) { T.ja = 0; ++T.pa; }
}

// one finished block: park its four values, publish at chunk ends, test the value that has reached kAge
__device__ __forceinline__ void tail_block(const ScanArgs& a, ScanTail& T, const TailLds& L, float f0, float s0, float f1, float s1,
                                           int pass, int j, int nbw, int lane, int wave, int nwaves, int cls, int64_t b0)
{
#pragma unroll
    for (int i = kAge; i > 0; --i) {
        T.cf[0][i] = T.cf[0][i - 1]; T.cs[0][i] = T.cs[0][i - 1];
        T.cf[1][i] = T.cf[1][i - 1]; T.cs[1][i] = T.cs[1][i - 1];
    }
    T.cf[0][0] = f0; T.cs[0][0] = s0; T.cf[1][0] = f1; T.cs[1][0] = s1;
    if (a.filter) {
        T.pm0 = fmaxf(T.pm0, f0);
        T.pm1 = fmaxf(T.pm1, f1);
        // after the FIRST block of a range (the bound of a pass can form as early as possible), then every kPublish blocks
        if ((j & (kPublish - 1)) == 0 || j == nbw - 1) publish_chunk(a, T, pass, lane, cls, T.ev % nwaves == wave);
    }
    if (T.t >= kAge) {
        const int b = lane & 31;
        u32 th0 = 0, th1 = 0;
        if (a.filter) {
            th0 = lds_get_bound(L, T.pa, b);
            th1 = lds_get_bound(L, T.pa, 32 + b);
            if (__ballot((th0 == 0 && T.pa * 64 + b < a.nq) || (th1 == 0 && T.pa * 64 + 32 + b < a.nq))) {
                wait_for_theta(a, L, T.pa, lane);
                th0 = lds_get_bound(L, T.pa, b);
                th1 = lds_get_bound(L, T.pa, 32 + b);
            }
        }
        test_aged(a, T, L, pass & 1, T.cf[0][kAge], T.cs[0][kAge], T.cf[1][kAge], T.cs[1][kAge], th0, th1, lane, b0, nbw);
    }
    if (T.rc_due >= 0 && T.t >= T.rc_due) { recompute_theta<false>(a, L, T.rc_pass, lane); T.rc_due = -1; }
    ++T.t;
}

// The workgroup's bounds of the last kThetaBack passes, refreshed from global memory (memory-side: fresh whatever the XCD's
// L2 holds) -- once per workgroup and pass, in the pass prologue.  (Read per staged entry instead, 150 k same-line atomics
// per pass saturated the four lines the bounds of a pass live on: ~88 requests per microsecond and line.)
__device__ __forceinline__ void refresh_bounds(const ScanArgs& a, const TailLds& L, int pass_now, int wave, int nwaves, int lane)
{
    if (!a.filter) return;
    for (int w = wave; w < kThetaBack; w += nwaves) {
        const int p = pass_now - w;
        if (p >= 0) lds_put_bound(L, p, lane, load_memside_u32(a.thetac + p * 64 + lane));
    }
}
// staged appends of one half -> the queries' global lists (whole workgroup; the half is not appended to meanwhile).  Every
// entry is tested AGAIN, against what its query's bound has become since it was staged: the in-loop test runs a few blocks
// behind the publishers, when the bound of a pass is still forming (a wave that runs ahead tests against the publishes of the
// other fast waves only -- measured 0.6-1.1 k staged entries per query at 1M rows where the final bound admits 0.3 k); the
// flush runs a pass later, or at the end of the launch.  Two phases, so that the round trip of the list counters' atomics
// hides behind whatever the caller does in between (the pass prologue stages the next query tile there).
struct FlushTicket {
    Cand e;
    u32 pos;
    bool live;
};
template <int NT>
__device__ __forceinline__ FlushTicket flush_begin(const ScanArgs& a, const TailLds& L, int half, int n, int tid)
{
    FlushTicket f;
    f.live = false;
    f.pos = 0;
    if (tid < n) {
        f.e = L.stage[half * kStageHalf + tid];
        f.live = (u32)(f.e.key >> 32) >= lds_get_bound(L, (int)(f.e.q >> 6), (int)(f.e.q & 63));
        if (f.live) f.pos = atomicAdd(a.count + f.e.q, 1u);
    }
    return f;
}
template <int NT>
__device__ __forceinline__ void flush_end(const ScanArgs& a, const TailLds& L, const FlushTicket& f, int half, int n, int tid)
{
    if (f.live && f.pos < (u32)kCandCap) a.list[(size_t)f.e.q * kCandCap + f.pos] = f.e;
    for (int i = tid + NT; i < n; i += NT) {     // more entries than threads: the front runners of a launch
        const Cand e = L.stage[half * kStageHalf + i];
        if ((u32)(e.key >> 32) >= lds_get_bound(L, (int)(e.q >> 6), (int)(e.q & 63))) append_direct(a, e.key, e.sec, e.q);
    }
}

// after the last block of the last pass: test what is still in the chains against the bound as it stands (no rendezvous: a
// wave that ends early misses the last chunks of the late ones -- a slightly lower bound, a few per cent more entries)
__device__ __forceinline__ void tail_drain(const ScanArgs& a, ScanTail& T, const TailLds& L, int half, int lane, int64_t b0, int nbw)
{
    const int R = T.t < kAge ? T.t : kAge;   // values still untested: chain positions R - 1 .. 0
    if (a.filter && T.rc_due >= 0) { recompute_theta<true>(a, L, T.rc_pass, lane); T.rc_due = -1; }
    int pa_loaded = -1;
    const int b = lane & 31;
#pragma unroll
    for (int pos = kAge - 1; pos >= 0; --pos) {
        if (pos < R) {
            if (a.filter && T.pa != pa_loaded) {   // the flush tests every staged entry again, against a refreshed copy
                lds_put_bound(L, T.pa, lane, load_memside_u32(a.thetac + T.pa * 64 + lane));
                pa_loaded = T.pa;
            }
            const u32 th0 = a.filter ? lds_get_bound(L, T.pa, b) : 0u, th1 = a.filter ? lds_get_bound(L, T.pa, 32 + b) : 0u;
            test_aged(a, T, L, half, T.cf[0][pos], T.cs[0][pos], T.cf[1][pos], T.cs[1][pos], th0, th1, lane, b0, nbw);
        }
    }
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo)
{
    const bf16x2 h = __builtin_convertvector(f32x2{v0, v1}, bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
    hi = __builtin_bit_cast(unsigned, h);
    const float r0 = v0 - __uint_as_float(hi << 16);
    const float r1 = v1 - __uint_as_float(hi & 0xFFFF0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
}

// Query tile of one pass in LDS, bf16, for BOTH scan kernels: piece p (16 k-values), lane (h, b): Q^[b][16p + 8h + 0..7] for
// queries qbase + 0..31, then the same for queries qbase + 32..63 -- the B fragment of v_mfma_f32_32x32x16_bf16.  This
// form converts the (L2-resident) row-major fp32 queries in the workgroup: one-pass launches and the q64 kernel; multi-pass
// launches of the bf16 kernel copy images made by qtile_kernel (below).
template <int NT>
__device__ __forceinline__ void stage_query_tile(const ScanArgs& a, bf16x8* qw, int P2, int qbase, int tid)
{
    const bool vec_ok = (a.d & 3) == 0;
    // opaque copy of the thread index: left visible, hipcc hoists this address arithmetic out of the pass loop and keeps ~60
    // registers of it alive through the ring loop
    int tid_p = tid;
    asm volatile("" : "+v"(tid_p));
    int idx0 = tid_p;
    if (vec_ok && a.d >= 4) {
        // four fragments per step, their eight 16-B loads issued together and UNCONDITIONALLY (clamped query and column,
        // masked afterwards): predicated, each load was a branch + load + s_waitcnt vmcnt(0) -- 32 dependent L2 round trips
        // per thread and pass
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

// The same tile images for EVERY pass of a launch, written once by a small kernel ahead of the scan (multi-pass launches of
// the bf16 kernel): out[pass][idx] = what stage_query_tile puts at qw[idx] for qbase = 64 pass.  The scan's workgroups then
// copy a finished 128 KiB image per pass instead of each converting it from the fp32 queries: with 512 threads the
// conversion was four dependent L2 round trips per pass, with the 256 threads of a small-shard workgroup eight -- 10 us of
// a 50 us pass at 125 k rows per GPU, sixteen times per launch.
__global__ __launch_bounds__(256) void qtile_kernel(const float* __restrict__ q, int nq, int d, int P2, bf16x8* __restrict__ out)
{
    const int per_pass = 2 * P2 * 64;
    const int pass = blockIdx.y, qbase = pass * 64;
    const bool vec_ok = (d & 3) == 0;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < per_pass; idx += gridDim.x * 256) {
        const int tile = idx >= P2 * 64;
        const int u = idx - tile * P2 * 64;
        const int p = u >> 6, l = u & 63;
        const int b = qbase + (l & 31) + 32 * tile;
        const int col = 16 * p + 8 * (l >> 5);
        float v[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c = col + 4 * half;
            if (b < nq && vec_ok && c + 3 < d) {
                const float4 t = *reinterpret_cast<const float4*>(q + (int64_t)b * d + c);
                v[4 * half + 0] = t.x; v[4 * half + 1] = t.y; v[4 * half + 2] = t.z; v[4 * half + 3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * half + j] = (b < nq && c + j < d) ? q[(int64_t)b * d + c + j] : 0.f;
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
        out[(size_t)pass * per_pass + idx] = o;
    }
}

// copy of one pass's image into LDS by LDS-DMA: a wave instruction moves 64 consecutive 16-B fragments (1 KiB, lane-linear on
// both sides), every wave issues its whole share before anyone waits -- one round trip per pass, no registers
typedef __attribute__((address_space(3))) void* scan_lds_ptr_t;
template <int NT>
__device__ __forceinline__ void stage_query_tile_image(const bf16x8* __restrict__ img, bf16x8* qw, int P2, int tid)
{
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nchunks = 2 * P2;                       // 1 KiB chunks in the image
    for (int c = wave; c < nchunks; c += NT / 64)
        __builtin_amdgcn_global_load_lds((const void*)(img + (size_t)c * 64 + lane), (scan_lds_ptr_t)(qw + (size_t)c * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the image has landed (and, in order before it, whatever the ring had in flight)
}

// ------------------------------------------------------------------------------------------------------
// K1, the two operand modes (HIPRAG_SCAN_MODE).  Both answer 64 queries per pass against bf16 query tiles, run
// ceil(nq / 64) passes back to back inside one launch (barrier, re-stage the tile, stream the wave's block range again; the
// piece stream is CYCLIC -- the last re-arms of pass p already fetch the first pieces of pass p + 1 -- so HBM keeps
// streaming across the pass boundary and there is no kernel boundary, which would cost 35-60 us of idle GPU), and share the
// tail above.  The X stream is driven by hand: loads are inline asm (invisible to hipcc's waitcnt pass, which otherwise
// drains the queue with vmcnt(0) at the loop back-edge) and every use is fenced by a counted s_waitcnt that takes the ring
// slot as an in/out operand, so no consumer can be scheduled above its wait.  vmcnt(RING - 1) before slot i is exact when
// only the ring is in flight and merely conservative when other operations are queued behind it (loads return in order
// among themselves).  The ring is armed BEFORE the query tile is staged, so HBM is streaming while the prologue runs.
//
//   bf16 (default)  scan_bf16_kernel streams the bf16 FILTER COPY of the rows -- half the bytes of the fp32 rows: per 1 KiB
//                   piece two v_mfma_f32_32x32x16_bf16 (one per 32-query tile), no conversion work.  Both truncations are
//                   carried by the certificate (scan_eps, mode 3): |<x,q> - <x^,q^>| <= |x| |q - q^| + |x - x^| |q^|, both
//                   deviations computed exactly (per query at search time, maximum over rows at add time).
//   q64             scan_split_kernel streams the fp32 rows themselves (N*d*4 bytes per pass, SURVEY 8(d)'s price) and
//                   splits every value on the fly into hi = bf16(v), lo = bf16(v - hi): four MFMAs per pair of fp32
//                   pieces; only the query truncation is left in eps (mode 2).  No extra memory.
// The fp64 re-score and the exhaustive path read the fp32 rows, so both modes return the same exact results.
// ------------------------------------------------------------------------------------------------------
#define HIPRAG_SCAN_PROLOGUE(PIECES_PER_BLOCK, BASE_PTR)                                                                 \
    extern __shared__ float4 qs[];                                                                                       \
    constexpr int NT = NWAVES * 64;                                                                                      \
    const int tid = threadIdx.x;                                                                                         \
    const int lane = tid & 63;                                                                                           \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                           \
    const int P2 = a.P / 2;   /* 16-k-value pieces per block = LDS fragments per query tile */                           \
    /* statically partitioned: a CU that also hosts a tail workgroup of an earlier launch becomes the straggler of the   \
       whole launch, so let the scan's waves win issue arbitration */                                                    \
    __builtin_amdgcn_s_setprio(3);                                                                                       \
    const int64_t gw = (int64_t)blockIdx.x * NWAVES + wave;                                                              \
    const int64_t W = (int64_t)gridDim.x * NWAVES;                                                                       \
    const int64_t bpw = scan_blocks_per_wave(a.nblocks, W);                                                              \
    const int64_t b0 = min(gw * bpw, a.nblocks);                                                                         \
    const int64_t b1 = min(b0 + bpw, a.nblocks);                                                                         \
    const int nbw = (int)(b1 - b0);                                                                                      \
    const int S = nbw * (PIECES_PER_BLOCK);                                                                              \
    const float4* base = (BASE_PTR);                                                                                     \
    if (a.stamps && lane == 0) a.stamps[2 * gw] = wall_clock64();                                                        \
    if (a.gate && tid == 0) {                                                                                            \
        const unsigned long long was = __hip_atomic_fetch_add(a.started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
        if (was + 1 == a.target) __hip_atomic_fetch_max(a.gate, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       \
    }                                                                                                                    \
    TailLds L;                                                                                                           \
    L.stage = reinterpret_cast<Cand*>(reinterpret_cast<char*>(qs) + (size_t)a.P * 1024);                                 \
    L.ctl = reinterpret_cast<u32*>(L.stage + 2 * kStageHalf);                                                            \
    L.bound = L.ctl + 16;                                                                                                \
    if (tid < 16) L.ctl[tid] = 0;                                                                                        \
    const int cls = (int)((gw + blockIdx.x) % a.ncls); /* members of a class on every XCD and every wave slot */           \
    ScanTail T;                                                                                                          \
    T.init()

// pass prologue shared by both kernels: barriers, staged appends of the previous pass -> global lists, new query tile
// pass prologue of both kernels: STAGE_TILE is the statement that writes the pass's query tile
#define HIPRAG_PASS_PROLOGUE(STAGE_TILE)                                                                                 \
    const int qbase = pass * 64;                                                                                         \
    const bool wrap = pass + 1 < npass;                                                                                  \
    int nprev = 0;                                                                                                       \
    u32 total_prev = 0;                                                                                                  \
    if (pass) {                                                                                                          \
        refresh_bounds(a, L, pass - 1, wave, NWAVES, lane); /* its round trip overlaps the wait for the slower waves */  \
        __syncthreads(); /* every wave is done with the previous tile and with the previous pass's staged appends */     \
        total_prev = L.ctl[(pass - 1) & 1];                                                                              \
        nprev = (int)min(total_prev - (((pass - 1) & 1) ? T.flushed1 : T.flushed0), (u32)kStageHalf);                    \
    }                                                                                                                    \
    const FlushTicket ft = flush_begin<NT>(a, L, (pass - 1) & 1, nprev, tid);                                            \
    STAGE_TILE;                                                                                                          \
    flush_end<NT>(a, L, ft, (pass - 1) & 1, nprev, tid);                                                                 \
    if (pass) { if ((pass - 1) & 1) T.flushed1 = total_prev; else T.flushed0 = total_prev; }                             \
    __syncthreads()

#define HIPRAG_SCAN_EPILOGUE()                                                                                           \
    if (S > 0) tail_drain(a, T, L, (npass - 1) & 1, lane, b0, nbw);                                                      \
    refresh_bounds(a, L, npass - 1, wave, NWAVES, lane);                                                                 \
    __syncthreads();                                                                                                     \
    {                                                                                                                    \
        const int hl = (npass - 1) & 1;                                                                                  \
        const int nl = (int)min(L.ctl[hl] - (hl ? T.flushed1 : T.flushed0), (u32)kStageHalf);                            \
        const FlushTicket fl = flush_begin<NT>(a, L, hl, nl, tid);                                                       \
        flush_end<NT>(a, L, fl, hl, nl, tid);                                                                            \
    }                                                                                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the tail's clamped re-arms are still in flight */                \
    if (a.stamps && lane == 0) a.stamps[2 * gw + 1] = wall_clock64()

template <int METRIC, int NWAVES, int RING, bool MULTI>
__global__ __launch_bounds__(NWAVES * 64) void scan_bf16_kernel(ScanArgs a)
{
    HIPRAG_SCAN_PROLOGUE(P2, reinterpret_cast<const float4*>(a.xh) + b0 * P2 * kPieceVec4);
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
        HIPRAG_PASS_PROLOGUE(if (MULTI && a.qtile) stage_query_tile_image<NT>(reinterpret_cast<const bf16x8*>(a.qtile) + (size_t)pass * 2 * P2 * 64,
                                                                             reinterpret_cast<bf16x8*>(qs), P2, tid);
                             else stage_query_tile<NT>(a, reinterpret_cast<bf16x8*>(qs), P2, qbase, tid));
        if (S <= 0) continue;

        int s = 0;
        bf16x8 n0v = q0[lane], n1v = q1[lane];   // query fragments are read one piece ahead
        for (int j = 0; j < nbw; ++j) {
            const int64_t blk = b0 + j;
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
                    // one step = one 1 KiB piece (16 k-values): two MFMAs (one per query tile), re-arm the ring slot
                    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ring[i]) : "n"(RING - 1) : "memory");
                    const bf16x8 av = __builtin_bit_cast(bf16x8, ring[i]);
                    const bf16x8 bv0 = n0v, bv1 = n1v;
                    int nx = pp + i + 1;
                    nx = nx == P2 ? 0 : nx;
                    n0v = q0[nx * 64 + lane];
                    n1v = q1[nx * 64 + lane];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv1, acc1, 0, 0, 0);
                    // re-arm: past the end of the range the stream wraps to the first pieces of the next pass (S >= RING
                    // whenever S > 0, so one subtraction is enough); the last pass repeats its final piece instead
                    int n0 = s + RING + i;
                    n0 = n0 < S ? n0 : (wrap ? n0 - S : S - 1);
                    const unsigned voff = lane16 + (unsigned)n0 * 1024u;
                    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(ring[i]) : "v"(voff), "s"(base) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                s += RING;
            }
            if (METRIC == HIPRAG_METRIC_L2)   // the norms were issued before this block's P2 >= RING re-arms
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(nrm[0]), "+v"(nrm[1]), "+v"(nrm[2]), "+v"(nrm[3]) : "n"(RING) : "memory");
            float sec0, sec1;
            int lane_b = lane;
            asm volatile("" : "+v"(lane_b));   // (as in stage_query_tile: keeps the epilogue's lane arithmetic out of the ring loop's live set)
            const float fst0 = block_lane_top2<METRIC>(acc0, nrm, blk, lane_b >> 5, a, sec0);
            const float fst1 = block_lane_top2<METRIC>(acc1, nrm, blk, lane_b >> 5, a, sec1);
            tail_block(a, T, L, fst0, sec0, fst1, sec1, pass, j, nbw, lane_b, wave, NWAVES, cls, b0);
        }
    }  // pass
    HIPRAG_SCAN_EPILOGUE();
}

// query tile of the q64 kernel: for piece pair pp and lane (h, b) the bf16 images of Q[b][16pp + 4h + 0..3] and
// Q[b][16pp + 8 + 4h + 0..3] (the k order the A fragment gets from a PAIR of fp32 pieces), queries qbase + 0..31 then + 32..63;
// hi parts only: |q - bf16(q)| is in the certificate's eps
template <int NT>
__device__ __forceinline__ void stage_query_tile_split(const ScanArgs& a, u32x4* qw, int npairs, int qbase, int tid)
{
    const bool vec_ok = (a.d & 3) == 0;
    int tid_p = tid;
    asm volatile("" : "+v"(tid_p));
    for (int idx = tid_p; idx < 2 * npairs * 64; idx += NT) {
        const int tile = idx >= npairs * 64;
        const int u = idx - tile * npairs * 64;
        const int pp = u >> 6, l = u & 63;
        const int hh = l >> 5;
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
                for (int jj = 0; jj < 4; ++jj)
                    v[4 * half + jj] = (b < a.nq && col + jj < a.d) ? a.q[(int64_t)b * a.d + col + jj] : 0.f;
            }
        }
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        split_pair(v[0], v[1], h0, l0);
        split_pair(v[2], v[3], h1, l1);
        split_pair(v[4], v[5], h2, l2);
        split_pair(v[6], v[7], h3, l3);
        (void)l0; (void)l1; (void)l2; (void)l3;
        qw[idx] = u32x4{h0, h1, h2, h3};
    }
}

// q64: the fp32 rows, split on the fly.  Data layout: block of 32 rows = P pieces of 1 KiB, piece p, lane slot
// piece_slot(h, r): X[32B + r][8p + 4h + 0..3]; a PAIR of pieces gives a lane the 8 k-values [16pp + 4h + 0..3] and
// [16pp + 8 + 4h + 0..3] of its row -- so the query tile of this kernel holds, for piece pair pp and lane (h, b), the
// bf16 images of Q[b][16pp + 4h + 0..3] and Q[b][16pp + 8 + 4h + 0..3] (k order of the A fragment), staged by its own
// prologue below (not stage_query_tile's 8-consecutive-values order).
template <int METRIC, int NWAVES, int RING, bool MULTI>
__global__ __launch_bounds__(NWAVES * 64) void scan_split_kernel(ScanArgs a)
{
    HIPRAG_SCAN_PROLOGUE(a.P, a.xb + b0 * a.P * kPieceVec4);
    const int npairs = P2;
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
    const int npass = MULTI ? (a.nq + 63) / 64 : 1;
    const bf16x8* q0 = reinterpret_cast<const bf16x8*>(qs);
    const bf16x8* q1 = q0 + npairs * 64;
    for (int pass = 0; pass < npass; ++pass) {
        HIPRAG_PASS_PROLOGUE(stage_query_tile_split<NT>(a, reinterpret_cast<u32x4*>(qs), npairs, qbase, tid));
        if (S <= 0) continue;

        int s = 0;
        bf16x8 n0v = q0[lane], n1v = q1[lane];
        for (int j = 0; j < nbw; ++j) {
            const int64_t blk = b0 + j;
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
            for (int pp = 0; pp < npairs; pp += RING / 2) {
#pragma unroll
                for (int i = 0; i < RING / 2; ++i) {
                    // one step = two 1 KiB pieces (16 k-values per lane half): split, 4 MFMAs, re-arm both ring slots
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ring[2 * i]), "+v"(ring[2 * i + 1]) : "n"(RING - 2) : "memory");
                    const f32x4 v0 = ring[2 * i], v1 = ring[2 * i + 1];
                    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                    split_pair(v0[0], v0[1], h0, l0);
                    split_pair(v0[2], v0[3], h1, l1);
                    split_pair(v1[0], v1[1], h2, l2);
                    split_pair(v1[2], v1[3], h3, l3);
                    const u32x4 ahi = {h0, h1, h2, h3}, alo = {l0, l1, l2, l3};
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, ahi), al = __builtin_bit_cast(bf16x8, alo);
                    const bf16x8 bv0 = n0v, bv1 = n1v;
                    int nx = pp + i + 1;
                    nx = nx == npairs ? 0 : nx;
                    n0v = q0[nx * 64 + lane];
                    n1v = q1[nx * 64 + lane];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bv0, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bv0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bv1, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bv1, acc1, 0, 0, 0);
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
            float sec0, sec1;
            int lane_b = lane;
            asm volatile("" : "+v"(lane_b));
            const float fst0 = block_lane_top2<METRIC>(acc0, nrm, blk, lane_b >> 5, a, sec0);
            const float fst1 = block_lane_top2<METRIC>(acc1, nrm, blk, lane_b >> 5, a, sec1);
            tail_block(a, T, L, fst0, sec0, fst1, sec1, pass, j, nbw, lane_b, wave, NWAVES, cls, b0);
        }
    }  // pass
    HIPRAG_SCAN_EPILOGUE();
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

// Certificate slack: |scan value - exact score| <= eps for every row, on the scale the scan selects by (IP: <x,q>;
// L2: 2<x,q> - |x|^2 = |q|^2 - dist).  Terms: fp32 accumulation ((d_pad + 80) u, u = 2^-24, incl. the split's extra
// roundings), the operand truncations of the mode, and the quad tag in the two low mantissa bits (kTagSlack of the value's
// magnitude).
constexpr double kTagSlack = 4.76837158203125e-07;  // 2^-21
template <int METRIC>
__device__ __forceinline__ double scan_eps(int dpad, int mode, double qn2, double xn2, double dq2, double dx2)
{
    const double xn = sqrt(xn2), qn = sqrt(qn2);
    const double u = 5.9604644775390625e-08;  // 2^-24
    double eps = 1.05 * (double)(dpad + 80) * u * qn * xn;
    // q64: the scan sees q^ = bf16(q) and x^ = hi + lo.  |<x, q - q^>| <= |x| |q - q^| (Cauchy-Schwarz) with |q - q^| computed
    // exactly per query (dq2, same RNE conversion as the scan prologue) -- about 0.3 * 2^-9 |q| for ordinary data instead of the
    // element-wise worst case 2^-9 |q|; |<x - x^, q^>| <= 2^-17 |x| |q^| for the two-term split of the rows.
    if (mode == 2) eps += 7.62939453125e-06 * 1.01 * qn * xn + 1.0001 * sqrt(dq2) * xn;
    // bf16 filter copy: |<x, q> - <x^, q^>| <= |x| |q - q^| + |x - x^| |q^|, |q^| <= |q| + |q - q^|; dx2 = max over rows
    if (mode == 3) eps += 1.0001 * (sqrt(dq2) * xn + sqrt(dx2) * (qn + sqrt(dq2)));
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
// K2: the finish -- ONE kernel, one 8-wave workgroup per query (rounds 1-2 took seven launches: select, re-score,
// final, three for "round B", exhaustive check):
//   1. the query's candidate list -> LDS, dropping entries below the final thetac (they cannot matter: either the
//      certificate holds with thetac as the bound of everything unseen, or the query goes to the exhaustive path);
//   2. rank by counting (keys are distinct: they carry the group id) -> sorted list;
//   3. re-score the tagged quad of the best R0 = k + max(8, k/2) groups in fp64 from the fp32 rows, exact top-k of those
//      rows under (score, id);
//   4. EXTEND: tau = (k-th exact score so far) - eps; every group on the list with first >= tau is re-scored as well
//      (the list is sorted, so that is a prefix; the k-th score only rises afterwards, so one extension suffices) -- what
//      rounds 1-2 did with a fixed K' and a second round of three kernels, and why k = 50 needed that round for every query;
//   5. groups whose `second` could still reach the k-th score get their other three quads re-scored (rare);
//   6. CERTIFICATE: every row that was not re-scored has scan value <= m = max(first of the best group not re-scored,
//      final thetac), hence exact score <= m + eps, eps a worst-case bound (scan_eps).  If the k-th exact score is not
//      > m + eps -- massive ties, e.g. the zero vector the reference returns for empty text (hf/embeddings.py:47-48), a list
//      that overflowed, or more than kMaxRescore groups within eps -- the query is flagged for the exhaustive path.
//   7. the workgroup leaves the query's scan state (count, thetac, class slots) zeroed for the next launch of the slot.
// ------------------------------------------------------------------------------------------------------
constexpr int kListLds = kNoFilterGroups;   // groups of one query the finish ranks in LDS
constexpr int kMaxRescore = 192;            // groups re-scored (tagged quads) per query
constexpr int kMaxExpand = 24;              // groups whose other three quads are re-scored
constexpr int kFinThreads = 512;            // 8 waves and <= 40 KiB of LDS, two workgroups per CU (see fin_kernel): the 1024 queries of a small-shard
                                            // launch are finished in ONE round of workgroups (16 waves / 55 KiB: two rounds, 0.14 ms)
constexpr int kCandRows = 4 * kMaxRescore + 12 * kMaxExpand;
constexpr int kMaxKFast = 128;              // deepest k of this path; beyond: exhaustive path for every query

struct FinArgs {
    const float4* xb;
    const float* q;            // [nq, d]
    const unsigned* max_norm2_bits;  // [0] max |x|^2, [1] max |x - bf16(x)|^2 (float bits)
    double* out64;             // [nq, k]
    float* out32;              // [nq, k] or null
    int64_t* out_ids;          // [nq, k]
    int* flags;                // [nq]
    int* host_flags;           // [nq] a copy of the flags in pinned host memory, or null (hipidx_search's one-query path)
    int* arrivals;             // [nq] exhaustive-path arrival counters, zeroed here
    unsigned long long* fallback_counter;   // queries sent to the exhaustive path
    unsigned long long* extend_counter;     // queries whose re-scored prefix had to be extended (step 4)
    unsigned long long* work_counters;      // [3] sums over queries: list entries written by the scan, entries ranked, groups re-scored
    u32* slots;
    u32* thetac;
    u32* count;
    const Cand* list;
    int64_t ntotal, id_base;
    int d, P, k, mode, filter;
};

// NT threads: 512 (8 waves) for shallow k, 1024 (16 waves) for k >= 32, where re-scoring k + k/2 groups one after the other
// in 8 waves is most of a finish workgroup's time (k = 50: 229 us per 256-query launch with 8 waves).
template <int METRIC, int NT>
// (NT, 4): four waves per SIMD, i.e. TWO of these 8-wave workgroups per CU.  Left alone the compiler takes 142
// VGPRs -- three waves per SIMD, ONE workgroup per CU -- and a finish workgroup is a ~0.15 ms chain of dependent steps
// that only other workgroups on the CU can hide: on the CUs the next scan leaves it the finish of 1024
// queries then lasted as long as the scan itself.  128 VGPRs cost 36 bytes of scratch per lane.
__global__ __launch_bounds__(NT, 4) void fin_kernel(FinArgs a)
{
    __shared__ float qv[kMaxDPad];
    // the unsorted list (lkey / lsec) is dead once it has been ranked into skey / ssec: the re-scored rows (ck / ci) overlay it
    constexpr int kOverlay = kCandRows * 16 > kListLds * 12 ? kCandRows * 16 : kListLds * 12;
    __shared__ __attribute__((aligned(16))) unsigned char overlay[kOverlay];
    u64* lkey = reinterpret_cast<u64*>(overlay);
    float* lsec = reinterpret_cast<float*>(overlay + kListLds * 8);
    u64* ck = reinterpret_cast<u64*>(overlay);
    i64* ci = reinterpret_cast<i64*>(overlay + kCandRows * 8);
    __shared__ u64 skey[kListLds];
    __shared__ float ssec[kListLds];
    __shared__ u64 tk[kMaxKFast];
    __shared__ i64 ti[kMaxKFast];
    __shared__ double dred[32];
    __shared__ int expand[kMaxExpand];
    __shared__ int s_n, s_r1, s_expand;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x;
    const int dpad = a.P * 8;
    const int k = a.k;

    // query into LDS (zero padded), its exact |q|^2 and |q - bf16(q)|^2 (the scan's query tile holds bf16(q), RNE)
    double qpart = 0.0, dpart = 0.0;
    for (int c = tid; c < dpad; c += NT) {
        const float vf = c < a.d ? a.q[(int64_t)q * a.d + c] : 0.f;
        qv[c] = vf;
        const double v = (double)vf, dv = v - (double)(float)(__bf16)vf;
        qpart += v * v;
        dpart += dv * dv;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { qpart += __shfl_xor(qpart, off); dpart += __shfl_xor(dpart, off); }
    if (lane == 0) { dred[wave] = qpart; dred[16 + wave] = dpart; }   // NT / 64 <= 16 waves
    if (tid == 0) { s_n = 0; s_r1 = 0; s_expand = 0; }
    if (tid < k) { tk[tid] = 0; ti[tid] = -1; }
    const u32 cnt_raw = a.count[q];
    const u32 theta = a.filter ? a.thetac[q] : 0u;
    __syncthreads();
    double qn2 = 0.0, dq2 = 0.0;
    for (int w = 0; w < NT / 64; ++w) { qn2 += dred[w]; dq2 += dred[16 + w]; }
    const double eps = scan_eps<METRIC>(dpad, a.mode, qn2, (double)__uint_as_float(a.max_norm2_bits[0]), dq2,
                                        (double)__uint_as_float(a.max_norm2_bits[1]));

    // 1. list -> LDS
    const int n_in = (int)min(cnt_raw, (u32)kCandCap);
    const Cand* src = a.list + (size_t)q * kCandCap;
    for (int i = tid; i < n_in; i += NT) {
        const Cand e = src[i];
        if ((u32)(e.key >> 32) >= theta) {
            const int p = atomicAdd(&s_n, 1);
            if (p < kListLds) { lkey[p] = e.key; lsec[p] = e.sec; }
        }
    }
    __syncthreads();
    // 7. (early: everything of the scan state has been read) leave the slot clean for its next launch
    if (tid == 0) { a.count[q] = 0; a.thetac[q] = 0; }
    if (tid < kClasses) a.slots[((size_t)(q >> 6) * kClasses + tid) * 64 + (q & 63)] = 0;
    const int n = s_n;
    bool flag = cnt_raw > (u32)kCandCap || n > kListLds;
    int R = 0;
    bool extended = false;
    u64 kth_key = 0;

    // rows of groups [g0, g1) of the sorted list -> ck / ci[4j ..]
    auto rescore_groups = [&](int g0, int g1) {
        for (int j = g0 + wave; j < g1; j += NT / 64) {
            const u64 e = skey[j];
            const u32 gid = packed_index(e);
            const int g = (int)(__float_as_uint(packed_value(e)) & 3u);
            const int64_t blk = gid >> 1;
            const int r0 = 8 * g + 4 * (int)(gid & 1);
            const double s = rescore4<METRIC>(a.xb, a.P, blk, r0, qv);
            const i64 row = blk * kRowsPerBlock + r0 + (lane & 3);
            if (lane < 4) {
                ck[4 * j + lane] = row < a.ntotal ? ord64(METRIC == HIPRAG_METRIC_IP ? s : -s) : 0ull;
                ci[4 * j + lane] = row;
            }
        }
    };
    // exact top-k of ck / ci[0 .. nc) under (key desc, id asc) by counting -> tk / ti (cleared first); returns the k-th key
    auto topk_rows = [&](int nc) -> u64 {
        __syncthreads();
        if (tid < k) { tk[tid] = 0; ti[tid] = -1; }
        __syncthreads();
        for (int i = tid; i < nc; i += NT) {
            const u64 mk = ck[i];
            if (mk == 0) continue;
            const i64 mi = ci[i];
            int r = 0;
            for (int j = 0; j < nc; ++j) {
                const u64 ok = ck[j];
                r += (ok > mk || (ok == mk && ci[j] < mi)) ? 1 : 0;
            }
            if (r < k) { tk[r] = mk; ti[r] = mi; }
        }
        __syncthreads();
        return tk[k - 1];
    };

    if (!flag) {
        // 2. rank by counting
        for (int i = tid; i < n; i += NT) {
            const u64 mine = lkey[i];
            int r = 0;
            for (int j = 0; j < n; ++j) r += lkey[j] > mine ? 1 : 0;
            skey[r] = mine;
            ssec[r] = lsec[i];
        }
        __syncthreads();
        // 3. first batch
        const int R0 = min(n, k + max(8, k >> 1));
        rescore_groups(0, R0);
        R = R0;
        kth_key = topk_rows(4 * R);
        // 4. extend to every listed group that can still matter
        {
            const double t = kth_on_scan_scale<METRIC>(kth_key, qn2) - eps;
            float tf = t > -3.0e38 ? (float)t : -FLT_MAX;
            if ((double)tf > t) tf = nextafterf(tf, -INFINITY);
            for (int j = R0 + tid; j < n; j += NT)
                if (packed_value(skey[j]) >= tf) atomicMax(&s_r1, j + 1);
            __syncthreads();
            const int R1 = s_r1;
            if (R1 > R0) {
                extended = true;
                if (R1 > kMaxRescore) flag = true;
                else {
                    rescore_groups(R0, R1);
                    R = R1;
                    kth_key = topk_rows(4 * R);
                }
            }
        }
    }
    if (!flag) {
        // 5. other quads of groups whose `second` could still reach the k-th score
        const double kth = kth_on_scan_scale<METRIC>(kth_key, qn2);
        for (int j = tid; j < R; j += NT) {
            const float m2 = ssec[j];
            if (m2 > -1.0e38f && !(kth > (double)m2 + eps)) {
                const int p = atomicAdd(&s_expand, 1);
                if (p < kMaxExpand) expand[p] = j;
            }
        }
        __syncthreads();
        const int ne = s_expand;
        if (ne > kMaxExpand) flag = true;
        else if (ne > 0) {
            for (int x = wave; x < ne; x += NT / 64) {
                const u64 e = skey[expand[x]];
                const u32 gid = packed_index(e);
                const int tagged = (int)(__float_as_uint(packed_value(e)) & 3u);
                const int64_t blk = gid >> 1;
                int o = 4 * R + 12 * x;
                for (int g = 0; g < 4; ++g) {
                    if (g == tagged) continue;
                    const int r0 = 8 * g + 4 * (int)(gid & 1);
                    const double s = rescore4<METRIC>(a.xb, a.P, blk, r0, qv);
                    const i64 row = blk * kRowsPerBlock + r0 + (lane & 3);
                    if (lane < 4) {
                        ck[o + lane] = row < a.ntotal ? ord64(METRIC == HIPRAG_METRIC_IP ? s : -s) : 0ull;
                        ci[o + lane] = row;
                    }
                    o += 4;
                }
            }
            kth_key = topk_rows(4 * R + 12 * ne);
        }
    }
    if (!flag) {
        // 6. certificate
        float m = R < n ? packed_value(skey[R]) : -FLT_MAX;
        if (theta != 0) m = fmaxf(m, unord32(theta));
        if (m > -1.0e38f && !(kth_on_scan_scale<METRIC>(kth_key, qn2) > (double)m + eps)) flag = true;
    }
    if (tid < k) write_result<METRIC>(a.out64, a.out32, a.out_ids, (int64_t)q * k + tid, tk[tid], ti[tid], a.id_base);
    if (tid == 0) {
        a.flags[q] = flag ? 1 : 0;
        if (a.host_flags) a.host_flags[q] = flag ? 1 : 0;
        a.arrivals[q] = 0;
        if (flag) atomicAdd(a.fallback_counter, 1ull);
        else if (extended) atomicAdd(a.extend_counter, 1ull);
        atomicAdd(a.work_counters + 0, (unsigned long long)cnt_raw);
        atomicAdd(a.work_counters + 1, (unsigned long long)n);
        atomicAdd(a.work_counters + 2, (unsigned long long)R);
    }
}

// For k beyond what the finish holds every query goes straight to the exhaustive path: exact, just not fast -- k in the
// hundreds is outside anything the reference asks for (top_chunks = 50, page_retriever.py:81).
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
    int mine = 0;
    for (int q = tid; q < a.nq; q += kSelThreads) mine |= a.flags[q];
    if (!__syncthreads_or(mine)) return;  // the common case: a <= n_cu-workgroup launch that reads nq flags and leaves
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

// ------------------------------------------------------------------------------------------------------
// host object
// ------------------------------------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (function, size) instead of on every launch
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

// The start gate's wait (hipidx_gate_tail_dev): one wave polls the gate word -- memory-side reads, a plain load could be
// served from a stale L2 line for ever -- until the awaited scan has raised it, or until the bound is up.  The bound is
// what makes the gate safe where kernels are serialised (a profiler collecting counters, AMD_SERIALIZE_KERNEL, a debugger):
// there the awaited scan cannot start while this kernel runs, and an unbounded wait (hipStreamWaitValue64, which this
// replaces: it hung a --pmc collection) would never end.  The gate opens ~30 us after the previous scan has ended -- which
// the stream has waited for before it gets here -- so 1 ms of the 100 MHz wall clock is far beyond any real wait; a gate that
// times out only costs the overlap it was there for.
constexpr unsigned long long kGateTimeoutTicks = 100000ull;
__global__ __launch_bounds__(64) void gate_wait_kernel(unsigned long long* gate, unsigned long long target)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_fetch_or(gate, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (wall_clock64() - t0 > kGateTimeoutTicks) break;
        __builtin_amdgcn_s_sleep(20);
    }
}

struct DenseIndex {
    std::mutex mu;
    int device = 0;
    int d = 0, P = 0, metric = 0;
    int64_t ntotal = 0, cap_blocks = 0, id_base = 0;
    int n_cu = 256;
    int scan_cus = 256;       // workgroups of a scan launch (one per CU); hipidx_set_spare_cus leaves some CUs to other streams
    int scan_mode = 3;        // HIPRAG_SCAN_MODE: bf16 = 3 (bf16 filter copy; default), q64 = 2 (fp32 rows split on the fly)
    DevBuf xb, xh, norms, scalars;  // xh: bf16 filter copy; scalars: [0] max |x|^2 bits (u32), [1] max |x - bf16(x)|^2 bits,
                                // [2..3] fallback counter (u64), [4..5] extended-prefix counter
    // search workspace of one launch in flight
    struct Workspace {
        DevBuf list, state, flags, ek, ei;   // state: count[Q] | thetac[Q] | slots[Q / 64][kClasses][64]
        DevBuf qtile;                        // bf16 mode: the query-tile images of a multi-pass launch (qtile_kernel)
        int k = 0, q = 0;
        int64_t blocks = 0;
        int ev_idx = -1;
        bool dirty = false;                  // a scan ran without its finish: the scan state is not clean
        int waves = 8;
        unsigned long long seq = 0;          // sequence number of the scan last launched into this slot (0: none)
    };
    static constexpr int kSlots = 8;   // launches in flight: the scan of step i+1 runs beside the tails of steps i, i-1, ...
    Workspace ws[kSlots];
    DevBuf qbuf, o64, o32, oid;
    // pinned host staging of hipidx_search's few-query path (device-visible under the same address): the query goes up with
    // an asynchronous copy, the finish writes scores, ids and flags straight into host memory
    PinBuf pin_q, pin_o32, pin_oid, pin_flags;
    // start gate: see ScanArgs.  `started` holds the workgroup counter and, on a line of its own, the gate word; both only
    // ever grow.
    unsigned long long* gate = nullptr;
    DevBuf started;
    unsigned long long scan_seq = 0, started_total = 0;
    static constexpr int kFewQueries = 16;
    int launch_q = 256;       // queries one begin/finish pair takes (a multiple of 64): update_launch_q
    int launch_env = 0;       // HIPRAG_LAUNCH_QUERIES (0 = size launches by the index)
    // stats
    int64_t passes = 0, queries = 0, launches = 0;
    // timing: a ring of event pairs around the scan kernel, averaged by get_stats (no sync inside the search path)
    static constexpr int kEvRing = 512;
    bool timing = false;
    std::vector<hipEvent_t> evs;   // 2*kEvRing once timing was enabled
    DevBuf stamps;                 // [kEvRing][n_cu * 8 waves][2] in-kernel wall-clock ticks of the same launches
    int wall_khz = 100000;
    int64_t ev_count = 0;          // launches since timing was (re)enabled
    int ev_every = 1;
    std::vector<char> ev_set;      // [kEvRing] whether the launch in that ring slot was bracketed by events
    std::vector<int> ev_waves;     // [kEvRing] waves of that launch (its stamps occupy the first 2 * waves words of the ring slot)

    int64_t nblocks() const { return (ntotal + kRowsPerBlock - 1) / kRowsPerBlock; }
    unsigned* max_norm2_bits() { return scalars.as<unsigned>(); }
    unsigned* max_dx2_bits() { return scalars.as<unsigned>() + 1; }
    unsigned long long* fallback_counter() { return reinterpret_cast<unsigned long long*>(scalars.as<unsigned>() + 2); }
    unsigned long long* extend_counter() { return reinterpret_cast<unsigned long long*>(scalars.as<unsigned>() + 4); }
    unsigned long long* work_counters() { return reinterpret_cast<unsigned long long*>(scalars.as<unsigned>() + 6); }

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
        if (pipe_in) (void)hipEventDestroy(pipe_in);
        for (int i = 0; i < kSlots; ++i) {
            if (pipe_scanned[i]) (void)hipEventDestroy(pipe_scanned[i]);
            if (pipe_done[i]) (void)hipEventDestroy(pipe_done[i]);
        }
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        if (add_ev) (void)hipEventDestroy(add_ev);
    }

    int32_t init()
    {
        HR_CHECK_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        HR_CHECK_HIP(hipGetDeviceProperties(&prop, device));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        scan_cus = n_cu;
        if (const char* ms = getenv("HIPRAG_SCAN_MODE")) {
            if (!strcmp(ms, "bf16")) scan_mode = 3;
            else if (!strcmp(ms, "q64")) scan_mode = 2;
            else { set_error("HIPRAG_SCAN_MODE=%s: the scan has two operand modes, bf16 (default) and q64", ms); return HIPRAG_E_INVALID; }
        }
        const char* lq = getenv("HIPRAG_LAUNCH_QUERIES");
        launch_env = lq ? atoi(lq) : 0;
        update_launch_q();
        int32_t rc = scalars.reserve(64);
        if (rc) return rc;
        if ((rc = started.reserve(256))) return rc;
        HR_CHECK_HIP(hipMemset(started.p, 0, 256));
        gate = started.as<unsigned long long>() + 16;   // its own 128-byte line
        HR_CHECK_HIP(hipMemset(scalars.p, 0, 64));
        HR_CHECK_HIP(hipStreamSynchronize(nullptr));   // hipMemset of device memory may return before the fill has run
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
        // The buffer the scan STREAMS is allocated first.  How fast 2048 waves stream a buffer depends on which physical pages
        // it got: the same binary scanned 1M rows at 2.77-2.94 ms per launch with the filter copy allocated behind the 4 GB
        // of fp32 rows and at 2.60-2.66 with it allocated first (A/B in fresh processes, several boxes; the "allocation
        // lottery" of rounds 1-2, which blamed the scan's output).  Probing several candidate allocations and keeping the
        // fastest was measured too and bought nothing beyond this order.
        const bool stream_h = scan_mode == 3;
        HR_CHECK_HIP(hipMalloc(stream_h ? &nh : &nx, stream_h ? hbytes : xbytes));
        hipError_t e = hipMalloc(stream_h ? &nx : &nh, stream_h ? xbytes : hbytes);
        if (e == hipSuccess) e = hipMalloc(&nn, nbytes);
        if (e != hipSuccess) { if (nx) (void)hipFree(nx); if (nh) (void)hipFree(nh); if (nn) (void)hipFree(nn); HR_CHECK_HIP(e); }
        HR_CHECK_HIP(hipMemset(nx, 0, xbytes));
        HR_CHECK_HIP(hipMemset(nn, 0, nbytes));
        HR_CHECK_HIP(hipMemset(nh, 0, hbytes));
        // hipMemset of device memory is ordered on the null stream but may return before the fill has run, and the kernels
        // that write and read these buffers run on the callers' (non-blocking) streams
        HR_CHECK_HIP(hipStreamSynchronize(nullptr));
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

    static constexpr int kPassQ = 64;   // queries that share one read of the index
    // Small shards (an 8-GPU row split of 1M rows leaves 125 k per GPU): with 8 waves per workgroup a wave streams two or
    // three 32-row blocks per pass; 4-wave workgroups stream twice as many each and leave half of every SIMD's registers to
    // the tail kernels of earlier steps.  Measured in rounds 1-2 (1024 queries per launch, pipelined): 125 k rows 945 -> 902
    // us per step, 250 k 1640 -> 1590, 500 k about equal, 1M equal: 4 waves below 9 blocks per wave of the 8-wave partition.
    int scan_waves(int64_t nb) const { return (scan_mode == 3 && P % 32 == 0 && nb < (int64_t)scan_cus * 8 * 9) ? 4 : 8; }
    bool fast_k(int k) const { return k <= kMaxKFast; }

    // Passes per launch.  A launch chained behind its predecessor pays ~45-60 us of dispatch bubble and the tail of a
    // launch is a fixed cost too, so launches are sized to last about as long as four passes over a 1M x 1024 fp32 index
    // (2.6 ms) whatever the index size: 8 passes of the bf16 copy there, 16 (the cap) at half a million rows and below --
    // where a short launch would spend a quarter of its time outside the scan.  HIPRAG_LAUNCH_QUERIES fixes the size.
    void update_launch_q()
    {
        if (launch_env > 0) { launch_q = std::max(kPassQ, std::min(kMaxQ, launch_env / kPassQ * kPassQ)); return; }
        const double pass_bytes = (double)std::max<int64_t>(nblocks(), 1) * P * (scan_mode == 3 ? 512.0 : 1024.0);
        const int np = (int)std::lround(4.0 * 4.096e9 / pass_bytes);
        launch_q = std::max(4, std::min(16, np)) * kPassQ;
    }

    // Workspace of one slot for (up to launch_q queries, k), allocated on first use: an unused slot costs nothing.
    int32_t reserve_slot(int slot, int k)
    {
        Workspace& w = ws[slot];
        const int64_t nb = std::max<int64_t>(nblocks(), 1);
        if (k <= w.k && nb <= w.blocks && launch_q <= w.q) return HIPRAG_OK;
        const int kk = std::max(k, w.k);
        const int64_t nbb = std::max(nb, w.blocks);
        const int64_t nslices = (nbb * kRowsPerBlock + kExRows - 1) / kExRows;
        const int ekk = std::min(kk, kExRows);
        const size_t Q = (size_t)std::max(launch_q, w.q);
        int32_t rc;
        if ((rc = w.list.reserve(Q * kCandCap * sizeof(Cand)))) return rc;
        const size_t state_bytes = state_words(Q) * sizeof(u32);
        const bool fresh = w.state.bytes < state_bytes;
        if ((rc = w.state.reserve(state_bytes))) return rc;
        if (fresh) {   // the finish keeps it clean from here on
            // the fill is ordered on the null stream only and may still be pending when hipMemset returns; the scan that reads
            // this state runs on a non-blocking stream (a garbage bound drops candidates: seen once as a two-rank mismatch)
            HR_CHECK_HIP(hipMemset(w.state.p, 0, w.state.bytes));
            HR_CHECK_HIP(hipStreamSynchronize(nullptr));
        }
        if ((rc = w.flags.reserve(2 * Q * sizeof(int)))) return rc;  // flags[Q] + arrivals[Q]
        if (scan_mode == 3 && (rc = w.qtile.reserve((Q / 64) * (size_t)P * 64 * 16))) return rc;   // passes x (2 * P2 * 64) fragments
        if ((rc = w.ek.reserve(Q * nslices * ekk * sizeof(u64)))) return rc;
        if ((rc = w.ei.reserve(Q * nslices * ekk * sizeof(i64)))) return rc;
        w.k = kk;
        w.blocks = nbb;
        w.q = (int)Q;
        return HIPRAG_OK;
    }
    static size_t state_words(size_t Q) { return 2 * Q + (Q / 64) * kClasses * 64; }
    static u32* st_count(const Workspace& w) { return w.state.as<u32>(); }
    static u32* st_thetac(const Workspace& w) { return w.state.as<u32>() + w.q; }
    static u32* st_slots(const Workspace& w) { return w.state.as<u32>() + 2 * (size_t)w.q; }

    // phase 1 of a launch (<= launch_q queries): the scan, into workspace `slot`
    template <int METRIC>
    int32_t scan_pass(const float* q_dev, int nq, int k, int slot, hipStream_t st)
    {
        Workspace& w = ws[slot];
        const int64_t nb = nblocks();
        const int ev = (int)(ev_count % kEvRing);
        // HIP events cost two barrier packets per launch on the scan's stream; hipidx_enable_timing(h, n) brackets every n-th
        // launch only (the in-kernel stamps cover every launch either way)
        const bool use_ev = timing && ev_count % ev_every == 0;
        const bool run_scan = nb > 0 && fast_k(k);   // deeper k: every query takes the exhaustive path, nothing to scan for
        w.seq = 0;
        if (run_scan) {
            if (w.dirty) HR_CHECK_HIP(hipMemsetAsync(w.state.p, 0, w.state.bytes, st));   // a scan without its finish came before
            w.dirty = true;
            const int nw = scan_waves(nb);
            w.waves = nw;
            ScanArgs sa;
            sa.xb = xb.as<float4>(); sa.xh = xh.p; sa.q = q_dev; sa.norms = norms.as<float>();
            sa.slots = st_slots(w); sa.thetac = st_thetac(w); sa.count = st_count(w);
            sa.list = w.list.as<Cand>();
            sa.nblocks = nb; sa.ntotal = ntotal; sa.nq = nq; sa.d = d; sa.P = P;
            sa.filter = 2 * nb > kNoFilterGroups ? 1 : 0;
            const int64_t bpw = scan_blocks_per_wave(nb, (int64_t)scan_cus * nw);
            sa.ncls = (int)std::max<int64_t>(1, std::min<int64_t>(kClasses, (nb + bpw - 1) / bpw));   // waves that own blocks
            // (no fill of the stamp slot ahead of the launch -- a kernel of its own between two scans, 5-15 us: the host knows
            // how many waves the launch has and reads exactly their words)
            sa.stamps = timing ? stamps.as<unsigned long long>() + (size_t)ev * n_cu * kMaxScanWaves * 2 : nullptr;
            w.seq = ++scan_seq;
            started_total += (unsigned long long)scan_cus;
            sa.started = started.as<unsigned long long>(); sa.target = started_total; sa.seq = w.seq; sa.gate = gate;
            const size_t scan_lds = (size_t)P * 1024 + (size_t)2 * kStageHalf * sizeof(Cand) + 64 + (size_t)kThetaBack * 64 * 4;  // query tile + staged appends + control words + bounds
            const bool one_pass = nq <= kPassQ;
            sa.qtile = nullptr;
            if (scan_mode == 3 && !one_pass) {
                const int npass = (nq + kPassQ - 1) / kPassQ, per_pass = P * 64;   // 2 * P2 * 64 fragments of 16 B
                hipLaunchKernelGGL(qtile_kernel, dim3((unsigned)((per_pass + 255) / 256), (unsigned)npass), dim3(256), 0, st, q_dev, nq, d,
                                   P / 2, w.qtile.as<bf16x8>());
                sa.qtile = w.qtile.p;
            }
            void (*scan)(ScanArgs);
            if (scan_mode == 3) {
                // ring depth: 16 pieces where that divides the pieces of a block (d_pad / 16), else 8
                const bool r16 = (P / 2) % 16 == 0;
                if (nw == 4) scan = one_pass ? scan_bf16_kernel<METRIC, 4, 16, false> : scan_bf16_kernel<METRIC, 4, 16, true>;
                else if (r16) scan = one_pass ? scan_bf16_kernel<METRIC, 8, 16, false> : scan_bf16_kernel<METRIC, 8, 16, true>;
                else scan = one_pass ? scan_bf16_kernel<METRIC, 8, 8, false> : scan_bf16_kernel<METRIC, 8, 8, true>;
            } else {
                scan = one_pass ? scan_split_kernel<METRIC, 8, 16, false> : scan_split_kernel<METRIC, 8, 16, true>;
            }
            { int32_t lrc = ensure_lds(reinterpret_cast<const void*>(scan), scan_lds); if (lrc) return lrc; }
            if (use_ev) HR_CHECK_HIP(hipEventRecord(evs[2 * ev], st));
            hipLaunchKernelGGL(scan, dim3(scan_cus), dim3(nw * 64), scan_lds, st, sa);
            if (use_ev) HR_CHECK_HIP(hipEventRecord(evs[2 * ev + 1], st));
        }
        w.ev_idx = timing ? ev : -1;
        if (timing) { ev_set[ev] = use_ev && run_scan; ev_waves[ev] = run_scan ? scan_cus * w.waves : 0; ++ev_count; }
        HR_CHECK_HIP(hipGetLastError());
        passes += (nq + kPassQ - 1) / kPassQ;
        ++launches;
        queries += nq;
        return HIPRAG_OK;
    }

    // phase 2: the finish (list ranking, fp64 re-score, extension, certificate) + the exhaustive path; reads workspace `slot`
    // host_flags (pinned host memory, nq ints) != null: the finish also writes the queries' flags there and the exhaustive
    // check is NOT launched -- the caller synchronises, looks at the flags and calls exhaustive_pass only if one is set
    // (hipidx_search's few-query path: one launch and one kernel's run time less on the way to the host)
    template <int METRIC>
    int32_t finish_pass(const float* q_dev, int nq, int k, int slot, double* o64p, float* o32p, int64_t* oidp, hipStream_t st,
                        int* host_flags = nullptr)
    {
        Workspace& w = ws[slot];
        const int64_t nb = nblocks();
        int* flags = w.flags.as<int>();
        int* arrivals = flags + w.q;
        if (nb > 0 && fast_k(k)) {
            FinArgs fa;
            fa.xb = xb.as<float4>(); fa.q = q_dev; fa.max_norm2_bits = max_norm2_bits();
            fa.out64 = o64p; fa.out32 = o32p; fa.out_ids = oidp; fa.flags = flags; fa.host_flags = host_flags; fa.arrivals = arrivals;
            fa.fallback_counter = fallback_counter(); fa.extend_counter = extend_counter(); fa.work_counters = work_counters();
            fa.slots = st_slots(w); fa.thetac = st_thetac(w); fa.count = st_count(w);
            fa.list = w.list.as<Cand>();
            fa.ntotal = ntotal; fa.id_base = id_base; fa.d = d; fa.P = P; fa.k = k; fa.mode = scan_mode;
            fa.filter = 2 * nb > kNoFilterGroups ? 1 : 0;
            if (k >= 32) hipLaunchKernelGGL((fin_kernel<METRIC, 1024>), dim3(nq), dim3(1024), 0, st, fa);
            else hipLaunchKernelGGL((fin_kernel<METRIC, kFinThreads>), dim3(nq), dim3(kFinThreads), 0, st, fa);
            w.dirty = false;
        } else {
            hipLaunchKernelGGL(flag_all_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, flags, arrivals, fallback_counter(), nq);
            host_flags = nullptr;   // every query is flagged: nothing to look at first
        }
        if (host_flags) { HR_CHECK_HIP(hipGetLastError()); return HIPRAG_OK; }
        return exhaustive_pass<METRIC>(q_dev, nq, k, slot, o64p, o32p, oidp, st);
    }

    template <int METRIC>
    int32_t exhaustive_pass(const float* q_dev, int nq, int k, int slot, double* o64p, float* o32p, int64_t* oidp, hipStream_t st)
    {
        Workspace& w = ws[slot];
        int* flags = w.flags.as<int>();
        int* arrivals = flags + w.q;
        ExArgs ea;
        ea.xb = xb.as<float4>(); ea.q = q_dev; ea.flags = flags; ea.arrivals = arrivals;
        ea.ek = w.ek.as<u64>(); ea.ei = w.ei.as<i64>();
        ea.out64 = o64p; ea.out32 = o32p; ea.out_ids = oidp; ea.ntotal = ntotal; ea.id_base = id_base;
        ea.d = d; ea.P = P; ea.k = k; ea.kk = std::min(k, kExRows); ea.nq = nq;
        ea.nslices = (int)std::max<int64_t>(1, (ntotal + kExRows - 1) / kExRows);
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
            ev_waves.assign(kEvRing, 0);
            for (auto& e : evs) HR_CHECK_HIP(hipEventCreate(&e));
            int32_t src = stamps.reserve((size_t)kEvRing * n_cu * kMaxScanWaves * 2 * sizeof(unsigned long long));
            if (src) return src;
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

    int32_t finish_dev(const float* q_dev, int nq, int k, int slot, double* o64p, float* o32p, int64_t* oidp, hipStream_t st,
                       int* host_flags = nullptr)
    {
        return metric == HIPRAG_METRIC_IP ? finish_pass<HIPRAG_METRIC_IP>(q_dev, nq, k, slot, o64p, o32p, oidp, st, host_flags)
                                          : finish_pass<HIPRAG_METRIC_L2>(q_dev, nq, k, slot, o64p, o32p, oidp, st, host_flags);
    }

    // hipidx_search for a handful of queries (the reference's call shape: ONE, rag/storage/faiss_index.py:81-83): what is
    // not the scan has to be short.  Query up through pinned staging with an asynchronous copy; the finish writes scores,
    // ids and its flags straight into pinned host memory; one synchronise; the exhaustive check is launched only if the
    // finish flagged a query (it almost never does) -- against the general path: two blocking D2H copies, one blocking H2D
    // copy and one kernel less between the scan and the caller.
    int32_t search_few_host(const float* q_host, int nq, int k, float* out_scores, int64_t* out_ids)
    {
        int32_t rc;
        const size_t nk = (size_t)nq * k;
        if ((rc = qbuf.reserve((size_t)nq * d * sizeof(float)))) return rc;
        if ((rc = o64.reserve(nk * sizeof(double)))) return rc;
        if ((rc = pin_q.reserve((size_t)nq * d * sizeof(float)))) return rc;
        if ((rc = pin_o32.reserve(nk * sizeof(float)))) return rc;
        if ((rc = pin_oid.reserve(nk * sizeof(int64_t)))) return rc;
        if ((rc = pin_flags.reserve((size_t)nq * sizeof(int)))) return rc;
        if ((rc = prepare(k, 0))) return rc;
        memcpy(pin_q.p, q_host, (size_t)nq * d * sizeof(float));
        HR_CHECK_HIP(hipMemcpyAsync(qbuf.p, pin_q.p, (size_t)nq * d * sizeof(float), hipMemcpyHostToDevice, nullptr));
        int* hf = reinterpret_cast<int*>(pin_flags.p);
        float* h32 = reinterpret_cast<float*>(pin_o32.p);
        int64_t* hid = reinterpret_cast<int64_t*>(pin_oid.p);
        if ((rc = begin_dev(qbuf.as<float>(), nq, k, 0, nullptr))) return rc;
        for (int i = 0; i < nq; ++i) hf[i] = 1;   // a finish that does not write them (k beyond the fast path) launches the check itself
        if ((rc = finish_dev(qbuf.as<float>(), nq, k, 0, o64.as<double>(), h32, hid, nullptr, hf))) return rc;
        HR_CHECK_HIP(hipStreamSynchronize(nullptr));
        if (nblocks() > 0 && fast_k(k)) {
            int any = 0;
            for (int i = 0; i < nq; ++i) any |= hf[i];
            if (any) {
                rc = metric == HIPRAG_METRIC_IP
                         ? exhaustive_pass<HIPRAG_METRIC_IP>(qbuf.as<float>(), nq, k, 0, o64.as<double>(), h32, hid, nullptr)
                         : exhaustive_pass<HIPRAG_METRIC_L2>(qbuf.as<float>(), nq, k, 0, o64.as<double>(), h32, hid, nullptr);
                if (rc) return rc;
                HR_CHECK_HIP(hipStreamSynchronize(nullptr));
            }
        }
        memcpy(out_scores, h32, nk * sizeof(float));
        memcpy(out_ids, hid, nk * sizeof(int64_t));
        return HIPRAG_OK;
    }

    // Streams and events of search_dev's own pipeline (batches of more than one launch): created on first use
    hipStream_t pipe_tail = nullptr;
    hipEvent_t pipe_in = nullptr, pipe_scanned[kSlots] = {}, pipe_done[kSlots] = {};
    static constexpr int kPipeSpareCus = 48;   // hiprag/sharded.py SPARE_CUS: 32-64 measure the same

    int32_t pipe_init()
    {
        if (pipe_tail) return HIPRAG_OK;
        {   // the device's tail stream 0 (library-owned, shared by every index of the device: lib.cpp)
            void* t = nullptr;
            const int32_t trc = hiprag_tail_stream(device, 0, &t);
            if (trc) return trc;
            pipe_tail = (hipStream_t)t;
        }
        HR_CHECK_HIP(hipEventCreateWithFlags(&pipe_in, hipEventDisableTiming));
        for (int i = 0; i < kSlots; ++i) {
            HR_CHECK_HIP(hipEventCreateWithFlags(&pipe_scanned[i], hipEventDisableTiming));
            HR_CHECK_HIP(hipEventCreateWithFlags(&pipe_done[i], hipEventDisableTiming));
        }
        return HIPRAG_OK;
    }

    // A batch of several launches, pipelined the way hiprag/sharded.py pipelines steps: the scans chained on the
    // device's high-priority scan stream with CUs left out of their grids, the finish of launch j on the index's tail stream
    // behind the end of scan j AND the start of scan j + 1 (the start gate), the last one ungated; `st` -- where the queries
    // come from and the results are wanted -- is ahead of the first scan and behind the last finish.  Nothing synchronises
    // the host.  Uses every workspace slot: not to be mixed with a begin / finish pipeline in flight on the same index.
    int32_t search_dev_pipelined(const float* q_dev, int nq, int k, double* o64p, float* o32p, int64_t* oidp, hipStream_t st)
    {
        int32_t rc = pipe_init();
        if (rc) return rc;
        void* hpv = nullptr;
        if ((rc = hiprag_scan_stream(device, &hpv))) return rc;
        hipStream_t hp = (hipStream_t)hpv;
        HR_CHECK_HIP(hipEventRecord(pipe_in, st));
        HR_CHECK_HIP(hipStreamWaitEvent(hp, pipe_in, 0));
        HR_CHECK_HIP(hipStreamWaitEvent(pipe_tail, pipe_in, 0));   // (the output buffers: whatever `st` still does with them is ahead)
        const int saved_cus = scan_cus;
        scan_cus = std::min(scan_cus, std::max(1, n_cu - kPipeSpareCus));
        const int L = (nq + launch_q - 1) / launch_q;
        auto tails = [&](int j, bool gated) -> int32_t {
            const int slot = j % kSlots, o = j * launch_q, m = std::min(launch_q, nq - o);
            HR_CHECK_HIP(hipStreamWaitEvent(pipe_tail, pipe_scanned[slot], 0));
            if (gated && ws[slot].seq != 0)
                hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(64), 0, pipe_tail, gate, ws[slot].seq + 1);
            const int32_t frc = finish_dev(q_dev + (int64_t)o * d, m, k, slot, o64p + (int64_t)o * k, o32p ? o32p + (int64_t)o * k : nullptr,
                                           oidp + (int64_t)o * k, pipe_tail);
            if (frc) return frc;
            HR_CHECK_HIP(hipEventRecord(pipe_done[slot], pipe_tail));
            return HIPRAG_OK;
        };
        rc = HIPRAG_OK;
        int launched = 0;
        for (int j = 0; j < L && !rc; ++j) {
            const int slot = j % kSlots, o = j * launch_q, m = std::min(launch_q, nq - o);
            if (j >= kSlots) rc = hipStreamWaitEvent(hp, pipe_done[slot], 0) == hipSuccess ? HIPRAG_OK : HIPRAG_E_HIP;   // the slot's previous finish
            if (!rc) rc = prepare(k, slot);
            if (!rc) rc = begin_dev(q_dev + (int64_t)o * d, m, k, slot, hp);
            if (!rc) rc = hipEventRecord(pipe_scanned[slot], hp) == hipSuccess ? HIPRAG_OK : HIPRAG_E_HIP;
            if (!rc) { launched = j + 1; if (j >= 1) rc = tails(j - 1, true); }
        }
        scan_cus = saved_cus;
        // the last launched scan's finish has no scan behind it: no gate (nothing would open it)
        if (launched > 0) { const int32_t trc = tails(launched - 1, false); if (!rc) rc = trc; }
        if (launched > 0) HR_CHECK_HIP(hipStreamWaitEvent(st, pipe_done[(launched - 1) % kSlots], 0));   // the tail stream is in order: the last finish is behind all others
        if (rc == HIPRAG_E_HIP) set_error("a HIP call failed while enqueueing a pipelined search: %s", hipGetErrorString(hipGetLastError()));
        return rc;
    }

    int32_t search_dev(const float* q_dev, int nq, int k, double* o64p, float* o32p, int64_t* oidp, hipStream_t st)
    {
        int32_t rc = prepare(k, 0);
        if (rc) return rc;
        if (nq > launch_q && nblocks() > 0 && fast_k(k)) return search_dev_pipelined(q_dev, nq, k, o64p, o32p, oidp, st);
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


// ------------------------------------------------------------------------------------------------------
// IVF-Flat on top of the flat index (BASELINE north_star: "the flat-IP / IVF distance scan"; the reference itself builds
// faiss.IndexFlatL2 only, rag/storage/faiss_index.py:123).  The rows are stored PERMUTED by inverted list in an ordinary
// flat index (every list starts on a 32-row block; padding rows carry the original id -1), the nlist centroids in a second
// one.  A search is: exact top-nprobe of the query among the centroids (the flat search above) -> every probed list is cut
// into slices of kIvfRows rows, one workgroup per (query, list, slice) re-scores its rows in fp64 straight from the fp32
// rows (the same rescore4 as the flat finish: a row's score is the same bits in both indexes) and keeps its best k ->
// the canonical merge of the partial lists (hiprag_merge_topk_dev).  Approximate by construction unless nprobe = nlist,
// where every row is scored and the result equals the flat index's bit for bit (tests/test_ivf_gpu.py).
// Bound: HBM -- rows probed x d_pad x 4 bytes per query; there is nothing for 64 queries to share (each probes its own
// lists), which is why the flat scan wins for batches and IVF for one query at a time.
// ------------------------------------------------------------------------------------------------------
constexpr int kIvfRows = 256;   // rows per workgroup of the probe kernel
struct IvfArgs {
    const float4* xb;
    const float* q;        // [nq, d]
    const i64* probe;      // [nq, nprobe] list ids from the centroid search (-1 = no such list)
    const i64* offs;       // [nlist + 1] first stored row of every list (multiples of 32)
    const i64* orig;       // [stored rows] original id, -1 for padding
    double* ps;            // [nprobe * smax][nq][k] partial scores
    i64* pi;               //                         partial ids
    int d, P, k, nq, nprobe, smax;
};

template <int METRIC>
__global__ __launch_bounds__(256) void ivf_probe_kernel(IvfArgs a)
{
    __shared__ u64 keys[kIvfRows];
    __shared__ i64 ids[kIvfRows];
    __shared__ KeyId red[2 * 4];
    __shared__ float qv[kMaxDPad];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.y, part = blockIdx.x;
    const int j = part / a.smax, sl = part - j * a.smax;
    const int dpad = a.P * 8;
    const i64 list = a.probe[(i64)q * a.nprobe + j];
    i64 lo = 0, hi = 0;
    if (list >= 0) {
        lo = a.offs[list] + (i64)sl * kIvfRows;
        hi = min(a.offs[list + 1], lo + kIvfRows);
    }
    const int n = hi > lo ? (int)(hi - lo) : 0;     // workgroup-uniform
    double* ps = a.ps + ((i64)part * a.nq + q) * a.k;
    i64* pi = a.pi + ((i64)part * a.nq + q) * a.k;
    if (n == 0) {
        for (int r = tid; r < a.k; r += 256) { ps[r] = METRIC == HIPRAG_METRIC_IP ? -DBL_MAX : DBL_MAX; pi[r] = -1; }
        return;
    }
    for (int c = tid; c < dpad; c += 256) qv[c] = c < a.d ? a.q[(i64)q * a.d + c] : 0.f;
    for (int c = tid; c < kIvfRows; c += 256) { keys[c] = 0; ids[c] = -1; }
    __syncthreads();
    for (int g = wave; g * 4 < n; g += 4) {
        const i64 row0 = lo + (i64)g * 4;               // lists start on 32-row blocks and slices on 256 rows: quad-aligned
        const double s = rescore4<METRIC>(a.xb, a.P, row0 / kRowsPerBlock, (int)(row0 % kRowsPerBlock), qv);
        const i64 row = row0 + (lane & 3);
        if (lane < 4) {
            const i64 oid = row < hi ? a.orig[row] : -1;
            keys[g * 4 + lane] = oid >= 0 ? ord64(METRIC == HIPRAG_METRIC_IP ? s : -s) : 0ull;
            ids[g * 4 + lane] = oid;
        }
    }
    __syncthreads();
    wg_topk_rounds<256>(keys, ids, (n + 3) & ~3, a.k, red, [&](int r, u64 kk, i64 id) {
        ps[r] = kk ? (METRIC == HIPRAG_METRIC_IP ? unord64(kk) : -unord64(kk)) : (METRIC == HIPRAG_METRIC_IP ? -DBL_MAX : DBL_MAX);
        pi[r] = kk ? id : -1;
    });
}

struct IvfIndex {
    std::mutex mu;
    std::shared_ptr<DenseIndex> rows, cents;
    DevBuf offs, orig, probe64, probe_ids, ps, pi;
    int nlist = 0;
    i64 maxlen = 0;        // longest list, in stored rows
    i64 probed_rows = 0, searches = 0;   // stats: stored rows of the probed lists, queries
    std::vector<i64> offs_host;
};

Registry<IvfIndex>& ivf_reg()
{
    static Registry<IvfIndex> r;
    return r;
}

#define GET_INDEX(h)                                                       \
    std::shared_ptr<DenseIndex> ix = reg().get(h);                         \
    if (!ix) { set_error("unknown dense index handle %llu", (unsigned long long)(h)); return HIPRAG_E_HANDLE; } \
    std::lock_guard<std::mutex> guard(ix->mu);                             \
    HR_CHECK_HIP(hipSetDevice(ix->device))

}  // namespace

size_t clear_dense_registry() { ivf_reg().clear(); return reg().clear(); }
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
    *out_n = DenseIndex::kPassQ;
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
    // takes effect with the next launch: the finish reads a launch's lists and bounds, never its partition, and the stamp
    // buffer is sized for the whole chip
    ix->scan_cus = ix->n_cu - n;
    return HIPRAG_OK;
}

int32_t hipidx_gate_tail_dev(uint64_t h, int32_t slot, void* stream)
{
    GET_INDEX(h);
    HR_REQUIRE(slot >= 0 && slot < DenseIndex::kSlots, "slot must be in 0..7");
    const unsigned long long seq = ix->ws[slot].seq;
    if (!ix->gate || seq == 0) return HIPRAG_OK;
    // only a scan that is already on its way can open the gate: a wait for one that has not been launched is a mistake of the
    // caller's (it would merely run into the wait's time limit)
    HR_REQUIRE(ix->scan_seq > seq, "hipidx_gate_tail_dev: no scan has been launched after the one of slot %d", slot);
    hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ix->gate, seq + 1);
    HR_CHECK_HIP(hipGetLastError());
    return HIPRAG_OK;
}

int32_t hipidx_get_spare_cus(uint64_t h, int32_t* out_n)
{
    GET_INDEX(h);
    HR_REQUIRE(out_n, "null out");
    *out_n = ix->n_cu - ix->scan_cus;
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
    if (nq <= DenseIndex::kFewQueries) return ix->search_few_host(q_host, nq, k, out_scores, out_ids);
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
        HR_CHECK_HIP(hipMemcpy(&rbq, ix->extend_counter(), sizeof(rbq), hipMemcpyDeviceToHost));
        out->roundb_queries = (int64_t)rbq;
        unsigned long long wc[3] = {0, 0, 0};
        HR_CHECK_HIP(hipMemcpy(wc, ix->work_counters(), sizeof(wc), hipMemcpyDeviceToHost));
        out->list_entries = (int64_t)wc[0];
        out->ranked_entries = (int64_t)wc[1];
        out->rescored_groups = (int64_t)wc[2];
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
            const size_t per = (size_t)ix->n_cu * kMaxScanWaves * 2;
            std::vector<unsigned long long> hst((size_t)n * per);
            HR_CHECK_HIP(hipMemcpy(hst.data(), ix->stamps.p, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<std::pair<unsigned long long, unsigned long long>> se((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                unsigned long long lo = ~0ull, hi = 0;
                const size_t nwaves = std::min<size_t>((size_t)std::max(ix->ev_waves[(size_t)i], 0), per / 2);   // the launch's own waves
                for (size_t w = 0; w < nwaves; ++w) {
                    lo = std::min(lo, hst[(size_t)i * per + 2 * w]);
                    hi = std::max(hi, hst[(size_t)i * per + 2 * w + 1]);
                }
                se[(size_t)i] = {lo, hi};
            }
            double dsum = 0.0, gsum = 0.0;
            for (int64_t i = 0; i < n; ++i) dsum += (double)(se[(size_t)i].second - se[(size_t)i].first);
            const bool ordered = ix->ev_count <= DenseIndex::kEvRing;
            for (int64_t i = 0; ordered && i + 1 < n; ++i) gsum += (double)((long long)se[(size_t)i + 1].first - (long long)se[(size_t)i].second);
            out->avg_scan_wall_ms = (float)(dsum / n / ix->wall_khz);
            out->avg_scan_gap_ms = ordered && n > 1 ? (float)(gsum / (n - 1) / ix->wall_khz) : 0.f;
        }
    }
    return HIPRAG_OK;
}

// ---- IVF-Flat ----------------------------------------------------------------------------------------------------------
int32_t hipivf_create(uint64_t rows_h, uint64_t centroids_h, const int64_t* list_offsets_host, const int64_t* orig_ids_host,
                      int32_t nlist, uint64_t* out_handle)
{
    HR_REQUIRE(out_handle && list_offsets_host && orig_ids_host && nlist > 0, "bad hipivf_create arguments");
    std::shared_ptr<DenseIndex> rows = reg().get(rows_h), cents = reg().get(centroids_h);
    if (!rows || !cents) { set_error("unknown dense index handle"); return HIPRAG_E_HANDLE; }
    HR_REQUIRE(rows->d == cents->d && rows->metric == cents->metric && rows->device == cents->device,
               "rows and centroids must agree in dimension, metric and device");
    HR_REQUIRE(cents->ntotal == nlist, "the centroid index holds %lld rows, nlist is %d", (long long)cents->ntotal, nlist);
    HR_REQUIRE(list_offsets_host[0] == 0 && list_offsets_host[nlist] == rows->ntotal, "list offsets must cover the stored rows [0, %lld)",
               (long long)rows->ntotal);
    auto iv = std::make_shared<IvfIndex>();
    iv->rows = rows; iv->cents = cents; iv->nlist = nlist;
    iv->offs_host.assign(list_offsets_host, list_offsets_host + nlist + 1);
    for (int l = 0; l < nlist; ++l) {
        const i64 len = list_offsets_host[l + 1] - list_offsets_host[l];
        HR_REQUIRE(len >= 0 && list_offsets_host[l] % kRowsPerBlock == 0, "list %d must start on a 32-row block and not be negative", l);
        iv->maxlen = std::max(iv->maxlen, len);
    }
    HR_CHECK_HIP(hipSetDevice(rows->device));
    int32_t rc;
    if ((rc = iv->offs.reserve((size_t)(nlist + 1) * 8))) return rc;
    if ((rc = iv->orig.reserve((size_t)std::max<i64>(rows->ntotal, 1) * 8))) return rc;
    HR_CHECK_HIP(hipMemcpy(iv->offs.p, list_offsets_host, (size_t)(nlist + 1) * 8, hipMemcpyHostToDevice));
    HR_CHECK_HIP(hipMemcpy(iv->orig.p, orig_ids_host, (size_t)rows->ntotal * 8, hipMemcpyHostToDevice));
    *out_handle = ivf_reg().put(iv);
    return HIPRAG_OK;
}

int32_t hipivf_destroy(uint64_t h)
{
    std::shared_ptr<IvfIndex> iv = ivf_reg().get(h);
    if (!iv) { set_error("unknown IVF handle"); return HIPRAG_E_HANDLE; }
    {
        std::lock_guard<std::mutex> guard(iv->mu);
        (void)hipSetDevice(iv->rows->device);
        (void)hipDeviceSynchronize();
    }
    ivf_reg().erase(h);
    return HIPRAG_OK;
}

int32_t hipivf_search_dev(uint64_t h, const float* q_dev, int32_t nq, int32_t k, int32_t nprobe, double* out_scores64_dev,
                          float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    std::shared_ptr<IvfIndex> iv = ivf_reg().get(h);
    if (!iv) { set_error("unknown IVF handle"); return HIPRAG_E_HANDLE; }
    std::lock_guard<std::mutex> guard(iv->mu);
    HR_REQUIRE(nq >= 0 && k > 0 && k <= kIvfRows, "k must be in 1..%d (got %d)", kIvfRows, k);
    HR_REQUIRE(nprobe > 0 && nprobe <= kMaxK, "nprobe must be in 1..%d (got %d)", kMaxK, nprobe);
    if (nq == 0) return HIPRAG_OK;
    HR_REQUIRE(q_dev && out_scores64_dev && out_ids_dev, "null device pointer");
    DenseIndex& R = *iv->rows;
    DenseIndex& C = *iv->cents;
    HR_CHECK_HIP(hipSetDevice(R.device));
    hipStream_t st = (hipStream_t)stream;
    const int np = std::min(nprobe, iv->nlist);
    const int smax = (int)std::max<i64>(1, (iv->maxlen + kIvfRows - 1) / kIvfRows);
    const int parts = np * smax;
    const int qchunk = std::max(1, std::min(nq, 1024));
    int32_t rc;
    if ((rc = iv->probe64.reserve((size_t)qchunk * np * 8))) return rc;
    if ((rc = iv->probe_ids.reserve((size_t)qchunk * np * 8))) return rc;
    if ((rc = iv->ps.reserve((size_t)parts * qchunk * k * 8))) return rc;
    if ((rc = iv->pi.reserve((size_t)parts * qchunk * k * 8))) return rc;
    {
        std::lock_guard<std::mutex> gr(R.mu);
        if ((rc = R.wait_adds_stream(st))) return rc;
    }
    for (int o = 0; o < nq; o += qchunk) {
        const int m = std::min(qchunk, nq - o);
        const float* qo = q_dev + (i64)o * R.d;
        {   // coarse quantiser: the exact flat search of the query among the centroids
            std::lock_guard<std::mutex> gc(C.mu);
            if ((rc = C.search_dev(qo, m, np, iv->probe64.as<double>(), nullptr, iv->probe_ids.as<int64_t>(), st))) return rc;
        }
        IvfArgs a;
        a.xb = R.xb.as<float4>(); a.q = qo; a.probe = iv->probe_ids.as<i64>(); a.offs = iv->offs.as<i64>(); a.orig = iv->orig.as<i64>();
        a.ps = iv->ps.as<double>(); a.pi = iv->pi.as<i64>(); a.d = R.d; a.P = R.P; a.k = k; a.nq = m; a.nprobe = np; a.smax = smax;
        if (R.metric == HIPRAG_METRIC_IP) hipLaunchKernelGGL(ivf_probe_kernel<HIPRAG_METRIC_IP>, dim3(parts, m), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(ivf_probe_kernel<HIPRAG_METRIC_L2>, dim3(parts, m), dim3(256), 0, st, a);
        HR_CHECK_HIP(hipGetLastError());
        if ((rc = hiprag_merge_topk_dev(iv->ps.as<double>(), iv->pi.as<int64_t>(), parts, m, k, k, (int64_t)m * k, R.metric,
                                        out_scores64_dev + (i64)o * k, out_scores_dev ? out_scores_dev + (i64)o * k : nullptr,
                                        out_ids_dev + (i64)o * k, stream)))
            return rc;
    }
    iv->searches += nq;
    return HIPRAG_OK;
}

int32_t hipivf_info(uint64_t h, int32_t* out_nlist, int64_t* out_stored_rows, int64_t* out_longest_list)
{
    std::shared_ptr<IvfIndex> iv = ivf_reg().get(h);
    if (!iv) { set_error("unknown IVF handle"); return HIPRAG_E_HANDLE; }
    HR_REQUIRE(out_nlist && out_stored_rows && out_longest_list, "null out");
    *out_nlist = iv->nlist;
    *out_stored_rows = iv->rows->ntotal;
    *out_longest_list = iv->maxlen;
    return HIPRAG_OK;
}

}  // extern "C"
