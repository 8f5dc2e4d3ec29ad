// gemm256.h -- the batch GEMM of the encoder: C = A[M,K] * W[N,K]^T on 256 x 256 x 64 tiles, included by encoder.hip.
//
// Why a second tiled kernel: the 128 x 128 kernel (gemm_bf16_kernel) reads 16 KiB of LDS fragments per 32 MFMAs, drains
// its one prefetch stage (`vmcnt(0)` + barrier) every k-tile and pays the first-tile latency and the C-tile pass through
// LDS once per 16 k-tiles at K = 1024 -- 0.61-0.96 PFLOP/s on the encoder's shapes (r01).  This one follows the CDNA4
// playbook for MFMA-bound loops at one workgroup per CU:
//   * 8 waves = 2 groups of 4 (one wave of each group per SIMD) run the k-loop HALF A PHASE APART: while one group issues
//     its LDS fragment reads and LDS-DMA staging (the "load" half-step), the other runs 16 MFMAs (the "compute" half-step);
//     two raw s_barriers per phase keep them interleaved, so each SIMD's matrix pipe always has one wave feeding it.
//   * a wave owns 128 x 64 outputs (acc = 128 registers): 24 ds_read_b128 per 64 MFMAs instead of 32 per 64.
//   * staging never drains: half-slabs (16 KiB) of k-tile t+1 / t+2 are issued at fixed phases of k-tile t and waited for
//     with ONE counted `s_waitcnt vmcnt(4)` per k-tile, 3-4 phases after their issue (hazard table below).
//   * the workgroup is PERSISTENT: it walks output tiles (XCD-aware order) and the staging stream runs straight on into the
//     next tile's first k-tiles, so the first-tile latency is paid once per workgroup, not once per tile.
//   * the epilogue is wave-private: each wave transposes its accumulators through its own 4 KiB of LDS (no workgroup
//     barrier) into 16-byte row-major vectors; both wave groups run it at the same time (the SIMD's vector pipe issues for
//     two waves), then re-stagger.
// MFMA orientation: D[p][q] = sum_k P[p][k] Q[q][k]; a lane ends up with 4 CONSECUTIVE p of one q.  Row-major outputs want
// 4 consecutive n per lane, so P = W (n) and Q = A (m); the transposed V output of the QKV projection wants 4 consecutive
// m, so those tiles swap the roles (P = A, Q = W) -- same main loop, other pointers.
//
// LDS: [buffer 0: P lo | P hi | Q lo | Q hi][buffer 1: ...][8 x 4 KiB epilogue scratch] = 160 KiB, ONE __shared__ array
// (a second object makes hipcc drain vmcnt before every fragment read).  Half-slab image = 128 rows x 128 B, chunk kc of
// row r at slot kc ^ (r & 7) (tile_unit): conflict-free for the DMA writes and the ds_read_b128 fragment reads.
//
// Hazards (V = k-tile being computed from buffer b = V & 1; "interval" = time between two consecutive barrier events;
// group 0 loads in interval 8V+2ph and computes in 8V+2ph+1, group 1 half a phase later):
//   reads of buffer b:  Q halves in load(ph0), load(ph1); P lo (group 0) / P hi (group 1) in load(ph0), load(ph2)
//   ph0 of V issues   Q hi, P hi of V+1 -> buffer b^1  (last read by group 1 in load(ph2) of V-1, two barriers earlier)
//   ph3 of V issues   Q lo, P lo of V+2 -> buffer b    (last read by group 0 in load(ph2) of V, done before its ph3)
//   ph3 of V waits    vmcnt(4): everything but the 4 loads just issued has landed = all of V+1; every reader of V+1
//                     (load(ph0) of V+1) has passed a barrier that every issuer reached after that wait.
#pragma once

constexpr int G2_T = 256, G2_BK = 64, G2_THREADS = 512;
constexpr int G2_SLAB = 65536;
constexpr int G2_SCRATCH = 2 * G2_SLAB;
constexpr int G2_LDS_BYTES = 2 * G2_SLAB + 8 * 4096;

#define G2_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

template <int EPI>
__global__ __launch_bounds__(G2_THREADS) void gemm256_kernel(GemmArgs g, int nbm, int nbn)
{
    __shared__ __attribute__((aligned(16))) char lds[G2_LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;      // wr: P half and wave group; wc: 64-row quarter of Q
    const int r16 = lane & 15, kq = lane >> 4;
    const int nk = g.K / G2_BK;                   // even (launch condition)
    const int tiles = nbm * nbn;
    const int G = gridDim.x;
    // XCD-aware persistent order: block ids go round-robin over the 8 XCDs; give each XCD a contiguous run of tile indices
    // (4 column tiles x 8 row tiles share 12 operand panels in one L2)
    const int vw = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;

    auto tile_origin = [&](int L, int& m0, int& n0) {
        const int per = 8 * nbn;
        const int sr = L / per, l = L - sr * per;
        const int h = min(8, nbm - sr * 8);
        const int c = l / h;
        m0 = (sr * 8 + (l - c * h)) * G2_T;
        n0 = c * G2_T;
    };
    auto tile_swapped = [&](int n0) { return EPI == EPI_QKV && n0 >= 2 * g.H; };

    // ---- staging stream ------------------------------------------------------------------------------------------------
    const int lane_g = (lane >> 3) * g.K + (((lane & 7) ^ (lane >> 3)) << 3);    // element offset of this lane's 16 B
    const bf16* sP = nullptr;      // row base (incl. k offset) of the cursor k-tile's P / Q operand
    const bf16* sQ = nullptr;
    int sL = vw, skt = 0;
    bool svalid = sL < tiles;
    auto cursor_tile = [&]() {
        int m0, n0;
        tile_origin(sL, m0, n0);
        const bf16* pa = g.A + (size_t)m0 * g.K;
        const bf16* pw = g.W + (size_t)n0 * g.K;
        const bool sw = tile_swapped(n0);
        sP = sw ? pa : pw;
        sQ = sw ? pw : pa;
    };
    auto cursor_next = [&]() {
        ++skt;
        sP += G2_BK;
        sQ += G2_BK;
        if (skt == nk) {
            skt = 0;
            sL += G;
            svalid = sL < tiles;
            if (svalid) cursor_tile();
        }
    };
    auto issue_half = [&](const bf16* rowbase, int half, char* dst) {   // 16 KiB = 16 wave-instructions, 2 per wave
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int inst = wave * 2 + u;
            const bf16* src = rowbase + (size_t)(half * 128 + inst * 8) * g.K + lane_g;
            __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr_t)(dst + inst * 1024), 16, 0, 0);
        }
    };
    if (!svalid) return;
    cursor_tile();
    // prologue: k-tile 0 whole, the lo halves of k-tile 1 (the state every later tile boundary is in as well)
    issue_half(sP, 0, lds + 0 * 16384);
    issue_half(sP, 1, lds + 1 * 16384);
    issue_half(sQ, 0, lds + 2 * 16384);
    issue_half(sQ, 1, lds + 3 * 16384);
    cursor_next();
    if (svalid) {
        issue_half(sQ, 0, lds + G2_SLAB + 2 * 16384);
        issue_half(sP, 0, lds + G2_SLAB + 0 * 16384);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G2_BAR();

    // fragment read addresses: row r16 of a 16-row tile, 16-B chunk (4 ks + kq) ^ (r16 & 7); ks = 1 flips bit 2 of the chunk
    const int loff0 = r16 * 128 + ((kq ^ (r16 & 7)) << 4);
    const char* pA[2][2];   // [buffer][ks]
    const char* pB[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            pA[b][ks] = lds + b * G2_SLAB + wr * 16384 + (loff0 ^ (ks << 6));
            pB[b][ks] = lds + b * G2_SLAB + 32768 + wc * 8192 + (loff0 ^ (ks << 6));
        }

    for (int L = vw; L < tiles; L += G) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (wr == 1) G2_BAR();   // stagger: group 1 runs half a phase behind group 0

#define G2_LDA(B, sub)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
        a[i][ks] = *reinterpret_cast<const bf16x8*>(pA[B][ks] + ((sub) * 4 + i) * 2048)
#define G2_LDB(B, dst, sub)                                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
        dst[j][ks] = *reinterpret_cast<const bf16x8*>(pB[B][ks] + ((sub) * 2 + j) * 2048)
#define G2_MFMA(as, bfrag, bs)                                                                  \
    do {                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                    \
                acc[(as) * 4 + i][(bs) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(      \
                    a[i][ks], bfrag[j][ks], acc[(as) * 4 + i][(bs) * 2 + j], 0, 0, 0);          \
        __builtin_amdgcn_s_setprio(0);                                                          \
    } while (0)
#define G2_KTILE(B)                                                                             \
    do {                                                                                        \
        /* ph0: fragments b0, a0; stage the hi halves of the next k-tile into the other buffer */ \
        G2_LDB(B, b0, 0);                                                                       \
        G2_LDA(B, 0);                                                                           \
        if (svalid) {                                                                           \
            issue_half(sQ, 1, lds + ((B) ^ 1) * G2_SLAB + 3 * 16384);                           \
            issue_half(sP, 1, lds + ((B) ^ 1) * G2_SLAB + 1 * 16384);                           \
        }                                                                                       \
        G2_BAR();                                                                               \
        G2_MFMA(0, b0, 0);                                                                      \
        G2_BAR();                                                                               \
        /* ph1 */                                                                               \
        G2_LDB(B, b1, 1);                                                                       \
        G2_BAR();                                                                               \
        G2_MFMA(0, b1, 1);                                                                      \
        G2_BAR();                                                                               \
        /* ph2 */                                                                               \
        G2_LDA(B, 1);                                                                           \
        G2_BAR();                                                                               \
        G2_MFMA(1, b1, 1);                                                                      \
        G2_BAR();                                                                               \
        /* ph3: no fragment reads (b0 is still in registers); stage the lo halves of the k-tile after next into THIS */ \
        /* buffer, then the one counted wait of the k-tile */                                   \
        cursor_next();                                                                          \
        if (svalid) {                                                                           \
            issue_half(sQ, 0, lds + (B) * G2_SLAB + 2 * 16384);                                 \
            issue_half(sP, 0, lds + (B) * G2_SLAB + 0 * 16384);                                 \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                    \
        } else {                                                                                \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
        }                                                                                       \
        G2_BAR();                                                                               \
        G2_MFMA(1, b0, 0);                                                                      \
        G2_BAR();                                                                               \
    } while (0)

        {
            bf16x8 a[4][2], b0[2][2], b1[2][2];
#pragma unroll 1
            for (int kt = 0; kt < nk; kt += 2) {
                G2_KTILE(0);
                G2_KTILE(1);
            }
        }
        if (wr == 0) G2_BAR();   // group 1 finishes its last compute half-step: both groups are level again

        // ---- epilogue: wave-private transposition through 4 KiB of LDS, 16-byte global vectors ------------------------------
        int m0, n0;
        tile_origin(L, m0, n0);
        char* sc = lds + G2_SCRATCH + wave * 4096;
        const int rrow = lane >> 4, rc = lane & 15;   // read-back: row 4t + rrow, 16-B chunk rc
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        if (EPI == EPI_RESID) {
            // normal orientation: acc[i][j][r] = C[m0 + wc*64 + j*16 + r16][n0 + wr*128 + i*16 + 4kq + r]
            const int mb = m0 + wc * 64, nb = n0 + wr * 128;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        *reinterpret_cast<f32x4*>(sc + r16 * 256 + (((ii * 4 + kq) ^ r16) << 4)) = acc[h2 * 4 + ii][j];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int row = t * 4 + rrow;
                        const f32x4 c = *reinterpret_cast<const f32x4*>(sc + row * 256 + ((rc ^ row) << 4));
                        const int m = mb + j * 16 + row, n = nb + h2 * 64 + rc * 4;
                        if (m < g.M) {
                            const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
                            const bf16x4 rs = *reinterpret_cast<const bf16x4*>(g.resid + (size_t)m * g.N + n);
                            float4 o;
                            o.x = (c[0] + b.x) + (float)rs[0];
                            o.y = (c[1] + b.y) + (float)rs[1];
                            o.z = (c[2] + b.z) + (float)rs[2];
                            o.w = (c[3] + b.w) + (float)rs[3];
                            *reinterpret_cast<float4*>(g.out_f32 + (size_t)m * g.N + n) = o;
                        }
                    }
                }
        } else if (EPI == EPI_QKV && tile_swapped(n0)) {
            // V third, swapped roles: acc[i][j][r] = C[m0 + wr*128 + i*16 + 4kq + r][n0 + wc*64 + j*16 + r16];
            // stored transposed [seq, head, d, S]: LDS rows = n (d), 128 consecutive m (s) per row
            const int mb = m0 + wr * 128, nb = n0 + wc * 64;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bias = g.bias[nb + j * 16 + r16];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    bf16x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (bf16)(acc[i][j][r] + bias);
                    *reinterpret_cast<bf16x4*>(sc + r16 * 256 + (((i * 2 + (kq >> 1)) ^ r16) << 4) + (kq & 1) * 8) = v;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = t * 4 + rrow;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(sc + row * 256 + ((rc ^ row) << 4));
                    const int n = nb + j * 16 + row, m = mb + rc * 8;
                    if (m < g.M) {
                        const int hn = n - 2 * g.H;
                        const int head = hn >> 6, dd = hn & 63;
                        const int seq = m / g.S, s0 = m - seq * g.S;
                        *reinterpret_cast<bf16x8*>(g.vt + (((size_t)seq * g.heads + head) * 64 + dd) * g.S + s0) = v;
                    }
                }
            }
        } else {
            // row-major bf16 outputs (GELU; q and k of QKV), normal orientation
            const int mb = m0 + wc * 64, nb = n0 + wr * 128;
            float4 bias4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) bias4[i] = *reinterpret_cast<const float4*>(g.bias + nb + i * 16 + 4 * kq);
            const int which = EPI == EPI_QKV ? n0 / g.H : 0;      // 0 = q (scaled by 1/8, exact), 1 = k
            const float scale = (EPI == EPI_QKV && which == 0) ? 0.125f : 1.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float x[4] = {acc[i][j][0] + bias4[i].x, acc[i][j][1] + bias4[i].y, acc[i][j][2] + bias4[i].z,
                                  acc[i][j][3] + bias4[i].w};
                    bf16x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = EPI == EPI_GELU ? (bf16)(0.5f * x[r] * (1.f + erf_as(x[r] * 0.70710678118654752f)))
                                               : (bf16)(x[r] * scale);
                    *reinterpret_cast<bf16x4*>(sc + r16 * 256 + (((i * 2 + (kq >> 1)) ^ r16) << 4) + (kq & 1) * 8) = v;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = t * 4 + rrow;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(sc + row * 256 + ((rc ^ row) << 4));
                    const int m = mb + j * 16 + row, n = nb + rc * 8;
                    if (m < g.M) {
                        if (EPI == EPI_GELU) {
                            *reinterpret_cast<bf16x8*>(g.out_bf16 + (size_t)m * g.N + n) = v;
                        } else {
                            const int hn = n - which * g.H;
                            const int head = hn >> 6, dd = hn & 63;
                            const int seq = m / g.S, s = m - seq * g.S;
                            bf16* dst = (which == 0 ? g.q : g.k) + ((((size_t)seq * g.heads + head) * g.S + s) * 64 + dd);
                            *reinterpret_cast<bf16x8*>(dst) = v;
                        }
                    }
                }
            }
        }
    }
#undef G2_LDA
#undef G2_LDB
#undef G2_MFMA
#undef G2_KTILE
}
