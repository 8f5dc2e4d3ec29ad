// gemm256.h -- the batch GEMM of the encoder: C = A[M,K] * W[N,K]^T on 256 x 256 x 64 tiles, included by encoder.hip.
//
// Why a second tiled kernel: the 128 x 128 kernel (gemm_bf16_kernel) reads 16 KiB of LDS fragments per 32 MFMAs, drains
// its one prefetch stage (`vmcnt(0)` + barrier) every k-tile and pays the first-tile latency and the C-tile pass through
// LDS once per 16 k-tiles at K = 1024 -- 0.61-0.96 PFLOP/s on the encoder's shapes (r01).  This one follows the CDNA4
// playbook for MFMA-bound loops at one workgroup per CU:
//   * 8 waves = 2 groups of 4 (one wave of each group per SIMD) run the k-loop HALF A PHASE APART: while one group issues
//     its LDS fragment reads and LDS-DMA staging (the "load" half-step), the other runs 32 MFMAs (the "compute" half-step);
//     two raw s_barriers per phase keep them interleaved, so each SIMD's matrix pipe always has one wave feeding it.
//   * a wave owns 128 x 64 outputs (acc = 128 registers): 24 ds_read_b128 per 64 MFMAs instead of 32 per 64.
//   * staging: the half-slabs (16 KiB) of k-tile t+1 are issued at fixed points of k-tile t and waited for with counted
//     `s_waitcnt vmcnt(N)` two or more barrier intervals after their issue (hazard table below).
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
// Schedule: TWO phases per k-tile (32 MFMAs per wave each, four barrier events per k-tile -- a barrier round trip of the eight
// waves costs ~100 cycles, measured with the MFMAs and the staging compiled out; four 16-MFMA phases spent a third of
// their time in it):
//   phase A  reads: all of the wave's Q fragments (8) + P row tiles 0-3 (8)       MFMAs: P tiles 0-3 x Q tiles 0-3
//   phase B  reads: P row tiles 4-7 (8)                                           MFMAs: P tiles 4-7 x Q tiles 0-3
// Hazards (V = k-tile being computed from buffer b = V & 1, cursor = V+1 staged into buffer b^1; "interval" = time between
// two consecutive barrier events; group 0 runs load(A), compute(A), load(B), compute(B) in intervals 4V .. 4V+3, group 1 one
// interval later):
//   reads of buffer b^1 for V-1 completed: Q during interval 4V-2, P lo during 4V-1, P hi (group 1) during 4V
//   load(A) of V issues Q lo, Q hi, P lo of V+1 (6 DMA instructions per wave); load(B) of V issues P hi of V+1 (2)
//   W1  end of load(B) of V, after the 2 new ones: vmcnt(2) -> Q and P lo of V+1 have landed (issued two intervals ago)
//   P hi of V+1 is read by group 1 only, first in its load(A) of V+1 (interval 4V+5):
//   W2  group 1, end of compute(B) of V: vmcnt(0)            (its own portion, issued in interval 4V+3)
//   W3  group 0, end of load(A) of V+1, after the 6 new ones: vmcnt(6)   (its portion, issued in interval 4V+2)
//   every reader passes a barrier that every issuer reached after its wait.
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
    int lane_g = (lane >> 3) * g.K + (((lane & 7) ^ (lane >> 3)) << 3);    // element offset of this lane's 16 B
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
    // prologue: k-tile 0 whole (the state every later tile boundary is in as well: the cursor k-tile has landed)
    issue_half(sP, 0, lds + 0 * 16384);
    issue_half(sP, 1, lds + 1 * 16384);
    issue_half(sQ, 0, lds + 2 * 16384);
    issue_half(sQ, 1, lds + 3 * 16384);
    cursor_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2_BAR();

    // fragment read addresses: row r16 of a 16-row tile, 16-B chunk (4 ks + kq) ^ (r16 & 7); ks = 1 flips bit 2 of the chunk
    int loff0 = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);

    for (int L = vw; L < tiles; L += G) {
        // the per-lane constants of the k-loop are RECOMPUTED at every tile start from an opaque copy of the lane id, so that
        // nothing of the k-loop has to survive the epilogue in a register: a spilled constant comes back through a scratch
        // load, hipcc answers that with `s_waitcnt vmcnt(0)` at its first use, and that use sits inside the k-loop
        {
            int lane_k = lane;
            asm volatile("" : "+v"(lane_k));
            lane_g = (lane_k >> 3) * g.K + (((lane_k & 7) ^ (lane_k >> 3)) << 3);
            loff0 = (lane_k & 15) * 128 + (((lane_k >> 4) ^ (lane_k & 7)) << 4);
        }
        const char* pA[2][2];   // [buffer][ks]
        const char* pB[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                pA[b][ks] = lds + b * G2_SLAB + wr * 16384 + (loff0 ^ (ks << 6));
                pB[b][ks] = lds + b * G2_SLAB + 32768 + wc * 8192 + (loff0 ^ (ks << 6));
            }
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (wr == 1) G2_BAR();   // stagger: group 1 runs half a phase behind group 0

#define G2_LDA(B, sub)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
        a[i][ks] = *reinterpret_cast<const bf16x8*>(pA[B][ks] + ((sub) * 4 + i) * 2048)
#define G2_LDB(B)                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
        b[j][ks] = *reinterpret_cast<const bf16x8*>(pB[B][ks] + j * 2048)
#define G2_MFMA(as)                                                                             \
    do {                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
                acc[(as) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], b[j][ks], acc[(as) * 4 + i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                          \
    } while (0)
#define G2_KTILE(B)                                                                             \
    do {                                                                                        \
        /* phase A */                                                                           \
        G2_LDB(B);                                                                              \
        G2_LDA(B, 0);                                                                           \
        if (svalid) {                                                                           \
            issue_half(sQ, 0, lds + ((B) ^ 1) * G2_SLAB + 2 * 16384);                           \
            issue_half(sQ, 1, lds + ((B) ^ 1) * G2_SLAB + 3 * 16384);                           \
            issue_half(sP, 0, lds + ((B) ^ 1) * G2_SLAB + 0 * 16384);                           \
            if (wr == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       /* W3 */        \
        } else if (wr == 0) {                                                                   \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
        }                                                                                       \
        G2_BAR();                                                                               \
        G2_MFMA(0);                                                                             \
        G2_BAR();                                                                               \
        /* phase B */                                                                           \
        G2_LDA(B, 1);                                                                           \
        if (svalid) {                                                                           \
            issue_half(sP, 1, lds + ((B) ^ 1) * G2_SLAB + 1 * 16384);                           \
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                    /* W1 */        \
            cursor_next();                                                                      \
        }                                                                                       \
        G2_BAR();                                                                               \
        G2_MFMA(1);                                                                             \
        if (wr == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           /* W2 */        \
        G2_BAR();                                                                               \
    } while (0)

        {
            bf16x8 a[4][2], b[4][2];
#pragma unroll 1
            for (int kt = 0; kt < nk; kt += 2) {
                G2_KTILE(0);
                G2_KTILE(1);
            }
        }
        // ---- epilogue: wave-private transposition through 4 KiB of LDS, 16-byte global vectors ------------------------------
        int m0, n0;
        tile_origin(L, m0, n0);
        char* sc = lds + G2_SCRATCH + wave * 4096;
        // Every lane-derived quantity of the epilogue is recomputed here from an OPAQUE copy of the lane id: left to itself
        // hipcc hoists the epilogue's address arithmetic out of the tile loop, keeps it live across the k-loop, spills, and
        // then answers the reload with an `s_waitcnt vmcnt(0)` inside the k-loop -- which drains the LDS-DMA stream every
        // k-tile.  (The k-loop itself needs 128 + 64 + ~20 registers.)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int r16 = lane_e & 15, kq = lane_e >> 4;
        const int rrow = lane_e >> 4, rc = lane_e & 15;   // read-back: row 4t + rrow, 16-B chunk rc
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        // bias of the bf16 epilogues: inline-asm loads (hand-counted, see the residual epilogue) issued BEFORE the levelling
        // barrier, so that for group 0 their round trip overlaps group 1's last compute half-step
        u32x4 bq8[8];
        if (EPI == EPI_GELU) {       // (the QKV epilogue has two forms and no registers to spare: its bias loads stay hipcc's)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float* bp = g.bias + n0 + wr * 128 + i * 16 + 4 * kq;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq8[i]) : "v"(bp) : "memory");
            }
        }
        if (wr == 0) G2_BAR();   // group 1 finishes its last compute half-step: both groups are level again

        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        (void)bq8; (void)rrow; (void)rc;
        if (EPI == EPI_RESID || EPI == EPI_RESID16) {
            // normal orientation: acc[i][j][r] = C[m0 + wc*64 + j*16 + r16][n0 + wr*128 + i*16 + 4kq + r].
            // Chunk (j, h2) = 16 rows x 64 columns of fp32 through the wave's scratch; read back as 8 consecutive columns per
            // lane (two ds_read_b128), so the residual comes in 16-byte loads and the bf16 form leaves in 16-byte stores: 16
            // loads + 16 (bf16) or 32 (fp32) stores per wave and tile -- the CU's vector-memory path takes ~70 clocks per
            // wave-instruction, which is what an epilogue that all eight waves run at once is bound by.
            // All global loads are inline asm with hand-counted waits: beside the LDS-DMA stream hipcc answers every ordinary
            // load with `vmcnt(0)`, which here also waits for every store issued so far -- one memory round trip per chunk.
            // Rows are never masked: M is a whole number of tiles for these two epilogues (launch condition).
            const int mb = m0 + wc * 64, nb = n0 + wr * 128;
            const int rb_row = lane_e >> 3, c8 = lane_e & 7;
            u32x4 bq[2][2];                 // bias[nb + h2*64 + c8*8 + 0..7] as two float4
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float* bp = g.bias + nb + h2 * 64 + c8 * 8 + u * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[h2][u]) : "v"(bp) : "memory");
                }
            u32x4 rq[2][2][2];              // residual of two row groups j in flight: [j & 1][h2][t], 8 bf16 each
            auto load_resid = [&](int j) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16* rp = g.resid + (size_t)(mb + j * 16 + t * 8 + rb_row) * g.N + nb + h2 * 64 + c8 * 8;
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rq[j & 1][h2][t]) : "v"(rp) : "memory");
                    }
            };
            load_resid(0);
            load_resid(1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // Wait for the residual of group j.  Younger than it in the queue are the four loads of group j+1 (if any) and
                // the stores of group j-1 (if any); the bias loads are older than everything.  Only the younger LOADS are
                // allowed for in the count: loads return in order among themselves, but a store's acknowledgement may overtake an
                // older load, so a count that also allowed for the stores could be met with a residual load still in flight.
                // With vmcnt(4): were a load of group j outstanding, the four of group j+1 behind it would be too -- five.
                if (j == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rq[1][0][0]), "+v"(rq[1][0][1]), "+v"(rq[1][1][0]), "+v"(rq[1][1][1]) :: "memory");
                else if (j == 0) asm volatile("s_waitcnt vmcnt(4)" : "+v"(rq[0][0][0]), "+v"(rq[0][0][1]), "+v"(rq[0][1][0]), "+v"(rq[0][1][1]), "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]) :: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" : "+v"(rq[j & 1][0][0]), "+v"(rq[j & 1][0][1]), "+v"(rq[j & 1][1][0]), "+v"(rq[j & 1][1][1]) :: "memory");
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        *reinterpret_cast<f32x4*>(sc + r16 * 256 + (((ii * 4 + kq) ^ r16) << 4)) = acc[h2 * 4 + ii][j];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = t * 8 + rb_row;
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(sc + row * 256 + (((2 * c8) ^ row) << 4));
                        const f32x4 c1 = *reinterpret_cast<const f32x4*>(sc + row * 256 + (((2 * c8 + 1) ^ row) << 4));
                        const u32x4 rr = rq[j & 1][h2][t];
                        const u32x4 b0 = bq[h2][0], b1 = bq[h2][1];
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // bf16 pair e of the residual: low half = element 2e, high half = element 2e + 1
                            const float r_lo = __uint_as_float(rr[e] << 16), r_hi = __uint_as_float(rr[e] & 0xffff0000u);
                            const float c_lo = (2 * e < 4) ? c0[2 * e] : c1[2 * e - 4];
                            const float c_hi = (2 * e + 1 < 4) ? c0[2 * e + 1] : c1[2 * e + 1 - 4];
                            const float b_lo = __uint_as_float((2 * e < 4) ? b0[2 * e] : b1[2 * e - 4]);
                            const float b_hi = __uint_as_float((2 * e + 1 < 4) ? b0[2 * e + 1] : b1[2 * e + 1 - 4]);
                            o[2 * e] = (c_lo + b_lo) + r_lo;
                            o[2 * e + 1] = (c_hi + b_hi) + r_hi;
                        }
                        const size_t at = (size_t)(mb + j * 16 + row) * g.N + nb + h2 * 64 + c8 * 8;
                        if (EPI == EPI_RESID16) {
                            bf16x8 v;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (bf16)o[e];
                            *reinterpret_cast<bf16x8*>(g.out_bf16 + at) = v;
                        } else {
                            *reinterpret_cast<float4*>(g.out_f32 + at) = make_float4(o[0], o[1], o[2], o[3]);
                            *reinterpret_cast<float4*>(g.out_f32 + at + 4) = make_float4(o[4], o[5], o[6], o[7]);
                        }
                    }
                }
                if (j + 2 < 4) load_resid(j + 2);
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
            if (EPI == EPI_GELU) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq8[0]), "+v"(bq8[1]), "+v"(bq8[2]), "+v"(bq8[3]), "+v"(bq8[4]), "+v"(bq8[5]),
                             "+v"(bq8[6]), "+v"(bq8[7]) :: "memory");
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + nb + i * 16 + 4 * kq);
                    bq8[i] = u32x4{__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w)};
                }
            }
            const int which = EPI == EPI_QKV ? n0 / g.H : 0;      // 0 = q (scaled by 1/8, exact), 1 = k
            const float scale = (EPI == EPI_QKV && which == 0) ? 0.125f : 1.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float x[4] = {acc[i][j][0] + __uint_as_float(bq8[i][0]), acc[i][j][1] + __uint_as_float(bq8[i][1]),
                                  acc[i][j][2] + __uint_as_float(bq8[i][2]), acc[i][j][3] + __uint_as_float(bq8[i][3])};
                    bf16x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = EPI == EPI_GELU ? (bf16)gelu_exact(x[r])
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
