// Step 5 of the destination-binned backward (boxattn_binned.h) on the matrix cores, for bf16
// storage: the scatter-add of a round of 64 records into the 32 pixels of a block IS a small
// matrix product
//
//     grad_value^T[c][pixel] += sum_k  G^T[c][k] * A^T[k][pixel]        k = record of the round
//
// with G = the records' upstream-gradient rows (bf16 as stored, exact) and A = the sparse
// 32 x 64 matrix of bilinear weight x attention weight (<= 4 non-zeros per record).  The VALU
// formulation (binned_accumulate_kernel) builds per-pixel entry lists with LDS atomics and walks
// them; its length is set by the LONGEST of the 32 lists of a round (2x the mean) and by ~750
// issued instructions per round.  Dense on v_mfma_f32_32x32x16_bf16 the same round is 8
// instructions of 32 cycles -- ten times the multiply-adds, a fraction of the issue slots, and
// no lists, ranks, atomics or imbalance at all.
//
// Precision: A is split into two bf16 terms, w = hi + lo (|w - hi - lo| <= 2^-17 |w|), and both
// products accumulate in the MFMA's fp32 accumulator; bf16 x bf16 products are exact in fp32.
// The result is the reference's sum to ~1e-5 relative per term, far inside the bf16 output
// rounding.  One difference in kind: a dense product multiplies every row of the round by
// every pixel's (mostly zero) weight, so a non-finite upstream gradient makes its whole block
// NaN instead of its <= 4 pixels.
//
// Operand layout (32x32x16, K = 16 records per instruction): lane l holds 8 consecutive k
// (k = 8 (l >> 5) + 0..7) of row / column l & 31 for BOTH operands, so both are kept
// K-contiguous in LDS: G^T[c][k] (the gathered rows are transposed while staging: the lanes of
// two neighbouring records swap halves with DPP and write whole dwords) and A^T[pixel][k] (each
// record lane scatters its <= 4 weights: hi term, product, lo term in place, product, clear --
// no zero-fill per round).  C/D: lane l = pixel l & 31, register r = channel
// (r & 3) + 8 (r >> 2) + 4 (l >> 5).
//
// Input: WIDE records {point id, x, y, attention weight} written by bin_kernel<.., WIDE> -- the
// record stream is read coalesced and carries everything but the upstream row, which is the
// only gather left (with 4-byte records the kernel gathered the location and the weight of
// every record too: 182 L1 line requests per round instead of 72, 62 us instead of 50, and
// 141 instead of 54 us on uniformly random locations).
// Measured at BoxeR-R50 COCO shapes (C2, bf16): 82 us (VALU kernel) -> 50 us; of the 50: row
// gathers + staging 17, weight scatter 8, MFMA 3.4 (BOXATTN_TUNE_MFMA_ABLATE).
//
// float32 storage (ST = float): the upstream rows are split into two bf16 terms as well,
// g = g_hi + g_lo (|g - g_hi - g_lo| <= 2^-16 |g|), staged one after the other through the same
// G^T tile; three products per K-step (g_hi a_hi, g_hi a_lo, g_lo a_hi; g_lo a_lo ~ 2^-25 is
// dropped).  Every term is accurate to ~2e-5 relative -- inside the operator's 1e-4 float32
// tolerance, not float32-exact like the VALU kernel.  Rows are fetched one round ahead (two would
// need 64 more registers).  Measured at C2: accumulate 103 -> 98 us, but the wide records cost
// 12 us more in the fill pass, so this flavour is opt-in (boxattn_set_variant(11)).
#pragma once
#include "boxattn_binned.h"

namespace boxattn {

#ifndef BOXATTN_TUNE_MFMA_ABLATE
#define BOXATTN_TUNE_MFMA_ABLATE 0   // timing experiments only (wrong results), bit mask: 1 no staging
#endif                               // writes, 2 no row gathers, 4 no MFMA, 8 no weight scatter, 16 rows gathered but not staged, 32 rows from a 4 KiB window
typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef float mfma_f32x16 __attribute__((ext_vector_type(16)));

// value of lane ^ D for D = 2, 4, 8 (inside a DPP row of 16 lanes; no LDS traffic)
template <int D> __device__ __forceinline__ unsigned pair_exchange(unsigned x)
{
    static_assert(D == 2 || D == 4 || D == 8, "lanes per upstream row");
    const int v = (int)x;
    if constexpr (D == 2) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    } else if constexpr (D == 8) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);   // row_ror:8
    } else {
        // row_shl:4 into the lanes 0-3 / 8-11 of each row, row_shr:4 into the lanes 4-7 / 12-15
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);
        return (unsigned)__builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);
    }
}

// bf16 rows of <= 32 channels: the register budget is held to 4 waves per SIMD (128 VGPRs; the
// kernel asked for 136 = 3 waves; no spills; C2 accumulate 56.8 -> 55.3 us, C2' 94.8 -> 92.6)
#ifndef BOXATTN_TUNE_MFMA_WPE
#define BOXATTN_TUNE_MFMA_WPE 4
#endif
template <typename ST, int C>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu((sizeof(ST) == 2 && C <= 32) ? BOXATTN_TUNE_MFMA_WPE : 1))) void binned_accumulate_mfma_kernel(
    const ST *__restrict__ grad_out, BinPlan plan, int S, int H, int Lq,
    const int4 *__restrict__ items, const int *__restrict__ n_items,
    const int *__restrict__ records, ST *__restrict__ grad_value,
    float *__restrict__ partials)
{
    constexpr bool F32 = sizeof(ST) == 4;
    static_assert(!F32 || C == 32, "float32 rows: 8 lanes per row");
    constexpr int BW = 8, PB = 32, R = 64;
    constexpr int CP = C < 32 ? 32 : C;            // operand rows: channels padded to the MFMA's 32
    constexpr int NCB = CP / 32;                   // 32-channel blocks
    constexpr int ROWB = C * (int)sizeof(ST);      // bytes of one upstream-gradient row
    constexpr int CPL = 16 / (int)sizeof(ST);      // channels per lane and fetch (8 bf16 / 4 fp32)
    constexpr int LPR = ROWB / 16;                 // lanes that fetch one row, 16 B each
    constexpr int RPP = 64 / LPR;                  // rows fetched per pass
    constexpr int NPASS = R / RPP;
    constexpr int GS = 68;    // ushorts per G^T row: 64 records + pad (34 dwords: transposing writes spread over the banks)
    constexpr int AS = 72;    // ushorts per A^T row (36 dwords: 16-byte aligned rows, conflict-free operand reads)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) unsigned short gt[CP * GS];
    // one A^T array: the hi term is scattered, multiplied, then overwritten in place by the lo term
    __shared__ __attribute__((aligned(16))) unsigned short at[PB * AS];

    // workgroup -> (slice, worker): all workers of a slice on one XCD (see binned_accumulate_kernel)
    const int n_slices = plan.n_slices, workers = gridDim.x;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = bid % 8, kq = bid / 8;
    const int per_xcd = (n_slices + 7) / 8;
    const int s = slice_on_xcd(xcd, kq % per_xcd, per_xcd);
    const int worker = kq / per_xcd;
    if (s >= n_slices || worker >= workers) return;
    const int b = s / H, h = s % H;
    const int lane = threadIdx.x;
    const int col = lane & 31, kb = lane >> 5;     // operand row / column, k-block
    const int n_it = n_items[2 * s];

    for (int i = lane; i < PB * AS / 2; i += 64) reinterpret_cast<unsigned int *>(at)[i] = 0u;
    if (C < CP)                                     // the padding channels stay zero
        for (int i = lane; i < CP * GS / 2; i += 64) reinterpret_cast<unsigned int *>(gt)[i] = 0u;
    wave_lds_sync();

    // (a workgroup with several items -- big, mostly empty maps -- has its next item in flight while
    // it works on the current one: an empty block is otherwise one dependent load per 2 KB stored)
    const int4 *my_items = items + (size_t)s * plan.item_cap;
    int4 item_n = my_items[min(worker, plan.item_cap - 1)];      // (list is heaviest first)
    for (int it = worker; it < n_it; it += workers) {
        const int4 item = item_n;                                               // heaviest first
        item_n = my_items[min(it + workers, plan.item_cap - 1)];
        const BlockGeo bg = unpack_block_geo((unsigned)item.x);
        BinLevel lv = plan.lv[0];
#pragma unroll
        for (int k = 1; k < kMaxBinLevels; ++k)
            if (k == bg.level) lv = plan.lv[k];
        const int oy = bg.oy, ox = bg.ox, bh = bg.bh, bw = bg.bw;
        // wide records {point id, x, y, attention weight} (bin_kernel<.., WIDE>), read coalesced
        const int4 *rec = reinterpret_cast<const int4 *>(records) + (size_t)s * plan.rec_cap;
        mfma_f32x16 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

        // software pipeline over rounds of 64 records: the records are read three rounds ahead,
        // their upstream-gradient rows two (into registers; staged into LDS once the current
        // round's operands have been read).  Idle lanes stage a zero row (and their
        // A columns stay zero).
        auto fetch_rec = [&](int rr) -> int4 {
            // (a non-temporal load here, to keep the stream out of the rows' way in L2: 50.7 -> 52.7 us)
            return rr + lane < item.z ? rec[rr + lane] : make_int4(-1, 0, 0, 0);   // id -1: idle lane
        };
        // bf16: rows of round r + 1 / r + 2 in flight; float32: `grow2` holds the CURRENT round's rows
        // (needed again for the lo term) and `grow` those of round r + 1
        u32x4 grow[NPASS], grow2[NPASS];
        auto fetch_rows = [&](const int4 &r, u32x4 (&grow)[NPASS]) {
            if (BOXATTN_TUNE_MFMA_ABLATE & 2) return;
            // (idle lanes, id -1, fetch row 0; it is zeroed when staged)
            const int row = (int)(((size_t)b * Lq + (max(r.x, 0) >> plan.lp_bits)) * H + h);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int j = ps * RPP + lane / LPR, piece = lane % LPR;
                int rj = __shfl(row, j, 64);
                if (BOXATTN_TUNE_MFMA_ABLATE & 32) rj = (rj & 63) + (int)((size_t)b * Lq * H);   // 64 hot rows
                grow[ps] = *reinterpret_cast<const u32x4 *>(grad_out + (size_t)rj * C + piece * CPL);
            }
        };
        // G^T[c][j] = row j, channel c.  Transposing 16-bit elements one ds_write_b16 at a time
        // was 17 of this kernel's 51 us; instead the lanes of records j and j + 1 (same 16-byte
        // piece) swap halves with DPP, and each writes whole dwords {G[j][c], G[j+1][c]}: the even
        // record's lane for the piece's even channels, the odd one's for the odd channels.
        // n_live: records of the round being staged; the rows of idle lanes are staged as ZEROS (in a
        // dense product 0 * Inf = NaN: a fetched row with a non-finite element would poison the block)
        auto stage_rows = [&](const u32x4 (&rows)[NPASS], int n_live, bool lo_term) {
            if (BOXATTN_TUNE_MFMA_ABLATE & 3) return;
            if (BOXATTN_TUNE_MFMA_ABLATE & 16) {    // rows gathered and waited for, not staged
                unsigned x = 0;
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) x ^= rows[ps][0] ^ rows[ps][1] ^ rows[ps][2] ^ rows[ps][3];
                if (x == 0x12345678u) gt[lane] = (unsigned short)x;
                return;
            }
            const int odd = (lane / LPR) & 1;
            const unsigned sel = odd ? 0x03020706u : 0x05040100u;   // v_perm_b32 byte selectors
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int j = ps * RPP + lane / LPR, piece = lane % LPR;
                unsigned int *dst =
                    reinterpret_cast<unsigned int *>(&gt[(piece * CPL + odd) * GS + (j & ~1)]);
                unsigned w[CPL / 2];               // the lane's channels as packed bf16 pairs
                if constexpr (F32) {
                    const float f0 = __uint_as_float(rows[ps][0]), f1 = __uint_as_float(rows[ps][1]);
                    const float f2 = __uint_as_float(rows[ps][2]), f3 = __uint_as_float(rows[ps][3]);
                    w[0] = pack_bf16x2(f0, f1);
                    w[1] = pack_bf16x2(f2, f3);
                    if (lo_term) {                 // g - g_hi, exact in float32
                        w[0] = pack_bf16x2(f0 - __uint_as_float(w[0] << 16), f1 - __uint_as_float(w[0] & 0xffff0000u));
                        w[1] = pack_bf16x2(f2 - __uint_as_float(w[1] << 16), f3 - __uint_as_float(w[1] & 0xffff0000u));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = rows[ps][i];
                }
                if (n_live < R) {                  // wave-uniform: only an item's last round has idle lanes
#pragma unroll
                    for (int i = 0; i < CPL / 2; ++i) w[i] = j < n_live ? w[i] : 0u;
                }
#pragma unroll
                for (int i = 0; i < CPL / 2; ++i) {
                    const unsigned own = w[i];
                    const unsigned oth = pair_exchange<LPR>(own);       // lane ^ LPR: record j ^ 1
                    dst[i * GS] = __builtin_amdgcn_perm(oth, own, sel);   // rows 2 i + odd, GS ushorts apart
                }
            }
        };
        // rows are gathered TWO rounds ahead (a wave with one round of rows in flight spent most
        // of its time waiting for them), records three
        int4 rec_c = fetch_rec(item.y), rec_n = fetch_rec(item.y + R), rec_n2 = fetch_rec(item.y + 2 * R);
        if constexpr (F32) {
            fetch_rows(rec_c, grow2);
            stage_rows(grow2, min(R, item.z - item.y), false);
        } else {
            fetch_rows(rec_c, grow);
            stage_rows(grow, min(R, item.z - item.y), false);
        }
        if (item.y + R < item.z) fetch_rows(rec_n, grow);
        for (int rr = item.y; rr < item.z; rr += R) {
            const int n = min(R, item.z - rr);
            const bool more = rr + R < item.z;     // wave-uniform
            int4 rec_n3 = make_int4(-1, 0, 0, 0);
            if (rr + 2 * R < item.z) {
                if constexpr (!F32) fetch_rows(rec_n2, grow2);
                rec_n3 = fetch_rec(rr + 3 * R);
            }
            const float2 xy_c = make_float2(__int_as_float(rec_c.y), __int_as_float(rec_c.z));
            const float a_c = __int_as_float(rec_c.w);
            // ---- lane = record: its <= 4 weights go to A^T[pixel][lane] as hi + lo bf16
            const Sample<float> sm = locate<float>(xy_c.x, xy_c.y, lv.H, lv.W);
            const float wk[4] = {sm.hh * sm.hw * a_c, sm.hh * sm.lw * a_c, sm.lh * sm.hw * a_c,
                                 sm.lh * sm.lw * a_c};
            const unsigned hi01 = pack_bf16x2(wk[0], wk[1]), hi23 = pack_bf16x2(wk[2], wk[3]);
            const unsigned lo01 = pack_bf16x2(wk[0] - __uint_as_float(hi01 << 16),
                                              wk[1] - __uint_as_float(hi01 & 0xffff0000u));
            const unsigned lo23 = pack_bf16x2(wk[2] - __uint_as_float(hi23 << 16),
                                              wk[3] - __uint_as_float(hi23 & 0xffff0000u));
            const unsigned short whi[4] = {(unsigned short)(hi01 & 0xffffu), (unsigned short)(hi01 >> 16),
                                           (unsigned short)(hi23 & 0xffffu), (unsigned short)(hi23 >> 16)};
            const unsigned short wlo[4] = {(unsigned short)(lo01 & 0xffffu), (unsigned short)(lo01 >> 16),
                                           (unsigned short)(lo23 & 0xffffu), (unsigned short)(lo23 >> 16)};
            int slot[4];                            // A^T element of corner k, -1: not in this block
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = sm.y0 + (k >> 1), xx = sm.x0 + (k & 1);
                const bool use = lane < n && sm.ok[k] && (unsigned)(yy - oy) < (unsigned)bh &&
                                 (unsigned)(xx - ox) < (unsigned)bw;
                slot[k] = use ? ((yy - oy) * BW + (xx - ox)) * AS + lane : -1;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (slot[k] >= 0 && !(BOXATTN_TUNE_MFMA_ABLATE & 8)) at[slot[k]] = whi[k];
            wave_lds_sync();
            // ---- the product: 4 K-steps of 16 records per 32-channel block, hi term then lo term
            mfma_bf16x8 g[R / 16][NCB];
#pragma unroll
            for (int t = 0; t < R / 16; ++t) {
                const int k0 = 16 * t + 8 * kb;
                const mfma_bf16x8 p_hi = __builtin_bit_cast(
                    mfma_bf16x8, *reinterpret_cast<const u32x4 *>(&at[col * AS + k0]));
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const u32x2 *gp = reinterpret_cast<const u32x2 *>(&gt[(cb * 32 + col) * GS + k0]);
                    const u32x2 g0 = gp[0], g1 = gp[1];
                    g[t][cb] = __builtin_bit_cast(mfma_bf16x8, u32x4{g0.x, g0.y, g1.x, g1.y});
                    if (!(BOXATTN_TUNE_MFMA_ABLATE & 4))
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[t][cb], p_hi, acc[cb], 0, 0, 0);
                    else
                        acc[cb][0] += __builtin_bit_cast(u32x4, g[t][cb])[0] + __builtin_bit_cast(u32x4, p_hi)[1];
                }
            }
            wave_lds_sync();                         // a wave's LDS operations execute in order
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (slot[k] >= 0 && !(BOXATTN_TUNE_MFMA_ABLATE & 8)) at[slot[k]] = wlo[k];
            wave_lds_sync();
#pragma unroll
            for (int t = 0; t < R / 16; ++t) {
                const int k0 = 16 * t + 8 * kb;
                const mfma_bf16x8 p_lo = __builtin_bit_cast(
                    mfma_bf16x8, *reinterpret_cast<const u32x4 *>(&at[col * AS + k0]));
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    if (!(BOXATTN_TUNE_MFMA_ABLATE & 4))
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[t][cb], p_lo, acc[cb], 0, 0, 0);
                    else
                        acc[cb][1] += __builtin_bit_cast(u32x4, p_lo)[2];
            }
            wave_lds_sync();
            if constexpr (F32) {
                // ---- third product: the lo term of the rows times the hi term of the weights
                stage_rows(grow2, n, true);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (slot[k] >= 0) at[slot[k]] = whi[k];
                wave_lds_sync();
#pragma unroll
                for (int t = 0; t < R / 16; ++t) {
                    const int k0 = 16 * t + 8 * kb;
                    const mfma_bf16x8 p_hi = __builtin_bit_cast(
                        mfma_bf16x8, *reinterpret_cast<const u32x4 *>(&at[col * AS + k0]));
                    const u32x2 *gp = reinterpret_cast<const u32x2 *>(&gt[col * GS + k0]);
                    const u32x2 g0 = gp[0], g1 = gp[1];
                    const mfma_bf16x8 g_lo = __builtin_bit_cast(mfma_bf16x8, u32x4{g0.x, g0.y, g1.x, g1.y});
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g_lo, p_hi, acc[0], 0, 0, 0);
                }
                wave_lds_sync();
            }
            // ---- clear this round's weights, stage the next round's rows (they have arrived)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (slot[k] >= 0 && !(BOXATTN_TUNE_MFMA_ABLATE & 8)) at[slot[k]] = 0;
            if (more) {
                if constexpr (F32) {
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) grow2[ps] = grow[ps];
                    stage_rows(grow2, min(R, item.z - rr - R), false);
                    if (rr + 2 * R < item.z) fetch_rows(rec_n2, grow);      // round r + 2, one round ahead
                } else {
                    stage_rows(grow, min(R, item.z - rr - R), false);
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) grow[ps] = grow2[ps];
                }
                rec_c = rec_n; rec_n = rec_n2; rec_n2 = rec_n3;
            }
            wave_lds_sync();
        }
        // ---- store.  Lane = pixel `col`; its registers hold channels 8 g + 4 kb + 0..3.
        if (item.w < 0) {
            // whole rows in the storage type: lanes l and l ^ 32 swap half of their packed
            // pairs, so that each writes two 16-byte pieces (8 channels) of the pixel's row
            const int py = col / BW, px = col % BW;
            const bool live = py < bh && px < bw;
            ST *dst = grad_value +
                      (((size_t)b * S + lv.start + (size_t)(oy + py) * lv.W + (ox + px)) * H + h) * C;
            if constexpr (F32) {                  // float32 rows: four 16-byte pieces per lane
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    if (live)
                        *reinterpret_cast<float4 *>(dst + 8 * g4 + 4 * kb) = make_float4(
                            acc[0][4 * g4], acc[0][4 * g4 + 1], acc[0][4 * g4 + 2], acc[0][4 * g4 + 3]);
            } else
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                unsigned pk[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pk[i] = pack_bf16x2(acc[cb][2 * i], acc[cb][2 * i + 1]);
                // kb = 0 keeps g = 0, 2 and needs the partner's halves of them; kb = 1: g = 1, 3
                const unsigned r0 = __shfl_xor(kb ? pk[0] : pk[2], 32, 64);
                const unsigned r1 = __shfl_xor(kb ? pk[1] : pk[3], 32, 64);
                const unsigned r2 = __shfl_xor(kb ? pk[4] : pk[6], 32, 64);
                const unsigned r3 = __shfl_xor(kb ? pk[5] : pk[7], 32, 64);
                const u32x4 lo_piece = kb ? u32x4{r0, r1, pk[2], pk[3]} : u32x4{pk[0], pk[1], r0, r1};
                const u32x4 hi_piece = kb ? u32x4{r2, r3, pk[6], pk[7]} : u32x4{pk[4], pk[5], r2, r3};
                const int c_lo = cb * 32 + 8 * kb, c_hi = cb * 32 + 16 + 8 * kb;
                if (live && c_lo < C) *reinterpret_cast<u32x4 *>(dst + c_lo) = lo_piece;
                if (live && c_hi < C) *reinterpret_cast<u32x4 *>(dst + c_hi) = hi_piece;
            }
        } else {
            float *dst = partials + (((size_t)s * plan.pslot_cap + item.w) * PB + col) * C;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = cb * 32 + 8 * g + 4 * kb;
                    if (c < C)
                        *reinterpret_cast<float4 *>(dst + c) =
                            make_float4(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2],
                                        acc[cb][4 * g + 3]);
                }
        }
    }
}

}  // namespace boxattn
