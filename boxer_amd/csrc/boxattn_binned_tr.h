// The matrix-core accumulate of the destination-binned backward for bf16 storage: one round of 64 records
// against the 32 pixels of a block is the product grad_value^T[c][pixel] += G^T[c][k] A^T[k][pixel] on
// v_mfma_f32_32x32x16_bf16 (A split into two bf16 terms), built around the instruction count.  The SQ
// counters of round 3 show every kernel of the step retiring one instruction per SIMD every
// ~4 cycles whatever its type -- scalar instructions included -- so the 420 instructions the first version
// of this kernel (rounds 1-2) issued per round WERE its 50 us.  Where they went, and what replaces them:
//
//   * G^T staging (~70): the gathered upstream rows were transposed in registers (DPP swaps + v_perm) and
//     written as dwords, because both MFMA operands want their K (= record) index contiguous per lane.
//     Here the rows go to LDS as they arrive -- G[record][channel], one ds_write_b128 per lane and pass,
//     4 KB with no padding -- and the operand is read with ds_read_b64_tr_b16, gfx950's transposing LDS
//     read: a 16-lane group reads a [4 records][16 channels] block (every lane supplies the address of 4
//     consecutive channels of one record) and lane i receives channel i of the 4 records.  Two reads =
//     the lane's 8 consecutive k.  Four records x 64 bytes are one 256-byte bank row: conflict-free.
//   * A^T scatter (~170): each of a record's 4 weights was stored under an exec mask of its own (in this
//     block? lane live?), 17 s_and_saveexec / s_or pairs a round, the validity flags combined in scalar
//     registers.  Here: a record exists only for a point that passed the window test (bin_kernel), and a
//     block lies inside its map, so "corner inside the block" is the only test left -- done on integer
//     row / column offsets that are `big` when outside, slot = min(row + column, dump slot).  All twelve
//     stores (hi term, lo term, clear) are unconditional; corners outside go to a row nobody reads.
//   * row addresses (~30): 32-bit offsets into a buffer resource instead of 64-bit pointer arithmetic.
//   * the store: lanes l and l + 32 exchange half of their packed rows with v_permlane32_swap (4
//     instructions) instead of 4 ds_bpermute + 12 selects.
//
#pragma once
#include "boxattn_binplan.h"
#include "boxattn_combine.h"
#include "boxattn_scan_tail.h"
#include "boxattn_binpass.h"      // redo_blocks (boxattn_spec.h)

namespace boxattn {

#ifndef BOXATTN_TUNE_TR_WPE
#define BOXATTN_TUNE_TR_WPE 4
#endif
#ifndef BOXATTN_TUNE_REC_AHEAD
#define BOXATTN_TUNE_REC_AHEAD 1       // float32 kernel: request the next item's first records while the current item is summed
#endif

typedef __bf16 tr_bf16x8 __attribute__((ext_vector_type(8)));
typedef float tr_f32x16 __attribute__((ext_vector_type(16)));
typedef short tr_i16x4 __attribute__((ext_vector_type(4)));
// 4 x 16 bits through the transposing read (see above); `p` is a byte address inside the workgroup's LDS
__device__ __forceinline__ uint2 lds_read_tr16(const unsigned short *base, unsigned byte_off)
{
    typedef __attribute__((address_space(3))) tr_i16x4 lds_vec;
    const tr_i16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_vec *)(reinterpret_cast<const char *>(base) + byte_off));
    return __builtin_bit_cast(uint2, v);
}

template <typename ST, int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(C <= 32 ? BOXATTN_TUNE_TR_WPE : 1))) void binned_accumulate_tr_kernel(
    const ST *__restrict__ grad_out, unsigned grad_out_bytes, BinPlan plan, int S, int H, int Lq,
    const int4 *__restrict__ items, const int *__restrict__ n_items,
    const int *__restrict__ records, ST *__restrict__ grad_value, float *__restrict__ partials, ChunkCombine cc,
    ZeroRole zr)
{
    static_assert(sizeof(ST) == 2, "bf16 storage");
    static_assert(C == 16 || C == 32 || C == 64, "channels per head");
    constexpr int BW = 8, PB = 32, R = 64;
    constexpr int CP = C < 32 ? 32 : C;            // operand rows: channels padded to the MFMA's 32
    constexpr int NCB = CP / 32;                   // 32-channel blocks
    constexpr int ROWB = C * 2;                    // bytes of one upstream-gradient row
    constexpr int LPR = ROWB / 16;                 // lanes that fetch one row, 16 B each
    constexpr int RPP = 64 / LPR;                  // rows fetched per pass
    constexpr int NPASS = R / RPP;
    constexpr int GPL = R * 64;                    // bytes of one G plane: [record][32 channels]
    constexpr int AS = 72;                         // ushorts per A^T row (36 dwords: 16-byte aligned rows, conflict-free operand reads)
    constexpr int ASB = AS * 2;
    constexpr int kDump = PB * ASB;                // byte offset of the row that takes the weights of corners outside the block
    constexpr int kBig = 1 << 20;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned short gs[NCB * GPL / 2];
    __shared__ __attribute__((aligned(16))) unsigned short at[(PB + 1) * AS];
    __shared__ int last_flag;

    // workgroup -> (slice, worker): all workers of a slice on one XCD (see binned_accumulate_kernel)
    const int n_slices = plan.n_slices, workers = (int)gridDim.x - plan.zero_workers;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = bid % 8, kq = bid / 8;
    const int per_xcd = (n_slices + 7) / 8;
    const int s = slice_on_xcd(xcd, kq % per_xcd, per_xcd);
    const int worker = kq / per_xcd - plan.zero_workers;        // the zero workers of a sparse map come first
    if (s >= n_slices || worker >= workers) return;
    const int b = s / H, h = s % H;
    const int lane = threadIdx.x;
    if (worker < 0) {          // front rows of the grid: zero workers of a sparse map, or redo workers of the one-pass fill
        if (zr.redo) redo_blocks<ST, C>(zr, plan, s, worker + plan.zero_workers, plan.zero_workers, S, H, Lq, grad_out, grad_value, lane);
        else zero_empty_blocks<ST, C>(zr, plan.nblk, s, worker + plan.zero_workers, S, H, grad_value, lane);
        return;
    }
    const int col = lane & 31, kb = lane >> 5;     // operand row / column, k-block
    const int n_it = n_items[2 * s];

    for (int i = lane; i < (PB + 1) * AS / 2; i += 64) reinterpret_cast<unsigned int *>(at)[i] = 0u;
    if (C < CP)                                     // the padding channels stay zero
        for (int i = lane; i < NCB * GPL / 4; i += 64) reinterpret_cast<unsigned int *>(gs)[i] = 0u;
    wave_lds_sync();

    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<ST *>(grad_out), 0, grad_out_bytes, 0x00020000);
    const unsigned slice_off = (unsigned)((b * Lq) * H + h) * (unsigned)ROWB;   // byte offset of (b, query 0, h)
    const unsigned q_stride = (unsigned)(H * ROWB);
    const int piece = lane % LPR, jrow = lane / LPR;
    // staging: record jrow (+ RPP per pass), 16-byte piece `piece` of its row -> plane piece / 4
    const unsigned stage_off = (unsigned)((piece >> 2) * GPL + jrow * 64 + (piece & 3) * 16);
    // operand read: 16-lane group g reads records 8 (g >> 1) + 0..3 (+ 4), channels 16 (g & 1) + 0..15
    const int g16 = lane >> 4, i16 = lane & 15;
    const unsigned tr_off = (unsigned)((8 * (g16 >> 1) + (i16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2);
    unsigned a_off = (unsigned)(col * ASB + 16 * kb);                 // this lane's 8 consecutive k of pixel `col`
    // (opaque: knowing its low bits are zero the compiler forms a_off | 32, | 64, | 96 in three more registers --
    // which it then spilled and reloaded every round -- instead of one address + the ds_read offset field)
    asm volatile("" : "+v"(a_off));

    const int4 *my_items = items + (size_t)s * plan.item_cap;
    int4 item_n = my_items[min(worker, plan.item_cap - 1)];      // (list is heaviest first)
    for (int it = worker; it < n_it; it += workers) {
        const int4 item = item_n;
        item_n = my_items[min(it + workers, plan.item_cap - 1)];
        const BlockGeo bg = unpack_block_geo((unsigned)item.x);
        int lvH = plan.lv[0].H, lvW = plan.lv[0].W, lv_start = plan.lv[0].start;
#pragma unroll
        for (int k = 1; k < kMaxBinLevels; ++k)
            if (k == bg.level) { lvH = plan.lv[k].H; lvW = plan.lv[k].W; lv_start = plan.lv[k].start; }
        const int oy = bg.oy, ox = bg.ox, bh = bg.bh, bw = bg.bw;
        const float Hf = (float)lvH, Wf = (float)lvW;
        const int4 *rec = reinterpret_cast<const int4 *>(records) + (size_t)s * plan.rec_cap;
        tr_f32x16 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

        // software pipeline over rounds of 64 records: the records are read three rounds ahead, their
        // upstream rows two (into registers; staged into LDS once the current round's operands have been
        // read).  Idle lanes of an item's last round (id -1) fetch from an offset outside the buffer: the
        // load returns ZEROS without touching memory, which is what they have to stage (0 * Inf = NaN: in a
        // dense product any row that happened to be fetched there could poison the block).
        constexpr unsigned kNoRow = 0x80000000u;
        auto fetch_rec = [&](int rr) -> int4 {
            if (rr + lane >= item.z) return make_int4(-1, 0, 0, 0);
            return rec[rr + lane];
        };
        auto fetch_rows = [&](const int4 &r, u32x4 (&rows)[NPASS]) {
            const unsigned id = (unsigned)r.x;
            const unsigned off = r.x < 0 ? kNoRow : __umul24(id >> plan.lp_bits, q_stride) + slice_off;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const unsigned oj = (unsigned)__shfl((int)off, ps * RPP + jrow, 64) + (unsigned)(piece * 16);
                rows[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs, oj, 0, 0);
            }
        };
        auto stage_rows = [&](const u32x4 (&rows)[NPASS]) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
                *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(gs) + stage_off + ps * RPP * 64) = rows[ps];
        };
        // (requesting the NEXT item's first records here -- an item starts with two dependent round trips, and a
        // level-0 block at BoxeR-R50 shapes is under four rounds -- costs this kernel seven spilled registers and
        // 1-3 %; the float32 kernel below, which has the registers, gains 1 % from it)
        int4 rec_c = fetch_rec(item.y), rec_n = fetch_rec(item.y + R), rec_n2 = fetch_rec(item.y + 2 * R);
        u32x4 grow_a[NPASS], grow_b[NPASS];
        fetch_rows(rec_c, grow_a);
        stage_rows(grow_a);
        if (item.y + R < item.z) fetch_rows(rec_n, grow_a);
        // one round: `next` holds the rows of round rr + R (staged at the end), `ahead` receives those of rr + 2 R
        auto round = [&](int rr, const u32x4 (&next)[NPASS], u32x4 (&ahead)[NPASS]) {
            const bool more = rr + R < item.z;     // wave-uniform
            int4 rec_n3 = make_int4(-1, 0, 0, 0);
            if (rr + 2 * R < item.z) {
                fetch_rows(rec_n2, ahead);
                rec_n3 = fetch_rec(rr + 3 * R);
            }
            // ---- lane = record: its <= 4 weights go to A^T[pixel][lane] as hi + lo bf16
            const float x = __int_as_float(rec_c.y), y = __int_as_float(rec_c.z), a = __int_as_float(rec_c.w);
            float h_im, w_im;
            {
#pragma clang fp contract(off)                   // two roundings, as in locate()
                h_im = y * Hf - 0.5f;
                w_im = x * Wf - 0.5f;
            }
            const float yf = floorf(h_im), xf = floorf(w_im);
            const float lh = h_im - yf, lw = w_im - xf;
            // footprint corner relative to the block; idle lanes: outside
            const int py = rec_c.x >= 0 ? (int)yf - oy : -2, px = (int)xf - ox;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float ha = hh * a, la = lh * a;
            const float w0 = ha * hw, w1 = ha * lw, w2 = la * hw, w3 = la * lw;
            const unsigned hi01 = pack_bf16x2(w0, w1), hi23 = pack_bf16x2(w2, w3);
            const unsigned lo01 = pack_bf16x2(w0 - __uint_as_float(hi01 << 16), w1 - __uint_as_float(hi01 & 0xffff0000u));
            const unsigned lo23 = pack_bf16x2(w2 - __uint_as_float(hi23 << 16), w3 - __uint_as_float(hi23 & 0xffff0000u));
            // byte offsets of the corners' rows / columns inside A^T, kBig when outside the block (idle lanes: all)
            const int r0 = (unsigned)py < (unsigned)bh ? __mul24(py, BW * ASB) : kBig;
            const int r1 = (unsigned)(py + 1) < (unsigned)bh ? __mul24(py, BW * ASB) + BW * ASB : kBig;
            const int c0 = (unsigned)px < (unsigned)bw ? __mul24(px, ASB) + 2 * lane : kBig;
            const int c1 = (unsigned)(px + 1) < (unsigned)bw ? __mul24(px, ASB) + 2 * lane + ASB : kBig;
            const int dump = kDump + 2 * lane;
            const int slot[4] = {min(r0 + c0, dump), min(r0 + c1, dump), min(r1 + c0, dump), min(r1 + c1, dump)};
            auto put = [&](int k, unsigned short v) {
                *reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(at) + slot[k]) = v;
            };
            put(0, (unsigned short)(hi01 & 0xffffu)); put(1, (unsigned short)(hi01 >> 16));
            put(2, (unsigned short)(hi23 & 0xffffu)); put(3, (unsigned short)(hi23 >> 16));
            wave_lds_sync();
            // ---- the product: 4 K-steps of 16 records per 32-channel block, hi term then lo term
            tr_bf16x8 g[R / 16][NCB];
#pragma unroll
            for (int t = 0; t < R / 16; ++t) {
                const tr_bf16x8 p_hi = __builtin_bit_cast(
                    tr_bf16x8, *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(at) + a_off + 32 * t));
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const uint2 g0 = lds_read_tr16(gs, tr_off + cb * GPL + t * 1024);
                    const uint2 g1 = lds_read_tr16(gs, tr_off + cb * GPL + t * 1024 + 256);
                    g[t][cb] = __builtin_bit_cast(tr_bf16x8, u32x4{g0.x, g0.y, g1.x, g1.y});
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[t][cb], p_hi, acc[cb], 0, 0, 0);
                }
            }
            wave_lds_sync();                         // a wave's LDS operations execute in order
            put(0, (unsigned short)(lo01 & 0xffffu)); put(1, (unsigned short)(lo01 >> 16));
            put(2, (unsigned short)(lo23 & 0xffffu)); put(3, (unsigned short)(lo23 >> 16));
            wave_lds_sync();
#pragma unroll
            for (int t = 0; t < R / 16; ++t) {
                const tr_bf16x8 p_lo = __builtin_bit_cast(
                    tr_bf16x8, *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(at) + a_off + 32 * t));
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[t][cb], p_lo, acc[cb], 0, 0, 0);
            }
            wave_lds_sync();
            // ---- clear this round's weights, stage the next round's rows (they have arrived)
#pragma unroll
            for (int k = 0; k < 4; ++k) put(k, (unsigned short)0);
            if (more) {
                stage_rows(next);
                rec_c = rec_n; rec_n = rec_n2; rec_n2 = rec_n3;
            }
            wave_lds_sync();
        };
        for (int rr = item.y; rr < item.z; rr += 2 * R) {       // (two rounds per trip: the row buffers swap roles, no copies)
            round(rr, grow_a, grow_b);
            if (rr + R >= item.z) break;
            round(rr + R, grow_b, grow_a);
        }
        // ---- store.  Lane = pixel `col`; its registers hold channels 8 g + 4 kb + 0..3.
        if (item.w < 0) {
            // whole rows in the storage type: lanes l and l + 32 swap half of their packed pairs
            // (v_permlane32_swap), so that each writes two 16-byte pieces (8 channels) of the pixel's row
            int ln = lane;
            asm volatile("" : "+v"(ln));              // (as for the partial tiles: nothing of this address is hoisted)
            const int py = (ln & 31) / BW, px = (ln & 31) % BW;
            const bool live = py < bh && px < bw;
            ST *dst = grad_value +
                      (((size_t)b * S + lv_start + (size_t)(oy + py) * lvW + (ox + px)) * H + h) * C;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                unsigned pk[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pk[i] = pack_bf16x2(acc[cb][2 * i], acc[cb][2 * i + 1]);
                // pk[2 g], pk[2 g + 1]: channels 8 g + 4 kb + 0..3.  After swap(g = 0, g = 1) the lanes of
                // kb = 0 hold channels 0-7 and those of kb = 1 channels 8-15; g = 2, 3 alike (+ 16).
                u32x4 piece_lo, piece_hi;
                {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
                    const auto s2 = __builtin_amdgcn_permlane32_swap(pk[4], pk[6], false, false);
                    const auto s3 = __builtin_amdgcn_permlane32_swap(pk[5], pk[7], false, false);
                    piece_lo = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    piece_hi = u32x4{s2[0], s3[0], s2[1], s3[1]};
                }
                const int c_lo = cb * 32 + 8 * kb, c_hi = cb * 32 + 16 + 8 * kb;
#ifndef BOXATTN_TUNE_GV_NT
#define BOXATTN_TUNE_GV_NT 0       // grad_value rows with non-temporal stores (nobody on the GPU reads them soon)
#endif
                if (BOXATTN_TUNE_GV_NT) {
                    if (live && c_lo < C) __builtin_nontemporal_store(piece_lo, reinterpret_cast<u32x4 *>(dst + c_lo));
                    if (live && c_hi < C) __builtin_nontemporal_store(piece_hi, reinterpret_cast<u32x4 *>(dst + c_hi));
                } else {
                    if (live && c_lo < C) *reinterpret_cast<u32x4 *>(dst + c_lo) = piece_lo;
                    if (live && c_hi < C) *reinterpret_cast<u32x4 *>(dst + c_hi) = piece_hi;
                }
            }
        } else {
            // a chunk: fp32 partial tile; the block's last chunk to finish sums them (chunk_finish)
            const bool publish = cc.tickets != nullptr;
            const __amdgpu_buffer_rsrc_t tile =
                partial_tile(partials, s, plan.pslot_cap, item.w & ((1 << kItemSlotBits) - 1), C);
            // (the lane's byte offset is formed HERE, from a lane index opaque to the compiler: hoisted out of the item loop the four
            // piece offsets sat in registers the round loop does not have -- spilled to scratch)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const unsigned lane_off = (unsigned)(((ln & 31) * C + 4 * (ln >> 5)) * 4);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c = cb * 32 + 8 * g4;
                    if (c + 4 * kb < C)
                        partial_store(tile, lane_off + (unsigned)(c * 4),
                                      make_float4(acc[cb][4 * g4], acc[cb][4 * g4 + 1], acc[cb][4 * g4 + 2],
                                                  acc[cb][4 * g4 + 3]), publish);
                }
            if (publish) chunk_finish<ST, C>(cc, partials, lv_start, lvW, S, H, grad_value, s, item.w, lane, &last_flag);
        }
    }
}

// FLOAT32 storage on the bf16 matrix cores, exactly: every float32 number is the sum of three bf16 numbers
// (x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2: 8 + 8 + 8 significand bits, each
// difference exact), bf16 x bf16 products are exact in float32, and the matrix cores accumulate in float32.  With the
// upstream rows g and the weights w both split, w g = sum_{i,j} w_i g_j; the six products with i + j <= 4 carry
// everything above 2^-24 |w g| -- what a float32 fused multiply-add keeps of the product -- and are summed into the
// float32 accumulators like the terms of the reference's float32 sum (in another, still unspecified, order).  Against
// the VALU list walk of binned_accumulate_kernel (float32 box attention, C = 32: 102 us at C2) a round of 64 records is
// 24 v_mfma_f32_32x32x16_bf16 (768 matrix-pipe cycles; the float32 MFMA flavour below needs 2 048) + the splitting of
// the 64 gathered rows on the VALU (the price of float32 storage: ~180 instructions a round).
//   rows:    gathered 128-byte rows -> registers -> split -> two bf16 planes G1, G2 in LDS (G3 follows into G1's place
//            once G1's operands have been read) -> ds_read_b64_tr_b16 -> the three operand sets stay in registers;
//   weights: A^T[pixel][record] as in the bf16 kernel, written three times (w1, w2, w3 over the same slots);
//   products: w1 (g1, g2, g3), w2 (g1, g2), w3 g1.
// Wide records {id, x, y, weight}; the next round's rows are requested as soon as this round's have been split (one
// register buffer).
// INST (round 6): instance attention.  A record's upstream row is t = a_s g[query] + a_l g_mask[query, point]
// (instance_attn_kernel.cuh:139): both rows are gathered, combined in float32 as they arrive and THEN split -- the
// bilinear weights go to A^T without an attention weight.  Records {point id, x, y, a_s} as for box attention (a_s = the
// spatial weight); a_l is gathered from `level_w` by the record's point id.  Two row buffers in flight: 2 waves per SIMD.
struct InstRows {
    const float *grad_mask;      // (B, Lq, P, H, C)
    unsigned grad_mask_bytes;
    const float *w_lv;           // (B, Lq, H, L, P)
    int P;
};
template <int C, bool INST = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(INST ? 2 : 3, INST ? 2 : 3))) void binned_accumulate_split_kernel(
    const float *__restrict__ grad_out, unsigned grad_out_bytes, BinPlan plan, int S, int H, int Lq,
    const int4 *__restrict__ items, const int *__restrict__ n_items,
    const int *__restrict__ records, float *__restrict__ grad_value, float *__restrict__ partials, ChunkCombine cc,
    ZeroRole zr, InstRows inst)
{
    static_assert(C == 32, "channels per head");
    constexpr int BW = 8, PB = 32, R = 64;
    constexpr int ROWB = C * 4;                    // bytes of one upstream-gradient row
    constexpr int LPR = ROWB / 16;                 // lanes that fetch one row, 16 B each (8)
    constexpr int RPP = 64 / LPR;                  // rows fetched per pass (8)
    constexpr int NPASS = R / RPP;                 // 8
    constexpr int GPL = R * 64;                    // bytes of one bf16 plane: [record][32 channels]
    constexpr int AS = 72, ASB = AS * 2;
    constexpr int kDump = PB * ASB;
    constexpr int kBig = 1 << 20;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned short gs[2 * GPL / 2];
    __shared__ __attribute__((aligned(16))) unsigned short at[(PB + 1) * AS];
    __shared__ int last_flag;

    const int n_slices = plan.n_slices, workers = (int)gridDim.x - plan.zero_workers;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = bid % 8, kq = bid / 8;
    const int per_xcd = (n_slices + 7) / 8;
    const int s = slice_on_xcd(xcd, kq % per_xcd, per_xcd);
    const int worker = kq / per_xcd - plan.zero_workers;
    if (s >= n_slices || worker >= workers) return;
    const int b = s / H, h = s % H;
    const int lane = threadIdx.x;
    if (worker < 0) {
        if (zr.redo) redo_blocks<float, C>(zr, plan, s, worker + plan.zero_workers, plan.zero_workers, S, H, Lq, grad_out, grad_value, lane);
        else zero_empty_blocks<float, C>(zr, plan.nblk, s, worker + plan.zero_workers, S, H, grad_value, lane);
        return;
    }
    const int col = lane & 31, kb = lane >> 5;
    const int n_it = n_items[2 * s];
    for (int i = lane; i < (PB + 1) * AS / 2; i += 64) reinterpret_cast<unsigned int *>(at)[i] = 0u;
    wave_lds_sync();

    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(grad_out), 0, grad_out_bytes, 0x00020000);
    const unsigned slice_off = (unsigned)((b * Lq) * H + h) * (unsigned)ROWB;
    const unsigned q_stride = (unsigned)(H * ROWB);
    // INST: rows of grad_mask (b, q, p, h, :) and the level weights of this slice's points
    const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(INST ? inst.grad_mask : grad_out), 0, INST ? inst.grad_mask_bytes : grad_out_bytes, 0x00020000);
    const int P = INST ? inst.P : 1, LP = plan.L * P;
    const unsigned m_slice_off = (unsigned)((b * Lq) * P * H + h) * (unsigned)ROWB;
    const unsigned m_q_stride = (unsigned)(P * H * ROWB), m_p_stride = (unsigned)(H * ROWB);
    const float rcp_p = 1.0f / (float)P;
    const int lp_mask = (1 << plan.lp_bits) - 1;
    const int piece = lane % LPR, jrow = lane / LPR;
    // staging: record jrow (+ 8 per pass), channels 4 piece .. 4 piece + 3 as 4 bf16 = 8 bytes of its plane row
    const unsigned stage_off = (unsigned)(jrow * 64 + piece * 8);
    const int g16 = lane >> 4, i16 = lane & 15;
    const unsigned tr_off = (unsigned)((8 * (g16 >> 1) + (i16 >> 2)) * 64 + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2);
    unsigned a_off = (unsigned)(col * ASB + 16 * kb);
    asm volatile("" : "+v"(a_off));

    const int4 *my_items = items + (size_t)s * plan.item_cap;
    int4 item_n = my_items[min(worker, plan.item_cap - 1)];
    // (the next item's first records are requested while the current item is summed: an item starts with two dependent
    // round trips -- its records, then the rows they name -- in front of its first product; C2 float32 step -1 %)
    const int4 *rec_s = reinterpret_cast<const int4 *>(records) + (size_t)s * plan.rec_cap;
    int4 rec_first = make_int4(-1, 0, 0, 0);
    if (BOXATTN_TUNE_REC_AHEAD && worker < n_it && item_n.y + lane < item_n.z) rec_first = rec_s[item_n.y + lane];
    for (int it = worker; it < n_it; it += workers) {
        const int4 item = item_n;
        item_n = my_items[min(it + workers, plan.item_cap - 1)];
        const BlockGeo bg = unpack_block_geo((unsigned)item.x);
        int lvH = plan.lv[0].H, lvW = plan.lv[0].W, lv_start = plan.lv[0].start;
#pragma unroll
        for (int k = 1; k < kMaxBinLevels; ++k)
            if (k == bg.level) { lvH = plan.lv[k].H; lvW = plan.lv[k].W; lv_start = plan.lv[k].start; }
        const int oy = bg.oy, ox = bg.ox, bh = bg.bh, bw = bg.bw;
        const float Hf = (float)lvH, Wf = (float)lvW;
        const int4 *rec = reinterpret_cast<const int4 *>(records) + (size_t)s * plan.rec_cap;
        tr_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;

        constexpr unsigned kNoRow = 0x80000000u;       // outside the buffer: zeros (idle lanes of a last round)
        auto fetch_rec = [&](int rr) -> int4 {
            if (rr + lane >= item.z) return make_int4(-1, 0, 0, 0);
            return rec[rr + lane];
        };
        u32x4 rows[NPASS], rows_m[INST ? NPASS : 1];
        float as_rows = 0.f, al_rows = 0.f;            // INST: the two attention weights of the records whose rows are in flight
        auto fetch_rows = [&](const int4 &r) {
            const unsigned off = r.x < 0 ? kNoRow : __umul24((unsigned)r.x >> plan.lp_bits, q_stride) + slice_off;
            unsigned off_m = kNoRow;
            if constexpr (INST) {
                const unsigned q = (unsigned)max(r.x, 0) >> plan.lp_bits;
                const int lp = max(r.x, 0) & lp_mask;
                int l_, p_;
                divmod_small(lp, P, rcp_p, l_, p_);
                off_m = r.x < 0 ? kNoRow : q * m_q_stride + (unsigned)p_ * m_p_stride + m_slice_off;
                as_rows = r.x < 0 ? 0.f : __int_as_float(r.w);
                al_rows = r.x < 0 ? 0.f : inst.w_lv[(((size_t)b * Lq + q) * H + h) * LP + lp];
            }
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const unsigned oj = (unsigned)__shfl((int)off, ps * RPP + jrow, 64) + (unsigned)(piece * 16);
                rows[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs, oj, 0, 0);
                if constexpr (INST) {
                    const unsigned om = (unsigned)__shfl((int)off_m, ps * RPP + jrow, 64) + (unsigned)(piece * 16);
                    rows_m[ps] = __builtin_amdgcn_raw_buffer_load_b128(rsm, om, 0, 0);
                }
            }
        };
        // x -> its leading bf16 term (packed pairs) and the exact remainder
        auto split2 = [](float &x0, float &x1) -> unsigned {
            const unsigned pk = pack_bf16x2(x0, x1);
            x0 -= __uint_as_float(pk << 16);
            x1 -= __uint_as_float(pk & 0xffff0000u);
            return pk;
        };
        int4 rec_c = BOXATTN_TUNE_REC_AHEAD ? rec_first : fetch_rec(item.y);
        int4 rec_n = fetch_rec(item.y + R), rec_n2 = fetch_rec(item.y + 2 * R);
        fetch_rows(rec_c);
        if (BOXATTN_TUNE_REC_AHEAD) {
            rec_first = make_int4(-1, 0, 0, 0);
            if (it + workers < n_it && item_n.y + lane < item_n.z) rec_first = rec_s[item_n.y + lane];
        }
        for (int rr = item.y; rr < item.z; rr += R) {
            const bool more = rr + R < item.z;         // wave-uniform
            int4 rec_n3 = make_int4(-1, 0, 0, 0);
            if (rr + 3 * R < item.z) rec_n3 = fetch_rec(rr + 3 * R);
            // ---- the rows of this round: split into three bf16 terms; planes G1, G2 to LDS, G3 kept packed
            uint2 g3pk[NPASS];
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                float x0 = __uint_as_float(rows[ps].x), x1 = __uint_as_float(rows[ps].y);
                float x2 = __uint_as_float(rows[ps].z), x3 = __uint_as_float(rows[ps].w);
                if constexpr (INST) {          // t = a_s g + a_l g_mask, in float32, before the split
                    const float as_j = __shfl(as_rows, ps * RPP + jrow, 64), al_j = __shfl(al_rows, ps * RPP + jrow, 64);
                    x0 = fmaf(al_j, __uint_as_float(rows_m[ps].x), as_j * x0);
                    x1 = fmaf(al_j, __uint_as_float(rows_m[ps].y), as_j * x1);
                    x2 = fmaf(al_j, __uint_as_float(rows_m[ps].z), as_j * x2);
                    x3 = fmaf(al_j, __uint_as_float(rows_m[ps].w), as_j * x3);
                }
                uint2 t1, t2;
                t1.x = split2(x0, x1); t1.y = split2(x2, x3);
                t2.x = split2(x0, x1); t2.y = split2(x2, x3);
                g3pk[ps] = uint2{pack_bf16x2(x0, x1), pack_bf16x2(x2, x3)};            // (exact: <= 8 bits are left)
                *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(gs) + stage_off + ps * RPP * 64) = t1;
                *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(gs) + GPL + stage_off + ps * RPP * 64) = t2;
            }
            if (more) fetch_rows(rec_n);               // (the row registers are free again: the next round's rows)
            // ---- lane = record: its <= 4 weights, three bf16 terms each
            // (INST: the attention weights are in the combined row already)
            const float x = __int_as_float(rec_c.y), y = __int_as_float(rec_c.z), a = INST ? 1.f : __int_as_float(rec_c.w);
            float h_im, w_im;
            {
#pragma clang fp contract(off)                   // two roundings, as in locate()
                h_im = y * Hf - 0.5f;
                w_im = x * Wf - 0.5f;
            }
            const float yf = floorf(h_im), xf = floorf(w_im);
            const float lh = h_im - yf, lw = w_im - xf, hh = 1.f - lh, hw = 1.f - lw;
            // (the reference's products: (hh hw) a etc. -- in that order)
            float w0 = hh * hw * a, w1 = hh * lw * a, w2 = lh * hw * a, w3 = lh * lw * a;
            const unsigned t1_01 = split2(w0, w1), t1_23 = split2(w2, w3);
            const unsigned t2_01 = split2(w0, w1), t2_23 = split2(w2, w3);
            const unsigned t3_01 = pack_bf16x2(w0, w1), t3_23 = pack_bf16x2(w2, w3);
            const int py = rec_c.x >= 0 ? (int)yf - oy : -2, px = (int)xf - ox;
            const int r0 = (unsigned)py < (unsigned)bh ? __mul24(py, BW * ASB) : kBig;
            const int r1 = (unsigned)(py + 1) < (unsigned)bh ? __mul24(py, BW * ASB) + BW * ASB : kBig;
            const int c0 = (unsigned)px < (unsigned)bw ? __mul24(px, ASB) + 2 * lane : kBig;
            const int c1 = (unsigned)(px + 1) < (unsigned)bw ? __mul24(px, ASB) + 2 * lane + ASB : kBig;
            const int dump = kDump + 2 * lane;
            const int slot[4] = {min(r0 + c0, dump), min(r0 + c1, dump), min(r1 + c0, dump), min(r1 + c1, dump)};
            auto put4 = [&](unsigned p01, unsigned p23) {
                *reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(at) + slot[0]) = (unsigned short)(p01 & 0xffffu);
                *reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(at) + slot[1]) = (unsigned short)(p01 >> 16);
                *reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(at) + slot[2]) = (unsigned short)(p23 & 0xffffu);
                *reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(at) + slot[3]) = (unsigned short)(p23 >> 16);
            };
            put4(t1_01, t1_23);
            wave_lds_sync();
            // ---- operands: G1, G2 now; G3 goes into G1's plane once G1 has been read
            tr_bf16x8 g1[R / 16], g2[R / 16], g3[R / 16];
#pragma unroll
            for (int t = 0; t < R / 16; ++t) {
                const uint2 a0 = lds_read_tr16(gs, tr_off + t * 1024), a1 = lds_read_tr16(gs, tr_off + t * 1024 + 256);
                g1[t] = __builtin_bit_cast(tr_bf16x8, u32x4{a0.x, a0.y, a1.x, a1.y});
                const uint2 b0 = lds_read_tr16(gs, GPL + tr_off + t * 1024), b1 = lds_read_tr16(gs, GPL + tr_off + t * 1024 + 256);
                g2[t] = __builtin_bit_cast(tr_bf16x8, u32x4{b0.x, b0.y, b1.x, b1.y});
            }
            wave_lds_sync();
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
                *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(gs) + stage_off + ps * RPP * 64) = g3pk[ps];
            wave_lds_sync();
#pragma unroll
            for (int t = 0; t < R / 16; ++t) {
                const uint2 a0 = lds_read_tr16(gs, tr_off + t * 1024), a1 = lds_read_tr16(gs, tr_off + t * 1024 + 256);
                g3[t] = __builtin_bit_cast(tr_bf16x8, u32x4{a0.x, a0.y, a1.x, a1.y});
            }
            // ---- products, the small ones first: w1 g3, w1 g2, w1 g1; w2 g2, w2 g1; w3 g1
            auto weights = [&](int t) -> tr_bf16x8 {
                return __builtin_bit_cast(
                    tr_bf16x8, *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(at) + a_off + 32 * t));
            };
#pragma unroll
            for (int t = 0; t < R / 16; ++t) {
                const tr_bf16x8 pw = weights(t);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g3[t], pw, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[t], pw, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[t], pw, acc, 0, 0, 0);
            }
            wave_lds_sync();
            put4(t2_01, t2_23);
            wave_lds_sync();
#pragma unroll
            for (int t = 0; t < R / 16; ++t) {
                const tr_bf16x8 pw = weights(t);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g2[t], pw, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[t], pw, acc, 0, 0, 0);
            }
            wave_lds_sync();
            put4(t3_01, t3_23);
            wave_lds_sync();
#pragma unroll
            for (int t = 0; t < R / 16; ++t)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[t], weights(t), acc, 0, 0, 0);
            wave_lds_sync();
            put4(0u, 0u);
            rec_c = rec_n; rec_n = rec_n2; rec_n2 = rec_n3;
            wave_lds_sync();
        }
        // ---- store.  Lane = pixel `col`; its registers hold channels 8 g + 4 kb + 0..3.
        if (item.w < 0) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int py = (ln & 31) / BW, px = (ln & 31) % BW, kbs = ln >> 5;
            const bool live = py < bh && px < bw;
            float *dst = grad_value + (((size_t)b * S + lv_start + (size_t)(oy + py) * lvW + (ox + px)) * H + h) * C;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                if (live)
                    *reinterpret_cast<float4 *>(dst + 8 * g4 + 4 * kbs) =
                        make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
        } else {
            const bool publish = cc.tickets != nullptr;
            const __amdgpu_buffer_rsrc_t tile =
                partial_tile(partials, s, plan.pslot_cap, item.w & ((1 << kItemSlotBits) - 1), C);
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const unsigned lane_off = (unsigned)(((ln & 31) * C + 4 * (ln >> 5)) * 4);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                partial_store(tile, lane_off + (unsigned)(32 * g4),
                              make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]), publish);
            if (publish) chunk_finish<float, C>(cc, partials, lv_start, lvW, S, H, grad_value, s, item.w, lane, &last_flag);
        }
    }
}

// float32 storage, 32 channels per head: the same product on v_mfma_f32_32x32x2_f32 -- float32 operands, float32
// products and accumulation: no splitting, no rounding beyond what any float32 summation order has (the VALU
// list walk of binned_accumulate_kernel is float32 too, in another order).  An operand of that instruction is ONE
// element per lane (row / column l & 31, k = l >> 5 of the 2 records of a step), so nothing has to be transposed:
// the gathered rows are staged as they arrive, G[record][channel] (8 KB), and read back one dword per lane (32
// consecutive dwords per half-wave: conflict-free); the weights are scattered as float32 into A^T[record][pixel]
// (lane = record writes inside its own 132-byte row; a 33rd column takes the corners outside the block).
// 32 MFMAs of 64 cycles per round: the kernel is bound by the matrix pipe (61 us at C2 for 73 k rounds) instead of
// the VALU kernel's list walk (103 us).  Wide records {id, x, y, weight} as the bf16 kernel.
template <int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void binned_accumulate_f32_kernel(
    const float *__restrict__ grad_out, unsigned grad_out_bytes, BinPlan plan, int S, int H, int Lq,
    const int4 *__restrict__ items, const int *__restrict__ n_items,
    const int *__restrict__ records, float *__restrict__ grad_value, float *__restrict__ partials, ChunkCombine cc,
    ZeroRole zr)
{
    static_assert(C == 32, "channels per head");
    constexpr int BW = 8, PB = 32, R = 64, RH = R / 2;
    constexpr int ROWB = C * 4;                    // bytes of one upstream-gradient row
    constexpr int LPR = ROWB / 16;                 // lanes that fetch one row, 16 B each (8)
    constexpr int RPP = 64 / LPR;                  // rows fetched per pass (8)
    constexpr int NPASS = R / RPP;                 // 8
    constexpr int AP = PB + 1;                     // floats per A^T row: 32 pixels + the dump column
    constexpr int kBig = 1 << 20;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // LDS tiles for HALF a round (32 records): the matrix pipe needs four waves per SIMD to stay fed across the
    // scatter / stage phases, and whole-round tiles (16.6 KB) allow two
    __shared__ __attribute__((aligned(16))) float gs[RH * C];
    __shared__ __attribute__((aligned(16))) float at[RH * AP];
    __shared__ int last_flag;

    const int n_slices = plan.n_slices, workers = (int)gridDim.x - plan.zero_workers;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = bid % 8, kq = bid / 8;
    const int per_xcd = (n_slices + 7) / 8;
    const int s = slice_on_xcd(xcd, kq % per_xcd, per_xcd);
    const int worker = kq / per_xcd - plan.zero_workers;        // the zero workers of a sparse map come first
    if (s >= n_slices || worker >= workers) return;
    const int b = s / H, h = s % H;
    const int lane = threadIdx.x;
    if (worker < 0) {
        if (zr.redo) redo_blocks<float, C>(zr, plan, s, worker + plan.zero_workers, plan.zero_workers, S, H, Lq, grad_out, grad_value, lane);
        else zero_empty_blocks<float, C>(zr, plan.nblk, s, worker + plan.zero_workers, S, H, grad_value, lane);
        return;
    }
    const int col = lane & 31, kb = lane >> 5;     // operand row / column, record of the K = 2 step
    const int n_it = n_items[2 * s];

    for (int i = lane; i < RH * AP; i += 64) at[i] = 0.f;
    wave_lds_sync();

    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(grad_out), 0, grad_out_bytes, 0x00020000);
    const unsigned slice_off = (unsigned)((b * Lq) * H + h) * (unsigned)ROWB;
    const unsigned q_stride = (unsigned)(H * ROWB);
    const int piece = lane % LPR, jrow = lane / LPR;

    const int4 *my_items = items + (size_t)s * plan.item_cap;
    int4 item_n = my_items[min(worker, plan.item_cap - 1)];
    for (int it = worker; it < n_it; it += workers) {
        const int4 item = item_n;
        item_n = my_items[min(it + workers, plan.item_cap - 1)];
        const BlockGeo bg = unpack_block_geo((unsigned)item.x);
        int lvH = plan.lv[0].H, lvW = plan.lv[0].W, lv_start = plan.lv[0].start;
#pragma unroll
        for (int k = 1; k < kMaxBinLevels; ++k)
            if (k == bg.level) { lvH = plan.lv[k].H; lvW = plan.lv[k].W; lv_start = plan.lv[k].start; }
        const int oy = bg.oy, ox = bg.ox, bh = bg.bh, bw = bg.bw;
        const float Hf = (float)lvH, Wf = (float)lvW;
        const int4 *rec = reinterpret_cast<const int4 *>(records) + (size_t)s * plan.rec_cap;
        tr_f32x16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }

        // records two rounds ahead, rows one (the kernel waits for the matrix pipe, not for memory); idle lanes of
        // the last round fetch from outside the buffer: zeros
        constexpr unsigned kNoRow = 0x80000000u;
        auto fetch_rec = [&](int rr) -> int4 {
            return rr + lane < item.z ? rec[rr + lane] : make_int4(-1, 0, 0, 0);
        };
        u32x4 rows[NPASS];
        auto fetch_rows = [&](const int4 &r) {
            const unsigned off = r.x < 0 ? kNoRow : __umul24((unsigned)r.x >> plan.lp_bits, q_stride) + slice_off;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const unsigned oj = (unsigned)__shfl((int)off, ps * RPP + jrow, 64) + (unsigned)(piece * 16);
                rows[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs, oj, 0, 0);
            }
        };
        int4 rec_c = fetch_rec(item.y), rec_n = fetch_rec(item.y + R);
        fetch_rows(rec_c);
        for (int rr = item.y; rr < item.z; rr += R) {
            const bool more = rr + R < item.z;     // wave-uniform
            // ---- lane = record: its <= 4 weights
            const float x = __int_as_float(rec_c.y), y = __int_as_float(rec_c.z), a = __int_as_float(rec_c.w);
            float h_im, w_im;
            {
#pragma clang fp contract(off)                   // two roundings, as in locate()
                h_im = y * Hf - 0.5f;
                w_im = x * Wf - 0.5f;
            }
            const float yf = floorf(h_im), xf = floorf(w_im);
            const float lh = h_im - yf, lw = w_im - xf, hh = 1.f - lh, hw = 1.f - lw;
            // (the reference's products: (hh hw) a etc. -- kept in that order)
            const float wk[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
            const int py = rec_c.x >= 0 ? (int)yf - oy : -2, px = (int)xf - ox;
            const int r0 = (unsigned)py < (unsigned)bh ? py * BW : kBig;
            const int r1 = (unsigned)(py + 1) < (unsigned)bh ? (py + 1) * BW : kBig;
            const int c0 = (unsigned)px < (unsigned)bw ? px : kBig;
            const int c1 = (unsigned)(px + 1) < (unsigned)bw ? px + 1 : kBig;
            float *my = at + (lane & (RH - 1)) * AP;
            const int slot[4] = {min(r0 + c0, PB), min(r0 + c1, PB), min(r1 + c0, PB), min(r1 + c1, PB)};
            // ---- two half rounds of 32 records: stage their rows, scatter their weights, 16 K-steps of 2 records
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int ps = 0; ps < NPASS / 2; ++ps)
                    *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(gs) + (ps * 64 + lane) * 16) = rows[hf * (NPASS / 2) + ps];
                if ((lane >> 5) == hf) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) my[slot[k]] = wk[k];
                }
                wave_lds_sync();
                if (hf == 1 && more) {             // the rows of this round are in LDS: request the next round's
                    fetch_rows(rec_n);
                }
                // (two accumulator chains: a dependent 32x32 MFMA waits for its predecessor's 64 cycles + latency)
#pragma unroll
                for (int t = 0; t < RH / 2; t += 2) {
                    const float g0 = gs[(2 * t + kb) * C + col], g1 = gs[(2 * t + 2 + kb) * C + col];
                    const float p0 = at[(2 * t + kb) * AP + col], p1 = at[(2 * t + 2 + kb) * AP + col];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, p0, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, p1, acc2, 0, 0, 0);
                }
                wave_lds_sync();
                if ((lane >> 5) == hf) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) my[slot[k]] = 0.f;
                }
                wave_lds_sync();
            }
            if (more) {
                rec_c = rec_n;
                rec_n = fetch_rec(rr + 2 * R);
            }
        }
        // ---- store.  Lane = pixel `col`; its registers hold channels 8 g + 4 kb + 0..3.
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
        if (item.w < 0) {
            int ln = lane;
            asm volatile("" : "+v"(ln));              // (store addresses formed here, not hoisted and spilled)
            const int py = (ln & 31) / BW, px = (ln & 31) % BW, kbs = ln >> 5;
            const bool live = py < bh && px < bw;
            float *dst = grad_value +
                         (((size_t)b * S + lv_start + (size_t)(oy + py) * lvW + (ox + px)) * H + h) * C;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                if (live)
                    *reinterpret_cast<float4 *>(dst + 8 * g4 + 4 * kbs) =
                        make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
        } else {
            const bool publish = cc.tickets != nullptr;
            const __amdgpu_buffer_rsrc_t tile =
                partial_tile(partials, s, plan.pslot_cap, item.w & ((1 << kItemSlotBits) - 1), C);
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const unsigned lane_off = (unsigned)(((ln & 31) * C + 4 * (ln >> 5)) * 4);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                partial_store(tile, lane_off + (unsigned)(32 * g4),
                              make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]), publish);
            if (publish) chunk_finish<float, C>(cc, partials, lv_start, lvW, S, H, grad_value, s, item.w, lane, &last_flag);
        }
    }
}

}  // namespace boxattn
