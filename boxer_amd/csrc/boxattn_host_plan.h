// Host-side planning of a call: shapes -> BinPlan / DensePlan, buffer layouts (plan, scratch, state), which
// accumulate kernel and record format a call uses.  Pure functions of the dimensions, the level tables and the
// option switches -- no launches.  Included by boxattn_capi.hip only, inside its unnamed namespace, behind the
// option switches (opt(), kOpt*), Dims and fast_group() it uses.
#pragma once

// ------------------------------------------------------- binned backward (boxattn_binned.h)
constexpr int kChunk = 1024, kChunkBig = 1536;   // records per work item (upper bound of the rule below; what it uses beyond)
// Records per work item.  One wavefront works an item off 64 records a round, and a round is a
// chain of dependent latencies (~3-5 us), so the kernel lasts at least rounds-per-item rounds:
// with few sample points (the decoders: 300 queries) 1 024-record items leave a handful of waves
// running 16 rounds while the rest of the chip idles.  Aim at ~256 items per (image, head) slice
// -- about the wave slots a slice gets -- between 128 and kChunk records.
inline int bin_chunk(const Dims &d)
{
    const int forced = opt(kOptBinChunk);
    if (forced > 0) return std::min(4096, std::max(64, (forced + 63) / 64 * 64));
    const long long rec_est = 3ll * d.Lq * d.L * d.P / 2;          // ~1.4 records per point
    const long long c = (rec_est / 256 + 63) / 64 * 64;
    // (encoder-sized problems: 1 536 -- fewer partial tiles for the combine step, measured at C2 / C2' against
    // 1 024 / 1 280 / 2 048: profiles/r04_chunk_sweep.log)
    return c >= kChunk ? kChunkBig : (int)std::max<long long>(128, c);
}

// Which accumulate kernel a call runs, and with it the record format of the bin passes:
//   kAccTr   bf16 box attention, C = 16 / 32 / 64: binned_accumulate_tr_kernel (v_mfma_f32_32x32x16_bf16) from
//            16-byte records {id, x, y, weight}, contiguous query ranges per bin workgroup;
//   kAccF32  float32 box attention, C = 32, opt-in (boxattn_set_option(19, 2)): binned_accumulate_f32_kernel
//            (v_mfma_f32_32x32x2_f32: float32-exact), 16-byte records -- measured at C2 97 us against the VALU
//            kernel's 102 on model-like inputs (the matrix pipe is busy 63 us of them), 100 against 133 on
//            uniformly random ones;
//   kAccValu everything else (float32, instance attention): binned_accumulate_kernel, 4-byte records, queries
//            interleaved over the bin workgroups.
//   kAccSplit float32 box AND instance attention, C = 32 (default): binned_accumulate_split_kernel -- the bf16 matrix cores
//            on exact three-term splits of rows and weights, 16-byte records; boxattn_set_option(19, 1) goes back to kAccValu.
enum AccKind { kAccValu = 0, kAccTr = 1, kAccF32 = 2, kAccSplit = 3 };
inline bool accumulate_tr_ok(const Dims &d)          // 32-bit row offsets: grad_out below 2 GB
{
    return (d.C == 16 || d.C == 32 || d.C == 64) && (size_t)d.B * d.Lq * d.H * d.C * 2 < kAccTrMaxBytes &&
           d.Lq < (1 << 24) && d.H * d.C * 2 < (1 << 24);
}
inline bool f32_matrix_shape_ok(const Dims &d)
{
    return d.C == 32 && (size_t)d.B * d.Lq * d.H * 32 * 4 < kAccTrMaxBytes && d.Lq < (1 << 24) && d.H * 128 < (1 << 24);
}
inline bool f32_mfma_ok(const Dims &d) { return opt(kOptAccF32) == 2 && f32_matrix_shape_ok(d); }
constexpr size_t kInstSplitMinPoints = 65536;
inline bool f32_split_ok(const Dims &d) { return opt(kOptAccF32) == 0 && f32_matrix_shape_ok(d); }
template <typename ST, bool INST> inline AccKind acc_kind(const Dims &d)
{
    if constexpr (!INST && std::is_same<ST, bf16_t>::value) {
        if (accumulate_tr_ok(d)) return kAccTr;
    }
    if constexpr (!INST && std::is_same<ST, float>::value) {
        if (f32_mfma_ok(d)) return kAccF32;
        if (f32_split_ok(d)) return kAccSplit;
    }
    if constexpr (INST && std::is_same<ST, float>::value) {
        // instance attention (round 6): the split kernel's INST flavour -- two upstream rows per record, combined before
        // the split; grad_mask rows by 32-bit offsets.  Only with many points a slice: the wide records cost the fill 3x
        // the bytes and the kernel runs at 2 waves per SIMD (C3, 19 k points a slice: 6 % slower than the VALU kernel;
        // C3', 235 k: accumulate -13 %, step -1.2 %)
        if (f32_split_ok(d) && (size_t)d.Lq * d.L * d.P >= kInstSplitMinPoints &&
            (size_t)d.B * d.Lq * d.P * d.H * 32 * 4 < kAccTrMaxBytes)
            return kAccSplit;
    }
    return kAccValu;
}
// the workspace query only knows the storage type and the dimensions: room for 16-byte records
// whenever a flavour of that type may write them
inline bool wide_workspace(bool is_bf16, const Dims &d)
{
    return is_bf16 ? accumulate_tr_ok(d) : f32_mfma_ok(d) || f32_split_ok(d);
}
constexpr int kMaxBlocks = 8192;      // per (image, head) slice: one LDS int each in bin_kernel

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

inline bool make_plan_blocks(const Dims &d, const int64_t *sh, const int64_t *ls, BinPlan &p)
{
    constexpr int BW = 8, BH = 4;
    p.L = d.L;
    long long blk0 = 0, next_start = 0;
    for (int l = 0; l < d.L; ++l) {
        const long long hl = sh[2 * l], wl = sh[2 * l + 1], st = ls[l];
        // standard packed layout only: level l starts where level l-1 ends (every grad_value
        // row then has exactly one owner block)
        if (hl < 0 || wl < 0 || hl > INT32_MAX || wl > INT32_MAX || st != next_start ||
            st + hl * wl > d.S)
            return false;
        next_start = st + hl * wl;
        p.lv[l].H = (int)hl;
        p.lv[l].W = (int)wl;
        p.lv[l].start = (int)st;
        p.lv[l].nbx = (int)((wl + BW - 1) / BW);
        p.lv[l].nby = (int)((hl + BH - 1) / BH);
        // blk_of(): floor(x nb / size) by multiply-high is exact while x nb < 2^32 / size
        if ((unsigned long long)wl * wl * p.lv[l].nbx >= (1ull << 32) ||
            (unsigned long long)hl * hl * p.lv[l].nby >= (1ull << 32))
            return false;
        p.lv[l].mw = wl > 1 ? (unsigned)((1ull << 32) / (unsigned long long)wl + 1) : 0u;
        p.lv[l].mh = hl > 1 ? (unsigned)((1ull << 32) / (unsigned long long)hl + 1) : 0u;
        // pack_block_geo(): n / nbx and n / nby by multiply-high for every n it can ask for, checked (once per
        // (range, divisor) pair of the process: make_plan runs on every call)
        const auto nb_magic = [](long long nmax, int nb) -> unsigned {
            if (nb <= 1 || nmax >= (1ll << 24)) return 0u;
            static std::mutex mu;
            static std::map<std::pair<long long, int>, unsigned> known;
            std::lock_guard<std::mutex> g(mu);
            const auto it = known.find({nmax, nb});
            if (it != known.end()) return it->second;
            unsigned m = (unsigned)((1ull << 32) / (unsigned long long)nb + 1);
            for (long long n = 0; n <= nmax && m; ++n)
                if ((unsigned)(((unsigned long long)n * m) >> 32) != (unsigned)(n / nb)) m = 0u;
            known[{nmax, nb}] = m;
            return m;
        };
        p.lv[l].mnx = nb_magic(std::max<long long>((long long)p.lv[l].nbx * p.lv[l].nby, (long long)p.lv[l].nbx * wl + p.lv[l].nbx), p.lv[l].nbx);
        p.lv[l].mny = nb_magic((long long)p.lv[l].nby * hl + p.lv[l].nby, p.lv[l].nby);
        p.lv[l].blk0 = (int)blk0;
        blk0 += (long long)p.lv[l].nbx * p.lv[l].nby;
    }
    // ... and the levels cover all of S: the binned kernels only store the rows a level block
    // owns, a padded tail (S > sum H_l W_l) would stay uninitialised (the atomic path zero-fills)
    if (next_start != d.S) return false;
    const long long rec_cap = 4ll * d.Lq * d.L * d.P;
    if (blk0 == 0 || blk0 > kMaxBlocks || rec_cap > INT32_MAX / 2 ||
        (long long)d.B * d.Lq * d.H > INT32_MAX || (long long)d.B * d.Lq * d.P * d.H > INT32_MAX)
        return false;
    int lp_bits = 0;
    while ((1ll << lp_bits) < (long long)d.L * d.P) ++lp_bits;
    if (((long long)d.Lq << lp_bits) > INT32_MAX) return false;
    if ((long long)d.L * d.P > (1 << 16)) return false;       // keeps per-workgroup point counts < 2^24
    p.lp_bits = lp_bits;
    p.n_slices = d.B * d.H;
    p.nblk = (int)blk0;
    p.rec_cap = (int)rec_cap;
    p.chunk = bin_chunk(d);
    p.item_cap = (int)(blk0 + rec_cap / p.chunk + 1);
    // sparse maps (fewer than ~2 expected records per block -- the BEV decoders: 1 000 queries against 468 x 468):
    // grad_value is zero-filled once and the empty blocks get no work item (BinPlan::min_items)
    // (such a plan is always built by launch_binning, which also fills the zero workers' geometry table: riders_ok
    // refuses it)
    p.min_items = blk0 > kScanThreads && 3ll * d.Lq * d.L * d.P / 2 < 2 * blk0 ? 0 : 1;
    p.zero_workers = p.min_items ? 0 : (int)((blk0 + kZeroPer - 1) / kZeroPer);
    // blocks with more than one chunk: sum of their chunk counts <= 2 * records / chunk
    p.pslot_cap = (int)std::min<long long>(2 * (rec_cap / p.chunk) + 2, blk0 + rec_cap / p.chunk + 1);
    // a chunk item carries {partial slot, ordinal of its block among the chunked ones} in one word (kItemSlotBits)
    if (p.pslot_cap >= (1 << kItemSlotBits) || blk0 >= (1 << (31 - kItemSlotBits))) return false;
    return true;
}

inline bool make_plan(const Dims &d, const int64_t *sh, const int64_t *ls, BinPlan &p)
{
    if (!sh || !ls || !d.valid() || fast_group(d) == 0 || d.L > kMaxBinLevels) return false;
    return make_plan_blocks(d, sh, ls, p);
}

std::atomic<float *> g_dense_dbg{nullptr};      // debugging aid: time stamps (boxattn_set_debug_buffer)

// The PLAN of a backward -- everything the binning knows before the records are written: per bin workgroup
// and block the first slot, per block the first record, the work-item list -- is what a training forward
// hands to its backward (a few hundred KB: 1.4 MB at BoxeR-R50 shapes).  The SCRATCH -- the records
// themselves, the fp32 partial tiles of chunked blocks -- only lives inside the backward call.
struct PlanLayout {
    size_t n_items, part, tickets, subtot, offsets, items, combos, scan_tmp, zgeo, total;
    int q_per_wg, n_wg;                                     // geometry of the bin passes
};
struct ScratchLayout { size_t records, partials, ctickets, total; };

// Is this a shape whose box-attention backward fills its bins in ONE pass (boxattn_spec.h)?  A property of the dimensions
// and the switches alone (the plan's layout -- the number of riders -- depends on it).  The ranges of that fill are
// planned from the PREVIOUS call's counts, so it is for maps whose blocks see about the same number of records from call to
// call: many records a block (an encoder's one query per pixel: 660 a block at BoxeR-R50 shapes, fixed by the geometry).
// 300 decoder queries leave ~9 records a block, anywhere: every call outgrows the ranges of the one before (measured, C3'':
// 104 us a step over changing inputs against 52 with the two-pass riders, profiles/r06_onepass_ab.log).
constexpr long long kSpecMinRecordsPerBlock = 192;
inline bool spec_shape(const Dims &d, const BinPlan &p)
{
    const int o = opt(kOptRiders);
    return (o == 0 || o == 2 || o == 3) && p.nblk <= kSpecMaxBlocks && p.min_items == 1 &&
           3ll * d.Lq * d.L * d.P / 2 >= kSpecMinRecordsPerBlock * p.nblk;
}

inline PlanLayout plan_layout(const Dims &d, const BinPlan &p)
{
    const size_t ns = (size_t)d.B * d.H;
    PlanLayout w;
    // ~2048 workgroups of 256 threads' worth of bin workgroups
    // ... and at most kScanSub * kScanWgPerSub per slice (the scan's two levels)
    // (a shape whose box-attention backward takes the one-pass fill gets more, shorter riders -- whatever the storage
    // type or operator of THIS call: the layout is a function of the dimensions and the switches alone)
    const long long wg_target = bin_wg_target(spec_shape(d, p), (long long)d.Lq * (long long)ns);
    w.q_per_wg = std::max(8, (int)(((long long)d.Lq * (long long)ns + wg_target - 1) / wg_target));
    w.q_per_wg = std::max(w.q_per_wg, (d.Lq + kScanSub * kScanWgPerSub - 1) /
                                          (kScanSub * kScanWgPerSub));
    // ... and every workgroup initialises, flushes and has scanned one counter per block: give it
    // at least 4 points per block (few queries on a big map: 1 000 queries on 468 x 468 = 6 903
    // blocks per slice now use 1 workgroup per slice instead of 67)
    const long long lp = (long long)d.L * d.P;
    w.q_per_wg = (int)std::max<long long>(w.q_per_wg, (4ll * p.nblk + lp - 1) / lp);
    w.n_wg = (d.Lq + w.q_per_wg - 1) / w.q_per_wg;
    size_t o = 0;
    w.n_items = o; o += align_up(ns * 2 * 4);
    w.tickets = o; o += align_up(ns * kRideTickets * 4);        // the riders' hand-offs (boxattn_scan_tail.h)
    w.part = o;    o += align_up(ns * (size_t)w.n_wg * (size_t)p.nblk * 4);
    w.subtot = o;  o += align_up(ns * kScanSub * (size_t)p.nblk * 4);
    w.offsets = o; o += align_up(ns * (p.nblk + 1) * 4);
    w.items = o;   o += align_up(ns * p.item_cap * 16);
    w.combos = o;  o += align_up(ns * (size_t)p.nblk * 16);
    // multi-workgroup block scan (more than kScanThreads blocks per slice): per-block prefixes
    // inside a segment + the segments' totals
    w.scan_tmp = o;
    if (p.nblk > kScanThreads) o += align_up(ns * ((size_t)p.nblk + kMaxBlocks / kScanThreads) * 16);
    w.zgeo = o;                                                 // sparse maps: block geometry for the zero workers
    if (p.zero_workers > 0) o += align_up((size_t)p.nblk * 8);
    w.total = o;
    return w;
}
// `wide`: 16-byte records {id, x, y, weight} instead of 4-byte point ids
inline ScratchLayout scratch_layout(const Dims &d, const BinPlan &p, bool wide)
{
    const size_t ns = (size_t)d.B * d.H;
    ScratchLayout w;
    size_t o = 0;
    w.ctickets = o; o += align_up(ns * (size_t)p.nblk * 4);      // in-launch combine (chunk_finish)
    w.records = o;  o += align_up(ns * (size_t)p.rec_cap * (wide ? 16 : 4));
    w.partials = o; o += align_up(ns * (size_t)p.pslot_cap * 32 * d.C * 4);
    w.total = o;
    return w;
}

// The caller's state buffer (include/boxattn.h): [kDenseStatSlots pairs of uint64 locality counters = 1 KiB]
// [tickets: B * H * kRideTickets ints].  The counters come FIRST, at a fixed place: one buffer serves calls of
// every shape on its stream, and what one shape's calls add to must never be another shape's tickets.
constexpr size_t kStatBytes = (size_t)kDenseStatSlots * 2 * sizeof(unsigned long long);

// May the count / fill passes and the scans of this plan run as riders (boxattn_ride.h)?
inline bool riders_ok(const BinPlan &plan, const PlanLayout &w)
{
    return opt(kOptRiders) != 1 && plan.nblk <= kRideMaxBlocks && plan.min_items == 1 &&
           w.n_wg <= kScanSub * kScanWgPerSub;
}

// ------------------------------------------------- the one-pass fill's state (boxattn_spec.h)
// Does the backward of this call fill its bins in one pass, into ranges kept in the caller's state buffer?  Box
// attention with 16-byte records (the matrix-core accumulates), maps the riders' two LDS arrays hold.
template <typename ST, bool INST> inline bool spec_ok(const Dims &d, const BinPlan &plan, const PlanLayout &pl)
{
    if (INST || std::is_same<ST, double>::value) return false;
    return spec_shape(d, plan) && acc_kind<ST, INST>(d) != kAccValu && riders_ok(plan, pl) &&
           (long long)d.Lq * d.P < (1 << 24) && plan.rec_cap < (1 << 28) &&
           d.n_qh() * (size_t)d.L * (size_t)d.P < ((size_t)1 << 29);
}

// The state buffer: [locality counters 1 KiB][one-pass counters 64 B][tickets][cursor][cbase][redo] -- the last three only
// for shapes the binned backward plans (every such shape: the size does not depend on the storage type or the switches).
struct StateLayout { size_t spec_stats, tickets, cursor, cbase, redo, total; };
inline StateLayout state_layout(const Dims &d, const BinPlan *plan)
{
    const size_t ns = (size_t)std::max(0, d.B) * (size_t)std::max(0, d.H);
    StateLayout w;
    size_t o = kStatBytes;
    w.spec_stats = o; o += 64;
    w.tickets = o;    o += align_up(ns * kRideTickets * sizeof(int));
    w.cursor = w.cbase = w.redo = o;
    if (plan && plan->nblk <= kSpecMaxBlocks) {
        w.cursor = o; o += align_up(ns * (size_t)plan->nblk * sizeof(int));
        w.cbase = o;  o += align_up(ns * ((size_t)plan->nblk + 1) * sizeof(int));
        w.redo = o;   o += align_up(ns * ((size_t)plan->nblk + 1) * sizeof(int2));
    }
    w.total = o;
    return w;
}

// What the library remembers (host side) of the state buffers it has seen: the shape a buffer serves and whether a call
// has planned its ranges.  A buffer is the caller's -- zeroed once, one per (stream, dimensions, level shapes) -- and a
// zeroed buffer is a valid state (every range empty), so nothing here decides RESULTS: an unknown or forgotten buffer is
// "cold" (the call runs the two-pass passes and plans the ranges), and a buffer that turns up with another shape is
// zeroed first (its tickets and ranges mean nothing to this shape).
struct StateShadow { Dims d; unsigned long long geo; bool learned; };
std::mutex g_state_mu;
std::map<const void *, StateShadow> g_state_shadow;
inline bool same_dims(const Dims &a, const Dims &b)
{
    return a.B == b.B && a.S == b.S && a.H == b.H && a.C == b.C && a.L == b.L && a.Lq == b.Lq && a.P == b.P;
}
// -> -1 unusable (misaligned / too small), 0 no state, 1 cold, 2 its ranges are planned
// fresh (BOXATTN_HINT_FRESH_STATE): the caller has zeroed the buffer since its last call -- a new buffer, possibly at an
// address another one had: whatever is remembered of the address is dropped
inline int state_check(void *state, size_t bytes, const StateLayout &sy, const Dims &d, const int64_t *sh, hipStream_t st,
                       bool fresh = false)
{
    if (!state) return 0;
    if (!aligned(state, 8) || bytes < sy.total) return -1;
    unsigned long long geo = 1469598103934665603ull;                    // FNV-1a over the level shapes
    for (int i = 0; sh && i < 2 * d.L; ++i) geo = (geo ^ (unsigned long long)sh[i]) * 1099511628211ull;
    std::lock_guard<std::mutex> g(g_state_mu);
    if (fresh) g_state_shadow.erase(state);
    const auto it = g_state_shadow.find(state);
    if (it == g_state_shadow.end()) {
        if (g_state_shadow.size() >= 4096) g_state_shadow.clear();
        g_state_shadow[state] = StateShadow{d, geo, false};
        return 1;
    }
    if (same_dims(it->second.d, d) && it->second.geo == geo) return it->second.learned ? 2 : 1;
    if (zero_async((char *)state + kStatBytes, bytes - kStatBytes, st) != hipSuccess) return -1;
    it->second = StateShadow{d, geo, false};
    return 1;
}
inline void state_learned(const void *state)
{
    std::lock_guard<std::mutex> g(g_state_mu);
    const auto it = g_state_shadow.find(state);
    if (it != g_state_shadow.end()) it->second.learned = true;
}

// ------------------------------------------------- window-staged encoder kernels (boxattn_dense.h)
// Encoder case: one query per pixel of packed levels, bf16 storage, C = 32, 2x2 points, <= 4 levels.
// elem: bytes per stored element -- 2: bf16 storage (64-byte pixels, geometry in 16-byte units, boxattn_dense.h),
// 4: float32 storage (128-byte pixels, 32-byte units, boxattn_dense_f32.h)
inline bool make_dense_plan(const Dims &d, const int64_t *sh, const int64_t *ls, DensePlan &p, int elem)
{
    if (!sh || !ls || !d.valid() || opt(kOptDense) == 1 || g_variant == 1 || g_variant == 2) return false;
    if (d.Lq != d.S || d.C != 32 || d.P != 4 || d.L > kDenseMaxLevels || d.B < 1 || d.S < 1) return false;
    if ((size_t)d.B * d.Lq * d.H * d.L * d.P >= (1ull << 31)) return false;       // 32-bit point ids
    if (d.n_value() * (size_t)elem >= kOobOffset) return false;
    const int unit = 8 * elem;                                  // bytes per unit of a window's pitch / offset
    const int budget = elem == 2 ? kDenseZeroOff : BOXATTN_DENSE_F32_LDS - 128;      // (minus the forwards' row of zeros)
    p = DensePlan{};
    p.dbg = g_dense_dbg.load();
    p.L = d.L; p.B = d.B; p.Lq = d.Lq; p.S = d.S; p.H = d.H;
    const auto magic = [](int dd) { return dd == 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned)dd); };
    p.mag_h = magic(d.H);
    long long next = 0;
    for (int l = 0; l < d.L; ++l) {
        const long long hl = sh[2 * l], wl = sh[2 * l + 1];
        if (hl <= 0 || wl <= 0 || hl > 32000 || wl > 32000 || ls[l] != next) return false;
        next += hl * wl;
        DenseLevel &v = p.lv[l];
        v.H = (int)hl; v.W = (int)wl; v.start = (int)ls[l];
        v.ntx = (int)((wl + kDenseTile - 1) / kDenseTile);
        v.ntiles = v.ntx * (int)((hl + kDenseTile - 1) / kDenseTile);
        if ((long long)v.ntiles * d.B * d.H >= (1 << 24)) return false;      // x * n and the block index < 2^31
        v.n_all = (unsigned)(v.ntiles * d.B);
        v.mag_ntx = magic(v.ntx);
        v.mag_ntiles = magic(v.ntiles);
    }
    if (next != d.S) return false;
    // windows: a tile's queries sit at pixel coordinate (qx + 0.5) r - 0.5 of the sampled level
    // (r = W_l / W_lq); their points lie a quarter box (ref / 4 pixels of the query's level, i.e.
    // ref / 4 * r here) to either side, the predicted offset may move them `jit` quarters further.
    // The windows share the workgroup's kDenseLdsBytes of LDS; what does not fit (a coarse tile's window on a
    // fine level) is not staged.
    constexpr float ref4 = 4.0f / 4.0f;     // expected box: 4 pixels of the query's own level (BoxeR's reference windows)
    constexpr float jit = 2.5f;             // margin for the predicted offsets, in box quarters
    for (int lq = 0; lq < d.L; ++lq) {
        int used = 0;
        // the tile's OWN level first -- its window serves the most points and must never be the one that does not
        // fit (odd map sizes make the coarser windows a column or a row larger: 99 x 167 next to 50 x 84 is a ratio
        // of 0.505, a 10 x 10 window instead of 9 x 9) -- then the others, coarsest first
        for (int k = 0; k < d.L; ++k) {
            const int l = k == 0 ? lq : (d.L - k <= lq ? d.L - k - 1 : d.L - k);
            DenseWin &w = p.win[lq][l];
            const float rx = (float)p.lv[l].W / (float)p.lv[lq].W, ry = (float)p.lv[l].H / (float)p.lv[lq].H;
            const float mx = rx * ref4 * (1.0f + jit), my = ry * ref4 * (1.0f + jit);
            const int cols = (int)std::ceil(rx * (kDenseTile - 1) + 2 * mx) + 2;
            const int rows = (int)std::ceil(ry * (kDenseTile - 1) + 2 * my) + 2;
            // 16.16 fixed point (a placement heuristic: any rounding will do, tile columns < 2^12)
            w.ax = (int)std::lround(kDenseTile * rx * 65536.0f); w.bx = (int)std::floor((0.5f * rx - 0.5f - mx) * 65536.0f);
            w.ay = (int)std::lround(kDenseTile * ry * 65536.0f); w.by = (int)std::floor((0.5f * ry - 0.5f - my) * 65536.0f);
            if (rx > 8.0f || ry > 8.0f) { w.ax = w.ay = 0; }      // (not staged anyway; keeps tx * ax inside 31 bits)
            const int need = rows * dense_win_pitch16(cols);       // in units (a pixel = 4 units, a row = its pixels + 1)
            const bool fits = cols <= kDenseWinMax && rows <= kDenseWinMax && unit * (used + need) <= budget;
            w.geo = dense_win_pack(fits ? rows : 0, fits ? cols : 0, used);
            if (fits) used += need;
        }
    }
    return true;
}

