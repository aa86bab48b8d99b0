// Riders: the overhead passes of the destination-binned backward (count, scan, fill, combine -- they
// move no algorithmic bytes) as extra workgroups INSIDE the launches of the kernels that do the
// operator's work, instead of launches of their own on the critical path:
//
//   training forward   [count riders -> per-slice scans by the last arrivers]  inside the forward kernel (two-pass
//                      binning; shapes on the one-pass fill of boxattn_spec.h carry nothing here)
//   backward, launch 1 [fill riders]                                           inside the point-gradient kernel
//   backward, launch 2 accumulate; the partial tiles of a chunked block are summed by the block's last
//                      arriver (boxattn_combine.h chunk_finish)
//
// The reference does its whole backward in one launch (box_attn_kernel.cuh:352-472, dispatch :1147-1200);
// round 3 ran count -> forward -> fill -> accumulate -> point gradients back to back on one stream, the
// two bin passes alone 13 + 25 us of a 143 us step.  As riders they overlap with their host kernel's tiles
// where two streams did not -- at a price: a rider has the host kernel's register / LDS footprint, so it
// takes a tile workgroup's place for as long as it lives, and it is bound by VALU latency (one wave per
// SIMD), not by memory.  FEW, FAT riders in front of the grid (one per CU, the next step's locations
// always in flight) cost least: C2 bf16 step 146 / 139 / 135 / 154 us with 1 024 / 512 / 256 / 128
// riders, 176 us interleaved with the tiles (profiles/r04_rider_sweep.log).
//
// Placement.  Rider workgroups come in GROUPS of 8 consecutive workgroup indices -- workgroup b runs on
// XCD b % 8 (observed; only speed depends on it), so a group puts one rider on every XCD and the host
// kernel's own "index mod 8 = XCD" mapping survives -- `lead` groups in front of the grid, then one group
// every 2^shift groups until `n_inter` more are placed.  No workgroup ever waits for another one: every
// hand-off inside a launch is a ticket whose LAST arriver does the next stage (no spinning, no assumption
// about dispatch order or residency).
#pragma once
#include "boxattn_device.h"

namespace boxattn {

struct RideGrid {
    unsigned lead;        // rider groups in front of everything
    unsigned n_inter;     // rider groups interleaved with the kernel's own groups, one every 2^shift groups
    unsigned shift;
    unsigned n_riders;    // rider workgroups that have work (<= 8 (lead + n_inter)); the surplus exits
};

struct RideRole {
    bool rider;
    unsigned id;          // rider index, or the host kernel's own workgroup index
};
// (blockIdx.x lives in a scalar register: this is a dozen scalar instructions)
__device__ __forceinline__ RideRole ride_role(unsigned b, const RideGrid rg)
{
    const unsigned g = b >> 3, x = b & 7u;
    RideRole r;
    if (g < rg.lead) { r.rider = true; r.id = b; return r; }
    const unsigned gi = g - rg.lead, span = rg.n_inter << rg.shift;
    if (gi < span) {
        const unsigned k = gi >> rg.shift, rem = gi & ((1u << rg.shift) - 1u);
        r.rider = rem == 0;
        r.id = r.rider ? (((rg.lead + k) << 3) | x) : (((gi - k - 1u) << 3) | x);
        return r;
    }
    r.rider = false;
    r.id = ((gi - rg.n_inter) << 3) | x;
    return r;
}

// host: place n_riders rider workgroups into a launch of own_blocks workgroups; *total = grid size
inline RideGrid ride_grid(unsigned n_riders, unsigned own_blocks, unsigned shift, unsigned *total)
{
    RideGrid rg{0u, 0u, shift, n_riders};
    const unsigned groups_r = (n_riders + 7u) / 8u, own_g = (own_blocks + 7u) / 8u;
    if (shift > 0) {
        const unsigned per = (1u << shift) - 1u;            // own groups behind every interleaved rider group
        rg.n_inter = own_g / per < groups_r ? own_g / per : groups_r;
    }
    rg.lead = groups_r - rg.n_inter;
    *total = 8u * (groups_r + own_g);
    return rg;
}

// ---- agent-scope (write-through / L1-bypassing) accesses for hand-offs inside a launch.  The form is
// MI355X_MICROARCH.md's "sc1 payload -> s_waitcnt vmcnt(0) by every storing wave -> [barrier] -> relaxed
// agent-scope ticket; the consumer reads the payload with sc1 loads": write-through stores need no
// release fence (a release writes back the XCD's whole L2, which the host kernel keeps full of dirty
// output lines: +4 us measured in round 3), sc1 loads no acquire.
typedef unsigned int ride_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void agent_store(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int agent_load(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stores_left()        // this wave's stores have left (every storing wave, before the ticket)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// Take a ticket on `t`: true for the last of `n` arrivers (workgroup-uniform), who also resets the
// ticket for the next call.  All waves of the workgroup must have passed stores_left() before.
template <int THREADS> __device__ __forceinline__ bool last_arriver(int *t, int n, int *lds_flag)
{
    if constexpr (THREADS > 64) __syncthreads();
    if (threadIdx.x == 0) {
        const int got = __hip_atomic_fetch_add(t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got == n - 1) agent_store(t, 0);
        *lds_flag = got == n - 1;
    }
    if constexpr (THREADS > 64) __syncthreads();
    else wave_lds_sync();
    return *lds_flag != 0;
}

}  // namespace boxattn
