// 8-byte {epoch, value} granules for hand-offs between the workgroups of one persistent cluster (Guideline 16 form R2:
// the data is the flag; ONE aligned 8-byte store per granule, polled with agent-scope relaxed loads = `global_load sc1`,
// which bypass the reader's L1 and are served by its XCD's L2 or the fabric behind it).
//
// Two store forms:
//   put_granule      agent-scope store (`global_store_dwordx2 sc1`): written through to the memory side, visible to every XCD.
//                    Placement-independent; ~0.55 us per exchange round even when all members share an XCD, because the
//                    store drops the line from the writer's L2 and every poll goes out to the fabric.
//   put_granule_xcd  workgroup-scope store (`global_store_dwordx2 sc0`): stays in the writer's XCD L2 -- the coherence point
//                    of all 32 CUs of that XCD -- where a same-XCD reader's sc1 load hits it.
//                    ONLY legal when writer and reader are on the same XCD.  That is never assumed from blockIdx: every
//                    cluster runs cluster_on_one_xcd() first, which exchanges the members' HW_REG_XCC_ID through the
//                    agent-scope form and returns true only if all of them match; otherwise the cluster keeps the
//                    agent-scope form for the whole launch.  A different placement therefore changes speed, never results.
#pragma once
#include <hip/hip_runtime.h>
typedef unsigned long long u64;

__device__ __forceinline__ void put_granule(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void put_granule_xcd(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}
#define TACO_XCD_TAG 0x58430000u     // 'XC..': epochs of the placement granules = this + the launch's first step (step epochs stay far below)
#define TACO_XCD_SPIN (1 << 22)
// Workgroup-wide form: lane i < cw polls member i's placement granule; the verdict is broadcast through the LDS int *flag_l.
// Contains two workgroup barriers: every thread of the workgroup must call it.  allow = 0 skips the exchange (agent scope).
// salt: distinguishes the launches that share one zero-filled granule buffer (the chunk launches of one recurrence pass: the
// chunk's first step), so a placement granule of an earlier launch is never taken for this launch's.
__device__ __forceinline__ bool cluster_shares_xcd(u64* slots, int w, int cw, int* err, int* flag_l, int tid, int allow,
                                                   unsigned salt = 0) {
    if (tid == 0) *flag_l = allow ? 1 : 0;
    __syncthreads();
    if (allow && tid < cw) {
        const unsigned mine = xcc_id();
        const unsigned tag = TACO_XCD_TAG + salt;
        if (tid == 0) __hip_atomic_store(slots + w, ((u64)tag << 32) | (u64)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64 x;
        int spins = 0;
        bool ok = true;
        for (;;) {
            x = __hip_atomic_load(slots + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(x >> 32) == tag) break;
            if (++spins > TACO_XCD_SPIN) { if (err) atomicExch(err, 1); ok = false; break; }
        }
        if (!ok || (unsigned)x != mine) *flag_l = 0;
    }
    __syncthreads();
    // placement statistics next to the error word (err[0] = hand-off timeout): err[2] counts the clusters that ran the placement
    // check, err[1] those that keep the agent-scope form although the L2-local form was allowed (members on different XCDs)
    if (allow && err && tid == 0 && w == 0) { atomicAdd(err + 2, 1); if (*flag_l == 0) atomicAdd(err + 1, 1); }
    return *flag_l != 0;
}

// One lane per workgroup calls this (then broadcasts the result through LDS).  slots: >= cw zeroed granules of this cluster.
__device__ __forceinline__ bool cluster_on_one_xcd(u64* slots, int w, int cw, int* err) {
    const unsigned mine = xcc_id();
    __hip_atomic_store(slots + w, ((u64)TACO_XCD_TAG << 32) | (u64)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool same = true;
    for (int i = 0; i < cw; ++i) {
        u64 x;
        int spins = 0;
        for (;;) {
            x = __hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(x >> 32) == TACO_XCD_TAG) break;
            if (++spins > TACO_XCD_SPIN) { if (err) atomicExch(err, 1); return false; }
        }
        same = same && ((unsigned)x == mine);
    }
    return same;
}
