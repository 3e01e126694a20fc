// Developer diagnostics compiled into libtaco_hip.so but NOT part of the product C-ABI (include/taco_hip.h):
// bound by hand in scripts/dev_*.py, never by tacotron_multispeaker_amd/_lib.py.
#pragma once
#include <hip/hip_runtime.h>
extern "C" {
// micro-benchmark of the cluster all-gather (scripts/dev_xchg.py): microseconds per exchange round.
// mode: 0 = agent-scope granules (product protocol), 1 = same-XCD L2 granules (plain store + sc1 load; only valid when
// every member of a cluster reports the same HW_REG_XCC_ID -- the kernel checks this and raises err = 2 otherwise)
int taco_dev_xchg_bench(void* xchg, int* err, float* sink, int nclus, int cw, int len, int iters, int same_xcd, int sleep,
                        int threads, int mode, hipStream_t stream);
}
extern "C" {
// synthetic background load for interference studies (scripts/dev_interfere.py; engine.py TACO_DEV_LOAD, developer knob):
// mode 0 = HBM/L2 streaming reads over [p, p + bytes) `reps` times, mode 1 = fp32 MFMA chain without memory traffic for `reps`
// iterations; wgs workgroups of 256 threads with lds_bytes of dynamic LDS each (occupancy control)
int taco_dev_load(const float* p, long bytes, int reps, int mode, int wgs, int lds_bytes, float* sink, hipStream_t stream);
}
