// shared between decoder.hip (per-step fallback + dispatcher) and attn_cluster.hip (persistent cluster kernels)
#pragma once
#include "common.hpp"
typedef unsigned long long u64;
struct AttnClu {
    const float *w1c, *f1, *w2, *b2, *wx, *whg, *whc, *bg, *wq, *v, *keys, *mem;
    float *p1, *p2, *r, *u, *c, *rh, *hc, *q, *align;
    u64* xchg; int* err;
    int N, S, Ti, s0, s1;
    int xcd_local;                   // 1: clusters whose members share an XCD use the L2-local granule form (set by the launcher)
};
int attn_cluster_fwd_launch(const AttnClu& p, hipStream_t st);
struct AttnCluB {
    const float *w1c, *w2, *wx, *whg, *whc, *wq, *v, *keys, *mem;
    const float *p1, *p2, *r, *u, *c, *hc, *q, *align, *dhc;
    const float* da_ext;             // optional [N,S,Ti]: extra gradient wrt the alignments (regularisers), or nullptr
    float *dxp, *dp2, *dp1, *dq, *de, *dctx;
    float *dhcarry, *dctxcarry;      // [N,256] each: state handed between chunk launches
    u64* xchg; int* err;
    int N, S, Ti, s0, s1;
    int xcd_local;
    int carry_flags;                 // TACO_ATTN_CARRY_*: hand the chunk-to-chunk carries over as granules (taco_attn_rnn_bwd_chunk)
    u64* carry_xchg;                 // exchange buffer whose carry region / residency counter is used (the POSTING launch's buffer)
};
int attn_cluster_bwd_launch(const AttnCluB& p, float* dkeys, float* dmem, float* dvpart, hipStream_t st);
int attn_cluster_bwd_reduce(const AttnCluB& p, float* dkeys, float* dmem, float* dvpart, hipStream_t st);
