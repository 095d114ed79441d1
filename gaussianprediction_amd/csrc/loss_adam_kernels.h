// loss_adam_kernels.h -- the multi-tensor Adam launch's table and chunk body (shared by gp_adam_multi_kernel and by the launch that
// carries the same chunks beside the keypoint MLP's data backward, deform_mlp_small.hip), and the "rider" slot through which
// gp_train_step_run hands that launch an optimizer table.  Everything else of loss_adam_kernels.hip is private to it.
#pragma once
#include "gp_common.h"

#define ADAM_MAX_TENSORS 32
#define ADAM_CHUNK 16384   // elements per workgroup-chunk
struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    unsigned long long n[ADAM_MAX_TENSORS];
    float step_size[ADAM_MAX_TENSORS];           // lr / (1 - beta1^t) with the TENSOR's step count t
    float bc2_sqrt[ADAM_MAX_TENSORS];            // sqrt(1 - beta2^t)
    unsigned keep_grad_mask;                     // bit k: leave tensor k's gradient as it is
    unsigned chunk_begin[ADAM_MAX_TENSORS + 1];   // prefix of chunk counts
    int count;
};

// One chunk (ADAM_CHUNK elements of one tensor) by a workgroup of THREADS threads: element-wise, so the result does not depend on
// THREADS or on which launch carries the chunk.
template <int THREADS>
__device__ __forceinline__ void adam_chunk_body(const AdamTable& t, unsigned chunk, int tid, float b1, float b2, float eps, int zero_grad,
                                                const uint32_t* __restrict__ skip_flag) {
    const bool skip = skip_flag && *skip_flag != 0;      // the frame that produced these gradients was invalid: no update
    int k = 0;
    while (k + 1 < t.count && chunk >= t.chunk_begin[k + 1]) ++k;
    const size_t base = (size_t)(chunk - t.chunk_begin[k]) * ADAM_CHUNK;
    const size_t n = t.n[k];
    float* __restrict__ p = t.p[k]; float* __restrict__ g = t.g[k]; float* __restrict__ m = t.m[k]; float* __restrict__ v = t.v[k];
    const float step_size = t.step_size[k], bc2_sqrt = t.bc2_sqrt[k];
    if ((t.keep_grad_mask >> k) & 1u) zero_grad = 0;
    const size_t end = base + ADAM_CHUNK < n ? base + ADAM_CHUNK : n;
    auto upd = [&](float4& pv, const float4& gv, float4& mv, float4& vv) {
        float* pp = (float*)&pv; const float* gg = (const float*)&gv; float* mm = (float*)&mv; float* vq = (float*)&vv;
#pragma unroll
        for (int u = 0; u < 4; ++u) gp_adam_update(pp[u], gg[u], mm[u], vq[u], b1, b2, eps, step_size, bc2_sqrt);
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t i = base + (size_t)tid * 4;
    if (skip) {             // discard the gradients (where this pass owns their zeroing), leave p / m / v untouched
        if (zero_grad)
            for (size_t j = base + tid; j < end; j += THREADS) g[j] = 0.f;
        return;
    }
    // two independent 16-byte streams per thread and iteration: 8 loads in flight per lane
    constexpr size_t S = (size_t)THREADS * 4;
    for (; i + S + 3 < end; i += 2 * S) {
        const size_t j = i + S;
        float4 pa = *(float4*)(p + i), ga = *(float4*)(g + i), ma = *(float4*)(m + i), va = *(float4*)(v + i);
        float4 pb = *(float4*)(p + j), gb = *(float4*)(g + j), mb = *(float4*)(m + j), vb = *(float4*)(v + j);
        upd(pa, ga, ma, va);
        upd(pb, gb, mb, vb);
        *(float4*)(p + i) = pa; *(float4*)(m + i) = ma; *(float4*)(v + i) = va;
        *(float4*)(p + j) = pb; *(float4*)(m + j) = mb; *(float4*)(v + j) = vb;
        if (zero_grad) { *(float4*)(g + i) = zero4; *(float4*)(g + j) = zero4; }
    }
    for (; i < end; i += S) {
        if (i + 3 < n) {
            float4 pv = *(float4*)(p + i), gv = *(float4*)(g + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
            upd(pv, gv, mv, vv);
            *(float4*)(p + i) = pv; *(float4*)(m + i) = mv; *(float4*)(v + i) = vv;
            if (zero_grad) *(float4*)(g + i) = zero4;
        } else {
            for (size_t j = i; j < n; ++j) {
                gp_adam_update(p[j], g[j], m[j], v[j], b1, b2, eps, step_size, bc2_sqrt);
                if (zero_grad) g[j] = 0.f;
            }
        }
    }
}

// The rider: an optimizer launch whose tensors need nothing the keypoint MLP's backward produces (the per-Gaussian tensors: HBM-bound,
// every CU) travels in the SAME launch as that backward's data kernel (16 workgroups, bound by the rate at which one CU takes the
// weights in) -- gp_train_step_run arms the slot, gp_mlp_backward's small-row path consumes it; an unconsumed rider is launched on
// its own (gp_adam_rider_flush).  One slot per host thread: arm, consume and flush happen inside ONE call of gp_train_step_run.
struct GpAdamRider {
    AdamTable t;
    float b1, b2, eps;
    int zero_grad;
    const uint32_t* skip_flag;
    unsigned chunks;
    bool armed;
};
GpAdamRider* gp_adam_rider_slot();
// fills the slot from the optimizer's arrays (the arguments of gp_adam_step_multi_steps); armed unless there is nothing to update
int gp_adam_rider_arm(int count, float* const* params, float* const* grads, float* const* exp_avgs, float* const* exp_avg_sqs,
                      const int64_t* numels, const float* lrs, const int64_t* steps, float beta1, float beta2, float eps, int zero_grad,
                      uint32_t keep_grad_mask, const uint32_t* skip_flag);
int gp_adam_rider_flush(hipStream_t s);     // launches an armed rider as a plain gp_adam_multi_kernel and disarms the slot
