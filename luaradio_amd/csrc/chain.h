// chain.h - lrhip_chain: fused linear runs of stages, pinned ring pipeline, host_execute
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// chain
// =====================================================================================================
struct lrhip_chain {
    struct Op {
        lrhip_stage *stage;
        bool owned;
    };
    std::vector<Op> ops;
    std::vector<std::unique_ptr<DeviceBuf>> edges;   // edges[i] = output of op i (all but the last)
    PinnedBuf h_in, h_out;
    DeviceBuf d_in, d_out;
    int last_launches = 0;
    unsigned flags = 0;                    // lrhip_chain_create_ex
    unsigned long long discard_in = 0;     // lrhip_chain_start_at: input samples whose output is still to be thrown away
    DeviceBuf d_discard;
    double max_latency = 0.0, fill_t0 = 0.0;   // lrhip_chain_set_latency: bound on how long a pushed sample waits for its batch
    // ---- pipelined ring (lrhip_chain_set_ring)
    struct Slot {
        PinnedBuf h_in, h_out;
        DeviceBuf d_in, d_out;
        hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;   // H2D done, kernels done, D2H done
        bool used = false;          // events have been recorded at least once
        long n_out = 0;
    };
    std::vector<std::unique_ptr<Slot>> ring;
    unsigned long ring_chunk = 0, ring_out_cap = 0;      // input samples per slot; output samples a slot can hold
    unsigned long fill = 0;                // lrhip_chain_push: samples accumulated in the head slot's pinned input, not yet launched
    unsigned head = 0, inflight = 0;       // next slot to submit into; chunks submitted and not collected
    hipStream_t s_in = nullptr, s_out = nullptr;
    ~lrhip_chain()
    {
        for (auto &sl : ring) {
            if (sl->ev_in) (void)hipEventDestroy(sl->ev_in);
            if (sl->ev_done) (void)hipEventDestroy(sl->ev_done);
            if (sl->ev_out) (void)hipEventDestroy(sl->ev_out);
        }
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_out) (void)hipStreamDestroy(s_out);
        for (auto &o : ops)
            if (o.owned) delete o.stage;
    }
};

// host-pointer path shared by stages and chains: pinned staging in, run, pinned staging out
template <typename Runner>
static long host_execute(PinnedBuf &h_in, PinnedBuf &h_out, DeviceBuf &d_in, DeviceBuf &d_out, int in_size, int out_size,
                         unsigned long max_out, const void *in_host, unsigned long n_in, void *out_host,
                         unsigned long out_capacity, Runner run)
{
    if (n_in && !in_host) return set_error("null input buffer");
    size_t in_bytes = (size_t)n_in * in_size;
    unsigned long cap = max_out < out_capacity ? max_out : out_capacity;
    if (max_out > out_capacity) return set_error("output capacity %lu < required %lu", out_capacity, max_out);
    if (max_out && !out_host) return set_error("null output buffer");
    if (h_in.reserve(in_bytes ? in_bytes : 16) || d_in.reserve(in_bytes ? in_bytes : 16)) return -1;
    if (h_out.reserve((size_t)cap * out_size + 16) || d_out.reserve((size_t)cap * out_size + 16)) return -1;
    if (in_bytes) {
        host_copy(h_in.p, in_host, in_bytes);
        LR_HIP(hipMemcpyAsync(d_in.p, h_in.p, in_bytes, hipMemcpyHostToDevice, ctx().stream));
    }
    long n_out = run(d_in.p, n_in, d_out.p, cap);
    if (n_out < 0) return n_out;
    if (n_out) LR_HIP(hipMemcpyAsync(h_out.p, d_out.p, (size_t)n_out * out_size, hipMemcpyDeviceToHost, ctx().stream));
    LR_HIP(hipStreamSynchronize(ctx().stream));
    if (n_out) host_copy(out_host, h_out.p, (size_t)n_out * out_size);
    return n_out;
}
