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

// ---- caller-owned host memory the driver may DMA from / to directly (lrhip_host_register) ------------------------------------------------
// LuaRadio's owning vectors and pipe read buffers are page-aligned and long-lived (radio/core/vector.lua:19-37, radio/core/pipe.lua:72-76), so the
// staging copies of the host-pointer path (caller vector -> pinned slot -> device and back) are two avoidable passes over host memory per direction:
// a registered range is pinned where it lies and hipMemcpyAsync reads / writes it.  Ranges are kept by the library only to answer "is this pointer inside
// one?" - the owner registers and unregisters (before it frees).
struct HostRanges {
    std::mutex m;
    std::vector<std::pair<const char *, size_t>> r;
    std::vector<char *> dev;         // the device-side address of each range (hipHostGetDevicePointer; null: not mapped)
    long pid = 0;
    // the device address of a host pointer inside a registered range, or null
    void *device_ptr(const void *p, size_t bytes)
    {
        std::lock_guard<std::mutex> lk(m);
        if (pid != (long)getpid()) return nullptr;
        const char *q = (const char *)p;
        for (size_t i = 0; i < r.size(); i++)
            if (q >= r[i].first && q + bytes <= r[i].first + r[i].second) return dev[i] ? dev[i] + (q - r[i].first) : nullptr;
        return nullptr;
    }
    bool has(const void *p, size_t bytes)
    {
        if (!bytes) return true;
        std::lock_guard<std::mutex> lk(m);
        if (pid != (long)getpid()) { r.clear(); dev.clear(); pid = (long)getpid(); }      // registrations do not survive fork()
        const char *q = (const char *)p;
        for (auto &e : r)
            if (q >= e.first && q + bytes <= e.first + e.second) return true;
        return false;
    }
};
static HostRanges &host_ranges()
{
    static HostRanges h;
    return h;
}
// copy streams of the piece-wise host path (per process, created on first use)
struct HostPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev;      // 3 per piece: H2D done, kernels done, D2H done
    std::mutex m;                    // held for the whole of a piece-wise call (host_execute)
    long pid = 0;
    int ensure(size_t pieces)
    {
        if (pid != (long)getpid()) { s_in = s_out = nullptr; ev.clear(); pid = (long)getpid(); }
        if (!s_in) LR_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
        if (!s_out) LR_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
        while (ev.size() < 3 * pieces) {
            hipEvent_t e = nullptr;
            LR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev.push_back(e);
        }
        return 0;
    }
};
static HostPipe &host_pipe()
{
    static HostPipe h;
    return h;
}

// host-pointer path shared by stages and chains: (pinned staging | registered caller memory) in, run, out.
// Round 4: a call of HOST_PIECE_MIN samples and more is cut into up to HOST_PIECES pieces that travel as a pipeline - H2D of piece k+1 (copy-in stream), the
// kernels of piece k (library stream), D2H of piece k-1 (copy-out stream), the CPU-side staging copies of an unregistered vector in between - so that both
// directions of the link and the host copies work at the same time instead of one after the other (a stand-alone LowpassFilter cf32 -> cf32: 1.9-2.4 GS/s as one piece, 3.3 / 4.3 / 5.2 GS/s at 2^20 / 2^22 / 2^24 samples per call on registered vectors).  Block state advances piece by piece exactly as it does chunk by chunk, so the values are those of any other chunking of
// the stream (bit for bit for direct-form blocks; overlap-save filters and the single-launch receiver to their stated Float32 rounding, include/lrhip.h).
constexpr unsigned long HOST_PIECE_MIN = 1ul << 19;      // same-box A/B (tools/ab_hostpath.py), 2^20-sample vectors: 2^17 2.6, 2^18 3.1, 2^19 3.3 GS/s registered
constexpr unsigned long HOST_PIECES = 8;
static unsigned long host_piece_min()
{
    static const unsigned long v = getenv("LRHIP_HOST_PIECE_MIN") ? strtoul(getenv("LRHIP_HOST_PIECE_MIN"), nullptr, 10) : HOST_PIECE_MIN;      // A/B knob
    return v < 4096 ? 4096 : v;
}
template <typename Runner>
static long host_execute(PinnedBuf &h_in, PinnedBuf &h_out, DeviceBuf &d_in, DeviceBuf &d_out, int in_size, int out_size,
                         unsigned long max_out, const void *in_host, unsigned long n_in, void *out_host,
                         unsigned long out_capacity, Runner run, unsigned long align = 1, bool direct_ok = false)
{
    if (n_in && !in_host) return set_error("null input buffer");
    size_t in_bytes = (size_t)n_in * in_size;
    unsigned long cap = max_out < out_capacity ? max_out : out_capacity;
    if (max_out > out_capacity) return set_error("output capacity %lu < required %lu", out_capacity, max_out);
    if (max_out && !out_host) return set_error("null output buffer");
    const bool in_reg = host_ranges().has(in_host, in_bytes), out_reg = host_ranges().has(out_host, (size_t)cap * out_size);
    // Round 5, DIRECT mode: both vectors registered and the stage / chain reads its input once and writes its output once (direct_io_ok) - the kernels load
    // the caller's vector and store into the caller's vector across the link, no device staging buffers, no copy engines, no pieces: ONE launch and one
    // synchronisation per call.  A stand-alone LowpassFilter cf32 -> cf32: 3.3 -> 4.4 GS/s at 2^20-sample vectors, 3.7 -> 5.5 at 2^22, 6.0 GS/s = 48 GB/s in
    // each direction at 2^24 (the link's measured both-way rate, tools/mb_link.py); kernels that know of it run a short persistent grid so that reads and
    // writes overlap (host_io_grid, common.h).  LRHIP_HOST_DIRECT=0 keeps the staged pipeline (A/B).
    static const bool direct_env = !getenv("LRHIP_HOST_DIRECT") || atoi(getenv("LRHIP_HOST_DIRECT")) != 0;
    // (an in-place call - out_host == in_host or overlapping slices of one registered buffer - was safe on the staged path and stays on it: tiles of a
    // persistent grid would store outputs over inputs other tiles have not loaded yet)
    const char *ia = (const char *)in_host, *oa = (const char *)out_host;
    const bool disjoint = !cap || ia + in_bytes <= oa || oa + (size_t)cap * out_size <= ia;
    if (direct_env && direct_ok && in_reg && out_reg && n_in && disjoint) {
        void *din = host_ranges().device_ptr(in_host, in_bytes), *dout = cap ? host_ranges().device_ptr(out_host, (size_t)cap * out_size) : (void *)nullptr;
        if (din && (dout || !cap)) {
            host_io_grid_ref() = 32;
            const long n_out = run(din, n_in, dout, cap);
            host_io_grid_ref() = 0;
            if (n_out < 0) return n_out;
            LR_HIP(hipStreamSynchronize(ctx().stream));
            return n_out;
        }
    }
    if ((!in_reg && h_in.reserve(in_bytes ? in_bytes : 16)) || d_in.reserve(in_bytes ? in_bytes : 16)) return -1;
    if ((!out_reg && h_out.reserve((size_t)cap * out_size + 16)) || d_out.reserve((size_t)cap * out_size + 16 * (size_t)HOST_PIECES * out_size + 16)) return -1;
    static const bool no_pieces = getenv("LRHIP_HOST_NO_PIECES") != nullptr;      // A/B knob: one piece, as in round 3
    unsigned long pieces = (no_pieces || n_in < 2 * host_piece_min()) ? 1 : n_in / host_piece_min();
    if (pieces > HOST_PIECES) pieces = HOST_PIECES;
    if (!disjoint) pieces = 1;      // overlapping vectors: the whole input is on the device before the first output byte comes back
    // ONE set of copy streams and events per process (HostPipe): a second host thread - another stage or chain; ctypes and LuaJIT release their lock around the
    // call - must not record and wait on them while this call's pieces are in flight, or its kernels could run before their own H2D has landed.  The thread
    // that does not get the lock takes the single-piece path, which touches only its own object's buffers and the library stream.
    HostPipe &hp = host_pipe();
    std::unique_lock<std::mutex> pipe_lock(hp.m, std::defer_lock);
    if (pieces > 1 && !pipe_lock.try_lock()) pieces = 1;
    if (pieces <= 1) {
        const void *src = in_reg ? in_host : h_in.p;
        if (in_bytes) {
            if (!in_reg) host_copy(h_in.p, in_host, in_bytes);
            LR_HIP(hipMemcpyAsync(d_in.p, src, in_bytes, hipMemcpyHostToDevice, ctx().stream));
        }
        long n_out = run(d_in.p, n_in, d_out.p, cap);
        if (n_out < 0) return n_out;
        if (n_out) LR_HIP(hipMemcpyAsync(out_reg ? out_host : h_out.p, d_out.p, (size_t)n_out * out_size, hipMemcpyDeviceToHost, ctx().stream));
        LR_HIP(hipStreamSynchronize(ctx().stream));
        if (n_out && !out_reg) host_copy(out_host, h_out.p, (size_t)n_out * out_size);
        return n_out;
    }
    if (hp.ensure(pieces)) return -1;
    // samples per piece: a multiple of the stage's / chain's own grid (lrhip_chain_shard_align: cuts on it reproduce the uncut run bit for bit where the
    // chain promises that at all - 128 000 for the FM receivers), else of 4096 (keeps the rows of the streaming kernels aligned); the last piece takes the rest
    const unsigned long grid = align > 1 ? align : 4096;
    unsigned long per = ((n_in / pieces) + grid - 1) / grid * grid;
    while (pieces > 1 && per * (pieces - 1) >= n_in) pieces--;
    std::vector<unsigned long> off_out(pieces + 1, 0), cnt_out(pieces, 0);
    unsigned long done_in = 0, total_out = 0, copied = 0;
    for (unsigned long k = 0; k < pieces; k++) {
        const unsigned long n_k = k + 1 == pieces ? n_in - done_in : per;
        const size_t ib = (size_t)done_in * in_size, nb = (size_t)n_k * in_size;
        if (!in_reg) host_copy((char *)h_in.p + ib, (const char *)in_host + ib, nb);
        LR_HIP(hipMemcpyAsync((char *)d_in.p + ib, (in_reg ? (const char *)in_host : (const char *)h_in.p) + ib, nb, hipMemcpyHostToDevice, hp.s_in));
        LR_HIP(hipEventRecord(hp.ev[3 * k], hp.s_in));
        LR_HIP(hipStreamWaitEvent(ctx().stream, hp.ev[3 * k], 0));
        // (capacity: what is left of the whole call's bound plus the slack reserved above - a piece's own bound may round up where the whole call's does not)
        const long m = run((const char *)d_in.p + ib, n_k, (char *)d_out.p + (size_t)total_out * out_size, cap - total_out + 16 * HOST_PIECES);
        if (m < 0) {
            (void)hipStreamSynchronize(hp.s_in); (void)hipStreamSynchronize(ctx().stream); (void)hipStreamSynchronize(hp.s_out);
            return m;
        }
        LR_HIP(hipEventRecord(hp.ev[3 * k + 1], ctx().stream));
        LR_HIP(hipStreamWaitEvent(hp.s_out, hp.ev[3 * k + 1], 0));
        if (m && total_out + (unsigned long)m <= cap) LR_HIP(hipMemcpyAsync((out_reg ? (char *)out_host : (char *)h_out.p) + (size_t)total_out * out_size, (char *)d_out.p + (size_t)total_out * out_size,
                                     (size_t)m * out_size, hipMemcpyDeviceToHost, hp.s_out));
        LR_HIP(hipEventRecord(hp.ev[3 * k + 2], hp.s_out));
        if (total_out + (unsigned long)m > cap) {
            (void)hipStreamSynchronize(hp.s_in); (void)hipStreamSynchronize(ctx().stream); (void)hipStreamSynchronize(hp.s_out);
            return set_error("host path: pieces produced %lu samples, the call's bound is %lu", total_out + (unsigned long)m, cap);
        }
        off_out[k] = total_out; cnt_out[k] = (unsigned long)m;
        total_out += (unsigned long)m;
        done_in += n_k;
        // unregistered output: hand over the pieces that have arrived while the later ones are still on their way
        if (!out_reg)
            while (copied < k && hipEventQuery(hp.ev[3 * copied + 2]) == hipSuccess) {
                if (cnt_out[copied]) host_copy((char *)out_host + (size_t)off_out[copied] * out_size, (char *)h_out.p + (size_t)off_out[copied] * out_size, (size_t)cnt_out[copied] * out_size);
                copied++;
            }
    }
    if (!out_reg) {
        for (; copied < pieces; copied++) {
            LR_HIP(hipEventSynchronize(hp.ev[3 * copied + 2]));
            if (cnt_out[copied]) host_copy((char *)out_host + (size_t)off_out[copied] * out_size, (char *)h_out.p + (size_t)off_out[copied] * out_size, (size_t)cnt_out[copied] * out_size);
        }
    } else {
        LR_HIP(hipStreamSynchronize(hp.s_out));
    }
    return (long)total_out;
}
