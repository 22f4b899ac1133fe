#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X DSP block engine.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.

Workloads (BASELINE.json configs):
  fir     (default; configs[1], the configuration the metric is quoted on) 128-tap real-taps FIR
          (LowpassFilterBlock(128, 15e3) at 220.5 kHz) on 2^28 synthetic ComplexFloat32 samples resident in
          HBM.  One step = one pass of FIRFilterBlock:process over the whole 2 GiB vector (one
          fir_mfma_kernel launch + the 1-block history-carry kernel).
  wbfm    (configs[2]) the examples/rtlsdr_wbfm_mono.lua chain, device-resident, on 2^26 RF samples.
  fanout  (configs[3]) one IQ slab broadcast from rank 0 to all ranks over RCCL, one Tuner branch per GPU.

N > 1: one process per GPU (torchrun), each rank filters its own independent stream of the same size
("scaling": "weak", no data-path collective for fir/wbfm); value = samples processed by all ranks / max time.

value = MSamples/s with inputs already in HBM.  roofline.achieved = algorithmic bytes (16 B/sample, SURVEY.md
section 8d) per launch / average launch duration measured with HIP events on the launch stream inside the timed region.
cpu_baseline = the CPU restatement (oracle, VOLK-style SIMD dot products) timed on this box's host cores on a
bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32 vector == f32 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="fir", choices=["fir", "wbfm", "fanout"])
    ap.add_argument("--log2-samples", type=int, default=None, help="per-GPU samples per step (default 28 fir, 26 wbfm/fanout)")
    ap.add_argument("--fir-mode", default="auto", choices=["auto", "direct", "fft"],
                    help="fir workload arithmetic: direct = Toeplitz MFMA (bit-exact fmaf chain); fft = fused overlap-save kernel; "
                         "auto = fft (the faster one; both are parity-tested)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for dry runs of the multi-rank plumbing)")
    ap.add_argument("--same-device", action="store_true", help="dry run: every rank uses cuda:0 (needs --dist-backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (seconds of CPU work)")
    return ap.parse_args()


def cpu_baseline_fir(taps, budget_s):
    """Oracle FIR (VOLK-style SIMD dot products) on a bounded sample of the same workload: 2^22-sample slabs of
    the same U(-1,1) IQ, repeated until ~budget_s of wall time; all host cores via OpenMP, and single core."""
    import numpy as np
    from oracle import oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    n = 1 << 22
    rng = np.random.default_rng(2)
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    y = np.empty(n, np.complex64)

    def rate(nt, seconds):
        f = O.FIR(taps, True)
        f.process_simd(x[:1 << 16], nt, out=y[:1 << 16])     # warm-up
        done, t0 = 0, time.perf_counter()
        while True:
            f.process_simd(x, nt, out=y)
            done += n
            dt = time.perf_counter() - t0
            if dt >= seconds:
                return done / dt / 1e6, done

    # pick the thread count that is actually fastest on this box (cgroup quotas make "all logical CPUs" a bad guess)
    cands = sorted({c for c in (2, 4, 8, 16, 32, 64, 128, avail) if c <= avail})
    probe = {c: rate(c, 0.25)[0] for c in cands}
    cores = max(probe, key=probe.get)
    out = {"single": rate(1, budget_s * 0.35), "all": rate(cores, budget_s * 0.4)}
    return {"value": round(out["all"][0], 2), "unit": "MSamples/s", "cores": cores, "kind": "port",
            "single_core_value": round(out["single"][0], 2),
            "sample": "oracle/lr_oracle.c lro_fir_process_simd (VOLK-style SIMD dot product per output, the form of "
                      "firfilter.lua:139-142), 128 real taps on %d x 2^22-sample slabs of the same U(-1,1) IQ, OpenMP %d threads (best of %s; %d logical CPUs available); "
                      "single_core_value = 1 thread (the reference gives a block one core)" % (out["all"][1] >> 22, cores, cands, avail)}


def wbfm_chain_report(lr, L, torch, dev, with_cpu):
    """second half of BASELINE.json's metric (configs[2]) inside the default bench line: the WBFM-mono receiver chain,
    device-resident, 2^26 synthetic FM IQ samples per step, HIP-event timed; with_cpu also runs the oracle chain (1 core)
    on the first 2^20 samples for the RMS error and the CPU rate."""
    import numpy as np
    fs, n = 1102500.0, 1 << 26
    t = torch.arange(n, dtype=torch.float64, device=dev) / fs
    m = 0.5 * torch.sin(2 * np.pi * 1e3 * t) + 0.5 * torch.sin(2 * np.pi * 5e3 * t)
    ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / fs * torch.cumsum(m, 0)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.stack([torch.cos(ph).float(), torch.sin(ph).float()], 1).reshape(-1)
    x += 0.01 * (torch.rand(2 * n, dtype=torch.float32, device=dev, generator=g) * 2 - 1)
    del t, m, ph
    rx = lr.wbfm_mono_receiver(fs, -250e3)
    cap = rx.max_output(n)
    y = torch.empty(cap + 16, dtype=torch.float32, device=dev)
    steps = 10
    for _ in range(2):
        rx.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    tm = L.lrhip_timer_create()
    L.lrhip_timer_start(tm)
    for _ in range(steps):
        rx.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    L.lrhip_timer_stop(tm)
    torch.cuda.synchronize()
    ms = L.lrhip_timer_elapsed_ms(tm) / steps
    L.lrhip_timer_destroy(tm)
    rep = {"workload": "configs[2]: Tuner(-250e3, 200e3, 5) -> FrequencyDiscriminator(1.25) -> Lowpass(128, 15e3) -> FMDeemphasis(75e-6) -> "
                       "Downsampler(5), 2^26 RF samples per step, device-resident",
           "value": round(n / ms / 1e3, 1), "unit": "MSamples/s (RF samples in)", "ms_per_step": round(ms, 4), "launches": rx.chain.last_launches,
           "algorithmic_GB/s": round(8.16 * n / ms / 1e6, 1)}
    if with_cpu:
        from oracle import oracle as O
        k = 1 << 20
        rx2 = lr.wbfm_mono_receiver(fs, -250e3)
        xs = x[:2 * k].cpu().numpy().view(np.complex64)
        got = rx2.process(xs)
        ch = O.wbfm_mono_chain(fs, -250e3, mode=O.MODE_LUA, rot_mode=O.MODE_F64)
        t0 = time.perf_counter()
        want = ch.process(xs)
        dt = time.perf_counter() - t0
        err = got.astype(np.float64) - want.astype(np.float64)
        rep["rms_err_vs_oracle"] = float(np.sqrt(np.mean(err ** 2)))
        rep["cpu_baseline"] = {"value": round(k / dt / 1e6, 2), "unit": "MSamples/s", "cores": 1, "kind": "port",
                               "sample": "oracle chain (per-block restatement of the reference's Lua/VOLK arithmetic) on the first 2^20 samples"}
    return rep


def main():
    args = parse()
    import numpy as np
    import torch
    import luaradio_amd as lr
    from luaradio_amd import types

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    lr.init(local_rank)
    L = lr._lib.load()
    # launch on torch's current stream so torch.cuda.synchronize()/barriers and the library's HIP events agree
    L.lrhip_set_stream(torch.cuda.current_stream().cuda_stream)

    log2n = args.log2_samples or (28 if args.workload == "fir" else 26)
    n = 1 << log2n
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev).manual_seed(2 + rank)

    extra = {}
    if args.workload == "fir":
        # synthetic IQ: re/im ~ U(-1,1), generated on the device in 2^24-sample slabs
        x = torch.empty(2 * n, dtype=torch.float32, device=dev)
        slab = 1 << 25
        for o in range(0, 2 * n, slab):
            x[o:o + slab] = torch.rand(min(slab, 2 * n - o), dtype=torch.float32, device=dev, generator=gen) * 2 - 1
        y = torch.empty(2 * n, dtype=torch.float32, device=dev)
        fir_mode = "fft" if args.fir_mode == "auto" else args.fir_mode
        blk = lr.LowpassFilterBlock(128, 15e3)
        blk.use_fft = 2 if fir_mode == "fft" else 0
        blk.rate = 220500.0
        blk.differentiate([types.ComplexFloat32])
        blk.initialize()
        taps = blk.taps.copy()

        def step():
            got = blk.process_device(x.data_ptr(), n, y.data_ptr(), n)
            assert got == n

        out_per_step = n
        alg_bytes = 16.0 * n
        flops = 4.0 * 128 * n if fir_mode == "direct" else 125.0 * n
        dominant = "fir_mfma_persistent_kernel<2,1,8,false,36>" if fir_mode == "direct" else "fir_fft_kernel<2,0>"
        config = {"workload": "configs[1]: 128-tap real-taps FIR (LowpassFilterBlock(128, 15e3) @ 220.5 kHz) on 2^%d synthetic "
                              "ComplexFloat32 IQ per GPU" % log2n,
                  "samples_per_step_per_gpu": n, "taps": 128, "kernel": dominant,
                  "algorithm": ("direct form, banded-Toeplitz product on the f32 matrix cores (bit-exact fmaf chain)" if fir_mode == "direct"
                                else "overlap-save, fused 1024-point FFT kernel (the reference's production form, firfilter.lua:320-398), <= 1e-6 of the f64 oracle"),
                  "parallelism": "independent streams x%d" % world}
    elif args.workload == "wbfm":
        fs = 1102500.0
        t = torch.arange(n, dtype=torch.float64, device=dev) / fs
        m = 0.5 * torch.sin(2 * np.pi * 1e3 * t) + 0.5 * torch.sin(2 * np.pi * 5e3 * t)
        ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / fs * torch.cumsum(m, 0)
        noise = torch.rand(2 * n, dtype=torch.float32, device=dev, generator=gen) * 2 - 1
        x = torch.stack([torch.cos(ph).float(), torch.sin(ph).float()], 1).reshape(-1) + 0.01 * noise
        del t, m, ph, noise
        rx = lr.wbfm_mono_receiver(fs, -250e3)
        cap = rx.max_output(n)
        y = torch.empty(cap + 16, dtype=torch.float32, device=dev)
        produced = []

        def step():
            produced.append(rx.process_device(x.data_ptr(), n, y.data_ptr(), cap))

        out_per_step = n      # metric counts RF samples in; audio samples out reported in config
        alg_bytes = 8.16 * n
        flops = 167.0 * n
        config = {"workload": "configs[2]: examples/rtlsdr_wbfm_mono.lua chain (Tuner -> FrequencyDiscriminator -> Lowpass -> "
                              "FMDeemphasis -> Downsampler) on 2^%d synthetic FM IQ samples @ 1.1025 MS/s, device-resident" % log2n,
                  "samples_per_step_per_gpu": n, "counted": "RF input samples", "parallelism": "independent streams x%d" % world}
        dominant = "fir_mfma_persistent_kernel<2,5,2,true,51>"
    else:
        # fan-out: rank 0 owns the IQ slab; every step it is broadcast over RCCL/xGMI and each rank runs its own
        # Tuner branch (offsets -350 kHz .. +350 kHz step 100 kHz)
        fs = 1102500.0
        x = torch.empty(2 * n, dtype=torch.float32, device=dev)
        if rank == 0:
            x.copy_(torch.rand(2 * n, dtype=torch.float32, device=dev, generator=gen) * 2 - 1)
        from luaradio_amd import fanout
        nbranches = world
        offs = fanout.branch_offsets(8 if world <= 8 else world)
        mine = {}
        for b in fanout.local_branches(nbranches, world, rank):
            tun = lr.TunerBlock(offs[b % len(offs)], 100e3, 5)
            tun.rate = fs
            tun.differentiate([types.ComplexFloat32])
            tun.initialize()
            mine[b] = fanout.DeviceBranch(tun, n)
        fo = fanout.FanOut(dist, rank, world, nbranches, mine, src=0)

        def step():
            fo.push(x)

        out_per_step = n
        alg_bytes = 9.6 * n
        flops = (4.0 * 128 / 5 + 6) * n
        config = {"workload": "configs[3]: one IQ slab of 2^%d samples broadcast from rank 0 (RCCL) to %d Tuner(offset_k, 100e3, 5) "
                              "branches, one per GPU" % (log2n, world),
                  "samples_per_step_per_gpu": n, "counted": "branch input samples", "parallelism": "fan-out x%d" % world}
        dominant = "fir_mfma_persistent_kernel<2,5,2,true,51>"

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    timer = L.lrhip_timer_create()
    t0 = time.perf_counter()
    L.lrhip_timer_start(timer)
    for _ in range(args.steps):
        step()
    L.lrhip_timer_stop(timer)
    sync_all()
    wall = time.perf_counter() - t0
    ev_ms = L.lrhip_timer_elapsed_ms(timer)
    L.lrhip_timer_destroy(timer)
    if dist is not None:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    # achievable-bandwidth yardstick on this box: the cheapest streaming kernel of the library (one multiply per scalar,
    # 8 B in + 8 B out per sample) over the same buffers, HIP-event timed like the workload
    yard_gbs = None
    if rank == 0 and args.workload == "fir":
        mc = lr.MultiplyConstantBlock(1.0)
        mc.differentiate([types.ComplexFloat32])
        mc.initialize()
        mc.process_device(x.data_ptr(), n, y.data_ptr(), n)
        yt = L.lrhip_timer_create()
        L.lrhip_timer_start(yt)
        for _ in range(5):
            mc.process_device(x.data_ptr(), n, y.data_ptr(), n)
        L.lrhip_timer_stop(yt)
        torch.cuda.synchronize()
        yard_gbs = 16.0 * n * 5 / (L.lrhip_timer_elapsed_ms(yt) / 1e3) / 1e9
        L.lrhip_timer_destroy(yt)

    if rank == 0:
        total = float(out_per_step) * world * args.steps
        launch_s = ev_ms / 1e3 / args.steps          # HIP-event time per step on the launch stream
        achieved = alg_bytes / launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get("%s:%d" % (args.workload if args.workload != "fir" else "fir-" + fir_mode, log2n))
                traffic = ent["bytes_per_launch"] if ent else None
            except Exception:
                traffic = None
        res = {
            "metric": "MSamples/s per block (128-tap complex FIR headline) + WBFM chain end-to-end",
            "value": round(total / wall / 1e6, 1), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": config,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": dominant,
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms_hip_events": round(launch_s * 1e3, 4),
                         "fp32_tflops": round(flops / launch_s / 1e12, 2), "fp32_peak_tflops": FP32_PEAK_TFLOPS,
                         "fp32_frac": round(flops / launch_s / 1e12 / FP32_PEAK_TFLOPS, 4)},
        }
        if yard_gbs:
            res["roofline"]["streaming_yardstick"] = {"kernel": "multiply_constant_kernel<1> (8 B in + 8 B out per sample, same buffers)",
                                                      "GB/s": round(yard_gbs, 1), "frac_of_yardstick": round(achieved / yard_gbs, 4)}
        if world == 1 and not args.no_cpu_baseline and args.workload == "fir":
            res["cpu_baseline"] = cpu_baseline_fir(taps, args.cpu_seconds)
        elif world == 1:
            res["cpu_baseline"] = None
        if world == 1 and args.workload == "fir" and log2n >= 26:
            del x, y
            torch.cuda.empty_cache()
            res["wbfm_chain"] = wbfm_chain_report(lr, L, torch, dev, not args.no_cpu_baseline)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
