"""Write-only ceiling on one MI355X: torch's fill kernel and the copy engine's memset on the Interpolator(5) output size (5 x 2^26 ComplexFloat32 = 2.68 GB),
and a 1 : 5 read/write mix (torch repeat_interleave-free: a strided copy) - what HBM takes when the traffic is mostly stores.  Yardstick for the write-heavy
rows of tools/bench_blocks.py (Upsampler / Interpolator)."""
import torch


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = 5 << 26
    y = torch.empty(2 * n, device="cuda")
    x = torch.rand(2 << 26, device="cuda")
    nbytes = y.numel() * 4
    ms = timed(lambda: y.fill_(1.5))
    print("torch fill_ %.2f GB: %.4f ms = %.0f GB/s" % (nbytes / 1e9, ms, nbytes / ms / 1e6))
    ms = timed(lambda: y.zero_())
    print("torch zero_ (memset) %.2f GB: %.4f ms = %.0f GB/s" % (nbytes / 1e9, ms, nbytes / ms / 1e6))
    # 1 : 5 mix: every input sample (8 B) is read once and written to five places
    yv = y.view(-1, 5, 2)
    xv = x.view(-1, 1, 2)
    ms = timed(lambda: yv.copy_(xv.expand(-1, 5, 2)))
    tot = nbytes + x.numel() * 4
    print("torch broadcast copy 8 B in : 40 B out, %.2f GB: %.4f ms = %.0f GB/s" % (tot / 1e9, ms, tot / ms / 1e6))
    z = torch.empty_like(x)
    ms = timed(lambda: z.copy_(x))
    print("torch copy 1 : 1, %.2f GB: %.4f ms = %.0f GB/s" % (2 * x.numel() * 4 / 1e9, ms, 2 * x.numel() * 4 / ms / 1e6))


if __name__ == "__main__":
    main()
