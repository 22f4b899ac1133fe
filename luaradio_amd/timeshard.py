"""Time-axis sharding of ONE stream across the GPUs of a node (second mode; include/lrhip.h "time-axis sharding").

A LuaRadio block is one process working through the whole stream (radio/core/block.lua:572-590).  Everything the hot-path blocks carry
from chunk to chunk is either a closed form of the absolute sample index (rotator phase, frequencytranslator.lua:93-110; decimation
phase, downsampler.lua:45-56) or a bounded memory of the input (filter history, firfilter.lua:244-250; the discriminator's previous
sample; a decaying recurrence), so a long recording can be cut into G contiguous partitions, one per GPU, with NO exchange between
them: partition g seeks its chain to (a_g - H), replays the H = chain.halo() samples in front of its first own sample with the output
thrown away, and then produces exactly the samples [a_g, b_g) of the single-stream run.

    bounds(n, G, align)            -> [(a_0, b_0), ...]: G contiguous partitions of n samples, boundaries on multiples of `align`
    run_partition(chain, x, a, b)  -> the chain's output for input samples [a, b) of the stream x (host arrays; the device-pointer form
                                      is run_partition_device)
    rank_partition(n, world, rank) -> (a, b) for one process per GPU (torchrun): no collective on the data path
Boundaries on multiples of chain.shard_align() input samples make the result bit-identical to the single-stream run as well (the
tile grids of the kernels then coincide: 128 000 for the WBFM receiver - 64 000 for its tail's scan, 5 120 for its tuner, whose tiles
rotate relative to their first sample in front of the discriminator - and 1 for plain filter / rotator / discriminator / downsampler
chains); any other boundary gives the same values to Float32 rounding of those two and exactly for everything else.
"""

ALIGN = 65536      # default cut granularity when the caller does not ask the chain (any multiple of chain.shard_align() is bit-exact)


def bounds(n, parts, align=ALIGN):
    """`parts` contiguous partitions of [0, n) with interior boundaries on multiples of `align` (as even as that allows; trailing
    partitions may be empty when n is small)."""
    if parts < 1 or align < 1 or n < 0:
        raise ValueError("need parts >= 1, align >= 1, n >= 0")
    units = (n + align - 1) // align
    cuts = [min(n, ((units * g) // parts) * align) for g in range(parts + 1)]
    cuts[-1] = n
    return [(cuts[g], cuts[g + 1]) for g in range(parts)]


def rank_partition(n, world, rank, align=ALIGN):
    return bounds(n, world, align)[rank]


def replay_start(a, halo, multiple=1):
    """first sample a partition starting at `a` has to read: max(0, a - halo), moved down to a multiple of `multiple` (the product of
    the chain's decimations keeps every stage's absolute index integral there - any value is correct, this one is tidy)"""
    s = max(0, a - halo)
    return s - s % multiple


def run_partition(chain, x, a, b, halo=None, align=None):
    """output of `chain` (luaradio_amd Chain / CompositeBlock) for samples [a, b) of the host array x, as part of the stream x[0:].
    The replay starts on a multiple of chain.shard_align() too, so that the state it leaves behind was computed on the tile grid of the
    uninterrupted run (at most one alignment unit of extra replay)."""
    h = chain.halo() if halo is None else halo
    s = replay_start(a, h, chain.shard_align() if align is None else align)
    chain.seek(s)
    if a > s:
        chain.process(x[s:a])          # replay: output discarded
    return chain.process(x[a:b])


def run_partition_device(chain, x_ptr, in_size, a, b, out_ptr, out_capacity, scratch_ptr, scratch_capacity, halo=None, align=None):
    """device-pointer form (asynchronous on the library stream): x_ptr addresses sample 0 of the stream, in_size its bytes per sample;
    the replayed outputs go to scratch_ptr (capacity >= chain.max_output(halo + align)).  Returns the output samples written to out_ptr."""
    h = chain.halo() if halo is None else halo
    s = replay_start(a, h, chain.shard_align() if align is None else align)
    chain.seek(s)
    if a > s:
        chain.process_device(x_ptr + s * in_size, a - s, scratch_ptr, scratch_capacity)
    return chain.process_device(x_ptr + a * in_size, b - a, out_ptr, out_capacity)
