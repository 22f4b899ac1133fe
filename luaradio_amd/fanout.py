"""Graph fan-out across the GPUs of one node (BASELINE.json configs[3]).

The reference's fan-out is one OutputPort writing the same vector to several pipes, each reader an independent
process (radio/core/block.lua:119-166, radio/core/pipe.lua:617-627).  The MI355X form of that one pattern:
the source GPU broadcasts each IQ slab to the other GPUs (RCCL over xGMI - `torch.distributed` backend "nccl"),
and every rank runs its own branch chain (e.g. TunerBlock(offset_k, bw, D)) on its own GPU.  There is no other
exchange: outputs stay per-GPU (each branch has its own sink).  One process per GPU.

Nothing here computes on the CPU: a branch executor is a device block/chain.  The distributed plumbing
(branch -> rank plan, slab broadcast, max-over-ranks timing) is backend-agnostic so it can be exercised with
gloo on CPU tensors in the tests, where the branch executor is injected by the test.
"""
import time


def branch_offsets(num_branches, step=100e3):
    """Centre offsets of the fan-out branches: symmetric comb, -350 kHz .. +350 kHz for 8 branches."""
    return [(-(num_branches - 1) / 2.0 + b) * step for b in range(num_branches)]


def plan(num_branches, world_size):
    """branch -> rank, round-robin (branch b on rank b when num_branches == world_size)."""
    if num_branches < 1 or world_size < 1:
        raise ValueError("need at least one branch and one rank")
    return [b % world_size for b in range(num_branches)]


def local_branches(num_branches, world_size, rank):
    return [b for b, r in enumerate(plan(num_branches, world_size)) if r == rank]


class DeviceBranch:
    """A device-resident branch: wraps an initialized luaradio_amd block/composite and a preallocated output."""

    def __init__(self, block, max_input_samples):
        import torch
        self.block = block
        self.cap = block.max_output(max_input_samples)
        self.out_floats = 2 if block.get_output_type().size == 8 else 1
        self.out = torch.empty(self.cap * self.out_floats + 16, dtype=torch.float32, device="cuda")
        self.produced = 0
        # a chain that starts with IQFileSource's format stage takes the raw file records: the slab then carries bytes (2 per sample for u8 / s8
        # instead of 8 - what crosses the xGMI link to every receiving GPU is a quarter, the link bound 76 GS/s instead of 19)
        self.in_record = int(getattr(block, "in_record", 0) or 0)

    def process(self, slab):
        """slab: 1-D float32 CUDA tensor of interleaved ComplexFloat32 samples (or, for a chain with a file-format head, any 1-D CUDA tensor
        holding whole raw records).  The kernels are enqueued on torch's CURRENT stream
        (lrhip_set_stream orders it behind whatever the library had queued before), so they follow the broadcast that filled the slab
        and precede any torch consumer of the returned view."""
        import torch
        from . import _lib
        if not slab.is_cuda:
            raise RuntimeError("DeviceBranch needs a CUDA tensor: there is no CPU path in luaradio_amd")
        _lib.adopt_torch_stream()
        n = slab.numel() * slab.element_size() // self.in_record if self.in_record else slab.numel() // 2
        got = self.block.process_device(slab.data_ptr(), n, self.out.data_ptr(), self.cap)
        self.produced += got
        return self.out[:got * self.out_floats]


class FanOut:
    """One source rank, `len(branches_by_index)` branches spread over the ranks of a process group.

    dist      torch.distributed (initialised) or None for a single process
    branches  {branch_index: executor} for the branches this rank owns (see local_branches()); an executor has
              .process(slab_tensor)
    """

    def __init__(self, dist, rank, world_size, num_branches, branches, src=0, device=None, always_broadcast=False):
        self.dist, self.rank, self.world, self.src = dist, rank, world_size, src
        self.always_broadcast = always_broadcast      # issue the collective even in a group of one rank (exercises RCCL + the communication stream on one GPU)
        self.device = device          # "cuda" / "cpu" for stream()'s slab buffers; None = cuda with RCCL or a single GPU process, cpu otherwise
        self.num_branches = num_branches
        mine = local_branches(num_branches, world_size, rank)
        if sorted(branches) != mine:
            raise ValueError("rank %d must own branches %s, got %s" % (rank, mine, sorted(branches)))
        self.branches = branches
        self.slabs = 0

    def push(self, slab):
        """Broadcast one slab from the source rank (in place into `slab` on the others) and run the local branches.
        Returns {branch_index: output}."""
        if self.dist is not None and (self.world > 1 or self.always_broadcast):
            self.dist.broadcast(slab, src=self.src)
        self.slabs += 1
        return {b: ex.process(slab) for b, ex in self.branches.items()}

    def stream(self, slabs, slab_floats):
        """Double-buffered fan-out: `slabs` yields the source's slabs (1-D float32 tensors of `slab_floats` values on the source rank;
        on the other ranks only the count matters - pass any iterable of the same length, e.g. range(k)).  The broadcast of slab k+1
        runs on its own communication stream while the branches work on slab k: two receive buffers, an event per buffer and direction
        (filled -> the compute stream may read; consumed -> the next broadcast may overwrite).  Yields {branch_index: output} per slab;
        an output view is valid until the next slab is requested.  With CPU tensors (gloo, tests) the same bookkeeping runs with
        asynchronous broadcast handles instead of streams."""
        import torch
        multi = self.dist is not None and (self.world > 1 or self.always_broadcast)
        if self.device is not None:
            cuda = self.device == "cuda"
        else:
            cuda = multi and self.dist.get_backend() == "nccl" or (not multi and torch.cuda.is_available())
        dev = "cuda" if cuda else "cpu"
        bufs = [torch.empty(slab_floats, dtype=torch.float32, device=dev) for _ in range(2)]
        comm = torch.cuda.Stream() if cuda else None
        filled = [torch.cuda.Event() if cuda else None for _ in range(2)]
        consumed = [torch.cuda.Event() if cuda else None for _ in range(2)]
        used = [False, False]
        handles = [None, None]
        it = iter(slabs)

        def launch(k, item):
            """copy / broadcast slab `item` into buffer k on the communication stream"""
            b = bufs[k]
            if cuda:
                if self.rank == self.src:
                    comm.wait_stream(torch.cuda.current_stream())      # the slab may still be being written on the compute stream
                with torch.cuda.stream(comm):
                    if used[k]:
                        comm.wait_event(consumed[k])
                    if self.rank == self.src:
                        b.copy_(item, non_blocking=True)
                    if multi:
                        self.dist.broadcast(b, src=self.src)
                    filled[k].record(comm)
            else:
                if self.rank == self.src:
                    b.copy_(item)
                handles[k] = self.dist.broadcast(b, src=self.src, async_op=True) if multi else None
            used[k] = True

        nxt = next(it, None)
        k = 0
        if nxt is not None:
            launch(0, nxt)
        while nxt is not None:
            cur_k = k
            nxt = next(it, None)
            if nxt is not None:
                launch(cur_k ^ 1, nxt)                  # slab k+1 travels while slab k is processed
            if cuda:
                torch.cuda.current_stream().wait_event(filled[cur_k])
            elif handles[cur_k] is not None:
                handles[cur_k].wait()
            self.slabs += 1
            out = {b: ex.process(bufs[cur_k]) for b, ex in self.branches.items()}
            if cuda:
                consumed[cur_k].record(torch.cuda.current_stream())
            yield out
            k ^= 1

    def timed(self, fn, sync):
        """max-over-ranks wall time of fn(), bracketed by sync() (barrier + device synchronize) on both sides"""
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        dt = time.perf_counter() - t0
        if self.dist is not None and self.world > 1:
            import torch
            t = torch.tensor([dt], dtype=torch.float64)
            if self.dist.get_backend() == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt
