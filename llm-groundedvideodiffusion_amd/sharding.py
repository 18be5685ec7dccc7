"""Multi-GPU layout of the sampler: independent (prompt, seed) videos, one process per GPU, no data-path collective.

Mirrors the manual sharding of the reference (`--skip_first_prompts/--num_prompts`, generate.py:255-262) with its seed rule
`seed = prompt_index + repeat_index * 6789 + seed_offset` (generate.py:325-335), so any sharding reproduces the
single-process run.  No collective sits on the data path: every rank writes its own videos.  torch.distributed ("nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests) is used for (i) the end-of-run tally in generate.py (all_reduce of the per-rank video
count) and (ii) `gather_frames`, the all_gather of decoded uint8 frames (13.3 MB per 24x320x576x3 video) that `bench.py --gpus N`
runs once, untimed, after the timed region, and that a host harness wanting all videos on rank 0 calls."""
import torch
import torch.distributed as dist


def owns(ind, rank=0, world=1):
    """The shard rule, in one place: global prompt index `ind` belongs to rank ind mod world (generate.py, scripts/*, bench.py)."""
    return ind % world == rank


def shard_jobs(num_prompts, repeats, seed_offset=0, rank=0, world=1, skip_first_prompts=0):
    """[(global_prompt_index, repeat_index, seed)] owned by `rank` (round-robin over the global prompt index)."""
    jobs = []
    for ind in range(skip_first_prompts, skip_first_prompts + num_prompts):
        if not owns(ind, rank, world):
            continue
        for rep in range(repeats):
            jobs.append((ind, rep, ind + rep * 6789 + seed_offset))
    return jobs


def gather_frames(frames):
    """all_gather of one tensor per rank (same shape); returns the list ordered by rank.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [frames]
    out = [torch.empty_like(frames) for _ in range(dist.get_world_size())]
    dist.all_gather(out, frames.contiguous())
    return out
