"""Data-parallel gradient exchange (net-new: the reference's --GPUs_id tower loop never averaged gradients,
SURVEY.md fact 5).  One process per GPU; the flat fp32 gradient buffer is all-reduced (RCCL over xGMI when the
process group backend is "nccl", gloo in the CPU tests) in a few contiguous buckets ordered as backward produces
them (post-net + linear first, embeddings last) so that the exchange of a bucket can be enqueued on a side
stream as soon as its last gradient has landed.  Replicas keep per-replica BatchNorm statistics."""
import torch
import torch.distributed as dist


def bucket_ranges(layout, n_buckets=4):
    """Contiguous [begin, end) ranges of the flat buffer, returned in BACKWARD order, cut at block boundaries:
    [post_cbhg + linear] [decoder] [attention + encoder_cbhg] [encoder prenet + embeddings]."""
    e = layout.entries
    cuts = [0, e['encoder_cbhg/conv_bank/kernel'].offset, e['decoder_prenet/dense_1/kernel'].offset,
            e['post_cbhg/conv_bank/kernel'].offset, layout.total]
    if n_buckets < 4:
        cuts = [0, layout.total] if n_buckets <= 1 else [0, e['post_cbhg/conv_bank/kernel'].offset, layout.total]
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 2, -1, -1)]


def allreduce_average(flat, world, buckets=None, scale_fn=None, async_op=False):
    """sum-all-reduce every bucket, then divide by `world` (scale_fn: in-place device scaling kernel)."""
    if world <= 1:
        return []
    works = []
    for b0, b1 in (buckets or [(0, flat.numel())]):
        works.append(dist.all_reduce(flat[b0:b1], async_op=True))
    if async_op:
        return works
    for w in works:
        w.wait()
    if scale_fn is not None:
        scale_fn(flat, 1.0 / world)
    else:
        flat.mul_(1.0 / world)
    return []
