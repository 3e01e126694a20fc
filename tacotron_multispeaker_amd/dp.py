"""Data-parallel gradient exchange (net-new: the reference's --GPUs_id tower loop never averaged gradients,
SURVEY.md fact 5).  One process per GPU; the flat fp32 gradient buffer is all-reduced (RCCL over xGMI when the
process group backend is "nccl", gloo in the CPU tests) in contiguous buckets cut at block boundaries and ordered as
backward completes them (post-net + linear first, embeddings last).

Overlap with backward (BASELINE.json north_star): BucketExchange.launch(i) is called by the engine the
moment the last producer of bucket i has been ENQUEUED (weight-gradient GEMMs run on side streams, see Engine.flush_side),
under a stream that waits for exactly those producers.  torch.distributed then orders its communication stream behind that
stream and returns an async work handle, so bucket i travels over xGMI while the BPTT kernels of the earlier layers still
run; finish() makes the optimizer's stream wait for all handles.  Only the last bucket (encoder prenet + embeddings, whose
gradients appear at the very end of backward) is exposed.

The average (1/world) is not a pass of its own: the optimizer kernels take the factor (Engine.optimizer_step), and the dense
global norm is computed on the summed gradient and scaled, so every rank applies the bit-identical update.
Replicas keep per-replica BatchNorm statistics (independent towers, reference train.py:101-111)."""
import torch
import torch.distributed as dist


def bucket_ranges(layout, n_buckets=4):
    """Contiguous [begin, end) ranges of the flat buffer (creation order: embeddings | encoder prenet | encoder CBHG [bank,
    proj_1, proj_2, highways, biGRU] | attention | decoder | post-net | linear), returned in the order backward COMPLETES them:
      0 [post_cbhg .. end)                 post-net + linear            ready while the decoder BPTT runs
      1 [attention .. post_cbhg)           attention + decoder          ready when the encoder backward starts
      2 [encoder proj_2 .. attention)      encoder proj_2, highways, biGRU   ready early in the encoder backward
      3 [0 .. encoder proj_2)              embeddings, encoder prenet, conv bank, proj_1   the tail of backward (exposed)
    n_buckets 2: [post_cbhg .. end) [0 .. post_cbhg); 1: one message."""
    e = layout.entries
    cuts = [0, e['encoder_cbhg/proj_2/kernel'].offset, e['attention/memory_layer/kernel'].offset,
            e['post_cbhg/conv_bank/kernel'].offset, layout.total]
    if n_buckets < 4:
        cuts = [0, layout.total] if n_buckets <= 1 else [0, e['post_cbhg/conv_bank/kernel'].offset, layout.total]
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 2, -1, -1)]


class BucketExchange:
    """Out-of-band all-reduce(sum) of the buckets of one flat gradient buffer.

        ex = BucketExchange(flat, ranges, world)
        ex.begin()                   # start of a backward pass
        ex.launch(0) ... ex.launch(k)   # each as soon as its bucket's producers are enqueued; SAME order on every rank
        ex.finish()                  # remaining buckets are launched, then the current stream / the host waits for all

    flat holds SUMS afterwards; the caller applies 1/world (Engine: inside the optimizer kernels)."""

    def __init__(self, flat, ranges, world, group=None):
        self.flat, self.ranges, self.world, self.group = flat, list(ranges), int(world), group
        self.works = {}
        self.order = []              # launch order of the last pass (tests / diagnostics)

    def begin(self):
        self.works = {}
        self.order = []

    def launch(self, i):
        if self.world <= 1 or i in self.works:
            return
        b0, b1 = self.ranges[i]
        # async_op: the collective is ordered behind the CURRENT stream's work (c10d records an event on it), runs on the
        # backend's own communication stream / thread and does not block the host
        self.works[i] = dist.all_reduce(self.flat[b0:b1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.order.append(i)

    def finish(self):
        if self.world <= 1:
            return
        for i in range(len(self.ranges)):
            self.launch(i)
        for i in self.order:
            self.works[i].wait()     # device tensors: the current stream waits (no host block); CPU tensors: the host waits


def allreduce_average(flat, world, buckets=None, scale_fn=None):
    """Blocking form: sum-all-reduce every bucket, then divide by `world` (scale_fn: in-place device scaling kernel)."""
    if world <= 1:
        return
    ex = BucketExchange(flat, buckets or [(0, flat.numel())], world)
    ex.begin()
    ex.finish()
    if scale_fn is not None:
        scale_fn(flat, 1.0 / world)
    else:
        flat.mul_(1.0 / world)
