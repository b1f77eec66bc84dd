"""Data-parallel host logic (the reference itself is single-GPU).

One process per GPU. Rank r takes the contiguous window slice [r*B/G, (r+1)*B/G) of the global batch
together with its (k+1) document ids; the multipliers carry 1/B_global, batch-norm statistics are summed
across ranks (sync_batch_norm), and ONE all-reduce of the dense projection gradient (d_e x d_w floats;
the bias gradient rides with the statistics) is issued per step. Embedding tables are updated rank-locally
from the rank's own shard ("sparse rows stay GPU-local"): replicas of the tables drift apart — a documented
deviation from the single-GPU trajectory; the dense gradients and the loss of a step are exact.

Transports: RCCL over xGMI (``init_rccl``; the production path) or any callable that sums a float64 numpy
buffer in place across ranks (``torch_allreduce``; used with gloo in the tests).
"""
import numpy as np


def shard_bounds(num_instances, rank, world_size):
    if num_instances % world_size != 0:
        raise ValueError("global batch (%d) must divide evenly over %d ranks" % (num_instances, world_size))
    per = num_instances // world_size
    return rank * per, (rank + 1) * per


def shard_batch(features, labels, feature_weights, weights, entity_ids, window_size, num_random_entities, rank, world_size):
    """Contiguous slice of a global batch (+ its entity ids laid out [label, neg_1..neg_k] per instance)."""
    lo, hi = shard_bounds(len(labels), rank, world_size)
    R = num_random_entities + 1
    sl = lambda a, m: None if a is None else a[lo * m:hi * m]
    return (sl(features, window_size), sl(labels, 1), sl(feature_weights, window_size), sl(weights, 1),
            sl(entity_ids, R))


def torch_allreduce(dist, group=None):
    """all-reduce callback over torch.distributed (gloo or nccl) for Model.set_allreduce_callback."""
    import torch

    def fn(buf):
        t = torch.from_numpy(np.ascontiguousarray(buf)).clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        buf[:] = t.numpy()
    return fn


def torch_allreduce_device(dist, device, group=None):
    """As torch_allreduce, for backends that only reduce device tensors (nccl): stages the host buffer through `device`."""
    import torch

    def fn(buf):
        t = torch.from_numpy(np.ascontiguousarray(buf)).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        buf[:] = t.cpu().numpy()
    return fn


def init_rccl(model, dist, rank):
    """Bootstraps the engine's own RCCL communicator: rank 0 creates the ncclUniqueId, torch.distributed
    (any backend) broadcasts the 128 bytes, every rank joins."""
    from .model import comm_unique_id
    obj = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(obj, src=0)
    model.comm_init(obj[0])
