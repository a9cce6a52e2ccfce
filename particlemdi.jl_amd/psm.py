"""Consumer side of the one exchange step of the multi-GPU path: retained allocation samples
of all chains are all-gathered (RCCL over xGMI when the backend is "nccl", gloo on CPU) and
turned into the posterior-similarity matrix of generate_psm (consensus_map.jl:31-65): element
(i, j), i > j, of dataset k = fraction of samples in which observations i and j share a label;
diagonal 1; for K > 1 an extra "Overall" matrix = mean of the K matrices.

Chains are independent, so this is the ONLY collective of the path.
"""
import numpy as np


def allgather_samples(samples):
    """samples: uint8 tensor (T, C, K, n) of this rank -> (world*T*C, K, n) on every rank."""
    import torch
    import torch.distributed as dist
    flat = samples.reshape(-1, samples.shape[-2], samples.shape[-1]).contiguous()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flat
    out = torch.empty((dist.get_world_size() * flat.shape[0],) + tuple(flat.shape[1:]),
                      dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat)
    return out


def psm_rows(samples, row_lo, row_hi):
    """Rows [row_lo, row_hi) of the K (+1) posterior-similarity matrices from pooled samples
    (S, K, n); lower triangle as the reference fills it, identity elsewhere.  Works on torch
    tensors (any device) or numpy arrays.  The rows of a matrix are independent, so ranks
    split them with no further exchange."""
    is_np = isinstance(samples, np.ndarray)
    if is_np:
        import torch
        samples = torch.from_numpy(samples)
    import torch
    S, K, n = samples.shape
    out = torch.zeros((K + (1 if K > 1 else 0), row_hi - row_lo, n), dtype=torch.float64, device=samples.device)
    rows = torch.arange(row_lo, row_hi, device=samples.device)
    cols = torch.arange(n, device=samples.device)
    lower = (rows[:, None] > cols[None, :])
    eye = (rows[:, None] == cols[None, :]).to(torch.float64)
    for k in range(K):
        acc = torch.zeros((row_hi - row_lo, n), dtype=torch.float64, device=samples.device)
        for t in range(S):
            lab = samples[t, k]
            acc += (lab[row_lo:row_hi, None] == lab[None, :]).to(torch.float64)
        out[k] = (acc / S) * lower + eye
    if K > 1:
        out[K] = eye
        for k in range(K):
            out[K] += out[k] / K
        out[K] = out[K] * (1.0 - eye) + eye       # diagind .= 1.0
    return out.numpy() if is_np else out
