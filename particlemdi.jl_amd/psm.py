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


def psm_counts_device(samples, row_lo, row_hi, n_labels=0):
    """Co-clustering counts (consensus_map.jl:50-56) on the MI355X: samples is a CUDA uint8 tensor
    (S, K, n); returns an int32 CUDA tensor (K, row_hi-row_lo, n) with
    counts[k, i-row_lo, j] = #{t : samples[t, k, i] == samples[t, k, j]} (libpmdi_hip.so, pmdi_psm_counts_device).
    n_labels: every label is < n_labels (the model's N); 1..64 selects the matrix-core kernel, 0 = unknown."""
    import ctypes as C
    import torch
    from ._lib import _check, lib
    if not samples.is_cuda or samples.dtype != torch.uint8:
        raise ValueError("psm_counts_device needs a CUDA uint8 tensor (S, K, n)")
    smp = samples.contiguous()
    S, K, n = smp.shape
    out = torch.empty((K, row_hi - row_lo, n), dtype=torch.int32, device=smp.device)
    st = torch.cuda.current_stream(smp.device)
    _check(lib().pmdi_psm_counts_device(smp.device.index or 0, C.c_void_p(smp.data_ptr()), S, K, n, int(row_lo), int(row_hi),
                                        int(n_labels), C.c_void_p(out.data_ptr()), C.c_void_p(st.cuda_stream)))
    return out


def psm_rows(samples, row_lo, row_hi, n_labels=0, host=False):
    """Rows [row_lo, row_hi) of the K (+1) posterior-similarity matrices from pooled samples
    (S, K, n); lower triangle as the reference fills it, identity elsewhere.  Works on torch
    tensors or numpy arrays.  CUDA tensors go through the HIP kernels (pmdi_psm_counts_device); host data is only
    accepted with host=True (the plain-torch mirror used by the CPU tests and the gloo rehearsal).  The rows of a matrix are independent, so ranks
    split them with no further exchange."""
    is_np = isinstance(samples, np.ndarray)
    import torch
    if is_np:
        samples = torch.from_numpy(samples)
    if not samples.is_cuda and not host:
        # no silent CPU path: the counts come from the HIP kernels unless the caller asks for the host mirror
        raise ValueError("psm_rows: samples are not on an MI355X; pass host=True for the host mirror (tests, gloo rehearsal)")
    S, K, n = samples.shape
    out = torch.zeros((K + (1 if K > 1 else 0), row_hi - row_lo, n), dtype=torch.float64, device=samples.device)
    rows = torch.arange(row_lo, row_hi, device=samples.device)
    cols = torch.arange(n, device=samples.device)
    lower = (rows[:, None] > cols[None, :])
    eye = (rows[:, None] == cols[None, :]).to(torch.float64)
    S_t = torch.full((), float(S), dtype=torch.float64, device=samples.device)
    K_t = torch.full((), float(K), dtype=torch.float64, device=samples.device)
    dev_counts = psm_counts_device(samples, row_lo, row_hi, n_labels) if samples.is_cuda else None   # the HIP kernels
    for k in range(K):
        if dev_counts is not None:
            acc = dev_counts[k].to(torch.float64)
        else:       # host tensors (the gloo rehearsal of the exchange step): plain torch
            acc = torch.zeros((row_hi - row_lo, n), dtype=torch.float64, device=samples.device)
            for t in range(S):
                lab = samples[t, k]
                acc += (lab[row_lo:row_hi, None] == lab[None, :]).to(torch.float64)
        out[k] = torch.div(acc, S_t) * lower + eye      # tensor divisor: an IEEE division on every backend
    if K > 1:
        out[K] = eye
        for k in range(K):
            out[K] += torch.div(out[k], K_t)
        out[K] = out[K] * (1.0 - eye) + eye       # diagind .= 1.0
    return out.numpy() if is_np else out


class PosteriorSimilarityMatrix:
    """`Posterior_similarity_matrix` of consensus_map.jl:6-11: `psm` = K (+1 "Overall" if K > 1) n x n Float64 matrices,
    `names` = the dataset names (+ "Overall")."""

    def __init__(self, psm, names):
        self.psm, self.names = psm, names


def generate_psm(outputFile, burnin=0, thin=1, host=False, device=None):
    """generate_psm(outputFile, burnin, thin) of src/output_analysis/consensus_map.jl:31-65 on a file written by pmdi():
    the native reader (pmdi_csv_read_allocations) takes the allocation samples, the co-clustering counts come from the HIP
    kernels (pmdi_psm_counts_device) and the division / identity / "Overall" average follow :50-63.  host=True runs the plain
    host mirror instead (tests, machines without an MI355X); there is no silent fallback."""
    import torch
    from ._lib import read_allocations
    samples, names = read_allocations(outputFile, burnin, thin)
    S, K, n = samples.shape
    if S < 1:
        raise ValueError("generate_psm: no rows left after burn-in and thinning")
    if host:
        rows = psm_rows(samples, 0, n, host=True)
    else:
        if not torch.cuda.is_available():
            raise RuntimeError("generate_psm: no MI355X visible; pass host=True for the host mirror")
        dev = torch.device("cuda", 0 if device is None else int(device))
        rows = psm_rows(torch.from_numpy(samples).to(dev), 0, n, n_labels=int(samples.max()) + 1).cpu().numpy()
    return PosteriorSimilarityMatrix([rows[k] for k in range(rows.shape[0])], list(names) + (["Overall"] if K > 1 else []))
