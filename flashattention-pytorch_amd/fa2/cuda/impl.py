"""FA2 front end of the HIP extension — same public names as /root/reference/src/fa2/cuda/impl.py
(`_load_ext`, `_merge_bh`, `_split_bh`, `_split_bh_lse`, `_FA2CudaFn`, `fa2_cuda`), built from the shared pieces
in `common/hip_autograd.py`.  Extension attributes used: `forward` / `backward` (csrc/common/torch.extension.cpp:73-83).
3-D (BH,N,d) and 4-D (B,H,N,d) inputs are both accepted and come back with the same rank (the 3-D case follows the
fixed `_merge_bh` of src/fa3/cuda/impl.py:18-22).
"""
from common.hip_autograd import load_extension, make_attention_function, merge_bh, run_attention, split_bh

_load_ext = load_extension
_merge_bh = merge_bh
_split_bh = split_bh
_split_bh_lse = split_bh
_FA2CudaFn = make_attention_function("_FA2CudaFn", "forward", "backward", n_extra=0)


def fa2_cuda(q, k, v, causal, softmax_scale, spec):
    return run_attention(_FA2CudaFn, q, k, v, causal, softmax_scale, spec.br, spec.bc)
