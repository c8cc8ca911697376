"""ctypes binding of libekaid_hip.so (the C ABI declared in include/ekaid_hip.h).

This is the stub a maintainer of the reference would add.  It fails loudly:
if the shared library is missing or a kernel returns an error, an exception is raised --
there is no CPU or ATen fallback behind it.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_longlong, c_size_t,
                    c_uint64, c_void_p)

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EKAID_HIP_LIB') or os.path.join(_HERE, 'lib', 'libekaid_hip.so')   # env: A/B builds

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
ERR_UNSUPPORTED = -3


class EkaidError(RuntimeError):
    pass


class GemmDesc(Structure):
    _fields_ = [
        ('A', c_void_p), ('B', c_void_p), ('C', c_void_p),
        ('M', c_int), ('N', c_int), ('K', c_int),
        ('lda', c_int), ('ldb', c_int), ('ldc', c_int),
        ('transA', c_int), ('transB', c_int),
        ('batch1', c_int), ('batch2', c_int),
        ('strideA1', c_longlong), ('strideA2', c_longlong),
        ('strideB1', c_longlong), ('strideB2', c_longlong),
        ('strideC1', c_longlong), ('strideC2', c_longlong),
        ('alpha', c_float), ('alpha_dev', c_void_p),
        ('accumulate', c_int), ('bias', c_void_p),
        ('R', c_void_p), ('ldr', c_int), ('rdiv', c_int), ('rowmask', c_void_p),
        ('act', c_int),
        ('drop_p', c_float), ('drop_seed', c_uint64),
    ]


P, I, L, F, U64 = c_void_p, c_int, c_longlong, c_float, c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/ekaid_hip.h 1:1
SIGNATURES = {
    'ekaid_version': [],
    'ekaid_last_error': [],
    'ekaid_init': [I],
    'ekaid_set_dropout_epoch_ptr': [P],
    'ekaid_dropout_epoch_advance': [P, U64, P],
    'ekaid_gemm_workspace_bytes': [POINTER(GemmDesc)],
    'ekaid_set_gemm_mode': [I],
    'ekaid_get_gemm_mode': [],
    'ekaid_gemm_f32': [POINTER(GemmDesc), P, c_size_t, P],
    'ekaid_wn_scale_fwd': [P, P, L, P, P, P],
    'ekaid_wn_scale_bwd': [P, P, P, P, L, P, I, P, P],
    'ekaid_scale_dot': [P, P, P, L, P, P],
    'ekaid_colsum': [P, I, I, I, P, P],
    'ekaid_colsum_acc': [P, I, I, I, P, P],
    'ekaid_sum_n': [P, I, P, L, P],
    'ekaid_dropout_steps': [P, P, L, I, L, F, U64, P],
    'ekaid_act_bwd': [P, P, P, L, I, P],
    'ekaid_dropout': [P, P, L, F, U64, P],
    'ekaid_axpby': [P, P, P, L, F, F, P],
    'ekaid_mul': [P, P, P, L, P],
    'ekaid_rowsum': [P, L, I, P, I, P],
    'ekaid_expand_rows_fwd': [P, P, L, I, I, I, P, F, U64, P],
    'ekaid_expand_rows_bwd': [P, P, L, I, I, I, P, F, U64, P],
    'ekaid_gat_fused_fwd': [P, P, P, P, P, P, I, I, I, I, I, I, P, P, P],
    'ekaid_masked_softmax_fwd': [P, P, P, P, I, I, I, I, I, P, P],
    'ekaid_masked_softmax_bwd': [P, P, P, P, I, I, I, I, I, P, P, P, P],
    'ekaid_geom_embed': [P, P, I, I, I, I, P, P],
    'ekaid_residual_relu2_fwd': [P, P, P, L, F, U64, P],
    'ekaid_residual_relu2_bwd': [P, P, P, L, F, U64, P],
    'ekaid_gate_fuse_fwd': [P, P, P, P, P, L, L, F, U64, U64, P],
    'ekaid_gate_fuse_bwd': [P, P, P, P, P, P, P, P, P, L, L, F, U64, U64, P],
    'ekaid_weighted_pool_fwd': [P, P, I, I, I, P, P],
    'ekaid_weighted_pool_bwd': [P, P, P, I, I, I, P, P, P],
    'ekaid_qatt_weights_fwd': [P, I, I, P, P, P],
    'ekaid_qatt_weights_bwd': [P, P, I, I, P, P],
    'ekaid_embedding_fwd': [P, L, P, I, P, I, I, I, P],
    'ekaid_embedding_bwd': [P, L, P, I, P, I, I, I, I, L, P, P],
    'ekaid_onehot': [P, L, I, L, P, P],
    'ekaid_masked_add': [P, P, I, P, L, P],
    'ekaid_gru_cell_fwd': [P, I, P, I, P, I, I, I, P, I, P, P],
    'ekaid_gru_cell_bwd': [P, I, P, P, P, I, I, I, P, I, P, P, P],
    'ekaid_lstm_cell_fwd': [P, P, I, P, I, I, P, P, P, P],
    'ekaid_lstm_cell_bwd': [P, P, P, P, P, P, I, I, P, I, P, P],
    'ekaid_bias_act_drop': [P, I, P, I, I, I, F, U64, P, P],
    'ekaid_drop_act_bwd': [P, P, I, I, I, I, F, U64, P, I, P],
    'ekaid_spk_modmix_fwd': [P, I, I, P, P, P, P, P, I, P, P, P, I, P],
    'ekaid_spk_modmix_bwd': [P, P, P, P, I, I, P, I, P, P, P, I, P, P, P, P, P, P],
    'ekaid_spk_pos_fwd': [P, I, P, I, I, P, P, I, P, P, F, U64, F, U64, P, P, P, P, I, P],
    'ekaid_spk_pos_bwd': [P, I, P, P, I, I, P, P, P, I, F, U64, F, U64, P, P, I, P],
    'ekaid_spk_gate_bwd': [P, P, P, L, P, P],
    'ekaid_softmax_fwd': [P, P, L, I, I, P],
    'ekaid_softmax_bwd': [P, P, P, L, I, I, P],
    'ekaid_mix3_fwd': [P, P, P, P, I, I, P, P],
    'ekaid_mix3_bwd': [P, P, P, P, P, I, I, P, P, P, P, I, P],
    'ekaid_nll_fwd': [P, P, I, P, I, I, I, I, P, P, P, P],
    'ekaid_total_loss_fwd': [P, P, P, L, F, P, P],
    'ekaid_fill_scaled': [P, F, P, L, P],
    'ekaid_nll_bwd': [P, I, P, I, P, P, I, I, I, P, P],
    'ekaid_layernorm_fwd': [P, P, P, L, I, F, P, P, P],
    'ekaid_layernorm_bwd': [P, P, P, P, L, I, P, P, P, P],
    'ekaid_process_matrix': [P, I, I, I, I, I, P, P],
    'ekaid_greedy_step': [P, I, I, I, P, P, P, P, I, P, P],
    'ekaid_transpose': [P, I, I, I, P, I, P],
    'ekaid_adam_step': [P, P, P, P, L, F, F, F, F, I, F, P],
    'ekaid_adam_step_dev': [P, P, P, P, L, P, P, P, P],
}
_RESTYPES = {'ekaid_last_error': c_char_p, 'ekaid_gemm_workspace_bytes': c_size_t}

_lib = None


def load(path=None):
    """dlopen the library and bind every symbol of SIGNATURES (missing symbol -> error)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise EkaidError(
            'HIP extension not built: %s is missing (run `python -c "import __graft_entry__ as g; '
            'g.build()"` or `make -C ekaid_amd/csrc`). There is no fallback path.' % path)
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if path == LIB_PATH or _lib is None:
        _lib = lib
    return lib


def lib():
    return _lib if _lib is not None else load()


def last_error():
    msg = lib().ekaid_last_error()
    return msg.decode() if msg else ''


_calls = 0


def calls_issued():
    """C-ABI calls checked so far in this process (every kernel-launching call goes through ``check``): a host-side clock
    for "was X enqueued before Y" assertions (tests/test_hip_dp.py: the first gradient bucket's all-reduce vs the last
    backward kernel)."""
    return _calls


def check(rc, what=''):
    global _calls
    _calls += 1
    if rc != 0:
        raise EkaidError('%s failed (rc=%d): %s' % (what or 'ekaid call', rc, last_error()))


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise EkaidError('ekaid_amd ops run on the HIP device only (got a %s tensor); the CPU '
                             'restatement lives under oracle/ and is test infrastructure' % t.device)


_workspaces = {}


def workspace(nbytes, device):
    """Grow-only per-device scratch buffer (split-K slabs)."""
    buf = _workspaces.get(device)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[device] = buf
    return buf


def reserve_workspace(nbytes, device):
    """Pre-size the scratch buffer (call before capturing a HIP graph)."""
    workspace(nbytes, device)
