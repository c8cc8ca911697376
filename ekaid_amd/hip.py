"""ctypes binding of libekaid_hip.so (the C ABI declared in include/ekaid_hip.h).

This is the stub a maintainer of the reference would add.  It fails loudly:
if the shared library is missing or a kernel returns an error, an exception is raised --
there is no CPU or ATen fallback behind it.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_longlong, c_size_t, c_uint,
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
        ('a_rowsum', c_void_p),
        ('post', c_void_p), ('ldpost', c_int),
        ('a_amax', c_void_p), ('b_amax', c_void_p),
    ]


class Slabs(Structure):
    """ekaid_slabs_t: element (row, col) of slab s = p[s * stride + row * ld + col]."""
    _fields_ = [('p', c_void_p), ('n', c_int), ('stride', c_longlong), ('ld', c_int)]


class SpkStepAFwd(Structure):
    """ekaid_spk_step_a_fwd_t (include/ekaid_hip.h)."""
    _fields_ = [('B', c_int), ('R', c_int), ('A', c_int), ('P', c_int),
                ('gxa', c_void_p), ('gates_src', Slabs * 3), ('n_gates_src', c_int), ('b_hh', c_void_p), ('c_prev', c_void_p),
                ('h_out', c_void_p), ('c_out', c_void_p), ('gate_save', c_void_p),
                ('w_fc', c_void_p), ('b_fc', c_void_p), ('bef', c_void_p), ('diff', c_void_p), ('aft', c_void_p),
                ('mw', c_void_p), ('att', c_void_p), ('att2', c_void_p), ('ld2', c_int),
                ('pos_src', Slabs), ('b_p1', c_void_p), ('w_wp', c_void_p), ('b_wp', c_void_p), ('w_p2', c_void_p),
                ('b_p2', c_void_p), ('p_v', c_float), ('seed_v', c_uint64), ('p_d', c_float), ('seed_d', c_uint64),
                ('vp', c_void_p), ('dp', c_void_p), ('pw', c_void_p), ('pp', c_void_p), ('ldpp', c_int)]


class SpkStepABwd(Structure):
    """ekaid_spk_step_a_bwd_t (include/ekaid_hip.h)."""
    _fields_ = [('B', c_int), ('R', c_int), ('A', c_int), ('P', c_int),
                ('dpa_src', Slabs), ('d_gated', c_void_p), ('gate', c_void_p),
                ('mw', c_void_p), ('bef', c_void_p), ('diff', c_void_p), ('aft', c_void_p), ('w_fc', c_void_p),
                ('dz_fc', c_void_p), ('d_bef', c_void_p), ('d_diff', c_void_p), ('d_aft', c_void_p),
                ('dh_carry', Slabs), ('dc_in', c_void_p), ('gate_save', c_void_p), ('c_prev', c_void_p), ('c_cur', c_void_p),
                ('dgates', c_void_p), ('lddg', c_int), ('dc_out', c_void_p),
                ('pw', c_void_p), ('dhead', c_void_p), ('w_p2', c_void_p), ('w_wp', c_void_p), ('vp', c_void_p),
                ('p_v', c_float), ('seed_v', c_uint64), ('p_d', c_float), ('seed_d', c_uint64),
                ('dpp_out', c_void_p), ('lddpp', c_int), ('dwp_out', c_void_p), ('dx_out', c_void_p), ('lddx', c_int)]


class SpeakerSeq(Structure):
    """ekaid_speaker_seq_t (include/ekaid_hip.h): the recurrent part of the teacher-forced decoder behind one call."""
    _fields_ = ([('T', c_int), ('B', c_int), ('R', c_int), ('A', c_int), ('P', c_int), ('G1', c_int), ('We', c_int),
                 ('p_lm', c_float), ('p_pos', c_float), ('seeds', c_void_p)] +
                [(k, c_void_p) for k in ('Wcat', 'whh_a', 'bhh_a', 'w_fc', 'b_fc', 'b_p1', 'w_wp', 'b_wp', 'w_p2', 'b_p2',
                                         'w_g1', 'b_g1', 'w_g2', 'b_g2', 'wih_l', 'bhh_l', 'bef', 'diff', 'aft', 'GXA', 'GXL',
                                         'HA', 'CA', 'HL', 'CL', 'SA', 'SL', 'G1D', 'MW', 'VP', 'DP', 'PW', 'PA', 'AT', 'GT', 'GA')])


class SpeakerSeqBwd(Structure):
    """ekaid_speaker_seq_bwd_t."""
    _fields_ = [('f', SpeakerSeq)] + [(k, c_void_p) for k in ('dHLo', 'dDPl', 'DCAT', 'DZFC', 'DWP', 'DPP', 'DZ2',
                                                               'd_bef', 'd_diff', 'd_aft')]


class SpeakerGreedy(Structure):
    """ekaid_speaker_greedy_t."""
    _fields_ = [('f', SpeakerSeq), ('emb', c_void_p), ('bih_l', c_void_p), ('w_lo', c_void_p), ('b_lo', c_void_p), ('V', c_int),
                ('xt', c_void_p), ('logits', c_void_p), ('logp', c_void_p), ('it', c_void_p), ('unfinished', c_void_p),
                ('seq', c_void_p), ('seq_logp', c_void_p), ('seq_len', c_int), ('n_unfinished', c_void_p)]


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
    'ekaid_gemm_pair_workspace_bytes': [POINTER(GemmDesc), POINTER(GemmDesc)],
    'ekaid_gemm_pair_f32': [POINTER(GemmDesc), POINTER(GemmDesc), P, c_size_t, P],
    'ekaid_gemm_slab_count': [POINTER(GemmDesc)],
    'ekaid_gemm_slabs': [POINTER(GemmDesc), P, c_size_t, P],
    'ekaid_gemm_slabs2': [POINTER(GemmDesc), P, c_size_t, POINTER(GemmDesc), P, c_size_t, P],
    'ekaid_slab_epilogue': [POINTER(Slabs), I, P, I, I, I, F, U64, P, I, P],
    'ekaid_drop_act_slabs_bwd': [POINTER(Slabs), P, I, I, I, I, F, U64, P, I, P],
    'ekaid_lstm_cell_slabs_fwd': [P, POINTER(Slabs), I, P, P, I, I, P, P, P, P],
    'ekaid_lstm_cell_slabs_bwd': [POINTER(Slabs), P, P, P, P, P, I, I, P, I, P, P],
    'ekaid_gru_cell_slabs_fwd': [P, I, POINTER(Slabs), P, P, I, I, I, P, I, P, P],
    'ekaid_gru_cell_slabs_bwd': [P, I, P, POINTER(Slabs), P, P, I, I, I, P, I, P, P, P],
    'ekaid_gate_mul_slabs_fwd': [POINTER(Slabs), P, P, I, I, P, P, P],
    'ekaid_gate_slabs_bwd': [POINTER(Slabs), P, P, I, I, P, P, P],
    'ekaid_gru_seq_workspace_bytes': [I, I, I, I],
    'ekaid_gru_seq_fwd': [P, P, P, I, I, I, P, P, P, c_size_t, P],
    'ekaid_gru_seq_bwd': [P, P, P, P, I, I, I, P, P, P, c_size_t, P],
    'ekaid_gru_seq_fwd_persistent': [P, P, P, I, I, I, P, P, P, c_size_t, P],
    'ekaid_gru_seq_bwd_persistent': [P, P, P, P, I, I, I, P, P, P, c_size_t, P],
    'ekaid_gru_seq_persistent_workspace_bytes': [I, I, I, I],
    'ekaid_speaker_seq_workspace_bytes': [POINTER(SpeakerSeq), I],
    'ekaid_speaker_seq_fwd': [POINTER(SpeakerSeq), P, c_size_t, P],
    'ekaid_speaker_seq_bwd': [POINTER(SpeakerSeqBwd), P, c_size_t, P],
    'ekaid_speaker_greedy_workspace_bytes': [POINTER(SpeakerGreedy)],
    'ekaid_speaker_greedy_steps': [POINTER(SpeakerGreedy), I, I, P, c_size_t, P],
    'ekaid_speaker_greedy_steps_persistent': [POINTER(SpeakerGreedy), I, I, P, c_size_t, P],
    'ekaid_speaker_greedy_persistent_workspace_bytes': [POINTER(SpeakerGreedy), I],
    'ekaid_spk_step_a_fwd': [POINTER(SpkStepAFwd), P],
    'ekaid_spk_step_a_bwd': [POINTER(SpkStepABwd), P],
    'ekaid_wn_scale_fwd': [P, P, L, P, P, P],
    'ekaid_wn_scale_bwd': [P, P, P, P, L, P, I, P, P],
    'ekaid_scale_dot': [P, P, P, L, P, P],
    'ekaid_wn_scale_multi': [P, P, P, I, P, P, P],
    'ekaid_wn_grad_fused': [P, P, P, P, P, L, P, P],
    'ekaid_wn_grad_multi': [P, P, P, P, P, P, P, I, P],
    'ekaid_colsum': [P, I, I, I, P, P],
    'ekaid_colsum_acc': [P, I, I, I, P, P],
    'ekaid_sum_n': [P, I, P, L, P],
    'ekaid_dropout_steps': [P, P, L, I, L, F, U64, P],
    'ekaid_act_bwd': [P, P, P, L, I, P],
    'ekaid_dropout': [P, P, L, F, U64, P],
    'ekaid_axpby': [P, P, P, L, F, F, P],
    'ekaid_mul': [P, P, P, L, P],
    'ekaid_rowsum': [P, L, I, P, I, P],
    'ekaid_rowmask_dropout': [P, L, I, P, P, F, U64, P],
    'ekaid_self_mix3': [P, P, L, F, F, F, P],
    'ekaid_expand_rows_fwd': [P, P, L, I, I, I, P, F, U64, P],
    'ekaid_expand_rows_bwd': [P, P, L, I, I, I, P, F, U64, P],
    'ekaid_gat_fused_fwd': [P, P, P, P, P, P, I, I, I, I, I, I, P, P, P],
    'ekaid_gat_dqk': [P, P, P, I, I, I, I, I, F, P, P, P],
    'ekaid_gat_bwd_core': [P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P],
    'ekaid_masked_softmax_fwd': [P, P, P, P, I, I, I, I, I, P, P],
    'ekaid_masked_softmax_bwd': [P, P, P, P, I, I, I, I, I, P, P, P, P],
    'ekaid_geom_embed': [P, P, I, I, I, I, P, P],
    'ekaid_geom_embed_dropout': [P, P, I, I, I, I, P, F, U64, P],
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
    'ekaid_ss_tokens': [P, P, I, I, I, F, U64, P, P],
    'ekaid_speaker_seq_persistent_workspace_bytes': [POINTER(SpeakerSeq)],
    'ekaid_speaker_seq_fwd_persistent': [POINTER(SpeakerSeq), P, c_size_t, P],
    'ekaid_speaker_seq_persistent_status_ptr': [P],
    'ekaid_speaker_seq_bwd_persistent_workspace_bytes': [POINTER(SpeakerSeq)],
    'ekaid_speaker_seq_bwd_persistent': [POINTER(SpeakerSeqBwd), P, P, c_size_t, P],
    'ekaid_gemm_plan_query': [POINTER(GemmDesc), POINTER(c_int), POINTER(c_int)],
    'ekaid_seq_status_read': [POINTER(c_uint), P],
    'ekaid_seq_status_clear': [P],
    'ekaid_set_seq_mode': [I],
    'ekaid_get_seq_mode': [],
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
    'ekaid_nonfinite_count': [P, L, P, P],
    'ekaid_adam_step': [P, P, P, P, L, F, F, F, F, I, F, P],
    'ekaid_adam_step_dev': [P, P, P, P, L, P, P, P, P, P],
    'ekaid_seq_fault_flag': [P, P],
    'ekaid_launch_count': [I],
    'ekaid_amax_rc': [P, L, I, I, P, P, I, P],
    'ekaid_amax_rc_multi': [P, P, P, P, P, P, I, I, P],
}
_RESTYPES = {'ekaid_last_error': c_char_p, 'ekaid_launch_count': c_longlong, 'ekaid_gemm_workspace_bytes': c_size_t, 'ekaid_gemm_pair_workspace_bytes': c_size_t, 'ekaid_gru_seq_workspace_bytes': c_size_t, 'ekaid_gru_seq_persistent_workspace_bytes': c_size_t,
             'ekaid_speaker_seq_workspace_bytes': c_size_t, 'ekaid_speaker_seq_persistent_workspace_bytes': c_size_t,
             'ekaid_speaker_seq_persistent_status_ptr': c_void_p, 'ekaid_speaker_seq_bwd_persistent_workspace_bytes': c_size_t, 'ekaid_speaker_greedy_workspace_bytes': c_size_t, 'ekaid_speaker_greedy_persistent_workspace_bytes': c_size_t}

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


def seq_status_check(what='persistent sequence kernel'):
    """Raise if a persistent sequence kernel (csrc/persist*.hip) abandoned a launch since the last clear: a seam timed out -- typically
    because not all of its workgroups were resident (another process sharing the GPU).  Synchronises the current stream."""
    v = c_uint(0)
    check(lib().ekaid_seq_status_read(ctypes.byref(v), stream()), 'seq_status_read')
    if v.value:
        lib().ekaid_seq_status_clear(stream())
        raise EkaidError('%s: a seam did not open in time (status 0x%x): the launch was abandoned and its outputs are invalid. Is another '
                         'process using this GPU? (EKAID_SEQ_MODE=stepwise selects the launch-per-product chains)' % (what, v.value))


def seq_status_take():
    """The status word of the persistent sequence kernels (0 = fine), cleared if it was set.  Synchronises the current stream."""
    v = c_uint(0)
    check(lib().ekaid_seq_status_read(ctypes.byref(v), stream()), 'seq_status_read')
    if v.value:
        check(lib().ekaid_seq_status_clear(stream()), 'seq_status_clear')
    return int(v.value)


# indices of ekaid_launch_count (include/ekaid_hip.h)
COUNT = dict(spk_fwd_persistent=0, spk_bwd_persistent=1, greedy_persistent=2, gru_fwd_persistent=3, gru_bwd_persistent=4,
             spk_fwd_stepwise=5, spk_bwd_stepwise=6, greedy_stepwise=7, gemm_f32=8, gemm_split3=9, gemm_split2=10, gat_bwd_core=11)


def launch_counts():
    """name -> launches issued so far by this process (ekaid_launch_count): which implementation served the calls made in between two
    readings (persistent sequence kernels vs step-wise chains, GEMM arithmetic)."""
    L = lib()
    return {k: int(L.ekaid_launch_count(i)) for k, i in COUNT.items()}


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
POISON = os.environ.get('EKAID_POISON', '0') not in ('0', '', 'off', 'false')


_retired = []           # outgrown scratch buffers: kept allocated, a captured hipGraph may still hold their address


def workspace(nbytes, device):
    """Grow-only per-device scratch buffer (split-K slabs, persistent-kernel hand-off buffers).  A buffer that has to grow is RETIRED,
    never freed: a hipGraph captured while it was current replays kernels that write through its address (a second shape captured
    later with a larger need -- more decode steps -- would otherwise leave the first graph scribbling over whatever the allocator put
    there next).  Growth is by at least 1.5x, so the retired buffers together stay below twice the current one."""
    buf = _workspaces.get(device)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise EkaidError('hip.workspace: the scratch buffer would have to grow (%d -> %d bytes) inside a hipGraph capture; run the '
                             'step once eagerly first (TrainStep warm-up) or call hip.reserve_workspace' % (0 if buf is None else buf.numel(), nbytes))
        want = max(int(nbytes), 1 << 20, 0 if buf is None else buf.numel() * 3 // 2)
        if buf is not None:
            _retired.append(buf)
        buf = torch.empty(want, dtype=torch.uint8, device=device)
        _workspaces[device] = buf
    if POISON:
        buf.fill_(0xFF)              # every float read before it was written is a NaN (debug mode, see enable_poison; inside a
                                     # capture the fill becomes a node of the graph: every replay poisons again)
    return buf


_real_empty = None


def enable_poison():
    """Debug mode (EKAID_POISON=1, tests/conftest.py): every float tensor ``torch.empty`` / ``empty_like`` hands out on the device is
    filled with NaN and the scratch workspace with 0xFF bytes before each use, so a kernel whose result depends on memory it did not
    write (an uninitialised read, an output only partly written) turns that result into NaN instead of into whatever the allocator
    happened to return -- the suspicion behind ADVICE r04 "replay numerics depend on block placement" (ekaid_amd/engine.py)."""
    global _real_empty, POISON
    POISON = True
    if _real_empty is not None:
        return
    _real_empty = (torch.empty, torch.empty_like)

    def empty(*a, **k):
        t = _real_empty[0](*a, **k)
        if t.is_cuda and t.dtype.is_floating_point:
            t.fill_(float('nan'))
        return t

    def empty_like(*a, **k):
        t = _real_empty[1](*a, **k)
        if t.is_cuda and t.dtype.is_floating_point:
            t.fill_(float('nan'))
        return t
    torch.empty, torch.empty_like = empty, empty_like


def reserve_workspace(nbytes, device):
    """Pre-size the scratch buffer (call before capturing a HIP graph)."""
    workspace(nbytes, device)
