"""Device-side counterparts of reference ``utils/mimic_utils.py:119-208`` (the three functions the hot
path calls; lines 1-118 of that file are CSV/caption helpers for dataset building and out of scope)."""
import torch

from .. import hip, ops
from ..hip import check, ptr, stream


def torch_broadcast_adj_matrix(adj_matrix, label_num=11, device=None):
    """[B, n, n] integer-valued labels -> [B, n, n, label_num] one-hot float32 (label 0 = no edge).
    One kernel; the reference issues label_num nonzero()+index-add launches with host syncs (:128-137)."""
    return process_matrix_labels(adj_matrix, adj_matrix.shape[1], label_num, device)


def process_matrix_labels(adj_matrix, num_objects, label_num, device=None, out=None):
    """``out``: optional (B, num_objects, num_objects, label_num) f32 destination (a half of a pair buffer)."""
    device = torch.device(device) if device is not None else adj_matrix.device
    adj = adj_matrix.to(device)
    if adj.dtype not in (torch.float64, torch.int64, torch.int8):
        adj = adj.to(torch.float64)
    adj = adj.contiguous()
    label_type = {torch.float64: 0, torch.int64: 1, torch.int8: 2}[adj.dtype]
    hip.require_cuda(adj)
    B, S = adj.shape[0], adj.shape[1]
    assert adj.shape[2] == S
    if out is None:
        out = torch.empty(B, num_objects, num_objects, label_num, device=device, dtype=torch.float32)
    assert out.shape == (B, num_objects, num_objects, label_num) and out.is_contiguous() and out.dtype == torch.float32
    check(hip.lib().ekaid_process_matrix(adj.data_ptr(), label_type, B, S, num_objects, label_num,
                                         ptr(out), stream()), 'process_matrix')
    return out


def process_matrix(adj_matrix, cfg, num_objects, device, type, out=None):
    """Same signature as the reference (mimic_utils.py:141-149) + an optional destination."""
    if type == 'spatial':
        label_num = cfg.model.change_detector.spa_label_num
    elif type == 'semantic':
        label_num = cfg.model.change_detector.sem_label_num
    else:
        raise ValueError('process_matrix: unknown type %r' % (type,))
    return process_matrix_labels(adj_matrix, num_objects, label_num, device, out)


def torch_extract_position_embedding_from_boxes(bbox, nongt_dim=36, feat_dim=64, wave_length=1000, device=None):
    """Fusion of torch_extract_position_matrix (:152-190) and torch_extract_position_embedding (:192-208):
    float64 boxes [R, N, 4] -> float32 [R, N, min(nongt, N), feat_dim] (float64 math on the device)."""
    device = torch.device(device) if device is not None else bbox.device
    feat_range = torch.arange(0, feat_dim / 8)
    dim_mat = torch.pow(torch.ones((1,)) * wave_length, (8. / feat_dim) * feat_range).to(device)
    bb = bbox.to(device=device, dtype=torch.float64)
    return ops.geom_embed(bb, min(nongt_dim, bb.shape[1]), feat_dim, dim_mat)
