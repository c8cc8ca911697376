"""GPU parity of every C-ABI kernel / autograd operator against plain float64/float32 PyTorch-CPU
restatements of the same formula (op level).  Whole-model parity lives in test_hip_model.py."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


def relerr(got, want):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-12))


def check(got, want, tol, what):
    e = relerr(got, want)
    assert e <= tol, '%s: rel-to-max err %.3e > %.1e' % (what, e, tol)


@pytest.fixture(scope='module', autouse=True)
def _lib():
    from ekaid_amd import hip
    hip.load()
    assert torch.cuda.is_available()
    yield


# --------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [
    # M, N, K
    (128, 128, 32), (256, 128, 64), (130, 70, 33), (1, 1, 1), (2, 148, 512), (576, 1024, 2048),
    (64, 64, 1324), (37, 3, 512), (700, 4, 64), (16, 64, 128), (512, 256, 96), (9, 5, 11),
]


@pytest.fixture(params=['bf16x3', 'f32'])
def gemm_arith(request):
    """Run a test under both GEMM arithmetics (exact bf16 split on the bf16 matrix pipe / f32 MFMA)."""
    from ekaid_amd import ops
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(ops.GEMM_BF16_SPLIT if request.param == 'bf16x3' else ops.GEMM_F32_MFMA)
    yield request.param
    ops.set_gemm_mode(old)


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
@pytest.mark.parametrize('mode', ['nt', 'nn', 'tn'])
def test_gemm_variants(M, N, K, mode, gemm_arith):
    from ekaid_amd import ops
    A = rnd(M, K, seed=1)
    B = rnd(K, N, seed=2)
    want = A.double() @ B.double()
    if mode == 'nt':      # A [M][K], B stored [N][K]
        a, b, ta, tb, lda, ldb = A, B.t().contiguous(), 0, 1, K, K
    elif mode == 'nn':    # A [M][K], B stored [K][N]
        a, b, ta, tb, lda, ldb = A, B, 0, 0, K, N
    else:                 # A stored [K][M], B stored [K][N]
        a, b, ta, tb, lda, ldb = A.t().contiguous(), B, 1, 0, M, N
    C = torch.full((M, N), float('nan'), device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), C, M, N, K, lda, ldb, N, ta, tb)
    check(C, want, 2e-6 * max(1.0, math.sqrt(K) / 4), 'gemm %s %dx%dx%d' % (mode, M, N, K))


@pytest.mark.parametrize('mode', ['nt', 'nn', 'tn'])
@pytest.mark.parametrize('M0,M1,N,K', [(2304, 2304, 1024, 1024), (18432, 18432, 1024, 1024), (1152, 2304, 1024, 1024), (1024, 1024, 1024, 2304),
                                       (256, 256, 2048, 1536), (300, 200, 500, 700), (4608, 4608, 128, 1024)])
def test_gemm_pair_is_bit_identical_to_two_launches(M0, M1, N, K, mode, gemm_arith):
    """ekaid_gemm_pair_f32: two independent products in one launch keep the plan (tile, K split, summation order) and the epilogue of
    each, so every output bit equals the two-launch result -- for the three operand layouts, full epilogues (bias, activation,
    dropout, post / accumulate), K-split products, the A row sums of the weight-gradient form, unequal row counts, ragged shapes, and
    the pairs the planner sends to two launches."""
    from ekaid_amd import hip as H, ops
    from ekaid_amd.hip import ACT_RELU, ACT_TANH

    def operands(M, seed):
        A, B = rnd(M, K, seed=seed), rnd(K, N, seed=seed + 1)
        if mode == 'nt':
            return A.to(DEV), B.t().contiguous().to(DEV), 0, 1, K, K
        if mode == 'nn':
            return A.to(DEV), B.to(DEV), 0, 0, K, N
        return A.t().contiguous().to(DEV), B.to(DEV), 1, 0, M, N
    a0, b0, ta, tb, lda0, ldb0 = operands(M0, 11)
    a1, b1, _, _, lda1, ldb1 = operands(M1, 21)
    bias = rnd(N, seed=31).to(DEV)
    post = rnd(M1, N, seed=32).to(DEV)
    base = rnd(M0, N, seed=33).to(DEV)
    rs0, rs1 = (torch.zeros(M0, device=DEV), torch.zeros(M1, device=DEV)) if mode == 'tn' else (None, None)

    def run(pair, one_sided=False):
        c0, c1 = base.clone(), torch.full((M1, N), float('nan'), device=DEV)
        r0 = rs0.clone() if rs0 is not None else None
        r1 = rs1.clone() if (rs1 is not None and not one_sided) else None
        g0 = ((a0, b0, c0, M0, N, K, lda0, ldb0, N, ta, tb), dict(accumulate=True, bias=bias, act=ACT_TANH, a_rowsum=r0))
        g1 = ((a1, b1, c1, M1, N, K, lda1, ldb1, N, ta, tb), dict(alpha=0.5, act=ACT_RELU, drop_p=0.25, drop_seed=1234, post=post, ldpost=N,
                                                                 a_rowsum=r1))
        if pair:
            ops.gemm_pair(g0, g1)
        else:
            ops.gemm(*g0[0], **g0[1])
            ops.gemm(*g1[0], **g1[1])
        torch.cuda.synchronize()
        return c0, c1, r0, r1
    solo, paired = run(False), run(True)
    for x, y, name in zip(solo, paired, ('C0', 'C1', 'rowsum0', 'rowsum1')):
        if x is not None:
            assert torch.equal(x, y), '%s differs between the pair launch and two launches (%s %d/%dx%dx%d)' % (name, mode, M0, M1, N, K)
    assert torch.isfinite(paired[1]).all()
    if mode == 'tn':                      # the A row sums on ONE side only: still the two solo launches, bit for bit
        for x, y, name in zip(run(False, True), run(True, True), ('C0', 'C1', 'rowsum0', 'rowsum1')):
            if x is not None:
                assert torch.equal(x, y), '%s differs with one-sided row sums (%s %d/%dx%dx%d)' % (name, mode, M0, M1, N, K)
    want = torch.tanh(base.double().cpu() + (rnd(M0, K, seed=11).double() @ rnd(K, N, seed=12).double()) + bias.double().cpu())
    check(paired[0], want, 2e-5 * max(1.0, math.sqrt(K) / 4), 'pair product 0')


def test_scratch_buffer_that_grows_stays_valid_for_graphs_captured_before():
    """hip.workspace: a hipGraph captured while a scratch buffer was current keeps writing through its address at every replay.  When
    a later, larger product makes the buffer grow (train_mimic.py --hip_graph meeting a batch with more decode steps), the old buffer is
    retired, not freed: the first graph's replay still gives its result and does not touch memory the allocator hands out afterwards."""
    from ekaid_amd import hip as H, ops
    dev = torch.empty(1, device=DEV).device
    saved = H._workspaces.pop(dev, None)
    retired0 = len(H._retired)
    try:
        import ctypes

        def need(M, N, K):
            d = ops._fill_desc(a, b, c_eager, M, N, K, M, N, N, 1, 0)
            return H.lib().ekaid_gemm_workspace_bytes(ctypes.byref(d))
        M, N, K = 256, 256, 9216                                      # weight-gradient form, K-split: slabs in the scratch buffer
        a, b = rnd(K, M, seed=5).to(DEV), rnd(K, N, seed=6).to(DEV)
        c_eager, c_graph = torch.empty(M, N, device=DEV), torch.zeros(M, N, device=DEV)
        assert need(M, N, K) > 0, 'the test product must need slabs (K-split)'
        ops.gemm(a, b, c_eager, M, N, K, M, N, N, 1, 0)               # sizes the buffer
        small = H._workspaces[dev]
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            ops.gemm(a, b, c_graph, M, N, K, M, N, N, 1, 0)
        M2, N2, K2 = 1024, 1024, 9216                                  # needs a larger buffer
        assert need(M2, N2, K2) > small.numel() * 3 // 2, 'the second product was meant to outgrow the buffer'
        a2, b2 = rnd(K2, M2, seed=7).to(DEV), rnd(K2, N2, seed=8).to(DEV)
        c2 = torch.empty(M2, N2, device=DEV)
        ops.gemm(a2, b2, c2, M2, N2, K2, M2, N2, N2, 1, 0)
        big = H._workspaces[dev]
        assert big is not small and big.numel() >= small.numel() * 3 // 2, 'the second product was meant to outgrow the buffer'
        assert any(r is small for r in H._retired[retired0:]), 'the outgrown buffer must stay allocated'
        small_n = small.numel()
        del small
        victims = [torch.zeros(small_n, dtype=torch.uint8, device=DEV) for _ in range(16)]           # would land on freed memory
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(c_graph, c_eager), 'replay through the retired scratch buffer'
        assert all(int(v.max()) == 0 for v in victims), 'the replay wrote outside its own scratch buffer'
        del g
    finally:
        torch.cuda.synchronize()
        H._workspaces.pop(dev, None)
        del H._retired[retired0:]
        if saved is not None:
            H._workspaces[dev] = saved


@pytest.mark.parametrize('mode', ['nt', 'nn', 'tn'])
@pytest.mark.parametrize('M,N,K', [(512, 384, 1024), (256, 2048, 512), (4608, 1024, 2048), (1000, 600, 4096)])
def test_gemm_split_accuracy(M, N, K, mode):
    """The 3-way bf16 split AND the two-term f16 split (row / column scaled) are f32-accurate: against float64, their error is no
    larger than 1.5x the error of the f32 MFMA (an exact f32 fma chain) on the same data -- including data with a wide dynamic range
    (exp(3 N(0, 1)) per row of A and column of B: ~2^26 between rows)."""
    from ekaid_amd import ops
    g = torch.Generator().manual_seed(77)
    A = torch.randn(M, K, generator=g) * torch.exp(3 * torch.randn(M, 1, generator=g))
    B = torch.randn(K, N, generator=g) * torch.exp(3 * torch.randn(1, N, generator=g))
    want = A.double() @ B.double()
    scale = (A.double().abs() @ B.double().abs())          # per-element condition scale
    if mode == 'nt':
        a, b, ta, tb, lda, ldb = A, B.t().contiguous(), 0, 1, K, K
    elif mode == 'nn':
        a, b, ta, tb, lda, ldb = A, B, 0, 0, K, N
    else:
        a, b, ta, tb, lda, ldb = A.t().contiguous(), B, 1, 0, M, N
    a, b = a.to(DEV), b.to(DEV)
    errs = {}
    old = ops.get_gemm_mode()
    try:
        for name, m in (('f32', ops.GEMM_F32_MFMA), ('bf16x3', ops.GEMM_BF16_SPLIT), ('f16x2', ops.GEMM_F16X2)):
            ops.set_gemm_mode(m)
            C = torch.full((M, N), float('nan'), device=DEV)
            kw = _bounds_kw(a, b, ta, tb) if m == ops.GEMM_F16X2 else {}
            ran = _gemm_counted(a, b, C, M, N, K, lda, ldb, ta, tb, **kw)
            assert ran[{'f32': 'gemm_f32', 'bf16x3': 'gemm_split3', 'f16x2': 'gemm_split2'}[name]] >= 1, (name, ran)
            e = (C.double().cpu() - want).abs() / scale
            errs[name] = (float(e.max()), float(e.mean()))
    finally:
        ops.set_gemm_mode(old)
    print('gemm accuracy %s %dx%dx%d: f32 max %.2e mean %.2e | bf16x3 max %.2e mean %.2e | f16x2 max %.2e mean %.2e' % (
        mode, M, N, K, errs['f32'][0], errs['f32'][1], errs['bf16x3'][0], errs['bf16x3'][1], errs['f16x2'][0], errs['f16x2'][1]))
    for name in ('bf16x3', 'f16x2'):
        assert errs[name][0] <= 1.5 * errs['f32'][0] + 1e-9, name
        assert errs[name][1] <= 1.5 * errs['f32'][1] + 1e-10, name
        assert errs[name][0] < 2e-6, name


def _bounds_kw(a, b, ta, tb):
    """a_amax / b_amax of a product from its stored operands (include/ekaid_hip.h: the row maxima of A stored [M][K] or its column maxima
    stored [K][M]; the row maxima of B stored [N][K] or its column maxima stored [K][N])."""
    from ekaid_amd import ops
    ar, ac = ops.amax_rc(a, a.shape[0], a.shape[1], a.shape[1])
    br, bc = ops.amax_rc(b, b.shape[0], b.shape[1], b.shape[1])
    return dict(a_amax=ac if ta else ar, b_amax=br if tb else bc)


def _gemm_counted(a, b, C, M, N, K, lda, ldb, ta, tb, **kw):
    """ops.gemm + which arithmetic the library ran it on (launch counters)."""
    from ekaid_amd import hip, ops
    c0 = hip.launch_counts()
    ops.gemm(a, b, C, M, N, K, lda, ldb, N, ta, tb, **kw)
    c1 = hip.launch_counts()
    return {k: c1[k] - c0[k] for k in c0}


def test_amax_rc_row_and_column_maxima():
    """ekaid_amax_rc(_multi): exact row / column maxima of |x| for contiguous and row-strided matrices, narrow and wide, with zero rows,
    a NaN, ragged row counts; the multi-matrix launch; refusal of matrices that are not whole 16-byte vectors."""
    import ctypes
    from ekaid_amd import hip, ops
    g = torch.Generator().manual_seed(21)
    for rows, cols, ld in [(1, 4, 4), (5, 64, 64), (37, 300, 300), (129, 1024, 1024), (600, 2048, 2052), (4100, 36, 40), (33, 4096, 4096), (8200, 256, 256),
                           (18500, 1028, 1032), (70000, 8, 8), (2304, 1024, 1024), (65, 260, 260)]:
        buf = (torch.randn(rows, ld, generator=g) * torch.exp(2 * torch.randn(rows, 1, generator=g))).to(DEV)
        if rows > 3:
            buf[2] = 0
        x = buf[:, :cols]
        rm, cm = ops.amax_rc(buf, rows, cols, ld)
        torch.cuda.synchronize()
        assert torch.equal(rm, x.abs().amax(1)), (rows, cols, ld)
        assert torch.equal(cm, x.abs().amax(0)), (rows, cols, ld)
        r_only, none = ops.amax_rc(buf, rows, cols, ld, want_cols=False)
        assert none is None and torch.equal(r_only, rm)
    buf = torch.randn(40, 128, generator=g).to(DEV)
    buf[7, 5] = float('nan')
    rm, cm = ops.amax_rc(buf, 40, 128, 128)
    assert torch.isnan(rm[7]) and torch.isnan(cm[5]) and int(torch.isnan(rm).sum()) == 1 and int(torch.isnan(cm).sum()) == 1
    assert ops.amax_rc(torch.randn(8, 30, device=DEV), 8, 30, 30) == (None, None)            # not whole float4s: the caller stays on the exact split
    L = hip.lib()
    x = torch.randn(8, 30, device=DEV)
    out = torch.empty(8, device=DEV)
    assert L.ekaid_amax_rc(x.data_ptr(), 8, 30, 30, out.data_ptr(), None, 0, hip.stream()) == hip.ERR_UNSUPPORTED
    # the pool of column outputs: cleared once, slices handed out once each; an exhausted pool falls back to a memset per pass
    ops.bounds_pool_reset(torch.device(DEV, torch.cuda.current_device()))
    xs = [torch.randn(64, 4096, generator=g).to(DEV) for _ in range(70)]                    # 70 x (4096 columns + 64 rows) > 2^18 floats
    full, ops._BOUNDS_POOL_FLOATS = ops._BOUNDS_POOL_FLOATS, 1 << 18
    try:
        outs = [ops.amax_rc(t, 64, 4096, 4096) for t in xs]
    finally:
        ops._BOUNDS_POOL_FLOATS = full
    for t, (rm, cm) in zip(xs, outs):
        assert torch.equal(rm, t.abs().amax(1)) and torch.equal(cm, t.abs().amax(0))
    # raw entry point, outputs NOT zeroed by the caller (garbage in them): the library clears what it accumulates into
    L = hip.lib()
    for rows, cols in ((300, 1024), (300, 128)):
        t = torch.randn(rows, cols, generator=g).to(DEV)
        rm, cm = torch.full((rows,), 9e9, device=DEV), torch.full((cols,), 9e9, device=DEV)
        assert L.ekaid_amax_rc(t.data_ptr(), rows, cols, cols, rm.data_ptr(), cm.data_ptr(), 0, hip.stream()) == 0
        assert torch.equal(rm, t.abs().amax(1)) and torch.equal(cm, t.abs().amax(0))
    ops.bounds_pool_reset(torch.device(DEV, torch.cuda.current_device()))
    rm, cm = ops.amax_rc(xs[3], 64, 4096, 4096)
    assert torch.equal(cm, xs[3].abs().amax(0))
    ws = [torch.randn(r, c, generator=g).to(DEV) for r, c in ((64, 128), (1024, 1024), (3, 2048), (512, 36))]
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(ops.GEMM_F16X2)
    try:
        ops.weight_bounds_prepare(ws)
        for w in ws:
            r, c = ops.weight_bounds(w)
            assert torch.equal(r, w.abs().amax(1)) and torch.equal(c, w.abs().amax(0))
    finally:
        ops.weight_bounds_release()
        ops.set_gemm_mode(old)


def test_gemm_f16x2_is_exact_where_the_dropped_term_vanishes():
    """Integer operands: one side with at most 11 significant bits (its low plane is zero), the other with up to 22 (both planes),
    sums exact in f32 -> the two-term f16 split must give the integer result bit for bit in all three layouts (catches a wrong lane / k
    mapping of the f16 fragments, a wrong row / column scale in the staging or in the epilogue), also with rows / columns scaled by
    2^+-40 (exact powers of two: the per-row scales absorb them)."""
    from ekaid_amd import ops
    g = torch.Generator().manual_seed(6)
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(ops.GEMM_F16X2)
    try:
        for (M, N, K, amax, bmax) in [(128, 128, 32, 1 << 13, 30), (192, 320, 32, 30, 1 << 13), (76, 44, 24, 1 << 13, 30),
                                      (256, 192, 64, 1000, 120), (132, 72, 96, 60, 1 << 10)]:
            A = torch.randint(-amax, amax + 1, (M, K), generator=g).float()
            B = torch.randint(-bmax, bmax + 1, (K, N), generator=g).float()
            ra = torch.pow(2.0, torch.randint(-40, 41, (M, 1), generator=g).float())
            cb = torch.pow(2.0, torch.randint(-40, 41, (1, N), generator=g).float())
            want = (A.double() @ B.double())
            for scaled in (False, True):
                A2, B2 = (A * ra, B * cb) if scaled else (A, B)
                w2 = (want * ra.double() * cb.double()).float() if scaled else want.float()
                for mode in ('nt', 'nn', 'tn'):
                    if mode == 'nt':
                        a, b, ta, tb, lda, ldb = A2, B2.t().contiguous(), 0, 1, K, K
                    elif mode == 'nn':
                        a, b, ta, tb, lda, ldb = A2, B2, 0, 0, K, N
                    else:
                        a, b, ta, tb, lda, ldb = A2.t().contiguous(), B2, 1, 0, M, N
                    a, b = a.to(DEV), b.to(DEV)
                    C = torch.full((M, N), float('nan'), device=DEV)
                    ran = _gemm_counted(a, b, C, M, N, K, lda, ldb, ta, tb, **_bounds_kw(a, b, ta, tb))
                    assert ran['gemm_split2'] == 1, ran
                    assert torch.equal(C.cpu(), w2), (mode, scaled, M, N, K, float((C.cpu() - w2).abs().max()))
    finally:
        ops.set_gemm_mode(old)


def test_gemm_split_is_exact_on_bf16_representable_products():
    """Integer-valued operands whose products and sums are exact in f32: both arithmetics must be bit-identical
    to the integer result (catches any wrong lane/k mapping of the bf16 fragments and the transposed LDS reads)."""
    from ekaid_amd import ops
    g = torch.Generator().manual_seed(5)
    # (M, N, K, |A| bound, |B| bound): bound_A * bound_B * K <= 2^24 keeps every partial sum exact in f32;
    # 2^18-sized integers need all three bf16 planes of that operand
    for (M, N, K, amax, bmax) in [(128, 128, 32, 1 << 18, 2), (192, 320, 32, 2, 1 << 18), (77, 45, 24, 1 << 18, 2),
                                  (256, 192, 64, 4000, 60), (130, 70, 96, 50, 3000)]:
        A = torch.randint(-amax, amax + 1, (M, K), generator=g).float()
        B = torch.randint(-bmax, bmax + 1, (K, N), generator=g).float()
        want = (A.double() @ B.double()).float()
        for mode in ('nt', 'nn', 'tn'):
            if mode == 'nt':
                a, b, ta, tb, lda, ldb = A, B.t().contiguous(), 0, 1, K, K
            elif mode == 'nn':
                a, b, ta, tb, lda, ldb = A, B, 0, 0, K, N
            else:
                a, b, ta, tb, lda, ldb = A.t().contiguous(), B, 1, 0, M, N
            old = ops.get_gemm_mode()
            try:
                ops.set_gemm_mode(ops.GEMM_BF16_SPLIT)
                C = torch.empty(M, N, device=DEV)
                ops.gemm(a.to(DEV), b.to(DEV), C, M, N, K, lda, ldb, N, ta, tb)
            finally:
                ops.set_gemm_mode(old)
            assert torch.equal(C.cpu(), want), (mode, M, N, K, float((C.cpu() - want).abs().max()))


def test_gemm_randomised_layouts_alignments_and_epilogues(gemm_arith):
    """60 seeded random problems per arithmetic: ragged and tile-aligned sizes (the aligned ones take the buffer-load /
    software-pipelined path), leading dimensions with and without 16-byte alignment, misaligned base offsets,
    accumulate / bias / activation / scalar combinations, all three operand layouts -- against float64."""
    from ekaid_amd import ops
    from ekaid_amd.hip import ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID
    rng = np.random.RandomState(2024)
    acts = {ACT_NONE: lambda t: t, ACT_RELU: torch.relu, ACT_TANH: torch.tanh, ACT_SIGMOID: torch.sigmoid}
    for case in range(60):
        aligned = case % 3 == 0
        if aligned:
            M, N, K = (int(rng.choice([128, 256, 384, 512, 640])), int(rng.choice([128, 256, 384])), int(rng.choice([32, 64, 96, 256, 512])))
        else:
            M, N, K = int(rng.randint(1, 700)), int(rng.randint(1, 400)), int(rng.randint(1, 600))
        mode = ['nt', 'nn', 'tn'][case % 3 if not aligned else (case // 3) % 3]
        pad = 0 if aligned else int(rng.choice([0, 1, 3, 4]))
        off = 0 if aligned else int(rng.choice([0, 1, 4]))
        A = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
        B = torch.from_numpy(rng.standard_normal((K, N)).astype(np.float32))

        def stage(mat):                       # row-major storage with a padded leading dim and a base offset
            r, c = mat.shape
            buf = torch.full((off + r * (c + pad),), float('nan'))
            buf[off:].view(r, c + pad)[:, :c] = mat
            return buf.to(DEV), c + pad
        if mode == 'nt':
            (a, lda), (b, ldb), ta, tb = stage(A), stage(B.t().contiguous()), 0, 1
        elif mode == 'nn':
            (a, lda), (b, ldb), ta, tb = stage(A), stage(B), 0, 0
        else:
            (a, lda), (b, ldb), ta, tb = stage(A.t().contiguous()), stage(B), 1, 0
        acc, with_bias = bool(rng.randint(2)), bool(rng.randint(2))
        act = int(rng.choice([ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID]))
        alpha = float(rng.choice([1.0, 0.5, -2.0]))
        ldc = N + (0 if aligned else int(rng.choice([0, 2, 4])))
        C0 = torch.from_numpy(rng.standard_normal((M, ldc)).astype(np.float32))
        bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
        C = C0.clone().to(DEV)
        bias_dev = bias.to(DEV)
        ops.gemm(a, b, C, M, N, K, lda, ldb, ldc, ta, tb, alpha=alpha, accumulate=acc, bias=bias_dev if with_bias else None,
                 act=act, a_off=off, b_off=off)
        want = alpha * (A.double() @ B.double())
        if acc:
            want = want + C0[:, :N].double()
        if with_bias:
            want = want + bias.double()
        want = acts[act](want)
        what = 'case %d %s %dx%dx%d pad %d off %d acc %d bias %d act %d alpha %g' % (case, mode, M, N, K, pad, off, acc, with_bias, act, alpha)
        scale = float((A.double().abs() @ B.double().abs()).max()) * abs(alpha) + 1.0
        err = float((C[:, :N].double().cpu() - want).abs().max())
        assert err <= 4e-6 * scale, '%s: abs err %.3e vs scale %.3e' % (what, err, scale)
        if ldc > N:
            assert torch.equal(C[:, N:].cpu(), C0[:, N:]), what + ': wrote outside the N columns'


def test_gemm_epilogue_and_strides():
    from ekaid_amd import ops
    from ekaid_amd.hip import ACT_RELU, ACT_SIGMOID, ACT_TANH
    M, N, K = 200, 96, 72
    A, W = rnd(M, K + 8, seed=3), rnd(N, K + 4, seed=4)     # padded leading dims
    bias, R, rowmask = rnd(N, seed=5), rnd(M // 4, N, seed=6), (torch.rand(M) > 0.3).float()
    alpha_dev = torch.tensor(0.37)
    C0 = rnd(M, N + 5, seed=7)
    for act, fn in ((ACT_RELU, torch.relu), (ACT_TANH, torch.tanh), (ACT_SIGMOID, torch.sigmoid)):
        C = C0.clone().to(DEV)
        ops.gemm(A.to(DEV), W.to(DEV), C, M, N, K, K + 8, K + 4, N + 5, 0, 1, alpha=2.0, alpha_dev=alpha_dev.to(DEV),
                 accumulate=True, bias=bias.to(DEV), R=R.to(DEV), ldr=N, rdiv=4, rowmask=rowmask.to(DEV), act=act)
        pre = 2.0 * 0.37 * (A[:, :K].double() @ W[:, :K].double().t()) + C0[:, :N].double() + bias.double() + \
            rowmask.double()[:, None] * R.double().repeat_interleave(4, 0)
        check(C[:, :N], fn(pre), 3e-6, 'epilogue act=%d' % act)
        assert torch.equal(C[:, N:].cpu(), C0[:, N:]), 'gemm wrote outside its ldc window'


def test_gemm_batched_two_level():
    from ekaid_amd import ops
    R, H, N, M, dh = 3, 4, 9, 7, 16
    A = H * dh
    q, k = rnd(R, N, A, seed=8), rnd(R, M, A, seed=9)
    aff = torch.empty(R, N, H, M, device=DEV)
    ops.gemm(q.to(DEV), k.to(DEV), aff, N, M, dh, A, A, H * M, 0, 1, alpha=0.25, batch=(R, H),
             strides=(N * A, dh, M * A, dh, N * H * M, M))
    want = 0.25 * torch.einsum('rnhd,rmhd->rnhm', q.view(R, N, H, dh).double(), k.view(R, M, H, dh).double())
    check(aff, want, 2e-6, 'batched QK^T')


def test_gemm_splitk_path_is_exercised():
    from ekaid_amd import hip, ops
    import ctypes
    d = hip.GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch2 = 16, 64, 4096, 1, 1
    assert hip.lib().ekaid_gemm_workspace_bytes(ctypes.byref(d)) > 0
    A, W = rnd(16, 4096, seed=10), rnd(64, 4096, seed=11)
    C = torch.empty(16, 64, device=DEV)
    ops.gemm(A.to(DEV), W.to(DEV), C, 16, 64, 4096, 4096, 4096, 64, 0, 1, bias=rnd(64, seed=12).to(DEV))
    check(C, A.double() @ W.double().t() + rnd(64, seed=12).double(), 4e-6, 'split-K + bias')


@pytest.mark.parametrize('M,N,K', [(1, 11, 70001), (4, 64, 100000), (3, 5, 8192), (16, 16, 9000), (1, 1, 33000), (2, 128, 20000)])
def test_gemm_tall_skinny_weight_gradient(M, N, K, gemm_arith):
    """C[M][N] = A^T . B with at most 256 outputs and a very long contraction (label-bias / pair-geometry weight gradients,
    graph_att.py:92, graph_att_layer.py:121): dedicated reduction kernels, deterministic, with alpha and accumulate."""
    from ekaid_amd import ops
    A, B = rnd(K, M, seed=41).to(DEV), rnd(K, N, seed=42).to(DEV)
    C0 = rnd(M, N, seed=43).to(DEV)
    want = A.double().t() @ B.double()
    C = torch.empty(M, N, device=DEV)
    ops.gemm(A, B, C, M, N, K, M, N, N, 1, 0)
    check(C, want.cpu(), 3e-6, 'tall-skinny wgrad')
    C2 = torch.empty(M, N, device=DEV)
    ops.gemm(A, B, C2, M, N, K, M, N, N, 1, 0)
    assert torch.equal(C, C2)                                   # fixed summation order
    Cacc = C0.clone()
    ops.gemm(A, B, Cacc, M, N, K, M, N, N, 1, 0, alpha=0.5, accumulate=True)
    check(Cacc, (0.5 * want + C0.double()).cpu(), 3e-6, 'tall-skinny wgrad, alpha + accumulate')
    # a column block of wider operands (leading dimensions larger than the extents)
    Aw, Bw = rnd(K, M + 3, seed=44).to(DEV), rnd(K, N + 5, seed=45).to(DEV)
    Cs = torch.empty(M, N, device=DEV)
    ops.gemm(Aw, Bw, Cs, M, N, K, M + 3, N + 5, N, 1, 0, a_off=2, b_off=1)
    check(Cs, (Aw[:, 2:2 + M].double().t() @ Bw[:, 1:1 + N].double()).cpu(), 3e-6, 'tall-skinny wgrad on views')


@pytest.mark.parametrize('M,N,K', [(1024, 1024, 18432), (1024, 4096, 2304), (2048, 512, 4096), (512, 1024, 300), (148, 1024, 4096),
                                   (3, 1024, 4096), (130, 70, 1000), (64, 64, 64), (4096, 600, 1000), (128, 1024, 9216)])
def test_gemm_weight_gradient_carries_the_bias_gradient(M, N, K, gemm_arith):
    """a_rowsum: dW = dY^T X with db[m] += sum_rows dY[row][m] from the same launch (A-operand staging of the split kernels; a
    separate column-sum launch under the f32 arithmetic and for tall-skinny shapes).  ref: autograd of every biased nn.Linear."""
    from ekaid_amd import ops
    dY, X = rnd(K, M, seed=51).to(DEV), rnd(K, N, seed=52).to(DEV)
    db0 = rnd(M, seed=53).to(DEV)
    want_w = (dY.double().t() @ X.double()).cpu()
    want_b = (db0.double() + dY.double().sum(0)).cpu()
    dW, db = torch.full((M, N), float('nan'), device=DEV), db0.clone()
    ops.gemm(dY, X, dW, M, N, K, M, N, N, 1, 0, a_rowsum=db)
    check(dW, want_w, 2e-6 * max(1.0, math.sqrt(K) / 4), 'wgrad with bias gradient')
    check(db, want_b, 2e-6 * max(1.0, math.sqrt(K) / 4), 'bias gradient from the wgrad product')
    dW2, db2 = torch.empty(M, N, device=DEV), db0.clone()
    ops.gemm(dY, X, dW2, M, N, K, M, N, N, 1, 0, a_rowsum=db2)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)        # fixed summation order
    dW3 = torch.empty(M, N, device=DEV)
    ops.gemm(dY, X, dW3, M, N, K, M, N, N, 1, 0)
    assert torch.equal(dW, dW3)                                 # the product itself is untouched
    # a column block of a wider dY (the decoder's DCAT layout) into a column block of the gradient
    if M >= 8:
        dYw = rnd(K, M + 24, seed=54).to(DEV)
        dbw = torch.zeros(M, device=DEV)
        dWw = torch.zeros(M, N + 8, device=DEV)
        ops.gemm(dYw, X, dWw, M, N, K, M + 24, N, N + 8, 1, 0, a_off=8, c_off=4, a_rowsum=dbw)
        check(dbw, dYw[:, 8:8 + M].double().sum(0).cpu(), 2e-6 * max(1.0, math.sqrt(K) / 4), 'bias gradient of a column block')
        check(dWw[:, 4:4 + N], (dYw[:, 8:8 + M].double().t() @ X.double()).cpu(), 2e-6 * max(1.0, math.sqrt(K) / 4), 'wgrad block')


def test_gemm_rejects_bad_args():
    from ekaid_amd import hip, ops
    A = torch.zeros(4, 4, device=DEV)
    with pytest.raises(hip.EkaidError):
        ops.gemm(A, A, A, 4, 4, 4, 4, 4, 4, 1, 1)          # transA && transB unsupported


# --------------------------------------------------------------------------- linear / weight norm
def _grads(outs, ins):
    loss = sum((o * rnd(*o.shape, seed=40 + i).to(o.device)).sum() for i, o in enumerate(outs))
    return torch.autograd.grad(loss, ins, allow_unused=True)


@pytest.mark.parametrize('act', ['none', 'relu', 'tanh', 'sigmoid'])
def test_linear_multi_input_fwd_bwd(act):
    from ekaid_amd import ops
    from ekaid_amd.hip import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH
    code = dict(none=ACT_NONE, relu=ACT_RELU, tanh=ACT_TANH, sigmoid=ACT_SIGMOID)[act]
    fn = dict(none=lambda t: t, relu=torch.relu, tanh=torch.tanh, sigmoid=torch.sigmoid)[act]
    x1, x2, x3 = rnd(5, 7, 24, seed=13), rnd(5, 7, 12, seed=14), rnd(5, 7, 300, seed=15, scale=0.2)
    W, b, v_g = rnd(40, 336, seed=16, scale=0.1), rnd(40, seed=17), torch.tensor(1.7)
    cpu = [t.clone().double().requires_grad_(True) for t in (x1, x2, x3, W, b, v_g)]
    w_eff = cpu[3] * (cpu[5] / cpu[3].norm())
    y_ref = fn(F.linear(torch.cat(cpu[:3], -1), w_eff, cpu[4]))
    gpu = [t.clone().to(DEV).requires_grad_(True) for t in (x1, x2, x3, W, b, v_g)]
    y = ops.wn_linear(tuple(gpu[:3]), gpu[3], gpu[5], gpu[4], code)
    check(y, y_ref, 3e-6, 'wn_linear fwd')
    g_ref = _grads([y_ref], cpu)
    g = _grads([y], gpu)
    for name, a, r in zip(('x1', 'x2', 'x3', 'v', 'b', 'g'), g, g_ref):
        check(a, r.float(), 2e-5, 'wn_linear grad ' + name)


def test_linear_strided_rows_and_rowstack():
    from ekaid_amd import ops
    big = rnd(6, 5, 48, seed=18)
    x = big[:, 2]                                   # (6, 48) with row stride 240
    W, b = rnd(10, 48, seed=19), rnd(10, seed=20)
    xg = big.to(DEV)[:, 2]
    y = ops.linear(xg, W.to(DEV), b.to(DEV))
    check(y, F.linear(x.double(), W.double(), b.double()), 3e-6, 'strided linear')
    a1, a2 = rnd(3, 4, 48, seed=21), rnd(3, 4, 48, seed=22)
    Wg = W.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True)
    y2 = ops.linear_rowstack((a1.to(DEV), a2.to(DEV)), Wg, bg)
    ref_in = torch.cat([a1.reshape(-1, 48), a2.reshape(-1, 48)]).double()
    Wc, bc = W.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.linear(ref_in, Wc, bc)
    check(y2, yr, 3e-6, 'rowstack fwd')
    gg = _grads([y2], [Wg, bg])
    gr = _grads([yr], [Wc, bc])
    check(gg[0], gr[0].float(), 2e-5, 'rowstack dW')
    check(gg[1], gr[1].float(), 2e-5, 'rowstack db')


# --------------------------------------------------------------------------- point-wise family
def test_dropout_statistics_and_backward_mask():
    from ekaid_amd import ops
    x = torch.ones(1 << 20, device=DEV, requires_grad=True)
    y = ops.DropoutFn.apply(x, 0.2, 1234)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.8) < 5e-3, kept
    assert torch.allclose(y[y != 0], torch.tensor(1.25, device=DEV))
    (g,) = torch.autograd.grad(y.sum(), x)
    assert torch.equal(g, y.detach()), 'backward must reuse the forward mask'
    y2 = ops.DropoutFn.apply(x, 0.2, 1235)
    assert (y2 != y).float().mean().item() > 0.2, 'different seeds must give different masks'
    assert ops.dropout(x, 0.5, training=False) is x


def test_residual_relu2_gate_fuse_mix3_pool():
    from ekaid_amd import ops
    v, o = rnd(6, 5, 32, seed=23), rnd(6, 5, 32, seed=24)
    cv, co = v.double().requires_grad_(True), o.double().requires_grad_(True)
    ref = cv + torch.relu(2 * co)
    gv, go = v.to(DEV).requires_grad_(True), o.to(DEV).requires_grad_(True)
    out = ops.residual_relu2(gv, go, 0.2, training=False)
    check(out, ref, 1e-6, 'residual_relu2')
    for a, r in zip(_grads([out], [gv, go]), _grads([ref], [cv, co])):
        check(a, r.float(), 1e-6, 'residual_relu2 grad')
    # with dropout: output = v + relu(2 o m/(1-p)); gradient consistent with the same mask
    out_d = ops.ResidualRelu2Fn.apply(gv, go, 0.3, 77)
    mask = ((out_d - gv).detach() != 0) | (go.detach() <= 0)
    (gd,) = torch.autograd.grad(out_d.sum(), go)
    assert torch.all((gd != 0) == ((out_d - gv).detach() != 0))
    assert mask.float().mean() > 0.5
    # gate fuse with the shared (difference) terms broadcast over two images
    c2, g2, c1, g1 = rnd(4, 3, 16, seed=25), rnd(4, 3, 16, seed=26), rnd(2, 3, 16, seed=27), rnd(2, 3, 16, seed=28)
    cc = [t.double().requires_grad_(True) for t in (c2, g2, c1, g1)]
    ref = torch.sigmoid(cc[1] + cc[3].repeat(2, 1, 1)) * torch.tanh(cc[0] + cc[2].repeat(2, 1, 1))
    gg = [t.to(DEV).requires_grad_(True) for t in (c2, g2, c1, g1)]
    out = ops.gate_fuse(gg[0], gg[1], gg[2], gg[3], 0.5, training=False)
    check(out, ref, 2e-6, 'gate_fuse')
    for a, r in zip(_grads([out], gg), _grads([ref], cc)):
        check(a, r.float(), 3e-6, 'gate_fuse grad')
    # mix3
    w, a, b, c = torch.softmax(rnd(7, 3, seed=29), 1), rnd(7, 20, seed=30), rnd(7, 20, seed=31), rnd(7, 20, seed=32)
    cc = [t.double().requires_grad_(True) for t in (w, a, b, c)]
    ref = cc[0][:, 0:1] * cc[1] + cc[0][:, 1:2] * cc[2] + cc[0][:, 2:3] * cc[3]
    gg = [t.to(DEV).requires_grad_(True) for t in (w, a, b, c)]
    out = ops.Mix3Fn.apply(*gg)
    check(out, ref, 1e-6, 'mix3')
    for x, r in zip(_grads([out], gg), _grads([ref], cc)):
        check(x, r.float(), 2e-6, 'mix3 grad')
    # weighted pool
    w, x = torch.rand(5, 9), rnd(5, 9, 33, seed=33)
    cw, cx = w.double().requires_grad_(True), x.double().requires_grad_(True)
    ref = (cw.unsqueeze(-1) * cx).sum(1)
    gw, gx = w.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
    out = ops.weighted_pool(gw, gx)
    check(out, ref, 2e-6, 'weighted_pool')
    for a, r in zip(_grads([out], [gw, gx]), _grads([ref], [cw, cx])):
        check(a, r.float(), 2e-6, 'weighted_pool grad')


def test_rowsum_expand_rows_axpby_mul():
    from ekaid_amd import ops
    x = rnd(4, 6, 10, seed=34)
    x[1, 2] = 0
    check(ops.rowsum(x.to(DEV)), x.sum(-1), 2e-6, 'rowsum')
    m = ops.rowsum(x.to(DEV), as_mask=True)
    assert torch.equal(m.cpu(), (x.sum(-1) != 0).float())
    q = rnd(2, 12, seed=35)                                  # 2 question rows shared by 4 slices (2 images)
    mask = (torch.rand(24) > 0.2).float()
    cq = q.double().requires_grad_(True)
    ref = mask.double()[:, None] * cq.repeat(2, 1).repeat_interleave(6, 0)
    gq = q.to(DEV).requires_grad_(True)
    out = ops.expand_rows(gq, mask.to(DEV), 6)
    check(out, ref, 1e-7, 'expand_rows')
    check(_grads([out], [gq])[0], _grads([ref], [cq])[0].float(), 2e-6, 'expand_rows grad')
    a, b = rnd(33, seed=36), rnd(33, seed=37)
    ga, gb = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    check(ops.sub(ga, gb), a - b, 1e-7, 'sub')
    check(ops.mul(ga, gb), a * b, 1e-7, 'mul')
    check(ops.scale(ga, 0.5), a * 0.5, 1e-7, 'scale')
    gr = _grads([ops.mul(ops.sub(ga, gb), ga)], [ga, gb])
    ca, cb = a.double().requires_grad_(True), b.double().requires_grad_(True)
    rr = _grads([(ca - cb) * ca], [ca, cb])
    check(gr[0], rr[0].float(), 2e-6, 'mul/sub grad a')
    check(gr[1], rr[1].float(), 2e-6, 'mul/sub grad b')


def test_softmax_logsoftmax_embedding_qatt():
    from ekaid_amd import ops
    for cols in (3, 16, 148, 200):
        x = rnd(11, cols, seed=38, scale=3)
        for log_mode, fn in ((False, torch.softmax), (True, torch.log_softmax)):
            cx = x.double().requires_grad_(True)
            ref = fn(cx, -1)
            gx = x.to(DEV).requires_grad_(True)
            out = ops.SoftmaxFn.apply(gx, log_mode)
            check(out, ref, 2e-6, 'softmax log=%s cols=%d' % (log_mode, cols))
            check(_grads([out], [gx])[0], _grads([ref], [cx])[0].float(), 5e-6, 'softmax grad')
    tok = torch.randint(0, 31, (4, 5))
    t1, t2 = rnd(31, 8, seed=39), rnd(31, 6, seed=40)
    ct1 = t1.double().requires_grad_(True)
    ref = torch.cat([F.embedding(tok, ct1, padding_idx=30), F.embedding(tok, t2.double())], -1)
    g1 = t1.to(DEV).requires_grad_(True)
    out = ops.embedding(tok.to(DEV), [g1, t2.to(DEV)], relu=False, padding_idx=30)
    check(out, ref, 0, 'embedding')
    check(_grads([out], [g1])[0], _grads([ref], [ct1])[0].float(), 2e-6, 'embedding grad (padding row excluded)')
    ref = torch.relu(F.embedding(tok, ct1))
    out = ops.embedding(tok.to(DEV), [g1], relu=True)
    check(out, ref, 0, 'embedding+relu')
    check(_grads([out], [g1])[0], _grads([ref], [ct1])[0].float(), 2e-6, 'embedding+relu grad')
    # quirk Q-C weights
    atten = rnd(6, 5, seed=41)
    ca = atten.double().requires_grad_(True)
    ref = torch.softmax(ca.t(), dim=1).contiguous().view(-1, 5)
    ga = atten.to(DEV).requires_grad_(True)
    out = ops.QAttWeightsFn.apply(ga)
    check(out, ref, 2e-6, 'qatt weights')
    check(_grads([out], [ga])[0], _grads([ref], [ca])[0].float(), 5e-6, 'qatt weights grad')


def test_losses_layernorm_adam_process_matrix_geometry():
    from ekaid_amd import hip, ops
    from ekaid_amd.hip import check as hcheck, ptr, stream
    from ekaid_amd.utils.mimic_utils import process_matrix_labels
    from oracle import ekaid_oracle as O
    B, T, V = 3, 6, 31
    logp = torch.log_softmax(rnd(B, T, V, seed=42), -1)
    labels = torch.randint(1, V, (B, T + 1))
    masks = torch.zeros(B, T + 1)
    masks[:, :4] = 1
    masks[0, :6] = 1
    cl = logp.double().requires_grad_(True)
    ref = O.lm_criterion(cl, labels[:, 1:], masks[:, 1:].double())
    gl = logp.to(DEV).requires_grad_(True)
    out = ops.lm_loss(gl, labels.to(DEV)[:, 1:], masks.to(DEV)[:, 1:])
    check(out, ref, 2e-6, 'lm_loss (strided target/mask views)')
    att_b, att_a = torch.rand(B, 1, 8), torch.rand(B, 1, 8)
    cb, ca = att_b.double().requires_grad_(True), att_a.double().requires_grad_(True)
    ref_t = ref + 2.5e-3 * (cb.sum() + ca.sum()) / (2 * B)
    gb, ga = att_b.to(DEV).requires_grad_(True), att_a.to(DEV).requires_grad_(True)
    tot = ops.total_loss(out, gb, ga)
    check(tot, ref_t, 2e-6, 'total_loss')
    gg = torch.autograd.grad(tot, [gl, gb, ga])
    gr = torch.autograd.grad(ref_t, [cl, cb, ca])
    for a, r, n in zip(gg, gr, ('dlogp', 'datt_b', 'datt_a')):
        check(a, r.float(), 2e-6, 'loss grad ' + n)
    # layernorm
    x, gam, bet = rnd(7, 5, 64, seed=43), rnd(64, seed=44), rnd(64, seed=45)
    cc = [t.double().requires_grad_(True) for t in (x, gam, bet)]
    ref = F.layer_norm(cc[0], (64,), cc[1], cc[2], 1e-6)
    gg = [t.to(DEV).requires_grad_(True) for t in (x, gam, bet)]
    out = ops.layer_norm(gg[0], gg[1], gg[2], 1e-6)
    check(out, ref, 3e-6, 'layernorm')
    for a, r in zip(_grads([out], gg), _grads([ref], cc)):
        check(a, r.float(), 2e-5, 'layernorm grad')
    # adam vs torch.optim.Adam
    p0, grads = rnd(1000, seed=46), [rnd(1000, seed=47), rnd(1000, seed=48)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], 1e-4, (0.9, 0.999), 1e-8)
    pg, m, v = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step, g in enumerate(grads, 1):
        pr.grad = g.clone()
        opt.step()
        hcheck(hip.lib().ekaid_adam_step(ptr(pg), ptr((2 * g).to(DEV)), ptr(m), ptr(v), 1000, 1e-4, 0.9, 0.999, 1e-8, step,
                                         0.5, stream()), 'adam')
    check(pg - p0.to(DEV), pr.detach() - p0, 1e-5, 'adam delta after 2 steps (grad_scale=0.5)')
    # process_matrix (float64 and int64 inputs) and geometry embedding
    lab = torch.randint(0, 12, (2, 100, 100))
    want = O.process_matrix(lab.double(), 8, 11)
    assert torch.equal(process_matrix_labels(lab.double().to(DEV), 8, 11).cpu(), want)
    assert torch.equal(process_matrix_labels(lab.to(DEV), 8, 11).cpu(), want)
    from ekaid_amd.synthetic import make_batch
    bb = make_batch(B=3, N=9, D=8, Lq=5, T=3, seq_length=4, V=31)['d_bb']
    want = O.position_embedding(O.position_matrix(bb, 52), 64).float()
    feat_range = torch.arange(0, 64 / 8)
    dim_mat = torch.pow(torch.ones((1,)) * 1000, (8. / 64) * feat_range)
    got = ops.geom_embed(bb.to(DEV), 9, 64, dim_mat.to(DEV))
    assert float((got.cpu() - want).abs().max()) <= 2e-6, 'geometry embedding'


# --------------------------------------------------------------------------- recurrent cells
def test_gru_sequence_and_lstm_cell():
    from ekaid_amd import ops
    from oracle import ekaid_oracle as O
    B, L, E, H = 5, 6, 20, 24
    x = rnd(B, L, E, seed=49)
    names = ['q_emb.rnn.weight_ih_l0', 'q_emb.rnn.weight_hh_l0', 'q_emb.rnn.bias_ih_l0', 'q_emb.rnn.bias_hh_l0']
    vals = [rnd(3 * H, E, seed=50, scale=0.3), rnd(3 * H, H, seed=51, scale=0.3), rnd(3 * H, seed=52), rnd(3 * H, seed=53)]
    P = {n: v.double().requires_grad_(True) for n, v in zip(names, vals)}
    cx = x.double().requires_grad_(True)
    ref = O.gru_forward_all(P, cx)
    # cross-check the oracle's GRU equations against nn.GRU once more, here, in float64
    gru = torch.nn.GRU(E, H, 1, batch_first=True).double()
    with torch.no_grad():
        for n, v in zip(('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0'), vals):
            getattr(gru, n).copy_(v.double())
    assert float((gru(x.double())[0] - ref).abs().max()) < 1e-12
    gp = [v.to(DEV).requires_grad_(True) for v in vals]
    gx = x.to(DEV).requires_grad_(True)
    out = ops.gru_sequence(gx, *gp)
    check(out, ref, 3e-6, 'gru fwd')
    gg = _grads([out], [gx] + gp)
    gr = _grads([ref], [cx] + [P[n] for n in names])
    for n, a, r in zip(['x'] + names, gg, gr):
        check(a, r.float(), 3e-5, 'gru grad ' + n)
    # LSTM cell
    Hh = 16
    gxs, ghs, c0 = rnd(B, 4 * Hh, seed=54), rnd(B, 4 * Hh, seed=55), rnd(B, Hh, seed=56)
    cc = [t.double().requires_grad_(True) for t in (gxs, ghs, c0)]
    gates = cc[0] + cc[1]
    i, f, g, o = (torch.sigmoid(gates[:, :Hh]), torch.sigmoid(gates[:, Hh:2 * Hh]), torch.tanh(gates[:, 2 * Hh:3 * Hh]),
                  torch.sigmoid(gates[:, 3 * Hh:]))
    c2r = f * cc[2] + i * g
    h2r = o * torch.tanh(c2r)
    gg_in = [t.to(DEV).requires_grad_(True) for t in (gxs, ghs, c0)]
    h2, c2 = ops.LSTMCellFn.apply(*gg_in)
    check(h2, h2r, 2e-6, 'lstm h')
    check(c2, c2r, 2e-6, 'lstm c')
    for a, r in zip(_grads([h2, c2], gg_in), _grads([h2r, c2r], cc)):
        check(a, r.float(), 5e-6, 'lstm grad')


# --------------------------------------------------------------------------- graph attention core
def _gat_reference(q, k, S, cond, lbias, geo, H, direction):
    R, N, A = q.shape
    M = k.shape[1]
    dh = A // H
    c = cond.transpose(1, 2) if direction else cond
    lb = lbias.transpose(1, 2) if direction else lbias
    aff = torch.einsum('rnhd,rmhd->rnhm', q.view(R, N, H, dh), k.view(R, M, H, dh)) / math.sqrt(dh)
    if geo is not None:
        aff = aff + torch.log(torch.clamp_min(geo, 1e-6)).permute(0, 1, 3, 2)
    # the mask constant and the bias add are float32 operations in the reference: -9e15 + bias == -9e15
    # there (uniform row for an isolated node), which a float64 add would not reproduce
    aff32 = aff.float()
    aff = (torch.where(c[:, :, None, :M] > 0, aff32, torch.full_like(aff32, -9e15)) + lb[:, :, None, :M].float()).double()
    P = torch.softmax(aff, -1)
    out = torch.einsum('rnhm,rma->rnha', P, S[:, :M]).reshape(R, N, H * A)
    return out, P


@pytest.mark.parametrize('N,M,A,H', [(8, 8, 64, 4), (36, 36, 128, 4), (52, 52, 64, 4), (9, 7, 32, 2), (36, 36, 1024, 4),
                                     (52, 52, 1024, 4), (36, 30, 256, 4), (64, 64, 128, 4), (13, 13, 64, 2), (40, 33, 192, 3),
                                     (10, 10, 48, 2), (36, 36, 40, 2)])     # (the last two: width not a multiple of 32 -> vector-ALU aggregation)
@pytest.mark.parametrize('direction', [0, 1])
@pytest.mark.parametrize('with_geo', [False, True])
def test_gat_attention_fused_and_composed(N, M, A, H, direction, with_geo):
    from ekaid_amd import ops
    R = 3
    sc = 1.0 / math.sqrt(A / H) ** 0.5
    q, k, S = rnd(R, N, A, seed=57, scale=sc), rnd(R, M, A, seed=58, scale=sc), rnd(R, N, A, seed=59)
    cond = (torch.rand(R, N, N) > 0.4).float()
    cond[0, 2, :] = 0
    cond[0, :, 2] = 0                                  # an isolated node in both orientations -> uniform row
    lbias = rnd(R, N, N, seed=60, scale=0.3)
    geo = torch.relu(rnd(R, N, M, H, seed=61)) if with_geo else None
    cpu = [t.double().requires_grad_(True) if t is not None else None for t in (q, k, S, lbias, geo)]
    out_r, P_r = _gat_reference(cpu[0], cpu[1], cpu[2], cond.double(), cpu[3], cpu[4], H, direction)
    assert torch.allclose(P_r[0, 2], torch.full_like(P_r[0, 2], 1.0 / M))
    gr = _grads([out_r], [t for t in cpu if t is not None])
    for impl in ('fused', 'gemm'):
        gpu = [t.to(DEV).requires_grad_(True) if t is not None else None for t in (q, k, S, lbias, geo)]
        out, P = ops.gat_attention(gpu[0], gpu[1], gpu[2], cond.to(DEV), gpu[3], gpu[4], H, direction, impl)
        check(P, P_r, 5e-6, 'P ' + impl)
        check(out, out_r, 5e-6, 'outT ' + impl)
        ins_g = [t for t in gpu if t is not None]
        gg = _grads([out], ins_g)
        for n, a, r in zip(('dq', 'dk', 'dS', 'dlbias', 'dgeo'), gg, gr):
            check(a, r.float(), 5e-5, '%s %s' % (n, impl))


@pytest.mark.parametrize('N,M,A,H,core', [(36, 36, 1024, 4, True), (52, 52, 1024, 4, False), (36, 20, 256, 4, True), (6, 6, 64, 16, True), (9, 7, 32, 2, True),
                                          (36, 36, 96, 4, True), (48, 48, 192, 3, True), (64, 64, 128, 4, False), (36, 36, 48, 2, False), (20, 20, 64, 16, False)])
@pytest.mark.parametrize('direction', [0, 1])
def test_gat_backward_in_one_launch_equals_the_composed_backward(N, M, A, H, core, direction):
    """ekaid_gat_bwd_core (d aggregation + d edge softmax, doutT read once, f32 matrix pipe) against the composition it replaces (two
    batched products on the exact split around ekaid_masked_softmax_bwd), with a gradient arriving on the returned softmax as well and
    with / without the geometry term; outside the kernel's envelope (more than 144 (node, head) rows, width not a multiple of 32) the
    composition runs -- asserted from the library's launch counter.  (The production routing also asks for >= 256 slices: lifted here.)"""
    from ekaid_amd import hip, ops
    R = 5
    sc = 1.0 / math.sqrt(A / H) ** 0.5
    cond = (torch.rand(R, N, N, generator=torch.Generator().manual_seed(5)) > 0.4).float().to(DEV)
    w = rnd(R, N, H * A, seed=71).to(DEV)
    wP = rnd(R, N, H, M, seed=72).to(DEV)
    for with_geo in (False, True):
        res = {}
        for use_core in (True, False):
            ins = [rnd(R, N, A, seed=57, scale=sc), rnd(R, M, A, seed=58, scale=sc), rnd(R, N, A, seed=59), rnd(R, N, N, seed=60, scale=0.3)]
            if with_geo:
                ins.append(torch.relu(rnd(R, N, M, H, seed=61)))
            ins = [t.to(DEV).requires_grad_(True) for t in ins]
            old = (ops._GAT_BWD_CORE, ops._GAT_BWD_CORE_MIN_SLICES)
            ops._GAT_BWD_CORE, ops._GAT_BWD_CORE_MIN_SLICES = use_core, 0
            try:
                c0 = hip.launch_counts()['gat_bwd_core']
                out, P = ops.gat_attention(ins[0], ins[1], ins[2], cond, ins[3], ins[4] if with_geo else None, H, direction)
                gs = torch.autograd.grad([out, P], ins, [w, wP])
                torch.cuda.synchronize()
                ran = hip.launch_counts()['gat_bwd_core'] - c0
            finally:
                ops._GAT_BWD_CORE, ops._GAT_BWD_CORE_MIN_SLICES = old
            assert ran == (1 if (use_core and core) else 0), (ran, use_core, core)
            res[use_core] = gs
        for name, a, b in zip(('dq', 'dk', 'dS', 'dlbias', 'dgeo'), res[True], res[False]):
            check(a, b.cpu(), 1e-5, '%s (geo %s)' % (name, with_geo))


@pytest.mark.parametrize('R', [1, 5, 200, 400])
@pytest.mark.parametrize('N,M,A,H', [(9, 7, 64, 2), (36, 36, 128, 4), (3, 3, 32, 1)])
def test_gat_fused_forward_query_node_split(R, N, M, A, H):
    """The fused forward splits the query nodes of a slice over 4 / 2 / 1 workgroups depending on the number of slices
    (R < 192 / < 384 / more): every split equals the composition of batched GEMMs + stand-alone softmax, for node counts
    that do not divide evenly (9 = 3+3+3+0, 3 = 1+1+1+0) and with both edge orientations."""
    from ekaid_amd import ops
    sc = 1.0 / math.sqrt(A / H) ** 0.5
    q, k, S = rnd(R, N, A, seed=157, scale=sc).to(DEV), rnd(R, M, A, seed=158, scale=sc).to(DEV), rnd(R, N, A, seed=159).to(DEV)
    cond = (torch.rand(R, N, N) > 0.4).float().to(DEV)
    lbias = rnd(R, N, N, seed=160, scale=0.3).to(DEV)
    geo = torch.relu(rnd(R, N, M, H, seed=161)).to(DEV)
    for direction in (0, 1):
        out_f, P_f = ops.gat_attention(q, k, S, cond, lbias, geo, H, direction, 'fused')
        out_g, P_g = ops.gat_attention(q, k, S, cond, lbias, geo, H, direction, 'gemm')
        check(P_f, P_g.cpu(), 5e-6, 'P split R=%d' % R)
        check(out_f, out_g.cpu(), 5e-6, 'outT split R=%d' % R)


# --------------------------------------------------------------------------- fused decoder-step chains (speaker.hip)
def _stream():
    from ekaid_amd import hip
    return hip.stream()


@pytest.mark.parametrize('B,R,A', [(5, 64, 96), (256, 512, 1024), (1, 300, 40)])
def test_spk_modmix_fwd_bwd(B, R, A):
    """fc -> softmax(3) -> mix3 and its backward (incl. the gated-product term and both accumulations) vs float64."""
    from ekaid_amd import hip
    L = hip.lib()
    h, w_fc, b_fc = rnd(B, R, seed=1), rnd(3, R, seed=2, scale=0.2), rnd(3, seed=3)
    feats = [rnd(B, A, seed=4 + i) for i in range(3)]
    hd, wd, bd = h.double().requires_grad_(), w_fc.double(), b_fc.double()
    fd = [f.double().requires_grad_() for f in feats]
    z = hd @ wd.t() + bd
    z.retain_grad()
    mwd = torch.softmax(z, 1)
    attd = mwd[:, 0:1] * fd[0] + mwd[:, 1:2] * fd[1] + mwd[:, 2:3] * fd[2]
    g = [t.to(DEV) for t in (h, w_fc, b_fc, *feats)]
    mw = torch.empty(B, 3, device=DEV)
    att = torch.empty(B, A, device=DEV)
    wide = torch.full((B, A + 7), float('nan'), device=DEV)
    hip.check(L.ekaid_spk_modmix_fwd(hip.ptr(g[0]), B, R, hip.ptr(g[1]), hip.ptr(g[2]), hip.ptr(g[3]), hip.ptr(g[4]), hip.ptr(g[5]), A,
                                     hip.ptr(mw), hip.ptr(att), wide.data_ptr() + 4 * 7, A + 7, _stream()), 'fwd')
    check(mw, mwd, 2e-6, 'module weights')
    check(att, attd, 2e-6, 'attended feature')
    assert torch.equal(wide[:, 7:], att) and torch.isnan(wide[:, :7]).all()
    # backward: d_att = d_att_a (strided) + d_gated * gate
    d_a, d_g, gate = rnd(B, A + 3, seed=20), rnd(B, A, seed=21), torch.sigmoid(rnd(B, A, seed=22))
    d_att = d_a[:, 3:].double() + d_g.double() * gate.double()
    attd.backward(d_att)
    dz = torch.empty(B, 3, device=DEV)
    acc0 = [rnd(B, A, seed=30 + i) for i in range(3)]
    dh0 = rnd(B, R, seed=40)
    dacc = [t.clone().to(DEV) for t in acc0]
    dh = dh0.clone().to(DEV)
    d_a_dev, d_g_dev, gate_dev = d_a.to(DEV), d_g.to(DEV), gate.to(DEV)      # keep the device copies alive across the call
    hip.check(L.ekaid_spk_modmix_bwd(hip.ptr(mw), hip.ptr(g[3]), hip.ptr(g[4]), hip.ptr(g[5]), B, A, d_a_dev.data_ptr() + 4 * 3, A + 3,
                                     hip.ptr(d_g_dev), hip.ptr(gate_dev), hip.ptr(g[1]), R, hip.ptr(dz), hip.ptr(dacc[0]),
                                     hip.ptr(dacc[1]), hip.ptr(dacc[2]), hip.ptr(dh), _stream()), 'bwd')
    check(dz, z.grad, 3e-5, 'd logits')        # s_c - <mw, s> cancels
    for i in range(3):
        check(dacc[i], acc0[i].double() + fd[i].grad, 5e-6, 'd feat %d (accumulated)' % i)
    check(dh, dh0.double() + hd.grad, 5e-6, 'd h (accumulated)')


@pytest.mark.parametrize('B,R,P', [(7, 64, 16), (256, 512, 16), (3, 100, 5)])
def test_spk_pos_chain_fwd_bwd_vs_float64(B, R, P):
    """relu(x+b) -> weight_pos -> softmax -> pos2 and its backward, dropout off, strided in/out, vs float64."""
    from ekaid_amd import hip
    L = hip.lib()
    x, b1 = rnd(B, R + 5, seed=1), rnd(R, seed=2)
    w_wp, b_wp, w_p2, b_p2 = rnd(P, R, seed=3, scale=0.3), rnd(P, seed=4), rnd(R, P, seed=5), rnd(R, seed=6)
    xd = x[:, 5:].double().requires_grad_()
    v = torch.relu(xd + b1.double())
    v.retain_grad()
    dposd = v @ w_wp.double().t() + b_wp.double()
    dposd.retain_grad()
    pwd = torch.softmax(dposd, 1)
    ppd = pwd @ w_p2.double().t() + b_p2.double()
    dev = lambda t: t.to(DEV)
    xg = dev(x)
    vp, dp, pw = torch.empty(B, R, device=DEV), torch.empty(B, P, device=DEV), torch.empty(B, P, device=DEV)
    pp = torch.full((B, R + 2), float('nan'), device=DEV)
    W = [dev(t) for t in (b1, w_wp, b_wp, w_p2, b_p2)]
    hip.check(L.ekaid_spk_pos_fwd(xg.data_ptr() + 4 * 5, R + 5, hip.ptr(W[0]), B, R, hip.ptr(W[1]), hip.ptr(W[2]), P, hip.ptr(W[3]),
                                  hip.ptr(W[4]), 0.0, 1, 0.0, 2, hip.ptr(vp), hip.ptr(dp), hip.ptr(pw), hip.ptr(pp), R + 2, _stream()),
              'pos fwd')
    check(vp, v, 1e-6, 'vpos'); check(dp, dposd, 3e-6, 'dpos'); check(pw, pwd, 3e-6, 'pos weights')
    check(pp[:, :R], ppd, 3e-6, 'pos feat')
    assert torch.isnan(pp[:, R:]).all()
    dpp, dhead = rnd(B, R + 4, seed=9), rnd(B, P, seed=10)
    ppd.backward(dpp[:, :R].double(), retain_graph=True)
    dposd.backward(dhead.double())
    dwp = torch.empty(B, P, device=DEV)
    dx = torch.full((B, R + 1), float('nan'), device=DEV)
    dpp_dev, dhead_dev = dev(dpp), dev(dhead)
    hip.check(L.ekaid_spk_pos_bwd(hip.ptr(dpp_dev), R + 4, hip.ptr(pw), hip.ptr(dhead_dev), B, P, hip.ptr(W[3]), hip.ptr(W[1]),
                                  hip.ptr(vp), R, 0.0, 1, 0.0, 2, hip.ptr(dwp), hip.ptr(dx), R + 1, _stream()), 'pos bwd')
    check(dwp, dposd.grad, 5e-6, 'd weight_pos output')
    check(dx[:, :R], xd.grad, 5e-6, 'd pos1 pre-activation')
    assert torch.isnan(dx[:, R:]).all()


def test_spk_pos_chain_with_dropout_equals_the_unfused_kernel_sequence():
    """With both dropouts on, the fused chain reproduces the sequence of stand-alone kernels it replaces (same
    counter-based masks: (seed_v, b*R + k) and (seed_d, b*P + j)), forward and backward."""
    from ekaid_amd import hip, ops
    from ekaid_amd.hip import ACT_RELU
    from ekaid_amd.speaker_seq import _nt, _nn, _drop
    L = hip.lib()
    B, R, P, pv, pd, sv, sd = 64, 256, 16, 0.3, 0.5, 1234567, 7654321
    dev = lambda t: t.to(DEV)
    x, b1 = dev(rnd(B, R, seed=1)), dev(rnd(R, seed=2))
    w_wp, b_wp, w_p2, b_p2 = dev(rnd(P, R, seed=3, scale=0.3)), dev(rnd(P, seed=4)), dev(rnd(R, P, seed=5)), dev(rnd(R, seed=6))
    e = lambda *s: torch.empty(*s, device=DEV)
    vp, dp, pw, pp = e(B, R), e(B, P), e(B, P), e(B, R)
    hip.check(L.ekaid_spk_pos_fwd(hip.ptr(x), R, hip.ptr(b1), B, R, hip.ptr(w_wp), hip.ptr(b_wp), P, hip.ptr(w_p2), hip.ptr(b_p2),
                                  pv, sv, pd, sd, hip.ptr(vp), hip.ptr(dp), hip.ptr(pw), hip.ptr(pp), R, _stream()), 'fused fwd')
    vp2, dp2, pw2, pp2 = e(B, R), e(B, P), e(B, P), e(B, R)
    hip.check(L.ekaid_bias_act_drop(hip.ptr(x), R, hip.ptr(b1), B, R, ACT_RELU, pv, sv, hip.ptr(vp2), _stream()), 'bias_act_drop')
    ops.gemm(vp2, w_wp, dp2, B, P, R, R, R, P, 0, 1, bias=b_wp, drop_p=pd, drop_seed=sd)
    hip.check(L.ekaid_softmax_fwd(hip.ptr(dp2), hip.ptr(pw2), B, P, 0, _stream()), 'softmax')
    _nt(pw2, 0, P, w_p2, 0, P, pp2, 0, R, B, R, P, bias=b_p2)
    assert torch.equal(vp, vp2) and 0.2 < float((vp == 0).float().mean()) < 0.9
    check(dp, dp2, 3e-6, 'dpos'); check(pw, pw2, 3e-6, 'pos weights'); check(pp, pp2, 3e-6, 'pos feat')
    assert torch.equal(dp == 0, dp2 == 0) and float((dp == 0).float().mean()) > 0.2
    dpp, dhead = dev(rnd(B, R, seed=9)), dev(rnd(B, P, seed=10))
    dwp, dx = e(B, P), e(B, R)
    hip.check(L.ekaid_spk_pos_bwd(hip.ptr(dpp), R, hip.ptr(pw), hip.ptr(dhead), B, P, hip.ptr(w_p2), hip.ptr(w_wp), hip.ptr(vp), R,
                                  pv, sv, pd, sd, hip.ptr(dwp), hip.ptr(dx), R, _stream()), 'fused bwd')
    t16, t16b, dwp2, tR, dx2 = e(B, P), e(B, P), e(B, P), e(B, R), e(B, R)
    _nn(dpp, 0, R, w_p2, 0, P, t16, 0, P, B, P, R)
    hip.check(L.ekaid_softmax_bwd(hip.ptr(pw), hip.ptr(t16), hip.ptr(t16b), B, P, 0, _stream()), 'softmax_bwd')
    hip.check(L.ekaid_axpby(hip.ptr(t16b), hip.ptr(dhead), hip.ptr(t16b), B * P, 1.0, 1.0, _stream()), 'axpby')
    _drop(t16b, dwp2, B * P, pd, sd)
    _nn(dwp2, 0, P, w_wp, 0, R, tR, 0, R, B, R, P)
    hip.check(L.ekaid_drop_act_bwd(hip.ptr(tR), hip.ptr(vp), R, B, R, ACT_RELU, pv, sv, hip.ptr(dx2), R, _stream()), 'drop_act_bwd')
    check(dwp, dwp2, 5e-6, 'd weight_pos output'); check(dx, dx2, 5e-6, 'd pos1 pre-activation')
    assert torch.equal(dx == 0, dx2 == 0)


def test_spk_gate_bwd():
    from ekaid_amd import hip
    n = 256 * 1024 + 3
    d, a, g = rnd(n, seed=1), rnd(n, seed=2), torch.sigmoid(rnd(n, seed=3))
    out = torch.empty(n, device=DEV)
    dd, ad, gd = d.to(DEV), a.to(DEV), g.to(DEV)
    hip.check(hip.lib().ekaid_spk_gate_bwd(hip.ptr(dd), hip.ptr(ad), hip.ptr(gd), n, hip.ptr(out), _stream()), 'gate')
    check(out, d.double() * a.double() * g.double() * (1 - g.double()), 1e-6, 'gate derivative')


def test_round2_kernels_int8_labels_loss_denominator_device_adam_dropout_epoch():
    """The entry points added for the real-data feed, data-parallel loss weighting and hipGraph replay."""
    from ekaid_amd import hip, ops
    from ekaid_amd.hip import check as hcheck, ptr, stream
    from ekaid_amd.utils.mimic_utils import process_matrix_labels
    from oracle import ekaid_oracle as O
    # int8 N x N label matrices (the compact PCIe form) == the reference's float64 100 x 100 path
    lab = torch.randint(0, 12, (3, 100, 100))
    want = O.process_matrix(lab.double(), 52, 11)
    got = process_matrix_labels(lab[:, :52, :52].to(torch.int8).contiguous().to(DEV), 52, 11)
    assert torch.equal(got.cpu(), want)
    # LM loss with an external denominator: value and gradient scale with sum(mask) / denom
    B, T, V = 4, 5, 19
    logp = torch.log_softmax(rnd(B, T, V, seed=7), -1)
    labels = torch.randint(1, V, (B, T + 1))
    masks = torch.zeros(B, T + 1)
    masks[:, :3] = 1
    masks[1, :5] = 1
    g0 = logp.to(DEV).requires_grad_(True)
    l0 = ops.lm_loss(g0, labels.to(DEV)[:, 1:], masks.to(DEV)[:, 1:])
    g1 = logp.to(DEV).requires_grad_(True)
    denom = torch.tensor([7.5], device=DEV)
    l1 = ops.lm_loss(g1, labels.to(DEV)[:, 1:], masks.to(DEV)[:, 1:], denom)
    ratio = float(masks[:, 1:].sum()) / 7.5
    assert float(l1) == pytest.approx(float(l0) * ratio, rel=1e-6)
    d0, = torch.autograd.grad(l0, g0)
    d1, = torch.autograd.grad(l1, g1)
    check(d1, d0 * ratio, 1e-6, 'lm_loss gradient with external denominator')
    # Adam with device-resident step / hyper-parameters vs torch.optim.Adam, learning rate changed between steps
    p0, grads = rnd(1000, seed=46), [rnd(1000, seed=47), rnd(1000, seed=48), rnd(1000, seed=49)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], 1e-4, (0.9, 0.999), 1e-8)
    pg, m, v = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    hyper = torch.tensor([1e-4, 0.9, 0.999, 1e-8, 0.5], device=DEV)
    stepd = torch.zeros(1, dtype=torch.int32, device=DEV)
    derived = torch.zeros(4, device=DEV)
    fault = torch.zeros(1, device=DEV)
    for i, g in enumerate(grads):
        if i == 2:
            opt.param_groups[0]['lr'] = 3e-5
            hyper[0] = 3e-5
        pr.grad = g.clone()
        opt.step()
        hcheck(hip.lib().ekaid_adam_step_dev(ptr(pg), ptr((2 * g).to(DEV)), ptr(m), ptr(v), 1000, ptr(hyper), ptr(stepd),
                                             ptr(derived), ptr(fault) if i else None, stream()), 'adam_dev')
    assert int(stepd.item()) == 3
    check(pg - p0.to(DEV), pr.detach() - p0, 1e-5, 'device-state adam delta after 3 steps')
    # fault guard (ABI 8): a step behind a persistent launch that abandoned itself is refused ON THE DEVICE -- weights, moments and
    # the step counter stay as they are -- (a) when any rank raised the all-reduced fault flag, (b) when this device's status word is set
    import ctypes
    snap = (pg.clone(), m.clone(), v.clone())
    gbad = torch.full((1000,), float('nan'), device=DEV)
    fault.fill_(2.0)                                           # e.g. two of the ranks reported a fault
    hcheck(hip.lib().ekaid_adam_step_dev(ptr(pg), ptr(gbad), ptr(m), ptr(v), 1000, ptr(hyper), ptr(stepd), ptr(derived), ptr(fault),
                                         stream()), 'adam_dev')
    assert int(stepd.item()) == 3 and all(torch.equal(a, b) for a, b in zip((pg, m, v), snap)), 'update applied despite the fault flag'
    fault.zero_()
    rt = ctypes.CDLL('libamdhip64.so')
    word = ctypes.c_void_p(hip.lib().ekaid_speaker_seq_persistent_status_ptr(None))
    one = ctypes.c_uint(0x101)
    torch.cuda.synchronize()
    assert rt.hipMemcpy(word, ctypes.byref(one), 4, 1) == 0     # host -> device: as if seam 1 of a forward launch had timed out
    try:
        flag = torch.full((1,), -1.0, device=DEV)
        hcheck(hip.lib().ekaid_seq_fault_flag(ptr(flag), stream()), 'seq_fault_flag')
        assert float(flag.item()) == 1.0
        hcheck(hip.lib().ekaid_adam_step_dev(ptr(pg), ptr(gbad), ptr(m), ptr(v), 1000, ptr(hyper), ptr(stepd), ptr(derived), None,
                                             stream()), 'adam_dev')
        assert int(stepd.item()) == 3 and all(torch.equal(a, b) for a, b in zip((pg, m, v), snap)), 'update applied despite the status word'
        with pytest.raises(hip.EkaidError):
            hip.seq_status_check()                              # the host-side check reports it (and clears the word)
    finally:
        hcheck(hip.lib().ekaid_seq_status_clear(stream()), 'seq_status_clear')
    hcheck(hip.lib().ekaid_seq_fault_flag(ptr(flag), stream()), 'seq_fault_flag')
    assert float(flag.item()) == 0.0
    hcheck(hip.lib().ekaid_adam_step_dev(ptr(pg), ptr((2 * grads[0]).to(DEV)), ptr(m), ptr(v), 1000, ptr(hyper), ptr(stepd), ptr(derived),
                                         ptr(fault), stream()), 'adam_dev')
    assert int(stepd.item()) == 4 and not torch.equal(pg, snap[0])
    # dropout epoch: NULL pointer = seed alone; with a pointer the mask is a function of (seed, epoch); GEMM epilogue too
    x = torch.ones(1 << 16, device=DEV)
    y0, y1, y2, y3 = (torch.empty_like(x) for _ in range(4))
    L = hip.lib()
    hcheck(L.ekaid_dropout(ptr(x), ptr(y0), x.numel(), 0.5, 1234, stream()), 'dropout')
    epoch = torch.zeros(1, dtype=torch.int64, device=DEV)
    try:
        hcheck(L.ekaid_set_dropout_epoch_ptr(epoch.data_ptr()), 'set epoch')
        hcheck(L.ekaid_dropout(ptr(x), ptr(y1), x.numel(), 0.5, 1234, stream()), 'dropout')
        hcheck(L.ekaid_dropout_epoch_advance(epoch.data_ptr(), 1, stream()), 'advance')
        hcheck(L.ekaid_dropout(ptr(x), ptr(y2), x.numel(), 0.5, 1234, stream()), 'dropout')
        hcheck(L.ekaid_dropout(ptr(x), ptr(y3), x.numel(), 0.5, 1234, stream()), 'dropout')
        A, Bm = rnd(64, 32, seed=1).to(DEV), rnd(48, 32, seed=2).to(DEV)
        c1, c2 = torch.empty(64, 48, device=DEV), torch.empty(64, 48, device=DEV)
        ops.gemm(A, Bm, c1, 64, 48, 32, 32, 32, 48, 0, 1, drop_p=0.5, drop_seed=99)
        hcheck(L.ekaid_dropout_epoch_advance(epoch.data_ptr(), 1, stream()), 'advance')
        ops.gemm(A, Bm, c2, 64, 48, 32, 32, 32, 48, 0, 1, drop_p=0.5, drop_seed=99)
    finally:
        hcheck(L.ekaid_set_dropout_epoch_ptr(None), 'unset epoch')
    assert int(epoch.item()) == 2
    assert torch.equal(y0, y1), 'epoch 0 must reproduce the plain seed'
    assert torch.equal(y2, y3) and not torch.equal(y1, y2)
    assert abs(float((y2 != 0).float().mean()) - 0.5) < 0.02 and abs(float(((y1 != 0) & (y2 != 0)).float().mean()) - 0.25) < 0.02
    assert not torch.equal(c1 != 0, c2 != 0)


def test_fanout_sums_bias_slots_and_batched_step_dropout():
    """Launch-diet entry points: n-ary gradient sum (ops.fanout / ops.halves), bias gradients accumulated into a zeroed
    slot (ekaid_colsum_acc), T dropout masks in one launch (ekaid_dropout_steps), Q-A mix, stacked adjacency view."""
    import ctypes
    from ekaid_amd import hip, ops
    from ekaid_amd.hip import check as hcheck, ptr, stream
    L = hip.lib()
    # ekaid_sum_n: 1..8 addends, vector path + unaligned tail + unaligned base
    for n in (1, 5, 1024, 4099):
        for k in (1, 2, 3, 8):
            xs = [rnd(n + 1, seed=100 + j).to(DEV) for j in range(k)]
            for shift in (0, 1):
                part = [x[shift:shift + n] for x in xs]
                out = torch.empty(n, device=DEV)
                arr = (ctypes.c_void_p * k)(*[p.data_ptr() for p in part])
                hcheck(L.ekaid_sum_n(arr, k, ptr(out), n, stream()), 'sum_n')
                want = part[0].clone()
                for p in part[1:]:
                    want = want + p                      # same left-to-right order: bit-identical
                assert torch.equal(out, want)
    assert L.ekaid_sum_n(None, 9, None, 4, None) != 0
    # fanout / halves: gradients equal autograd's own accumulation over plain slices
    x = rnd(6, 5, 8, seed=1).to(DEV).requires_grad_(True)
    w = [rnd(6, 5, 8, seed=10 + i).to(DEV) for i in range(3)]
    wh = [rnd(3, 5, 8, seed=20 + i).to(DEV) for i in range(4)]
    a, b, c = ops.fanout(x, 3)
    b0, b1, a0, a1 = ops.halves(c, 2)
    y = (a * w[0]).sum() + (b * w[1]).sum() + (b0 * wh[0]).sum() + (b1 * wh[1]).sum() + (a0 * wh[2]).sum() + (a1 * wh[3]).sum()
    g, = torch.autograd.grad(y, x)
    want = w[0] + w[1] + torch.cat((wh[0] + wh[1], wh[2] + wh[3]), 0)
    check(g, want, 1e-6, 'fanout/halves gradient')
    # unused branches arrive as None
    x2 = rnd(4, 3, seed=2).to(DEV).requires_grad_(True)
    u, v, _unused = ops.fanout(x2, 3)
    h0, h1 = ops.halves(v)
    g2, = torch.autograd.grad((u * 2).sum() + (h1 * 3).sum(), x2)
    want2 = torch.full((4, 3), 2.0)
    want2[2:] += 3.0
    assert torch.equal(g2.cpu(), want2)
    # mix3 == a*x + b*x + c*x, value and gradient
    x3 = rnd(1000, seed=3).to(DEV).requires_grad_(True)
    y3 = ops.self_mix3(x3, 0.3, 0.5, 1 - 0.3 - 0.5)
    check(y3, 0.3 * x3.detach().cpu() + 0.5 * x3.detach().cpu() + 0.2 * x3.detach().cpu(), 1e-6, 'mix3')
    g3, = torch.autograd.grad(y3.sum(), x3)
    check(g3, torch.ones(1000), 1e-6, 'mix3 gradient')
    # colsum_acc accumulates on top of what is there (a zeroed slot, or an earlier contribution of a shared bias)
    m = rnd(300, 70, seed=4).to(DEV)
    out = torch.full((70,), 1.5, device=DEV)
    hcheck(L.ekaid_colsum_acc(ptr(m), 300, 70, 70, ptr(out), stream()), 'colsum_acc')
    check(out, 1.5 + m.double().sum(0).cpu(), 2e-6, 'colsum_acc')
    sub = torch.zeros(20, device=DEV)
    hcheck(L.ekaid_colsum_acc(m.data_ptr() + 4 * 10, 300, 20, 70, ptr(sub), stream()), 'colsum_acc')
    check(sub, m[:, 10:30].double().sum(0).cpu(), 2e-6, 'colsum_acc column block')
    # dropout_steps == one ekaid_dropout per step with seed + t (shared input, and per-step input in place)
    T, n, seed, p = 5, 777, 4242, 0.3
    src = rnd(n, seed=5).to(DEV)
    many = torch.empty(T, n, device=DEV)
    hcheck(L.ekaid_dropout_steps(ptr(src), ptr(many), n, T, 0, p, seed, stream()), 'dropout_steps')
    one = torch.empty(T, n, device=DEV)
    for t in range(T):
        hcheck(L.ekaid_dropout(ptr(src), one.data_ptr() + 4 * t * n, n, p, seed + t, stream()), 'dropout')
    assert torch.equal(many, one)
    assert 0.2 < float((many == 0).float().mean()) < 0.4 and not torch.equal(many[0], many[1])
    grads = rnd(T, n, seed=6).to(DEV)
    inplace = grads.clone()
    hcheck(L.ekaid_dropout_steps(ptr(inplace), ptr(inplace), n, T, n, p, seed, stream()), 'dropout_steps')
    want_g = torch.empty(T, n, device=DEV)
    for t in range(T):
        hcheck(L.ekaid_dropout(grads.data_ptr() + 4 * t * n, want_g.data_ptr() + 4 * t * n, n, p, seed + t, stream()), 'dropout')
    assert torch.equal(inplace, want_g)
    # stack2: a view when the halves are adjacent in one buffer, a copy otherwise
    pair = torch.arange(24, device=DEV, dtype=torch.float32).view(4, 3, 2)
    v2 = ops.stack2(pair[:2], pair[2:])
    assert v2.data_ptr() == pair.data_ptr() and torch.equal(v2, pair)
    cpy = ops.stack2(pair[2:], pair[:2])
    assert cpy.data_ptr() != pair.data_ptr() and torch.equal(cpy, torch.cat((pair[2:], pair[:2]), 0))
    # zeros_pool: aligned, zero, disjoint
    z = ops.zeros_pool(DEV, (3, 5), (7,), (2, 2))
    assert [tuple(t.shape) for t in z] == [(3, 5), (7,), (2, 2)] and all(float(t.abs().sum()) == 0 for t in z)
    assert all(t.data_ptr() % 16 == 0 for t in z)


@pytest.mark.parametrize('M,N,K', [(100, 70, 33), (1152, 1024, 1024), (256, 512, 4096), (64, 1024, 2304)])
def test_gemm_epilogue_post_add(M, N, K, gemm_arith):
    """ekaid_gemm_t.post (a gradient that joins the product behind its dropout) against the separate kernels it replaces: bit-identical
    (same product, same mask function); tiles stored directly, edge tiles, and split-K shapes finished by the reduce kernel; with an
    accumulating product (old C and post both fetched ahead of the stores)."""
    from ekaid_amd import hip, ops
    L = hip.lib()
    A, W = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2).to(DEV)
    post, bias = rnd(M, N, seed=3).to(DEV), rnd(N, seed=4).to(DEV)
    plain = torch.empty(M, N, device=DEV)
    ops.gemm(A, W, plain, M, N, K, K, K, N, 0, 1, bias=bias)
    # post: C = dropout(product) + post
    c = torch.empty(M, N, device=DEV)
    ops.gemm(A, W, c, M, N, K, K, K, N, 0, 1, bias=bias, drop_p=0.3, drop_seed=77, post=post, ldpost=N)
    d = torch.empty(M, N, device=DEV)
    hip.check(L.ekaid_dropout(hip.ptr(plain), hip.ptr(d), plain.numel(), 0.3, 77, hip.stream()), 'dropout')
    assert torch.equal(c, d + post)
    # accumulate (old C added before the dropout) together with post
    c0 = rnd(M, N, seed=5).to(DEV)
    c3 = c0.clone()
    ops.gemm(A, W, c3, M, N, K, K, K, N, 0, 1, accumulate=True, post=post, ldpost=N)
    plain2 = torch.empty(M, N, device=DEV)
    ops.gemm(A, W, plain2, M, N, K, K, K, N, 0, 1)
    check(c3, (plain2 + c0) + post, 1e-6, 'accumulate + post')


def test_rowmask_dropout_and_self_mix3_equal_the_launches_they_replace():
    from ekaid_amd import hip, ops
    L = hip.lib()
    x = rnd(300, 1024, seed=1).to(DEV)
    x[7] = 0.0
    x[100, :] = 0.0
    x[11, :2] = torch.tensor([1.0, -1.0], device=DEV)
    x[11, 2:] = 0.0                                     # sums to exactly zero without being a zero row: masked, as in the reference
    keep, seed, xd = ops.rowmask_dropout(x, 0.2)
    assert torch.equal(keep, ops.rowsum(x, as_mask=True).view(-1)) and float(keep[7]) == 0.0 and float(keep[11]) == 0.0
    want = torch.empty_like(x)
    hip.check(L.ekaid_dropout(hip.ptr(x), hip.ptr(want), x.numel(), 0.2, seed, hip.stream()), 'dropout')
    assert torch.equal(xd, want)
    y = torch.empty(1000 * 37 + 3, device=DEV)
    src = rnd(1000 * 37 + 3, seed=2).to(DEV)
    hip.check(L.ekaid_self_mix3(hip.ptr(src), hip.ptr(y), src.numel(), 0.3, 0.5, 0.2, hip.stream()), 'self_mix3')
    t, y2 = torch.empty_like(src), torch.empty_like(src)
    hip.check(L.ekaid_axpby(hip.ptr(src), hip.ptr(src), hip.ptr(t), src.numel(), 0.3, 0.5, hip.stream()), 'axpby')
    hip.check(L.ekaid_axpby(hip.ptr(t), hip.ptr(src), hip.ptr(y2), src.numel(), 1.0, 0.2, hip.stream()), 'axpby')
    check(y, y2, 1.5e-7, 'self_mix3 vs two axpby launches')


def test_fanout_gradients_handed_from_reader_to_reader():
    """ops.GradSlot: readers of a fan-out tensor that know slots (linear layers with input dropout, the residual of
    residual_relu2, the value operand of the attention core) pass ONE running gradient along instead of leaving n tensors to a sum
    kernel; any mix with ordinary readers gives the same gradient as the unfused graph."""
    from ekaid_amd import hip, ops
    from ekaid_amd.models.fc import FCNet
    torch.manual_seed(0)
    R, N, A = 6, 9, 64
    x0 = rnd(R, N, A, seed=1).to(DEV)
    fq, fk = FCNet([A, A], None, 0.2).to(DEV), FCNet([A, A], None, 0.2).to(DEV)
    fq.train(); fk.train()
    wq, wk, wo = rnd(R * N, A, seed=2).to(DEV), rnd(R * N, A, seed=3).to(DEV), rnd(R, N, A, seed=4).to(DEV)
    o = rnd(R, N, A, seed=5).to(DEV)

    def run(slots):
        x = x0.clone().requires_grad_(True)
        ops.manual_seed(123)
        if slots:
            a, b, c, d = ops.fanout(x, 4)
        else:
            a = b = c = d = x
        y = (fq(a.view(R * N, A)) * wq).sum() + (fk(ops.pass_slot(b, b.view(R * N, A))) * wk).sum()
        y = y + (ops.residual_relu2(c, o, 0.0, True) * wo).sum() + (d * d).sum()          # d: an ordinary reader
        g, = torch.autograd.grad(y, x)
        return g, None

    g_plain, _ = run(False)
    g_slots, _ = run(True)
    check(g_slots, g_plain, 2e-6, 'gradient through the slot hand-over')
    # the slot-aware readers alone: nothing is left to sum
    x = x0.clone().requires_grad_(True)
    a, b, c = ops.fanout(x, 3)
    slot = ops.slot_of(a)
    y = (fq(ops.pass_slot(a, a.view(R * N, A))) * wq).sum() + (fk(ops.pass_slot(b, b.view(R * N, A))) * wk).sum()
    y = y + (ops.residual_relu2(c, o, 0.0, True) * wo).sum()
    import ekaid_amd.ops as O
    seen = []
    real = O._sum_into
    O._sum_into = lambda grads, out: (seen.append(len(grads)), real(grads, out))[1]
    try:
        g, = torch.autograd.grad(y, x)
    finally:
        O._sum_into = real
    assert not seen, 'a sum kernel still ran: %r' % (seen,)
    assert slot.lent is None and g.shape == x.shape and float(g.abs().sum()) > 0


# --------------------------------------------------------------------------- slabs: GEMMs whose consumers reduce
@pytest.mark.parametrize('mode,M,N,K', [('nt', 32, 6656, 512), ('nt', 256, 2048, 1536), ('nn', 32, 512, 6656), ('nn', 7, 100, 50),
                                        ('nt', 2, 64, 31), ('nn', 256, 1024, 2048), ('nt', 600, 256, 512)])
def test_gemm_slabs_sum_to_the_product(mode, M, N, K, gemm_arith):
    """ekaid_gemm_slabs: the K-slabs add up to op(A).op(B) (float64 reference), also when the planner does not split."""
    from ekaid_amd import ops
    A, Bm = rnd(M, K, seed=3), rnd(K, N, seed=4)
    want = A.double() @ Bm.double()
    if mode == 'nt':
        a, b, ta, tb, lda, ldb = A, Bm.t().contiguous(), 0, 1, K, K
    else:
        a, b, ta, tb, lda, ldb = A, Bm, 0, 0, K, N
    S = ops.gemm_slabs(a.to(DEV), b.to(DEV), M, N, K, lda, ldb, ta, tb)
    assert S.n >= 1 and S.buf.numel() == S.n * M * N
    check(S.total(), want, 2e-6 * max(1.0, math.sqrt(K) / 4), 'slabs %s %dx%dx%d (%d slabs)' % (mode, M, N, K, S.n))


@pytest.mark.parametrize('mode', ['nt', 'nn'])
def test_gemm_slab_pairs_in_one_launch(mode, gemm_arith):
    """ekaid_gemm_slabs2: two independent products (different N, K and split counts, operand offsets) from one launch equal
    the two single launches bit for bit."""
    from ekaid_amd import ops
    M = 48
    shapes = [(6656, 512), (2048, 512)] if mode == 'nt' else [(512, 2048), (512, 6656)]
    gs, singles = [], []
    for i, (N, K) in enumerate(shapes):
        Awide = rnd(3, M, K, seed=10 + i).to(DEV)                       # the product reads slice 1 (element offset)
        Bm = (rnd(N, K, seed=20 + i) if mode == 'nt' else rnd(K, N, seed=20 + i)).to(DEV)
        g = (Awide, Bm, M, N, K, K, K if mode == 'nt' else N, 0, 1 if mode == 'nt' else 0, M * K, 0)
        gs.append(g)
        singles.append(ops.gemm_slabs(*g[:9], a_off=g[9], b_off=g[10]))
    p0, p1 = ops.gemm_slabs2(gs[0], gs[1])
    for s, p, g in zip(singles, (p0, p1), gs):
        assert p.n == s.n
        assert torch.equal(p.buf, s.buf)
        Aw, Bm = g[0][1].double().cpu(), g[1].double().cpu()
        check(p.total(), Aw @ (Bm.t() if mode == 'nt' else Bm), 5e-6, 'pair member')


def _rand_slabs(n, M, N, seed):
    """n random slabs [n][M][N] on the device + their float64 sum."""
    from ekaid_amd import ops
    buf = rnd(n, M, N, seed=seed).to(DEV)
    return ops.SlabSet(buf.view(-1), n, M, N), buf.double().cpu().sum(0)


def test_slab_consumers_equal_the_plain_kernels():
    """Every slab-consuming kernel == summing the slabs first and calling the plain kernel it replaces."""
    import ctypes
    from ekaid_amd import hip, ops
    from ekaid_amd.hip import ACT_RELU, ACT_TANH, check as hcheck, ptr, stream
    L = hip.lib()
    B, H = 37, 24
    f = dict(device=DEV, dtype=torch.float32)
    # ---- slab_epilogue (two sources, a column block of a wider slab set, bias, relu, dropout) vs bias_act_drop
    S0, t0 = _rand_slabs(5, B, 3 * H, 1)
    S1, t1 = _rand_slabs(2, B, H, 2)
    bias = rnd(H, seed=3).to(DEV)
    for p, seed in ((0.0, 0), (0.5, 77)):
        arr, n = ops.slab_array(S0.cols(H), S1.cols(0))
        out = torch.empty(B, H, **f)
        hcheck(L.ekaid_slab_epilogue(arr, n, ptr(bias), B, H, ACT_RELU, p, seed, ptr(out), H, stream()), 'slab_epilogue')
        x = (t0[:, H:2 * H] + t1).float().to(DEV).contiguous()
        want = torch.empty(B, H, **f)
        hcheck(L.ekaid_bias_act_drop(ptr(x), H, ptr(bias), B, H, ACT_RELU, p, seed, ptr(want), stream()), 'bias_act_drop')
        assert float((out - want).abs().max()) < 2e-5
        assert torch.equal(out == 0, want == 0) or p == 0.0
    # odd width -> scalar path
    S2, t2 = _rand_slabs(3, B, 7, 4)
    arr, n = ops.slab_array(S2.cols(0))
    out = torch.empty(B, 7, **f)
    hcheck(L.ekaid_slab_epilogue(arr, n, None, B, 7, ACT_TANH, 0.0, 0, ptr(out), 7, stream()), 'slab_epilogue')
    check(out, torch.tanh(t2), 2e-6, 'slab_epilogue scalar path')
    # ---- drop_act_slabs_bwd vs drop_act_bwd
    y = torch.relu(rnd(B, H, seed=5)).to(DEV)
    S3, t3 = _rand_slabs(4, B, H, 6)
    s3 = S3.cols(0)
    out = torch.empty(B, 2 * H, **f)
    want = torch.empty(B, 2 * H, **f)
    hcheck(L.ekaid_drop_act_slabs_bwd(ctypes.byref(s3), ptr(y), H, B, H, ACT_RELU, 0.3, 5, ptr(out), 2 * H, stream()), 'drop_act_slabs_bwd')
    dy = t3.float().to(DEV).contiguous()
    hcheck(L.ekaid_drop_act_bwd(ptr(dy), ptr(y), H, B, H, ACT_RELU, 0.3, 5, ptr(want), 2 * H, stream()), 'drop_act_bwd')
    assert float((out[:, :H] - want[:, :H]).abs().max()) < 2e-5
    # ---- LSTM cell forward / backward
    gx, c0 = rnd(B, 4 * H, seed=7).to(DEV), rnd(B, H, seed=8).to(DEV)
    Sg0, tg0 = _rand_slabs(3, B, 6 * H, 9)            # gates sit in columns [H, 5H) of a wider product
    Sg1, tg1 = _rand_slabs(6, B, 4 * H, 10)
    b4 = rnd(4 * H, seed=11).to(DEV)
    h2, c2, sv = torch.empty(B, H, **f), torch.empty(B, H, **f), torch.empty(B, 4 * H, **f)
    arr, n = ops.slab_array(Sg0.cols(H), Sg1.cols(0))
    hcheck(L.ekaid_lstm_cell_slabs_fwd(ptr(gx), arr, n, ptr(b4), ptr(c0), B, H, ptr(h2), ptr(c2), ptr(sv), stream()), 'lstm_cell_slabs_fwd')
    gh = (tg0[:, H:5 * H] + tg1).float().to(DEV) + b4
    h2w, c2w, svw = torch.empty(B, H, **f), torch.empty(B, H, **f), torch.empty(B, 4 * H, **f)
    hcheck(L.ekaid_lstm_cell_fwd(ptr(gx), ptr(gh), 4 * H, ptr(c0), B, H, ptr(h2w), ptr(c2w), ptr(svw), stream()), 'lstm_cell_fwd')
    for a, w, nm in ((h2, h2w, 'h'), (c2, c2w, 'c'), (sv, svw, 'gates')):
        assert float((a - w).abs().max()) < 3e-6, nm
    Sc, tc = _rand_slabs(5, B, H, 12)
    dhb, dc2 = rnd(B, H, seed=13).to(DEV), rnd(B, H, seed=14).to(DEV)
    for carry, dh_b in ((Sc.cols(0), dhb), (ops.NO_SLABS, dhb), (Sc.cols(0), None)):
        dg, dc = torch.zeros(B, 5 * H, **f), torch.empty(B, H, **f)
        hcheck(L.ekaid_lstm_cell_slabs_bwd(ctypes.byref(carry), ptr(dh_b), ptr(dc2), ptr(svw), ptr(c0), ptr(c2w), B, H, ptr(dg), 5 * H,
                                           ptr(dc), stream()), 'lstm_cell_slabs_bwd')
        dh_a = tc.float().to(DEV).contiguous() if carry.n else None
        dgw, dcw = torch.zeros(B, 5 * H, **f), torch.empty(B, H, **f)
        hcheck(L.ekaid_lstm_cell_bwd(ptr(dh_a), ptr(dh_b), ptr(dc2), ptr(svw), ptr(c0), ptr(c2w), B, H, ptr(dgw), 5 * H, ptr(dcw),
                                     stream()), 'lstm_cell_bwd')
        assert float((dg - dgw).abs().max()) < 3e-6 and float((dc - dcw).abs().max()) < 3e-6
    # ---- GRU cell forward / backward
    gi, h0 = rnd(B, 3 * H, seed=15).to(DEV), rnd(B, H, seed=16).to(DEV)
    Sh, th = _rand_slabs(4, B, 3 * H, 17)
    b3 = rnd(3 * H, seed=18).to(DEV)
    hn, sg = torch.empty(B, H, **f), torch.empty(B, 4 * H, **f)
    sh = Sh.cols(0)
    hcheck(L.ekaid_gru_cell_slabs_fwd(ptr(gi), 3 * H, ctypes.byref(sh), ptr(b3), ptr(h0), H, B, H, ptr(hn), H, ptr(sg), stream()),
           'gru_cell_slabs_fwd')
    ghw = th.float().to(DEV) + b3
    hnw, sgw = torch.empty(B, H, **f), torch.empty(B, 4 * H, **f)
    hcheck(L.ekaid_gru_cell_fwd(ptr(gi), 3 * H, ptr(ghw), 3 * H, ptr(h0), H, B, H, ptr(hnw), H, ptr(sgw), stream()), 'gru_cell_fwd')
    assert float((hn - hnw).abs().max()) < 3e-6 and float((sg - sgw).abs().max()) < 3e-6
    dhn, dhd = rnd(B, H, seed=19).to(DEV), rnd(B, H, seed=20).to(DEV)
    Sk, tk = _rand_slabs(3, B, H, 21)
    sk = Sk.cols(0)
    dgi, dgh, dho = torch.empty(B, 3 * H, **f), torch.empty(B, 3 * H, **f), torch.empty(B, H, **f)
    hcheck(L.ekaid_gru_cell_slabs_bwd(ptr(dhn), H, ptr(dhd), ctypes.byref(sk), ptr(sgw), ptr(h0), H, B, H, ptr(dgi), 3 * H, ptr(dgh),
                                      ptr(dho), stream()), 'gru_cell_slabs_bwd')
    carry = (dhd + tk.float().to(DEV)).contiguous()
    dgiw, dghw, dhw = torch.empty(B, 3 * H, **f), torch.empty(B, 3 * H, **f), torch.empty(B, H, **f)
    hcheck(L.ekaid_gru_cell_bwd(ptr(dhn), H, ptr(carry), ptr(sgw), ptr(h0), H, B, H, ptr(dgiw), 3 * H, ptr(dghw), ptr(dhw), stream()),
           'gru_cell_bwd')
    for a, w in ((dgi, dgiw), (dgh, dghw), (dho, dhw)):
        assert float((a - w).abs().max()) < 3e-6
    # ---- gate: sigmoid(sum + b) * att, and its backward
    A = 40
    Sa, ta = _rand_slabs(4, B, A, 22)
    ba, att = rnd(A, seed=23).to(DEV), rnd(B, A, seed=24).to(DEV)
    gate, gated = torch.empty(B, A, **f), torch.empty(B, A, **f)
    sa = Sa.cols(0)
    hcheck(L.ekaid_gate_mul_slabs_fwd(ctypes.byref(sa), ptr(ba), ptr(att), B, A, ptr(gate), ptr(gated), stream()), 'gate_mul_slabs_fwd')
    gw = torch.sigmoid(ta + ba.double().cpu())
    check(gate, gw, 2e-6, 'gate')
    check(gated, gw * att.double().cpu(), 2e-6, 'gated')
    dz, dgd = torch.empty(B, A, **f), torch.empty(B, A, **f)
    hcheck(L.ekaid_gate_slabs_bwd(ctypes.byref(sa), ptr(att), ptr(gate), B, A, ptr(dz), ptr(dgd), stream()), 'gate_slabs_bwd')
    check(dgd, ta, 2e-6, 'd_gated')
    check(dz, ta * att.double().cpu() * gw * (1 - gw), 5e-6, 'dz gate')


@pytest.mark.parametrize('B,R,A,P,drop', [(5, 64, 96, 16, 0.0), (33, 512, 1024, 16, 0.5), (3, 100, 40, 5, 0.3)])
def test_spk_step_a_equals_the_kernel_sequence_it_fuses(B, R, A, P, drop):
    """spk_step_a_fwd / _bwd (att-LSTM cell + module mix + POS chain in one row-local kernel, slab inputs) == the plain
    kernels lstm_cell / spk_modmix / spk_pos applied to the summed slabs, forward and backward, with dropout."""
    import ctypes
    from ekaid_amd import hip, ops
    from ekaid_amd.hip import check as hcheck, ptr, stream
    L = hip.lib()
    f = dict(device=DEV, dtype=torch.float32)
    NW = 6 * R + 8
    gxa, c_prev = rnd(B, 4 * R, seed=1).to(DEV), rnd(B, R, seed=2).to(DEV)
    S0, t0 = _rand_slabs(3, B, NW, 3)                 # block [0, 4R): gates; block [4R + 8, 5R + 8): pos1 pre-activation
    S1, t1 = _rand_slabs(4, B, 4 * R, 4)
    oP = 4 * R + 8
    bhh, w_fc, b_fc = rnd(4 * R, seed=5).to(DEV), rnd(3, R, seed=6, scale=0.2).to(DEV), rnd(3, seed=7).to(DEV)
    bef, diff, aft = (rnd(B, A, seed=8 + i).to(DEV) for i in range(3))
    b1, w_wp, b_wp = rnd(R, seed=11).to(DEV), rnd(P, R, seed=12, scale=0.2).to(DEV), rnd(P, seed=13).to(DEV)
    w_p2, b_p2 = rnd(R, P, seed=14, scale=0.3).to(DEV), rnd(R, seed=15).to(DEV)
    sv, sd = 1234567, 7654321
    d = hip.SpkStepAFwd()
    d.B, d.R, d.A, d.P = B, R, A, P
    h_out, c_out, save = torch.empty(B, R, **f), torch.empty(B, R, **f), torch.empty(B, 4 * R, **f)
    mw, att, PA = torch.empty(B, 3, **f), torch.empty(B, A, **f), torch.zeros(B, R + A, **f)
    vp, dp, pw = torch.empty(B, R, **f), torch.empty(B, P, **f), torch.empty(B, P, **f)
    d.gxa, d.b_hh, d.c_prev = ptr(gxa), ptr(bhh), ptr(c_prev)
    d.gates_src[0], d.gates_src[1], d.n_gates_src = S0.cols(0), S1.cols(0), 2
    d.h_out, d.c_out, d.gate_save = ptr(h_out), ptr(c_out), ptr(save)
    d.w_fc, d.b_fc, d.bef, d.diff, d.aft = ptr(w_fc), ptr(b_fc), ptr(bef), ptr(diff), ptr(aft)
    d.mw, d.att, d.att2, d.ld2 = ptr(mw), ptr(att), PA.data_ptr() + 4 * R, R + A
    d.pos_src = S0.cols(oP)
    d.b_p1, d.w_wp, d.b_wp, d.w_p2, d.b_p2 = ptr(b1), ptr(w_wp), ptr(b_wp), ptr(w_p2), ptr(b_p2)
    d.p_v, d.seed_v, d.p_d, d.seed_d = drop, sv, drop, sd
    d.vp, d.dp, d.pw, d.pp, d.ldpp = ptr(vp), ptr(dp), ptr(pw), ptr(PA), R + A
    hcheck(L.ekaid_spk_step_a_fwd(ctypes.byref(d), stream()), 'spk_step_a_fwd')
    # the sequence it replaces
    gh = ((t0[:, :4 * R] + t1).float().to(DEV) + bhh).contiguous()
    h_w, c_w, save_w = torch.empty(B, R, **f), torch.empty(B, R, **f), torch.empty(B, 4 * R, **f)
    hcheck(L.ekaid_lstm_cell_fwd(ptr(gxa), ptr(gh), 4 * R, ptr(c_prev), B, R, ptr(h_w), ptr(c_w), ptr(save_w), stream()), 'lstm_cell_fwd')
    mw_w, att_w, PA_w = torch.empty(B, 3, **f), torch.empty(B, A, **f), torch.zeros(B, R + A, **f)
    hcheck(L.ekaid_spk_modmix_fwd(ptr(h_w), B, R, ptr(w_fc), ptr(b_fc), ptr(bef), ptr(diff), ptr(aft), A, ptr(mw_w), ptr(att_w),
                                  PA_w.data_ptr() + 4 * R, R + A, stream()), 'spk_modmix_fwd')
    x = t0[:, oP:oP + R].float().to(DEV).contiguous()
    vp_w, dp_w, pw_w = torch.empty(B, R, **f), torch.empty(B, P, **f), torch.empty(B, P, **f)
    hcheck(L.ekaid_spk_pos_fwd(ptr(x), R, ptr(b1), B, R, ptr(w_wp), ptr(b_wp), P, ptr(w_p2), ptr(b_p2), drop, sv, drop, sd, ptr(vp_w),
                               ptr(dp_w), ptr(pw_w), ptr(PA_w), R + A, stream()), 'spk_pos_fwd')
    for nm, a, w in (('h', h_out, h_w), ('c', c_out, c_w), ('gates', save, save_w), ('mw', mw, mw_w), ('att', att, att_w),
                     ('PA', PA, PA_w), ('vp', vp, vp_w), ('dp', dp, dp_w), ('pw', pw, pw_w)):
        assert float((a - w).abs().max()) <= 2e-5 * max(1.0, float(w.abs().max())), nm
    # ---- backward
    Sd, td = _rand_slabs(3, B, R + A, 20)
    Sc, tcar = _rand_slabs(2, B, R, 21)
    d_gated, gate = rnd(B, A, seed=22).to(DEV), torch.sigmoid(rnd(B, A, seed=23)).to(DEV)
    dc_in, dhead = rnd(B, R, seed=24).to(DEV), rnd(B, P, seed=25).to(DEV)
    for carry in (Sc.cols(0), ops.NO_SLABS):
        q = hip.SpkStepABwd()
        q.B, q.R, q.A, q.P = B, R, A, P
        dfe = [rnd(B, A, seed=30 + i).to(DEV) for i in range(3)]              # accumulated in place: start from non-zero
        dfe_w = [t.clone() for t in dfe]
        dz_fc, dgates, dc_out = torch.empty(B, 3, **f), torch.zeros(B, NW, **f), torch.empty(B, R, **f)
        dpp, dwp = torch.empty(B, R, **f), torch.empty(B, P, **f)
        q.dpa_src, q.d_gated, q.gate = Sd.cols(0), ptr(d_gated), ptr(gate)
        q.mw, q.bef, q.diff, q.aft, q.w_fc = ptr(mw_w), ptr(bef), ptr(diff), ptr(aft), ptr(w_fc)
        q.dz_fc, q.d_bef, q.d_diff, q.d_aft = ptr(dz_fc), ptr(dfe[0]), ptr(dfe[1]), ptr(dfe[2])
        q.dh_carry, q.dc_in, q.gate_save, q.c_prev, q.c_cur = carry, ptr(dc_in), ptr(save_w), ptr(c_prev), ptr(c_w)
        q.dgates, q.lddg, q.dc_out = ptr(dgates), NW, ptr(dc_out)
        q.pw, q.dhead, q.w_p2, q.w_wp, q.vp = ptr(pw_w), ptr(dhead), ptr(w_p2), ptr(w_wp), ptr(vp_w)
        q.p_v, q.seed_v, q.p_d, q.seed_d = drop, sv, drop, sd
        q.dpp_out, q.lddpp, q.dwp_out = ptr(dpp), R, ptr(dwp)
        q.dx_out, q.lddx = dgates.data_ptr() + 4 * oP, NW
        hcheck(L.ekaid_spk_step_a_bwd(ctypes.byref(q), stream()), 'spk_step_a_bwd')
        DPA = td.float().to(DEV).contiguous()
        dh = (tcar.float().to(DEV).contiguous() if carry.n else torch.zeros(B, R, **f))
        dz_w = torch.empty(B, 3, **f)
        hcheck(L.ekaid_spk_modmix_bwd(ptr(mw_w), ptr(bef), ptr(diff), ptr(aft), B, A, DPA.data_ptr() + 4 * R, R + A, ptr(d_gated), ptr(gate),
                                      ptr(w_fc), R, ptr(dz_w), ptr(dfe_w[0]), ptr(dfe_w[1]), ptr(dfe_w[2]), ptr(dh), stream()), 'spk_modmix_bwd')
        dwp_w, dg_w, dc_w = torch.empty(B, P, **f), torch.zeros(B, NW, **f), torch.empty(B, R, **f)
        hcheck(L.ekaid_spk_pos_bwd(ptr(DPA), R + A, ptr(pw_w), ptr(dhead), B, P, ptr(w_p2), ptr(w_wp), ptr(vp_w), R, drop, sv, drop, sd,
                                   ptr(dwp_w), dg_w.data_ptr() + 4 * oP, NW, stream()), 'spk_pos_bwd')
        hcheck(L.ekaid_lstm_cell_bwd(ptr(dh), None, ptr(dc_in), ptr(save_w), ptr(c_prev), ptr(c_w), B, R, ptr(dg_w), NW, ptr(dc_w), stream()),
               'lstm_cell_bwd')
        for nm, a, w in (('dz_fc', dz_fc, dz_w), ('d_bef', dfe[0], dfe_w[0]), ('d_diff', dfe[1], dfe_w[1]), ('d_aft', dfe[2], dfe_w[2]),
                         ('dgates', dgates, dg_w), ('dc', dc_out, dc_w), ('dwp', dwp, dwp_w), ('dpp', dpp, DPA[:, :R])):
            assert float((a - w).abs().max()) <= 3e-5 * max(1.0, float(w.abs().max())), nm


def test_wn_linear_rowbcast_equals_the_expanded_form():
    """Inference form of the question-conditioned FCNet: q . Wq^T once per sample + row-broadcast residual in the epilogue ==
    the layer applied to the materialised [v | mask * q_expand] rows, incl. masked (all-zero) node rows and the two images
    sharing the question rows.  ref: relation_encoder.py:19-29, graph_att.py:80."""
    from ekaid_amd import ops
    from ekaid_amd.hip import ACT_NONE, ACT_RELU
    B, N, A, Q, O = 3, 5, 24, 16, 20
    R = 2 * B
    x = rnd(R * N, A, seed=1).to(DEV)
    q = rnd(B, Q, seed=2).to(DEV)
    mask = (torch.rand(R * N, generator=torch.Generator().manual_seed(3)) > 0.3).float().to(DEV)
    v, g, b = rnd(O, A + Q, seed=4, scale=0.3).to(DEV), torch.tensor(1.7, device=DEV), rnd(O, seed=5).to(DEV)
    for act in (ACT_NONE, ACT_RELU):
        with torch.no_grad():
            lazy = ops.wn_linear_rowbcast(x, q, mask, N, v, g, b, act)
            full = ops.wn_linear((x, ops.expand_rows(q, mask, N)), v, g, b, act)
        W = (v * (g / v.norm())).double().cpu()
        qe = (q.double().cpu().repeat(2, 1).repeat_interleave(N, 0)) * mask.double().cpu()[:, None]
        want = torch.cat([x.double().cpu(), qe], 1) @ W.t() + b.double().cpu()
        if act == ACT_RELU:
            want = torch.relu(want)
        check(full, want, 3e-6, 'expanded form')
        check(lazy, want, 3e-6, 'row-broadcast form')


def test_check_finite_debug_mode():
    """EKAID_CHECK_FINITE=1 (ops.CHECK_FINITE): a NaN / Inf in a layer output raises AssertionError naming the layer, like the
    reference's per-layer `assert not x.isnan().any()` (fc.py:42-48); clean tensors pass; off by default."""
    from ekaid_amd import ops
    x = rnd(64, 32, seed=1).to(DEV)
    W = rnd(16, 32, seed=2).to(DEV)
    was = ops.CHECK_FINITE
    try:
        ops.CHECK_FINITE = True
        ops.linear(x, W)
        bad = x.clone()
        bad[3, 5] = float('nan')
        with pytest.raises(AssertionError, match='linear layer'):
            ops.linear(bad, W)
        ops.assert_finite(x, 'clean')
        with pytest.raises(AssertionError, match='2 non-finite'):
            t = x.clone()
            t[0, 0], t[1, 1] = float('inf'), float('-inf')
            ops.assert_finite(t, 'probe')
        ops.CHECK_FINITE = False
        ops.linear(bad, W)                  # off: no check, no synchronisation
    finally:
        ops.CHECK_FINITE = was


@pytest.mark.parametrize('mode', ['nt', 'nn', 'tn'])
@pytest.mark.parametrize('M,N,K', [(256, 2048, 512), (4608, 1024, 2048), (700, 132, 336), (36, 36, 256), (2304, 1024, 1024), (1024, 1024, 2304)])
def test_gemm_f16x2_variants(M, N, K, mode):
    """EKAID_GEMM_F16X2 over the tile shapes, K splits and epilogues the planner picks (64x64, 128x128, split-K + reduce, ragged M / N,
    bias + activation + accumulate, the A-row-sum of the weight-gradient form, two products in one launch), with operands whose rows
    span 2^+-20, a zero row, and a product WITHOUT bounds in the same mode (it must run on the exact split): <= 5e-7 of the element's
    condition scale against float64, like the exact split."""
    from ekaid_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.pow(2.0, torch.randint(-20, 21, (M, 1), generator=g).float())
    Bm = torch.randn(K, N, generator=g) * torch.pow(2.0, torch.randint(-20, 21, (1, N), generator=g).float())
    A[M // 2] = 0
    bias = torch.randn(N, generator=g)
    want = A.double() @ Bm.double()
    mag = A.double().abs() @ Bm.double().abs() + 1e-300
    if mode == 'nt':
        a, b, ta, tb, lda, ldb = A, Bm.t().contiguous(), 0, 1, K, K
    elif mode == 'nn':
        a, b, ta, tb, lda, ldb = A, Bm, 0, 0, K, N
    else:
        a, b, ta, tb, lda, ldb = A.t().contiguous(), Bm, 1, 0, M, N
    a, b, bias_d = a.to(DEV), b.to(DEV), bias.to(DEV)
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(ops.GEMM_F16X2)
    try:
        kw = _bounds_kw(a, b, ta, tb)
        C = torch.full((M, N), float('nan'), device=DEV)
        ran = _gemm_counted(a, b, C, M, N, K, lda, ldb, ta, tb, **kw)
        assert ran['gemm_split2'] == 1 and ran['gemm_split3'] == 0, ran
        assert float(((C.double().cpu() - want).abs() / mag).max()) <= 5e-7
        assert float(C[M // 2].abs().max()) == 0.0                                     # the zero row stays exactly zero
        C2 = torch.full((M, N), float('nan'), device=DEV)                              # no bounds: the exact split, same mode
        ran = _gemm_counted(a, b, C2, M, N, K, lda, ldb, ta, tb)
        assert ran['gemm_split3'] == 1 and ran['gemm_split2'] == 0, ran
        assert float(((C2.double().cpu() - want).abs() / mag).max()) <= 5e-7
        # epilogue: relu(0.5 * A.B + C0 + bias), the scales must leave before it
        C0 = torch.randn(M, N, generator=g) * mag.float() * 0.1
        C3 = C0.clone().to(DEV)
        ops.gemm(a, b, C3, M, N, K, lda, ldb, N, ta, tb, alpha=0.5, accumulate=True, bias=bias_d, act=ops.ACT_RELU, **kw)
        w3 = torch.relu(0.5 * want + C0.double() + bias.double())
        assert float(((C3.double().cpu() - w3).abs() / (mag + C0.double().abs() + bias.double().abs())).max()) <= 1e-6
        if ta:                                                                         # bias gradient riding on the A operand (unscaled sums)
            rs = torch.zeros(M, device=DEV)
            C4 = torch.empty(M, N, device=DEV)
            ops.gemm(a, b, C4, M, N, K, lda, ldb, N, ta, tb, a_rowsum=rs, **kw)
            assert torch.equal(C4, C)
            sw = A.double().sum(1)
            assert float((rs.double().cpu() - sw).abs().max() / A.double().abs().sum(1).max()) <= 1e-6
        # two products in one launch = the two alone, bit for bit
        Ca, Cb = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        b2 = (b * 0.75).contiguous()
        kw2 = _bounds_kw(a, b2, ta, tb)
        ops.gemm(a, b2, Cb, M, N, K, lda, ldb, N, ta, tb, **kw2)
        want_b = Cb.clone()
        Ca.fill_(float('nan'))
        Cb.fill_(float('nan'))
        ops.gemm_pair(((a, b, Ca, M, N, K, lda, ldb, N, ta, tb), dict(kw)), ((a, b2, Cb, M, N, K, lda, ldb, N, ta, tb), dict(kw2)))
        assert torch.equal(Ca, C) and torch.equal(Cb, want_b)
    finally:
        ops.set_gemm_mode(old)


@pytest.mark.parametrize('multi', [True, False])
def test_weight_norm_backward_of_gradients_outside_their_flat_slots(multi, monkeypatch):
    """the deferred weight-norm backward (ekaid_wn_grad_multi at the end of the pass) keeps the raw gradient by ADDRESS, which
    is only sound for a gradient that lives in its flat-buffer slot.  The cases where ``grad_out`` hands out a temporary -- a second
    backward without zero_grad (p.grad already set: gradient accumulation), a weight-normed layer used twice in one step, no optimizer
    attached at all -- must be transformed inside their node.  All three against the weight_norm(dim=None) formula in float64, under
    both settings of the A/B knob."""
    from ekaid_amd import ops
    from ekaid_amd.dp import FlatAdam
    from ekaid_amd.models.fc import FCNet
    monkeypatch.setattr(ops, '_WN_MULTI', multi)
    g = torch.Generator().manual_seed(3)
    fa, fb = FCNet([96, 64], dropout=0).to(DEV), FCNet([64, 64], dropout=0).to(DEV)
    with torch.no_grad():
        for p in list(fa.parameters()) + list(fb.parameters()):
            p.copy_(torch.randn(p.shape, generator=g).to(DEV) * (0.3 if p.dim() > 0 else 1.0))
            if p.dim() == 0:
                p.abs_().add_(0.5)                                                   # weight_g: a positive magnitude
    x1, x2 = torch.randn(40, 96, generator=g).to(DEV), torch.randn(40, 96, generator=g).to(DEV)

    def loss_of(xa, xb):                       # fb (and fa) used TWICE in one step
        return (fb(fa(xa)) * 0.7).sum() + (fb(fa(xb)) ** 2).sum() * 0.1

    def want(times):
        P = {k: v.detach().double().cpu().requires_grad_(True) for k, v in list(fa.named_parameters()) + [('b.' + k, v) for k, v in fb.named_parameters()]}

        def layer(pre, x):
            v, gm, b = P[pre + 'main.0.weight_v'], P[pre + 'main.0.weight_g'], P[pre + 'main.0.bias']
            return torch.relu(x @ (v * (gm / v.norm())).t() + b)
        tot = 0.0
        for _ in range(times):
            ya, yb = layer('b.', layer('', x1.double().cpu())), layer('b.', layer('', x2.double().cpu()))
            tot = tot + (ya * 0.7).sum() + (yb ** 2).sum() * 0.1
        tot.backward()
        return {k: v.grad for k, v in P.items()}

    def got():
        return {**{k: p.grad for k, p in fa.named_parameters()}, **{'b.' + k: p.grad for k, p in fb.named_parameters()}}

    def compare(times, what):
        torch.cuda.synchronize()
        w, h = want(times), got()
        for k in w:
            assert h[k] is not None, (what, k)
            check(h[k], w[k], 2e-5, '%s: grad %s' % (what, k))
    # (1) no optimizer: every gradient buffer is a temporary
    loss_of(x1, x2).backward()
    compare(1, 'no slots')
    # (2) flat slots (FlatAdam discovers the layout on a first backward), the layers used twice: first use in the slot, second a temporary
    opt = FlatAdam([fa, fb], lr=0.0)
    opt.zero_grad()
    loss_of(x1, x2).backward()
    opt.step()
    opt.zero_grad()
    loss_of(x1, x2).backward()
    compare(1, 'slots, shared layers')
    # (3) a second backward WITHOUT zero_grad: p.grad is set, every buffer is a temporary that AccumulateGrad adds and frees
    loss_of(x1, x2).backward()
    compare(2, 'accumulation into set gradients')
    ops.clear_grad_claims()


def test_bounds_kept_for_backward_survive_another_forward_pass(monkeypatch):
    """The column maxima of a step's bounds passes live in one pool that the detector's forward clears (ops.bounds_pool_reset).  A layer
    keeps the bounds of x and W for its backward pass; if ANOTHER forward pass recycles the pool in between (two forward passes, then
    the first one's backward), the kept vectors hold the second pass's maxima -- a million times too small here, i.e. f16 overflow -- unless
    the backward notices the pool's generation and takes them afresh."""
    from ekaid_amd import hip, ops
    monkeypatch.setattr(ops, 'F16X2_MIN_WORK', 0)
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(ops.GEMM_F16X2)
    dev = torch.device(DEV, torch.cuda.current_device())
    try:
        g = torch.Generator().manual_seed(9)
        W = (torch.randn(256, 512, generator=g) * 0.05).to(DEV).requires_grad_(True)
        b = torch.zeros(256, device=DEV, requires_grad=True)
        x1 = torch.randn(1024, 512, generator=g).to(DEV).requires_grad_(True)
        x2 = (x1.detach() * 1e-6).requires_grad_(True)
        ops.bounds_pool_reset(dev)
        c0 = hip.launch_counts()
        y1 = ops.linear(x1, W, b)
        ops.bounds_pool_reset(dev)                         # "the next forward pass begins"
        y2 = ops.linear(x2, W * 1e-6, b)
        (y1 * torch.linspace(-1, 1, 256, device=DEV)).sum().backward()
        c1 = hip.launch_counts()
        assert c1['gemm_split2'] - c0['gemm_split2'] >= 4          # both forward products, the input- and the weight-gradient product
        dy = torch.linspace(-1, 1, 256).double().expand(1024, 256)
        check(W.grad, dy.t() @ x1.detach().double().cpu(), 1e-5, 'dW after another forward pass recycled the bounds pool')
        check(x1.grad, dy @ W.detach().double().cpu(), 1e-5, 'dx')
        assert torch.isfinite(y2).all()
    finally:
        ops.set_gemm_mode(old)
