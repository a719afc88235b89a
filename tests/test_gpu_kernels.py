"""GPU parity of every C-ABI entry point against the canonical oracle: bit-exact (==) on all
forward tensors, on seeded inputs at sizes the oracle finishes in seconds."""
import numpy as np
import pytest
import torch

from oracle import lipvq_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    import lipvq_vae_amd
    return lipvq_vae_amd.ops


def test_math_device_equals_host(ops, oracle):
    """lq_gelu/lq_sigmoid evaluated on the GPU (through a 1-layer-wide MLP is awkward; use the
    Lipschitz kernel for softplus and the MLP with identity-like weights for gelu/sigmoid)."""
    # softplus + division via lipschitz_scale: W = 1 row of H ones -> scale = min(1, softplus(ci)/H)
    ci = np.linspace(-30, 30, 4001).astype(np.float32)
    W = np.ones((ci.size, 4), np.float32)
    sc_ref, wn_ref = oracle.lipschitz_scale(W, ci)
    sc, wn = ops.lipschitz_scale(dev(W), dev(ci))
    assert np.array_equal(sc.cpu().numpy(), sc_ref)
    assert np.array_equal(wn.cpu().numpy(), wn_ref)


@pytest.mark.parametrize("N,K0,J0,J1,J2,acts", [
    (1000, 7, 64, 128, 64, (O.ACT_GELU, O.ACT_GELU, O.ACT_SIGMOID)),
    (77, 12, 64, 128, 208, (O.ACT_GELU, O.ACT_GELU, O.ACT_SIGMOID)),
    (513, 64, 64, 128, 7, (O.ACT_GELU, O.ACT_GELU, O.ACT_NONE)),
    (80, 208, 64, 128, 12, (O.ACT_GELU, O.ACT_GELU, O.ACT_NONE)),
    (300, 7, 64, 128, 32, (O.ACT_RELU, O.ACT_RELU, O.ACT_RELU)),
    (300, 32, 128, 64, 7, (O.ACT_RELU, O.ACT_RELU, O.ACT_RELU)),
    (1, 7, 64, 128, 32, (O.ACT_GELU, O.ACT_GELU, O.ACT_SIGMOID)),
    (33, 5, 32, 32, 3, (O.ACT_GELU, O.ACT_RELU, O.ACT_NONE)),
])
def test_mlp3_bit_exact(ops, oracle, N, K0, J0, J1, J2, acts):
    rng = np.random.default_rng(N * 7 + K0)
    W0 = rng.standard_normal((J0, K0)).astype(np.float32) * 0.5
    W1 = rng.standard_normal((J1, J0)).astype(np.float32) * 0.2
    W2 = rng.standard_normal((J2, J1)).astype(np.float32) * 0.2
    b0, b1, b2 = (rng.standard_normal(J).astype(np.float32) for J in (J0, J1, J2))
    x = rng.standard_normal((N, K0)).astype(np.float32)
    y_ref, pre_ref = oracle.mlp3(x, W0, b0, W1, b1, W2, b2, acts, save_pre=True)
    packed = ops.mlp3_pack(*(dev(a) for a in (W0, b0, W1, b1, W2, b2)))
    y, pre = ops.mlp3(dev(x), packed, acts, save_pre=True)
    torch.cuda.synchronize()
    for got, ref, name in ((pre[0], pre_ref[0], "pre0"), (pre[1], pre_ref[1], "pre1"), (pre[2], pre_ref[2], "pre2"),
                           (y, y_ref, "y")):
        g = got.cpu().numpy()
        bad = np.argwhere(g != ref)
        assert bad.size == 0, f"{name}: {len(bad)} mismatches, first {bad[:3].tolist()} got {g[tuple(bad[0])]} ref {ref[tuple(bad[0])]}"
    # without saving
    y2 = ops.mlp3(dev(x), packed, acts)
    assert torch.equal(y2, y)


def test_mlp3_gather(ops, oracle):
    rng = np.random.default_rng(5)
    K, D, N = 50, 64, 333
    table = rng.uniform(0, 1, (K, D)).astype(np.float32)
    idx = rng.integers(0, K, N).astype(np.int64)
    W0 = rng.standard_normal((64, D)).astype(np.float32) * 0.2
    W1 = rng.standard_normal((128, 64)).astype(np.float32) * 0.2
    W2 = rng.standard_normal((7, 128)).astype(np.float32) * 0.2
    b0, b1, b2 = (rng.standard_normal(J).astype(np.float32) for J in (64, 128, 7))
    acts = (O.ACT_GELU, O.ACT_GELU, O.ACT_NONE)
    y_ref = oracle.mlp3(table[idx], W0, b0, W1, b1, W2, b2, acts)
    packed = ops.mlp3_pack(*(dev(a) for a in (W0, b0, W1, b1, W2, b2)))
    y = ops.mlp3(dev(table), packed, acts, gather_idx=dev(idx))
    assert np.array_equal(y.cpu().numpy(), y_ref)


@pytest.mark.parametrize("N,J,Kd,act,gather", [
    (5000, 128, 64, O.ACT_GELU, False), (70000, 64, 128, O.ACT_GELU, False), (33000, 64, 7, O.ACT_NONE, False),
    (4097, 7, 128, O.ACT_GELU, False), (20000, 64, 64, O.ACT_NONE, True), (80, 208, 128, O.ACT_GELU, False),
    # a gradient operand wider than 224 columns: column blocks of 128 through the four-tile kernel (the embedding Linear's 512)
    (9000, 512, 64, O.ACT_NONE, False), (40000, 256, 64, O.ACT_RELU, False), (3000, 384, 32, O.ACT_NONE, False),
    (2500, 320, 64, O.ACT_NONE, False),                            # 320 % 128 != 0: the per-tile kernel
])
def test_wgrad_against_float64(ops, N, J, Kd, act, gather):
    """gW = G^T act(H), gb = column sums of G, for every kernel family behind lipvq_wgrad_f32, against float64 torch."""
    g = torch.Generator(device="cuda").manual_seed(N + J)
    G = torch.randn(N, J, device="cuda", generator=g)
    if gather:
        table = torch.randn(300, Kd, device="cuda", generator=g)
        hidx = torch.randint(0, 300, (N,), device="cuda", generator=g)
        gW, gb = ops.wgrad(G, table, h_act=act, hidx=hidx)
        Hd = table[hidx].double()
    else:
        H = torch.randn(N, Kd, device="cuda", generator=g)
        gW, gb = ops.wgrad(G, H, h_act=act)
        Hd = H.double()
    if act == O.ACT_GELU:
        Hd = torch.nn.functional.gelu(Hd)
    elif act == O.ACT_RELU:
        Hd = torch.relu(Hd)
    refW, refb = G.double().t() @ Hd, G.double().sum(0)
    assert float((gW.double() - refW).abs().max()) <= 2e-5 * float(refW.abs().max())
    assert float((gb.double() - refb).abs().max()) <= 2e-5 * max(float(refb.abs().max()), N ** 0.5)


LDS_ROWS = 65536          # csrc/lipvq_mlp.hip: from this many rows on the LDS-resident persistent kernel runs the stack


@pytest.mark.parametrize("K0,J0,J1,J2,acts,gather", [
    (64, 64, 128, 7, (O.ACT_GELU, O.ACT_GELU, O.ACT_NONE), True),          # the decoder, fed by codebook rows
    (7, 64, 128, 64, (O.ACT_GELU, O.ACT_GELU, O.ACT_SIGMOID), False),      # the encoder
    (208, 64, 128, 12, (O.ACT_GELU, O.ACT_GELU, O.ACT_NONE), False),       # D = 208 decoder
    (12, 64, 128, 208, (O.ACT_GELU, O.ACT_GELU, O.ACT_SIGMOID), False),    # D = 208 encoder: 7 output tiles
    (33, 128, 64, 5, (O.ACT_RELU, O.ACT_GELU, O.ACT_RELU), False),         # odd fan-in, two input slices
])
def test_mlp3_large_batch_kernel_bit_exact(ops, oracle, K0, J0, J1, J2, acts, gather):
    """The persistent weights-in-LDS kernel (N >= 65 536) against the canonical oracle, ragged last tile included."""
    N = LDS_ROWS + 45
    rng = np.random.default_rng(K0 * 31 + J2)
    W0 = rng.standard_normal((J0, K0)).astype(np.float32) * 0.4
    W1 = rng.standard_normal((J1, J0)).astype(np.float32) * 0.2
    W2 = rng.standard_normal((J2, J1)).astype(np.float32) * 0.2
    b0, b1, b2 = (rng.standard_normal(J).astype(np.float32) for J in (J0, J1, J2))
    packed = ops.mlp3_pack(*(dev(a) for a in (W0, b0, W1, b1, W2, b2)))
    if gather:
        table = rng.uniform(-1, 1, (300, K0)).astype(np.float32)
        idx = rng.integers(0, 300, N).astype(np.int64)
        x = table[idx]
        y, pre = ops.mlp3(dev(table), packed, acts, gather_idx=dev(idx), save_pre=True)
    else:
        x = rng.standard_normal((N, K0)).astype(np.float32) * 1.5
        x[5, :] = 40.0                                    # pre-activations beyond the GELU polynomial's range
        y, pre = ops.mlp3(dev(x), packed, acts, save_pre=True)
    y_ref, pre_ref = oracle.mlp3(x, W0, b0, W1, b1, W2, b2, acts, save_pre=True)
    for got, ref, name in ((pre[0], pre_ref[0], "pre0"), (pre[1], pre_ref[1], "pre1"), (pre[2], pre_ref[2], "pre2"), (y, y_ref, "y")):
        assert np.array_equal(got.cpu().numpy(), ref), name
    y2 = ops.mlp3(dev(table), packed, acts, gather_idx=dev(idx)) if gather else ops.mlp3(dev(x), packed, acts)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("K0,J0,J1,J2,acts,want_gx", [
    (7, 64, 128, 64, (O.ACT_GELU, O.ACT_GELU, O.ACT_SIGMOID), False),      # encoder backward (no d/d input)
    (64, 64, 128, 7, (O.ACT_GELU, O.ACT_GELU, O.ACT_NONE), True),          # decoder backward
    (208, 64, 128, 12, (O.ACT_GELU, O.ACT_GELU, O.ACT_NONE), True),
    (12, 64, 128, 208, (O.ACT_GELU, O.ACT_GELU, O.ACT_SIGMOID), True),
])
def test_mlp3_bwd_large_batch_kernel_equals_small_batch_kernel(ops, K0, J0, J1, J2, acts, want_gx):
    """Backward-data of N >= 65 536 rows (persistent weights-in-LDS kernel) == the same rows sent in pieces below that size
    (workgroup-per-tile kernel, which the training fixtures pin): the same chains, the same bits."""
    N = LDS_ROWS + 77
    g = torch.Generator(device="cuda").manual_seed(K0 + J2)
    W0 = torch.randn(J0, K0, device="cuda", generator=g) * 0.3
    W1 = torch.randn(J1, J0, device="cuda", generator=g) * 0.2
    W2 = torch.randn(J2, J1, device="cuda", generator=g) * 0.2
    pk = ops.mlp3_pack_bwd(W0, W1, W2)
    gy = torch.randn(N, J2, device="cuda", generator=g)
    pre = [torch.randn(N, J, device="cuda", generator=g) * 2.0 for J in (J0, J1, J2)]
    pre[1][3, :] = 9.0                                     # beyond the straight-line GELU' range
    if acts[2] == O.ACT_NONE:
        pre[2] = None
    big = ops.mlp3_bwd(gy, pre, pk, acts, want_gx=want_gx)
    cut = 40000
    parts = [ops.mlp3_bwd(gy[a:b].contiguous(), [None if p is None else p[a:b].contiguous() for p in pre], pk, acts, want_gx=want_gx)
             for a, b in ((0, cut), (cut, N))]
    for i, name in enumerate(("g2", "g1", "g0", "gx")):
        if big[i] is None:
            assert parts[0][i] is None
            continue
        assert torch.equal(big[i], torch.cat([parts[0][i], parts[1][i]])), name


@pytest.mark.parametrize("N,K,D,dist", [
    (1000, 256, 32, O.DIST_NORM), (777, 1024, 64, O.DIST_NORM), (300, 1000, 128, O.DIST_NORM),
    (80, 1024, 208, O.DIST_NORM), (100, 37, 24, O.DIST_NORM), (64, 50, 7, O.DIST_NORM),
    (500, 128, 32, O.DIST_SQSUM), (300, 512, 64, O.DIST_SQSUM), (90, 100, 208, O.DIST_SQSUM),
    (1, 5, 64, O.DIST_NORM),
])
def test_nearest_bit_exact(ops, oracle, N, K, D, dist):
    rng = np.random.default_rng(N + K + D)
    cb = rng.uniform(0, 1, (K, D)).astype(np.float32)
    z = rng.uniform(0, 1, (N, D)).astype(np.float32)
    z[: min(N, K) // 2] = cb[rng.permutation(K)[: min(N, K) // 2]] + (0.01 * rng.standard_normal((min(N, K) // 2, D))).astype(np.float32)
    if K > 3:
        cb[K - 1] = cb[1]                      # duplicate code: lowest index must win
    idx_ref, zq_ref, usage_ref, best_ref = oracle.nearest(z, cb, dist, want_best=True)
    usage = torch.zeros(K, dtype=torch.int64, device="cuda")
    idx, zq, best = ops.nearest(dev(z), dev(cb), dist, usage=usage, want_best=True)
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(zq.cpu().numpy(), zq_ref)
    assert np.array_equal(best.cpu().numpy(), best_ref)
    assert np.array_equal(usage.cpu().numpy(), usage_ref)


def test_nearest_golden_edge(ops, golden_dir):
    g = np.load(golden_dir / "llfq_nearest_edge.npz")
    idx, zq, best = ops.nearest(dev(g["z_e"]), dev(g["codebook"]), O.DIST_NORM, want_best=True)
    assert np.array_equal(idx.cpu().numpy(), g["indices"].astype(np.int64))
    assert np.array_equal(zq.cpu().numpy(), g["z_q"])
    d = g["distances"]
    assert np.array_equal(best.cpu().numpy(), d[np.arange(d.shape[0]), g["indices"]])


def test_ste_and_mse(ops, oracle):
    rng = np.random.default_rng(3)
    ze = rng.uniform(0, 1, (1000, 64)).astype(np.float32)
    zq = rng.uniform(0, 1, (1000, 64)).astype(np.float32)
    xr = rng.standard_normal((1000, 7)).astype(np.float32)
    x = rng.standard_normal((1000, 7)).astype(np.float32)
    assert np.array_equal(ops.ste(dev(ze), dev(zq)).cpu().numpy(), oracle.ste(ze, zq))
    a, b = oracle.mse_pair(xr, x, zq, ze)
    out = ops.mse_pair(dev(xr), dev(x), dev(zq), dev(ze)).cpu().numpy()
    # means are order-dependent sums: tolerance 1e-6 relative (double accumulation on both sides)
    assert abs(out[0] - a) <= 1e-6 * abs(a) and abs(out[1] - b) <= 1e-6 * abs(b)
    # the loss built on the device from the two means: the reference's fp32 association, bit for bit
    for w, form in ((0.25, ops.LOSS_LLFQ), (0.25, ops.LOSS_VQ), (0.37, ops.LOSS_VQ)):
        o3 = ops.mse_pair_loss(dev(xr), dev(x), dev(zq), dev(ze), w, form)
        m0, m1 = o3[0], o3[1]
        want = (m0 + m1 * w) + m1 * w if form == ops.LOSS_LLFQ else m0 + (m1 + w * m1)
        assert torch.equal(o3[:2].cpu(), torch.from_numpy(out)) and torch.equal(o3[2], want)


@pytest.mark.parametrize("N,K,D", [(5000, 37, 64), (80, 1024, 208), (70001, 256, 32), (40000, 1024, 208), (33001, 5000, 7)])
def test_scatter_add_deterministic_is_sequential_fp32(N, K, D):
    """Deterministic mode equals a sequential fp32 index_add_ over rows 0..N-1, bit for bit, on every run -- both the
    scanning kernel (lipvq_scatter_add_det_f32) and, from 32 768 rows on, the counting-sort route in sequential mode."""
    from lipvq_vae_amd import ops
    rng = np.random.default_rng(N)
    g = rng.standard_normal((N, D)).astype(np.float32)
    idx = rng.integers(0, K, N).astype(np.int64)
    idx[: N // 3] = 5                                   # one heavily used code: long ordered chains
    want = np.zeros((K, D), np.float32)
    for n in range(N):                                  # fp32 adds in row order
        want[idx[n]] += g[n]
    gt, it = torch.from_numpy(g).cuda(), torch.from_numpy(idx).cuda()
    a = ops.scatter_add(gt, it, K, deterministic=True)
    b = ops.scatter_add(gt, it, K, deterministic=True)
    assert torch.equal(a, b) and np.array_equal(a.cpu().numpy(), want)
    assert np.array_equal(ops.scatter_add(gt, it, K, route="sequential_scan").cpu().numpy(), want)
    if N >= 32768:
        assert np.array_equal(ops.scatter_add(gt, it, K, route="sequential_sorted").cpu().numpy(), want)
    c = ops.scatter_add(gt, it, K, deterministic=False).cpu().numpy()          # atomics / segments: same sum up to fp32 ordering
    assert np.abs(c - want).max() <= 1e-4 * (1 + np.abs(want).max())


@pytest.mark.parametrize("N,K,D,skew", [
    (32768 + 13, 1024, 64, False), (70001, 37, 64, True), (262144, 1024, 208, True), (40000, 2048, 32, False),
    (65536, 1000, 7, True), (100000, 8192, 128, True), (50000, 16384, 16, False), (33000, 2049, 64, True),
])
def test_scatter_add_sorted_route(N, K, D, skew):
    """Counting-sort route of large batches (csrc/lipvq_scatter.hip): equals a float64 index_add_ to fp32 summation accuracy, is
    bit-identical run after run (no atomics), equals the strictly sequential kernel bit for bit on codes whose rows fit one
    256-row segment; codes without rows stay zero; hot codes span many segments."""
    from lipvq_vae_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(N + K)
    g = torch.randn(N, D, device="cuda", generator=gen)
    if skew:                                                   # a few hot codes, many empty ones
        w = torch.rand(K, device="cuda", generator=gen) ** 8
        w[K // 2:] *= (torch.rand(K - K // 2, device="cuda", generator=gen) > 0.5)
        idx = torch.multinomial(w / w.sum(), N, replacement=True, generator=gen)
    else:
        idx = torch.randint(0, K, (N,), device="cuda", generator=gen)
    ref = torch.zeros(K, D, device="cuda", dtype=torch.float64).index_add_(0, idx, g.double())
    a = ops.scatter_add(g, idx, K, deterministic=False, route="sorted")
    b = ops.scatter_add(g, idx, K, deterministic=False, route="sorted")
    assert torch.equal(a, b)
    counts = torch.bincount(idx, minlength=K)
    scale = (g.abs().max() * counts.max().clamp(min=1).sqrt()).item()
    assert float((a.double() - ref).abs().max()) <= 2e-6 * max(scale, float(ref.abs().max()))
    assert torch.all(a[counts == 0] == 0)
    # codes whose rows fit one segment are summed in plain ascending row order: the same bits as the sequential kernel
    seq = ops.scatter_add(g, idx, K, deterministic=True)
    small = (counts > 0) & (counts <= 256)
    assert torch.equal(a[small], seq[small])
    if N >= 65536:
        assert torch.equal(ops.scatter_add(g, idx, K, deterministic=False), a)    # the default route of a large batch


def test_training_gradients_are_reproducible_in_deterministic_mode(oracle):
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    A, D, K, N = 7, 64, 256, 4000
    p = O.make_params(3, A, D, K, oracle=oracle)
    x = torch.from_numpy(O.make_inputs(3, N, A)).cuda()
    grads = []
    prev = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True)
    try:
        for _ in range(2):
            m = LLFQVAE_V4(A, D, num_codes=K).cuda()
            m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
            _, loss = m(x)
            loss.backward()
            grads.append([q.grad.clone() for q in m.parameters()])
    finally:
        torch.use_deterministic_algorithms(prev)
    assert all(torch.equal(a, b) for a, b in zip(*grads))


def test_gelu_derivative_on_device_tracks_the_canonical_one(ops, oracle):
    """lipvq_act_bwd_f32 evaluates the GELU derivative in a straight-line form on the device (lq_gelu_grad_dev); the oracle
    runs lq_gelu_grad (general erf / exp branches).  Dense sweep incl. the hand-over at |x| = sqrt(18), large |x|, 0, NaN."""
    x = np.concatenate([np.linspace(-12, 12, 200001), [4.2426405, 4.2426410, -4.2426405, 0.0, 1e-30, 88.0, -88.0]]).astype(np.float32)
    ref = oracle.math_probe(x, 5)                                  # lq_gelu_grad
    got = ops.act_bwd(torch.ones(x.size, 1, device="cuda"), dev(x.reshape(-1, 1)), O.ACT_GELU).cpu().numpy().ravel()
    assert np.abs(got - ref).max() <= 3e-7, np.abs(got - ref).max()
    nan = ops.act_bwd(torch.ones(1, 1, device="cuda"), torch.full((1, 1), float("nan"), device="cuda"), O.ACT_GELU)
    assert torch.isnan(nan).all()
