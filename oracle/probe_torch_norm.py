"""Development probe (not part of any product or test path): establishes empirically which
summation order torch's CPU kernels use for `torch.norm(x, dim=-1)` (reference
backbone_lfqvae_v5.py:43-45) and for `x.pow(2).sum(-1)` (reference backbone.py:58-60), by
comparing them bit for bit with candidate orders written in numpy.  Findings (torch
2.10.0 CPU, AVX512 host; the same kernels are used on AVX2):
  norm : 8 accumulators acc[j] = fma(d[8i+j], d[8i+j], acc[j]); lanes added left to right; sqrt.
  sum  : 4x8 accumulators over 32-wide chunks, left-over 8-vectors into accumulator 0,
         accumulators added left to right, then lanes left to right.
lipvq-vae_amd/csrc/lipvq_math.h (lq_sqdist8 / lq_sqdist32) restates exactly these orders.
Uses torch only; does not touch the reference.
"""
import torch, numpy as np
torch.manual_seed(0)
def cand(diff, L, fma, tail_seq=True):
    # diff: [M, D] float32 numpy. L lanes; acc[j] += d*d sequentially over chunks; then buffer[0]+=buffer[j] seq.
    M, D = diff.shape
    nfull = D - D % L
    acc = np.zeros((M, L), np.float32)
    for i in range(0, nfull, L):
        d = diff[:, i:i+L]
        if fma:
            acc = (acc.astype(np.float64) + d.astype(np.float64)*d.astype(np.float64)).astype(np.float32)  # fma: single rounding (double has enough bits: 24+24=48<53, plus add -> may double round? acc+prod exact in f64? not always but nearly)
        else:
            acc = acc + d*d
    s = acc[:, 0].copy()
    for j in range(1, L):
        s = s + acc[:, j]
    for i in range(nfull, D):
        d = diff[:, i]
        s = s + d*d   # non-fma tail
    return s
for D in (64, 32, 128, 208, 7):
    z = torch.rand(300, D); c = torch.rand(257, D)
    diff = (z.unsqueeze(1) - c.unsqueeze(0))
    ref = torch.norm(diff, dim=-1).numpy().reshape(-1)
    dn = diff.numpy().reshape(-1, D)
    print("D", D)
    for L in (4, 8, 16, 32):
        for fma in (False, True):
            s = cand(dn, L, fma)
            r = np.sqrt(s)
            print("  L", L, "fma", fma, "mismatch", int((r != ref).sum()), "of", r.size)
    # sequential
    s = np.zeros(dn.shape[0], np.float32)
    for i in range(D): s = s + dn[:, i]*dn[:, i]
    print("  seq mismatch", int((np.sqrt(s) != ref).sum()))


# ---- second probe: pow(2).sum(-1) tail handling ----
def cand(sq, variant):
    M, D = sq.shape; L=8; nacc=4; W=32
    nfull = D - D % W
    acc = np.zeros((M, nacc, L), np.float32)
    for i in range(0, nfull, W):
        acc = acc + sq[:, i:i+W].reshape(M, nacc, L)
    rem = list(range(nfull, D - D % L, L))
    if variant == 'rem_into_acc':
        for a, i in enumerate(rem): acc[:, a] = acc[:, a] + sq[:, i:i+L]
        v = acc[:,0]
        for a in range(1,nacc): v = v + acc[:,a]
    elif variant == 'rem_into_acc0':
        for i in rem: acc[:, 0] = acc[:, 0] + sq[:, i:i+L]
        v = acc[:,0]
        for a in range(1,nacc): v = v + acc[:,a]
    elif variant == 'comb_then_rem':
        v = acc[:,0]
        for a in range(1,nacc): v = v + acc[:,a]
        for i in rem: v = v + sq[:, i:i+L]
    s = v[:,0].copy()
    for j in range(1,L): s = s + v[:,j]
    for i in range(D - D % L, D): s = s + sq[:, i]
    return s
for D in (208, 40, 48, 72, 100, 12):
    z = torch.rand(200, D); c = torch.rand(129, D)
    sq = (z.unsqueeze(1) - c).pow(2)
    ref = sq.sum(-1).numpy().reshape(-1); sqn = sq.numpy().reshape(-1, D)
    for v in ('rem_into_acc','rem_into_acc0','comb_then_rem'):
        print(D, v, int((cand(sqn, v) != ref).sum()))
