"""Dev measurement (GPU): the default action branch (DefaultActionNetwork) against the stock torch modules the reference
builds (obs_nets.py:1244-1260), same parameters, same GPU: eval forward and a training step (forward + backward)."""
import sys, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.default_branch import DefaultActionNetwork, GraphedDefaultBranch


def stock(A, D):
    layer = nn.TransformerEncoderLayer(d_model=D, nhead=8, dim_feedforward=256, activation="gelu")
    return nn.Sequential(spectral_norm(nn.Linear(A, 64)), nn.GELU(), spectral_norm(nn.Linear(64, 128)), nn.GELU(),
                         spectral_norm(nn.Linear(128, D)), nn.TransformerEncoder(layer, num_layers=4), nn.Linear(D, D))


def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


warnings.simplefilter("ignore")
for N, A, D in ((80, 12, 208), (500, 12, 208), (2048, 7, 64)):
    torch.manual_seed(0)
    ours = DefaultActionNetwork(A, D).cuda()
    ref = stock(A, D).cuda()
    ref.load_state_dict(ours.state_dict())
    x = torch.randn(N, A, device="cuda")
    ours.eval(); ref.eval()
    with torch.no_grad():
        d = (ours(x) - ref(x)).abs().max().item() / ref(x).abs().max().item()
        t_o, t_r = timed(lambda: ours(x)), timed(lambda: ref(x))
        gr = GraphedDefaultBranch(ours, x)
        dg = (gr(x) - ours(x)).abs().max().item()
        t_g = timed(lambda: gr(x))
    ours.train(); ref.train()
    def step(m):
        for p in m.parameters(): p.grad = None
        m(x).square().mean().backward()
    s_o, s_r = timed(lambda: step(ours), 20), timed(lambda: step(ref), 20)
    print(f"N={N} A={A} D={D}: eval forward {t_o:.3f} ms, HIP-graph replay {t_g:.3f} ms (== eager: {dg == 0.0}) (stock torch {t_r:.3f} ms, max rel diff {d:.1e}); "
          f"train forward+backward {s_o:.3f} ms (stock torch {s_r:.3f} ms)")
