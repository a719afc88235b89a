import sys, copy
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import lipvq_vae_amd
from lipvq_vae_amd.icl import GraphedTokenizerStep, VQTokenizerTrainer
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from lipvq_vae_amd.autograd import forward_backward
from oracle import lipvq_oracle as O
orc = O.CanonicalOracle()
A, D, K, N = 12, 208, 1024, 80
p = O.make_params(77, A, D, K, oracle=orc)
def mk():
    m = LLFQVAE_V4(A, D, num_codes=K).cuda()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    return m
xs = [torch.from_numpy(O.make_inputs(300 + i, N, A)).cuda() for i in range(8)]
# path 1: trainer eager, 6 steps
m1 = mk(); t1 = VQTokenizerTrainer(m1)
l1 = [float(t1.train_on_actions(xs[i])[1]) for i in (0, 1, 2, 3, 3, 4)]
# path 2: 3 trainer steps, then forward_backward + capturable AdamW eager (no graph)
m2 = mk(); t2 = VQTokenizerTrainer(m2)
l2 = [float(t2.train_on_actions(xs[i])[1]) for i in (0, 1, 2)]
opt = torch.optim.AdamW(m2.parameters(), lr=1e-3, weight_decay=1e-4, capturable=True)
opt.load_state_dict(t2.vq_optimizer.state_dict())
for grp in opt.param_groups:
    grp["capturable"] = True
    for prm in grp["params"]:
        st = opt.state.get(prm)
        st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=prm.device)
for i in (3, 3, 4):
    z, loss, params, grads = forward_backward(m2, xs[i])
    for pp, g in zip(params, grads): pp.grad = g
    opt.step()
    l2.append(float(loss))
print("trainer   :", l1)
print("fb+captAdamW:", l2)
print([ (k, float((a-b).abs().max())) for (k,a),(_,b) in zip(m1.state_dict().items(), m2.state_dict().items())][:6])
# path 3: 3 trainer steps, deepcopy -> twin trainer with loaded optimizer state
m3 = mk(); t3 = VQTokenizerTrainer(m3)
l3 = [float(t3.train_on_actions(xs[i])[1]) for i in (0, 1, 2)]
tw = copy.deepcopy(m3); tw.invalidate_caches()
t4 = VQTokenizerTrainer(tw)
t4.vq_optimizer.load_state_dict(t3.vq_optimizer.state_dict())
print("state steps:", [float(s["step"]) for s in t4.vq_optimizer.state.values()][:3], len(t4.vq_optimizer.state))
l3 += [float(t4.train_on_actions(xs[i])[1]) for i in (3, 3, 4)]
print("twin      :", l3)
print("same params object?", [a is b for a, b in zip(m3.parameters(), tw.parameters())][:2])
# path 4: exactly the test
m5 = mk(); t5 = VQTokenizerTrainer(m5)
for i in range(3): t5.train_on_actions(xs[i])
twin = copy.deepcopy(m5); twin.invalidate_caches()
tw2 = VQTokenizerTrainer(twin)
tw2.vq_optimizer.load_state_dict(t5.vq_optimizer.state_dict())
g = GraphedTokenizerStep(m5, xs[3], optimizer_state=t5.vq_optimizer.state_dict(), warmup=2)
print("max |model - twin| right after graph construction (model did 2 more steps):", max(float((a-b).abs().max()) for a,b in zip(m5.parameters(), twin.parameters())))
lt = [float(tw2.train_on_actions(xs[3])[1]) for _ in range(2)]
print("twin warm:", lt)
print("max |model - twin| after twin's 2 steps:", max(float((a-b).abs().max()) for a,b in zip(m5.parameters(), twin.parameters())))
for i in range(4, 7):
    _, loss = g.step(xs[i]); lg = float(loss)
    _, rl = tw2.train_on_actions(xs[i])
    print(i, "graph", lg, "twin", float(rl), "max dparam", max(float((a-b).abs().max()) for a,b in zip(m5.parameters(), twin.parameters())))
