import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_policy_grad_gpu import _rows, _autograd64, _relu_masks
from iago_amd import network, train_rl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
own, opp, act, z = _rows(n, seed=n)
torch.manual_seed(5)
model = network.SLPolicy().cuda()
if len(sys.argv) > 2:
    model.load_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sl_model.npz"))
masks = _relu_masks(model, own, opp)
loss64, ref = _autograd64(model, own, opp, act, z, masks=masks)
probs = torch.empty(n, 64, device="cuda")
loss = model.reinforce_grads(own, opp, act, z, probs=probs)
torch.cuda.synchronize()
got = {k: p.grad.clone() for k, p in model.named_parameters()}
model.train()
for p in model.parameters():
    p.grad = None
train_rl.reinforce_loss(model, own, opp, act, z).backward()
got32 = {k: p.grad for k, p in model.named_parameters()}
print("n", n, "loss", float(loss), float(loss64))
from iago_amd import ops
with torch.no_grad():
    m64 = __import__("copy").deepcopy(model).double()
    p64 = torch.softmax(m64.logits(ops.encode_planes(own, opp).double()), 1)
    p32 = torch.softmax(model.logits(ops.encode_planes(own, opp)), 1)
print("probs: split-f16 forward max |dp| %.3e, float32 %.3e" % (float((probs.double() - p64).abs().max()), float((p32.double() - p64).abs().max())))
for k in ref:
    s = float(ref[k].abs().max())
    print("  %-20s native %.3e  f32 autograd %.3e (max %.3e)" % (k, float((got[k].double() - ref[k]).abs().max()) / s, float((got32[k].double() - ref[k]).abs().max()) / s, s))
