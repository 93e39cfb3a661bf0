import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from conftest import load_golden, T
from oracle import objnerf_oracle as O
from openobj_amd import ops, synthetic
dev = torch.device('cuda:0')
g = load_golden("g5_step_s10_feat")
K, R, n1, n2, feat_on = [int(x) for x in g["meta"]]
arena = ops.ParamArena(K, ops.NetShape(), dev)
arena.load_stacked([T(g[f"fc0_{i}"]) for i in range(18)] + [T(g["B0"])])
ws = ops.TrainWorkspace(arena, K, R, n1+n2, True)
b = synthetic.random_batch(K, R, n1, n2, seed=500, feat_dim=512)
batch = {k: T(b[k]).to(dev) for k in ["pts","z","gt_depth","gt_rgb","labels","gt_feat"]}
ops.train_step(arena, ws, batch, with_feat=True); torch.cuda.synchronize()
fc = [T(g[f"fc0_{i}"]).clone().requires_grad_(True) for i in range(18)]
B = T(g["B0"]).clone().requires_grad_(True)
loss, terms = O.train_forward_loss(fc, B, torch.full((K,),2.0), T(b["pts"]), T(b["gt_depth"]), T(b["gt_rgb"]), T(b["labels"]), T(b["z"]), gt_feat=T(b["gt_feat"]), return_terms=True)
print("hip terms\n", ws.loss_terms.cpu().numpy())
print("ref terms", [terms[k].detach().numpy() for k in ["depth","color","opacity","feat"]])
# per-ray check of cos using oracle pieces
rf = terms["render_feat"].detach(); gt = T(b["gt_feat"])
cos = torch.nn.functional.cosine_similarity(rf, gt, dim=-1)
print("ref cos[0,:6]", cos[0,:6].numpy(), "labels", b["labels"][0,:6])
grads = torch.autograd.grad(loss, fc+[B])
gv = arena.views(ws.grads)
for i in range(19):
    e = (gv[i].cpu()-grads[i]).abs().max().item(); s = grads[i].abs().max().item()
    print(i, ops.TENSOR_NAMES[i], "err %.3e scale %.3e rel %.2e" % (e, s, e/max(s,1e-12)))
# ---- inspect workspace intermediates
import torch.nn.functional as F
ps = arena.p_stride; ncu = 256
def a256(x): return (x + 255) & ~255
off = a256(K*ncu*ps*4) + a256(K*ncu*16) + a256(ps) + 256
rayin = ws.buf[off:off + K*R*34*4].view(torch.float32).reshape(K, R, 34).cpu(); off += a256(K*R*34*4)
gram = ws.buf[off:off + K*1088*4].view(torch.float32).reshape(K, 1088).cpu(); off += a256(K*1088*4)
rayfeat = ws.buf[off:off + K*R*36*4].view(torch.float32).reshape(K, R, 36).cpu()
p = [T(g[f"fc0_{i}"]) for i in range(18)]
Wof, bof = p[16], p[17]      # [K,512,32], [K,512]
gt = T(b["gt_feat"])
u_ref = torch.einsum('kch,krc->krh', Wof, gt)
print("u err", (rayin[..., :32]-u_ref).abs().max().item(), "beta err", (rayin[..., 32]-torch.einsum('kc,krc->kr', bof, gt)).abs().max().item(), "ng", rayin[...,33].min().item(), rayin[...,33].max().item())
G_ref = torch.einsum('kch,kcj->khj', Wof, Wof)
print("G err", (gram[:, :1024].reshape(K,32,32)-G_ref).abs().max().item(), "wb err", (gram[:,1024:1056]-torch.einsum('kch,kc->kh', Wof, bof)).abs().max().item(), "bb err", (gram[:,1056]-(bof*bof).sum(-1)).abs().max().item())
emb = O.embed_stacked(T(g["B0"]), torch.full((K,),2.0), T(b["pts"]))
hf = []
for k in range(K):
    pk = [t[k] for t in p]
    x1 = emb[k][..., :87]; x2 = emb[k][..., 87:]
    fc1 = F.relu(F.linear(x1, pk[0], pk[1])); fc2 = F.relu(F.linear(fc1, pk[2], pk[3]))
    fc3 = F.relu(F.linear(torch.cat((fc2, x1), -1), pk[4], pk[5])); fc4 = F.relu(F.linear(fc3, pk[6], pk[7]))
    hf.append(F.relu(F.linear(torch.cat((fc4, x2), -1), pk[14], pk[15])))
hf = torch.stack(hf)
term = terms["term"].detach()
fh_ref = (term[..., None]*hf).sum(-2)
print("fh err", (rayfeat[..., :32]-fh_ref).abs().max().item(), "fh scale", fh_ref.abs().max().item())
print("O err", (rayfeat[..., 34]-term.sum(-1)).abs().max().item())
Fr = torch.einsum('kch,krh->krc', Wof, fh_ref) + bof[:, None, :]*term.sum(-1)[..., None]
print("F vs render_feat", (Fr-terms["render_feat"].detach()).abs().max().item())
