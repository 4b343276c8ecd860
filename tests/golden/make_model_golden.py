"""Generates tests/golden/model_golden.npz by importing the reference's
model/model.py and model/loss.py from /root/reference (container-only; the
reference never travels to the GPU box, the fixture does).

``config.py`` needs ``easydict`` (absent from the image, not installable): an
attribute-dict stand-in with no arithmetic is put on sys.path for the import.
``sklearn`` (imported, unused, by model/loss.py) is present.

Run:  python tests/golden/make_model_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_golden.npz")

tmp = tempfile.mkdtemp()
with open(os.path.join(tmp, "easydict.py"), "w") as f:
    f.write("class EasyDict(dict):\n"
            "    def __getattr__(self, k):\n"
            "        try:\n            return self[k]\n"
            "        except KeyError:\n            raise AttributeError(k)\n"
            "    def __setattr__(self, k, v):\n        self[k] = v\n")
sys.path.insert(0, tmp)
sys.path.insert(0, REF)

from config import cfg  # noqa: E402
import model.model as ref_model  # noqa: E402
import model.loss as ref_loss  # noqa: E402

CANVAS, C, P, N, A_PER = 32, 8, 64, 8, 2
cfg.DATA.CANVAS_HEIGHT = CANVAS   # read at forward time, model/model.py:55
cfg.DATA.CANVAS_WIDTH = CANVAS

torch.manual_seed(1234)
net = ref_model.PPModel(9, C, A_PER * 9, A_PER * 8, "cpu")
# non-trivial BN statistics so that eval mode differs from identity
for m in net.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.2)
        m.running_var.uniform_(0.5, 1.5)
        m.weight.data.uniform_(0.5, 1.5)
        m.bias.data.normal_(0, 0.1)
g = torch.Generator().manual_seed(99)
B = 2
pillars = torch.randn(B, 9, P, N, generator=g)
pillars[:, :, :, 5:] = 0                       # zero padded slots
inds = torch.zeros(B, P, 3, dtype=torch.int64)
for b in range(B):
    cells = torch.randperm(CANVAS * CANVAS, generator=g)[:P - 10 * (b + 1)]
    k = cells.numel()
    inds[b, :k, 0] = 1
    inds[b, :k, 1] = cells % CANVAS            # column
    inds[b, :k, 2] = cells // CANVAS           # row
    pillars[b, :, k:, :] = 0

out = {}
state = {k: v.detach().clone() for k, v in net.state_dict().items()}
net.eval()
with torch.no_grad():
    cls_e, reg_e = net(pillars, inds)
out["cls_eval"], out["reg_eval"] = cls_e.numpy(), reg_e.numpy()
net.load_state_dict(state)
net.train()
cls_t, reg_t = net(pillars, inds)
out["cls_train"], out["reg_train"] = cls_t.detach().numpy(), reg_t.detach().numpy()

# loss: synthetic targets with a handful of positives (model/loss.py:24-63)
FM = CANVAS // 2
A = FM * FM * A_PER
cls_targets = torch.zeros(B, A, 9)
reg_targets = torch.zeros(B, A, 9)
pos = torch.randperm(A, generator=g)[:17]
for b in range(B):
    cls_targets[b, pos, torch.randint(0, 9, (17,), generator=g)] = 1
    reg_targets[b, pos, 0] = 1
    reg_targets[b, pos, 1:8] = torch.randn(17, 7, generator=g) * 0.5
    reg_targets[b, pos, 8] = torch.randint(0, 2, (17,), generator=g).float()
loss_fn = ref_loss.PPLoss(cfg.NET.B_ORT, cfg.NET.B_REG, cfg.NET.B_CLS, cfg.NET.GAMMA, "cpu")
cls_in = cls_t.detach().clone().requires_grad_(True)
reg_in = reg_t.detach().clone().requires_grad_(True)
p, cls_loss, reg_loss, ort_loss, total = loss_fn(cls_in, reg_in * 1.0, cls_targets, reg_targets)
total.backward()
out.update(loss_p=p.detach().numpy(), loss_vals=np.array([cls_loss.item(), reg_loss.item(),
                                                           ort_loss.item(), total.item()]),
           loss_grad_cls=cls_in.grad.numpy(), loss_grad_reg=reg_in.grad.numpy(),
           cls_targets=cls_targets.numpy(), reg_targets=reg_targets.numpy(),
           loss_weights=np.array([cfg.NET.B_ORT, cfg.NET.B_REG, cfg.NET.B_CLS, cfg.NET.GAMMA], np.float64))
out.update(pillars=pillars.numpy(), inds=inds.numpy(),
           dims=np.array([CANVAS, C, P, N, A_PER]))
for k, v in state.items():
    out["sd/" + k] = v.numpy()
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB;", len(state), "state tensors;",
      "params", sum(v.numel() for k, v in state.items() if "num_batches" not in k and "running" not in k))
print("state keys sample:", list(state)[:6], "...", list(state)[-4:])
print("loss vals", out["loss_vals"])
