import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from oracle import stil_oracle as O
from oracle.make_golden import build_case
from stil_tta_amd import STiLModel
name = sys.argv[1] if len(sys.argv) > 1 else "dvm_r18_noeman"
hp, sd, batch, epoch, mr, mm = build_case(name)
d = dict(vars(hp)); d["mi_dropout"] = False
m = STiLModel(d); m.load_state_dict(sd); m.setup_device("cuda"); m.train()
vx_img = torch.cat((batch["l"][0][1], batch["u"][0][1])); vx_tab = torch.cat((batch["l"][1][1], batch["u"][1][1]))
vy = torch.cat((batch["l"][2], batch["u"][2]))
ov = O.validation_step(sd, vx_img, vx_tab, vy, hp)
v = m.validation_step(([vx_img.cuda(), vx_tab.cuda()], vy.cuda()), 0)
print("pre-step  val loss hip", float(v), "oracle", float(ov["loss"]))
for k in ("multimodal.val.CEloss", "multimodal.val.ITCloss", "multimodal.val.CLUBloss_imaging", "multimodal.val.CLUBloss_imaging_est", "multimodal.val.CLUBloss_tabular", "multimodal.val.CLUBloss_tabular_est"):
    print(k, float(m.logged[k]))
print("oracle ce", float(ov["loss_ce"]), "itc", float(ov["loss_itc"]))
with torch.no_grad():
    o = O.backbone_forward_all(sd, "model.", vx_img, vx_tab, hp, train=False)
    g = m.model.forward_all((vx_img.cuda(), vx_tab.cuda()), train=False)
for i, (a, b) in enumerate(zip(g, o)):
    print(i, float((a.cpu() - b).abs().max()), float(b.abs().max()))
