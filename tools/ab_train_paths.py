"""Which training-side kernel path moves the gradients: one training step with all paths on the torch modules, then with each A/B
switch turned on alone (and all together); prints loss and the relative L2 distance of all parameter gradients to the all-off step.
Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import cnn, ops, train_lm, synthetic, settings
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch

M, N, B = 512, 1024, int(os.environ.get("B", 2))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev).train()
state = {k: v.clone() for k, v in model.state_dict().items()}
ds = train_lm.SyntheticCrops(B, N, M, seed=5)
batch = torch.utils.data.default_collate([ds[i] for i in range(B)])
flags = [(settings, "USE_LOWRES_UPCONV_TRAIN"), (settings, "USE_SPLIT_PSP_TRAIN"), (settings, "USE_MFMA_CONV_TRAIN"), (settings, "USE_FUSED_BN_TRAIN")]


def run(on):
    for mod, name in flags:
        setattr(mod, name, name in on)
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    out, _ = train_lm.model_fn_dec(model, batch, dev)
    out["loss"].backward()
    return float(out["loss"].detach()), {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.grad is not None}


l0, g0 = run(())
l0b, g0b = run(())
den = sum((v ** 2).sum().item() for v in g0.values()) ** 0.5
dist = lambda g: sum(((g[k] - g0[k]) ** 2).sum().item() for k in g0) ** 0.5 / den
print("all off: loss %.6f; repeated: loss %.6f, gradient distance %.3e (run-to-run noise)" % (l0, l0b, dist(g0b)), flush=True)
for case in [(n,) for _, n in flags] + [tuple(n for _, n in flags)]:
    l, g = run(case)
    worst = max(((g[k] - g0[k]).norm().item() / den, k) for k in g0)
    print("%-60s loss %.6f  gradient distance %.3e  (largest share: %s %.3e)" % ("+".join(case), l, dist(g), worst[1], worst[0]), flush=True)
