"""Pin the differentiable CPU port (oracle/torch_port.py) to G9, captured from the reference's
NerfRenderer.forward / backward."""
import numpy as np
import torch

from conftest import load_golden
from oracle import torch_port as tp


def test_port_matches_reference_renderer():
    g = load_golden("G9_renderer_kplanes")
    sd = {k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("sd.")}
    packed, info = torch.as_tensor(g["packed"]), torch.as_tensor(g["info"])
    bg, target = torch.as_tensor(g["bg"]), torch.as_tensor(g["target"])
    out = tp.render(sd, packed, info, bg)
    np.testing.assert_allclose(out.numpy(), g["rendered"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(tp.render(sd, packed, info, None).numpy(), g["rendered_nobg"], rtol=0, atol=1e-6)
    grads, loss = tp.grads_of(sd, lambda p: torch.nn.functional.mse_loss(tp.render(p, packed, info, bg), target))
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-6)
    for k, v in grads.items():
        np.testing.assert_allclose(v, g["grad." + k], rtol=1e-4, atol=1e-9, err_msg=k)
    assert len(grads) == sum(1 for k in g if k.startswith("grad."))
