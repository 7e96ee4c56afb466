import os, sys, subprocess, torch
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tinynerf_amd import models as m
    torch.manual_seed(0)
    out = {}
    for name, net, n in (("van", m.VanillaFeatureMLP(10, 256, 8), 40037), ("h128", m.MLP(36, 128, 5, 128), 5000), ("h128b", m.MLP(36, 128, 5, 96), 777)):
        net = net.cuda()
        x = (torch.rand(n, 3 if name == "van" else 36, device="cuda") * 2 - 1).requires_grad_(name != "van")
        y = net(x) if name == "van" else net.fused(x, None, 0, 0, 0)
        gy = torch.randn_like(y)
        y.backward(gy)
        out[name] = [y.detach().cpu()] + [p.grad.cpu() for p in net.parameters()] + ([x.grad.cpu()] if x.grad is not None else [])
    torch.save(out, sys.argv[1])
else:
    for v in ("0", "1"):
        subprocess.check_call([sys.executable, __file__, f"/tmp/ab{v}.pt"], env=dict(os.environ, TN_MLP_WREG=v))
    a, b = torch.load("/tmp/ab0.pt"), torch.load("/tmp/ab1.pt")
    for k in a:
        for i, (u, w) in enumerate(zip(a[k], b[k])):
            d = (u - w).abs().max().item()
            print(k, i, tuple(u.shape), "max diff", d, "rel", d / (u.abs().max().item() + 1e-30), "EQUAL" if torch.equal(u, w) else "")
