import sys, importlib, torch
sys.path.insert(0, '.')
PKG = "gan-calibrated-semi-supervised-learning_amd"
ops = importlib.import_module(PKG + ".ops")
from oracle import manual_step as M
for H in (16, 32, 64, 128):
    N, C = 2, 64
    g = torch.Generator().manual_seed(H)
    z = torch.randn(N, C, H, H, generator=g) * 2 + 0.3
    dab = torch.randn(N, C, generator=g)
    da = torch.randn(N, C, H, H, generator=g)
    zd = z.permute(0, 2, 3, 1).contiguous().cuda()
    mean = torch.empty(N, C, device="cuda"); rstd = torch.empty(N, C, device="cuda")
    a = torch.empty(N, H, H, C, device="cuda")
    ops.in_act_fwd(zd, a, mean, rstd, C, 2)
    mu, r = M.in_stats(z); xh = (z - mu) * r
    print(H, "mean err", float((mean.cpu() - mu.view(N, C)).abs().max()), "rstd relerr", float(((rstd.cpu() - r.view(N, C)) / r.view(N, C)).abs().max()))
    for mode in ("bcast", "dense"):
        dzs = torch.empty(N, H, H, C, device="cuda")
        if mode == "bcast":
            ops.in_act_bwd(zd, mean, rstd, dzs, C, 2, da_bcast=dab.cuda())
            dn = dab.view(N, C, 1, 1) * (xh > 0).float()
        else:
            ops.in_act_bwd(zd, mean, rstd, dzs, C, 2, da=da.permute(0, 2, 3, 1).contiguous().cuda())
            dn = da * (xh > 0).float()
        ref = M.in_bwd(xh, r, dn)
        # fp64 reference too
        mu64, r64 = M.in_stats(z.double()); xh64 = (z.double() - mu64) * r64
        dn64 = (dab.double().view(N, C, 1, 1) if mode == "bcast" else da.double()) * (xh64 > 0).double()
        ref64 = M.in_bwd(xh64, r64, dn64)
        got = dzs.cpu().permute(0, 3, 1, 2)
        print("   ", mode, "err vs fp32 oracle", float((got - ref).abs().max() / ref.abs().max()),
              "kernel vs fp64", float((got.double() - ref64).abs().max() / ref64.abs().max()),
              "fp32 oracle vs fp64", float((ref.double() - ref64).abs().max() / ref64.abs().max()))
