import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    d = (a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-30)
    print(f"{k:32s} rel diff {float(d):.3e}   norm {float(a[k].double().norm()):.3e}")
