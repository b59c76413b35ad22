#!/usr/bin/env python3
"""Which kernels the vendor library launches for the yardstick's plain bf16 GEMMs (a comparison baseline only; never on the product
path).  Run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/hipblaslt_names.py`; each GEMM is
bracketed by a named marker kernel-free gap: the script prints the order it ran them in, tools/hipblaslt_names_parse.py maps the
trace's kernel names onto it."""
import sys, torch
dev = torch.device("cuda:0")
B, T = 32, 512
N = B * T
order = []
for C, hid in ((256, 1024), (512, 2048)):
    g = torch.Generator(device=dev).manual_seed(C)
    W1 = torch.randn(hid, C, device=dev, generator=g).bfloat16(); W2 = torch.randn(C, hid, device=dev, generator=g).bfloat16()
    xc = torch.randn(C, N, device=dev, generator=g).bfloat16(); xh = torch.randn(hid, N, device=dev, generator=g).bfloat16()
    xcb = xc.view(C, B, T).transpose(0, 1).contiguous(); xhb = xh.view(hid, B, T).transpose(0, 1).contiguous()
    cases = [("expand fwd flat        Y[hid x N] = W1[hid x C] X[C x N]", lambda: torch.matmul(W1, xc)),
             ("project fwd flat       Y[C x N] = W2[C x hid] X[hid x N]", lambda: torch.matmul(W2, xh)),
             ("project bwd-data flat  Y[hid x N] = W2^T X[C x N]", lambda: torch.matmul(W2.t(), xc)),
             ("expand bwd-data flat   Y[C x N] = W1^T X[hid x N]", lambda: torch.matmul(W1.t(), xh)),
             ("expand fwd batched     32 x (W1[hid x C] X[C x T])", lambda: torch.matmul(W1, xcb)),
             ("project fwd batched    32 x (W2[C x hid] X[hid x T])", lambda: torch.matmul(W2, xhb)),
             ("expand wgrad flat      dW1[hid x C] = G[hid x N] X[C x N]^T", lambda: torch.matmul(xh, xc.t())),
             ("project wgrad flat     dW2[C x hid] = G[C x N] X[hid x N]^T", lambda: torch.matmul(xc, xh.t()))]
    for name, fn in cases:
        fn(); torch.cuda.synchronize()                 # warm-up (heuristic / tuning lookup)
        mark = torch.zeros(1, device=dev); mark.add_(1.0); torch.cuda.synchronize()     # an elementwise kernel separates the cases in the trace
        for _ in range(3): fn()
        torch.cuda.synchronize()
        order.append(f"C={C} {name}")
print("\n".join(order))
