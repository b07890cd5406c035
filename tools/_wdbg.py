import sys, torch
sys.path.insert(0, "3d-sis_amd")
from sis3d import ops
dev = torch.device("cuda")
cin, cout, dims = 32, 32, (8, 4, 8)
torch.manual_seed(0)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
pc = ops.PackedConv(w, torch.zeros(cout, device=dev))
for c0 in range(0, cin, 4):
    full = torch.randn(1, cin, *dims, device=dev)
    m = torch.zeros(1, cin, 1, 1, 1, device=dev); m[:, c0:c0 + 4] = 1
    xs = ops.to_cl(full * m)
    ops.set_winograd(True)
    a = ops.conv3d_k3wino([xs], [pc], relu=False)[0]
    ref = torch.nn.functional.conv3d(full * m, w, padding=1)
    a_ = a
    print("channels %2d..%2d: max err %.3e  (ref max %.3f)" % (c0, c0 + 3, float((a_.float() - ref).abs().max()), float(ref.abs().max())))
