"""Shared test helpers (oracle-side constructions of the benchmark workload)."""
import torch


class LinearField(torch.nn.Module):
    """README vector field (reference README.md:42-49): Linear(H, H*C) viewed (..., H, C), optional tanh."""

    def __init__(self, H, C, dtype=torch.float32, scale=1.0, tanh=False, seed=0):
        super().__init__()
        self.H, self.C, self.tanh = H, C, tanh
        self.linear = torch.nn.Linear(H, H * C)
        gen = torch.Generator().manual_seed(seed)
        bound = 1 / H ** 0.5
        with torch.no_grad():
            w = (torch.rand(H * C, H, generator=gen, dtype=torch.float64) * 2 - 1) * bound * scale
            b = (torch.rand(H * C, generator=gen, dtype=torch.float64) * 2 - 1) * bound * scale
        self.linear = self.linear.to(dtype)
        with torch.no_grad():
            self.linear.weight.copy_(w)
            self.linear.bias.copy_(b)

    def forward(self, t, z):
        y = self.linear(z)
        if self.tanh:
            y = y.tanh()
        return y.view(*z.shape[:-1], self.H, self.C)


def make_series(B, L, C, dtype=torch.float32, seed=0):
    """SURVEY section 8(d) synthetic input: channel 0 = time linspace(0,1,L), others 0.5*randn."""
    gen = torch.Generator().manual_seed(seed)
    x = 0.5 * torch.randn(B, L, C, generator=gen, dtype=torch.float64)
    x[..., 0] = torch.linspace(0, 1, L, dtype=torch.float64)
    return x.to(dtype)


def golden_field(case, dtype=None, device="cpu"):
    H, C = case["H"], case["C"]
    f = LinearField(H, C, dtype=case["W"].dtype)
    with torch.no_grad():
        f.linear.weight.copy_(case["W"])
        f.linear.bias.copy_(case["b"])
    return f.to(device)


class TwoLayerField(torch.nn.Module):
    """reference example/time_series_classification.py:20-51"""

    def __init__(self, H, C, width, dtype=torch.float32, seed=0, final_tanh=True):
        super().__init__()
        self.H, self.C, self.final_tanh = H, C, final_tanh
        gen = torch.Generator().manual_seed(seed)
        self.linear1 = torch.nn.Linear(H, width).to(dtype)
        self.linear2 = torch.nn.Linear(width, H * C).to(dtype)
        with torch.no_grad():
            for lin in (self.linear1, self.linear2):
                bound = 1 / lin.in_features ** 0.5
                lin.weight.copy_((torch.rand(lin.weight.shape, generator=gen, dtype=torch.float64) * 2 - 1) * bound)
                lin.bias.copy_((torch.rand(lin.bias.shape, generator=gen, dtype=torch.float64) * 2 - 1) * bound)

    def forward(self, t, z):
        y = self.linear2(self.linear1(z).relu())
        if self.final_tanh:
            y = y.tanh()
        return y.view(*z.shape[:-1], self.H, self.C)
