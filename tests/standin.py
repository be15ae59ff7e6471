"""CPU stand-in for the multi-rank control-flow tests (gloo) and `bench.py --cpu-dry-run`.

The HIP model cannot run on CPU (there is no fallback), so the data-parallel algebra -- FlatParams' backward-completion
ordering, layer_ranges, the bucket tiling, the pre-scale by local/global graphs, bench.py's world > 1 branch -- is
exercised with a plain torch module that carries PAMNet's parameter NAMES (`embeddings`, `rbf_g.freq`,
`mlp_rbf_g.0.0.*`, `global_layer.k.*`, `local_layer.k.*`: SURVEY.md 8b) and the same `model(data) -> [num_graphs]`
contract.  The maths is an arbitrary residual stack with per-layer heads; it is test infrastructure, never a product path.
"""
import torch
import torch.nn as nn


class _Block(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.mlp_x1 = nn.Sequential(nn.Sequential(nn.Linear(d, d)))
        self.W_out = nn.Linear(d, 1)
        self.W = nn.Parameter(torch.randn(d, 1) * 0.1)


class LayeredStandIn(nn.Module):
    def __init__(self, n_layer=4, d=8):
        super().__init__()
        self.embeddings = nn.Parameter(torch.randn(6, d) * 0.3)
        self.rbf_g = nn.Module()
        self.rbf_g.freq = nn.Parameter(torch.arange(1.0, 5.0))
        self.mlp_rbf_g = nn.Sequential(nn.Sequential(nn.Linear(4, d)))
        self.global_layer = nn.ModuleList([_Block(d) for _ in range(n_layer)])
        self.local_layer = nn.ModuleList([_Block(d) for _ in range(n_layer)])

    def forward(self, data):
        if data.x.dim() == 1:                       # QM9-schema batch: atom types + positions
            x = self.embeddings[data.x.long()]
            r = data.pos.norm(dim=1, keepdim=True)
        else:                                       # feature rows
            x = data.x @ self.embeddings
            r = data.x[:, :1]
        x = x + self.mlp_rbf_g[0][0](torch.sin(r * self.rbf_g.freq))
        out = 0
        for g, l in zip(self.global_layer, self.local_layer):
            for blk in (g, l):
                x = x + torch.tanh(blk.mlp_x1[0][0](x))
                out = out + blk.W_out(x).view(-1) * torch.sigmoid(x @ blk.W).view(-1)
        return torch.zeros(data.num_graphs, dtype=x.dtype).index_add_(0, data.batch, out)
