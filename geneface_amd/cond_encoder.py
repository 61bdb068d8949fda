"""Condition (landmark / audio window) encoders and the bias-free ReLU MLP.

State-dict compatible re-implementation of modules/radnerfs/cond_encoder.py: AudioNet :7-52,
AudioAttNet :55-89, MLP :92-111 (same sub-module names, so `cond_prenet.encoder_conv.0.weight`,
`cond_att_net.attentionNet.0.bias`, `sigma_net.net.2.weight`, ... load unchanged).  These hold the
weights; on the fused render path the per-sample MLPs never run through torch.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_STRIDES = {1: (1, 1, 1, 1), 2: (2, 1, 1, 1), 3: (2, 2, 1, 1), 4: (2, 2, 1, 1), 16: (2, 2, 2, 2)}


class AudioNet(nn.Module):
    """[b, t_window, c] -> [b, dim_aud]: four k=3 Conv1d (LeakyReLU 0.02) that shrink the window to 1, then two FC."""

    def __init__(self, dim_in=29, dim_aud=64, win_size=16):
        super().__init__()
        if win_size not in _STRIDES:  # the reference's `win_size == [5, 8]` branch can never match an int
            raise ValueError("unsupported win_size")
        self.win_size, self.dim_aud = win_size, dim_aud
        chans = (dim_in, 32, 32, 64, 64)
        layers = []
        for i, s in enumerate(_STRIDES[win_size]):
            layers += [nn.Conv1d(chans[i], chans[i + 1], kernel_size=3, stride=s, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.encoder_conv = nn.Sequential(*layers)
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), nn.LeakyReLU(0.02, True), nn.Linear(64, dim_aud))

    def forward(self, x):
        x = self.encoder_conv(x.permute(0, 2, 1)).squeeze(-1)
        return self.encoder_fc1(x).squeeze()


class AudioAttNet(nn.Module):
    """[seq_len, c] -> [c]: attention weights over the window from a 1-D conv stack + Linear + softmax."""

    def __init__(self, in_out_dim=64, seq_len=8):
        super().__init__()
        self.seq_len, self.in_out_dim = seq_len, in_out_dim
        chans = (in_out_dim, 16, 8, 4, 2, 1)
        layers = []
        for i in range(5):
            layers += [nn.Conv1d(chans[i], chans[i + 1], kernel_size=3, stride=1, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.attentionConvNet = nn.Sequential(*layers)
        self.attentionNet = nn.Sequential(nn.Linear(seq_len, seq_len, bias=True), nn.Softmax(dim=1))

    def forward(self, x):
        y = self.attentionConvNet(x[:, :self.in_out_dim].permute(1, 0).unsqueeze(0))
        y = self.attentionNet(y.view(1, self.seq_len)).view(self.seq_len, 1)
        return torch.sum(y * x, dim=0)


_TALL = 1 << 16      # rows from which the weight gradient is worth splitting


class _linear_tall(torch.autograd.Function):
    """y = x W^T for a tall x [B, I] (B ~ 10^6 samples of a training batch).  Forward and dX are ordinary GEMMs.  dW = dY^T x reduces
    over B into an [O, I] <= 129 x 148 output: as one GEMM that is a few dozen output tiles, i.e. a few dozen of 256 CUs busy for 2 ms
    per layer (rocprofv3: 40 % of a training step).  Split the reduction instead: S batched [O, B/S] x [B/S, I] products fill the
    chip, followed by a sum over S (S * O * I floats)."""

    @staticmethod
    def forward(ctx, x, w):
        if torch.is_autocast_enabled():                       # what nn.Linear does under autocast: 16-bit operands, fp32 accumulate
            dt = torch.get_autocast_dtype("cuda")
            x, w = x.to(dt), w.to(dt)
        ctx.save_for_backward(x, w)
        return F.linear(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            B, O, I = x.shape[0], g.shape[1], x.shape[1]
            S = max(1, B // 4096)
            rows = B // S
            main = S * rows
            x = x.contiguous()
            gw = torch.bmm(g[:main].view(S, rows, O).transpose(1, 2), x[:main].view(S, rows, I)).sum(0, dtype=torch.float32)
            if main < B:
                gw = gw + (g[main:].t() @ x[main:]).float()
        return gx, gw


class MLP(nn.Module):
    def __init__(self, dim_in, dim_out, dim_hidden, num_layers):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.net = nn.ModuleList([
            nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=False)
            for l in range(num_layers)])

    def forward(self, x):
        for l, layer in enumerate(self.net):
            if x.dim() == 2 and x.shape[0] >= _TALL and torch.is_grad_enabled() and (x.requires_grad or layer.weight.requires_grad):
                x = _linear_tall.apply(x, layer.weight)
            else:
                x = layer(x)
            if l != self.num_layers - 1:
                x = F.relu(x, inplace=True)
        return x
