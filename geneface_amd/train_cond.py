"""The condition encoder under training: RADNeRF.cal_cond_feat (modules/radnerfs/radnerf.py:61-71 = AudioNet.forward + AudioAttNet.forward,
cond_encoder.py:44-52, 79-89) as ONE autograd node over three HIP launches (gf_cond_train_forward / _backward, csrc/cond_train.hip).

Through torch the encoder is ~50 launches forward and ~50 backward per training step (MIOpen convolutions of a [5, 204, 1] window, GEMVs,
activations, softmax; under autocast a cast of every weight on top): 0.4 ms of kernels and about as much host time in a 5.5 ms step.  The
node computes in fp32 whatever the autocast state (the reference's half arithmetic here is a property of torch.autocast, not of the model:
master weights and the gradients the optimizer sees are fp32 either way) and returns the gradient of all 24 parameter tensors; the window
itself is data and gets none.  Any encoder the kernel does not cover (gf_cond_check) keeps the torch modules.
"""
import ctypes as C

import torch

from .fused import GfCond
from .lib import check, current_stream, lib, ptr

_vp = C.c_void_p


class GfCondTrain(C.Structure):
    """ctypes mirror of gf_cond_train_t (include/geneface_hip.h)."""
    _fields_ = [("enc", C.POINTER(GfCond)), ("acts", _vp), ("grads", _vp), ("g_feat", _vp), ("g_conv_w", _vp * 4), ("g_conv_b", _vp * 4),
                ("g_fc1_w", _vp), ("g_fc1_b", _vp), ("g_fc2_w", _vp), ("g_fc2_b", _vp), ("g_att_w", _vp * 5), ("g_att_b", _vp * 5),
                ("g_att_lin_w", _vp), ("g_att_lin_b", _vp)]


def _modules(model):
    """(convs, fcs, att convs or None, att linear or None) when the encoder has the AudioNet (+ AudioAttNet) structure, else None."""
    pre = getattr(model, "cond_prenet", None)
    if pre is None or not hasattr(pre, "encoder_conv") or not hasattr(pre, "encoder_fc1"):
        return None
    convs = [m for m in pre.encoder_conv if isinstance(m, torch.nn.Conv1d)]
    fcs = [m for m in pre.encoder_fc1 if isinstance(m, torch.nn.Linear)]
    if len(convs) != 4 or len(fcs) != 2 or any(m.kernel_size != (3,) or m.padding != (1,) or m.bias is None for m in convs):
        return None
    if getattr(model, "with_att", False):
        att = getattr(model, "cond_att_net", None)
        if att is None:
            return None
        aconvs = [m for m in att.attentionConvNet if isinstance(m, torch.nn.Conv1d)]
        if len(aconvs) != 5 or [c.out_channels for c in aconvs] != [16, 8, 4, 2, 1]:
            return None
        return convs, fcs, aconvs, att.attentionNet[0]
    return convs, fcs, None, None


def _params(mods):
    convs, fcs, aconvs, lin = mods
    ps = []
    for m in convs + fcs + (aconvs or []) + ([lin] if lin is not None else []):
        ps += [m.weight, m.bias]
    return ps


def _describe(mods, model, cond, params, feat):
    """gf_cond_t over `params` (the tensors autograd handed the node, in _params order)."""
    convs, fcs, aconvs, lin = mods
    c = GfCond()
    f32 = torch.float32
    it = iter(params)
    for i, m in enumerate(convs):
        w, b = next(it), next(it)
        c.conv_w[i], c.conv_b[i], c.conv_stride[i], c.conv_ch[i] = ptr(w, f32), ptr(b, f32), int(m.stride[0]), int(m.in_channels)
    c.conv_ch[4] = int(convs[3].out_channels)
    c.fc1_w, c.fc1_b, c.fc2_w, c.fc2_b = (ptr(next(it), f32) for _ in range(4))
    if aconvs is not None:
        for i in range(5):
            c.att_w[i], c.att_b[i] = ptr(next(it), f32), ptr(next(it), f32)
        c.att_lin_w, c.att_lin_b = ptr(next(it), f32), ptr(next(it), f32)
    c.S, c.T, c.C, c.dim_aud = int(cond.shape[0]), int(cond.shape[1]), int(cond.shape[2]), int(model.cond_prenet.dim_aud)
    c.cond, c.cond_feat = ptr(cond, f32), ptr(feat, f32)
    return c


def supported(model, cond):
    """Can the node serve this model and window?  (structure, limits of the kernel, the attention net's window length)"""
    mods = _modules(model)
    if mods is None or cond.dim() != 3 or not cond.is_cuda:
        return False
    convs, fcs, aconvs, lin = mods
    S, T, Cc = (int(v) for v in cond.shape)
    if T != int(model.cond_prenet.win_size) or Cc != convs[0].in_channels or S > 16 or int(model.cond_prenet.dim_aud) > 64:
        return False
    if aconvs is not None and (int(model.cond_att_net.seq_len) != S or aconvs[0].in_channels != int(model.cond_prenet.dim_aud)):
        return False
    if aconvs is None and S != 1:
        return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in _params(mods))


class _CondEncoderTrain(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, model, cond, *params):
        mods = _modules(model)
        dev = cond.device
        cond = cond.detach().float().contiguous()
        params = tuple(p.detach().contiguous() for p in params)
        feat = torch.empty(int(model.cond_prenet.dim_aud), dtype=torch.float32, device=dev)
        c = _describe(mods, model, cond, params, feat)
        n = lib().gf_cond_train_scratch_floats(C.byref(c))
        acts = torch.empty(2 * n, dtype=torch.float32, device=dev)      # [0:n] activations (kept for the backward), [n:2n] its gradient scratch
        t = GfCondTrain(enc=C.pointer(c), acts=acts.data_ptr(), grads=acts.data_ptr() + 4 * n)
        check(lib().gf_cond_train_forward(C.byref(t), current_stream(dev)))
        ctx.model, ctx.n = model, n
        ctx.save_for_backward(cond, acts, feat, *params)
        return feat.clone()      # (the kernel's buffer stays the node's: a caller writing into its result must not change what backward reads)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_feat):
        cond, acts, feat, *params = ctx.saved_tensors
        model, n = ctx.model, ctx.n
        mods = _modules(model)
        dev = cond.device
        g_feat = g_feat.detach().float().contiguous()
        c = _describe(mods, model, cond, params, feat)
        grads = [torch.empty_like(p) for p in params]
        t = GfCondTrain(enc=C.pointer(c), acts=acts.data_ptr(), grads=acts.data_ptr() + 4 * n, g_feat=g_feat.data_ptr())
        it = iter(grads)
        for i in range(4):
            t.g_conv_w[i], t.g_conv_b[i] = next(it).data_ptr(), next(it).data_ptr()
        t.g_fc1_w, t.g_fc1_b, t.g_fc2_w, t.g_fc2_b = (next(it).data_ptr() for _ in range(4))
        if mods[2] is not None:
            for i in range(5):
                t.g_att_w[i], t.g_att_b[i] = next(it).data_ptr(), next(it).data_ptr()
            t.g_att_lin_w, t.g_att_lin_b = next(it).data_ptr(), next(it).data_ptr()
        check(lib().gf_cond_train_backward(C.byref(t), current_stream(dev)))
        return (None, None, *grads)


def cond_feat_train(model, cond):
    """cal_cond_feat(cond) [dim_aud] with gradients to every parameter of the encoder: two launches."""
    return _CondEncoderTrain.apply(model, cond, *_params(_modules(model)))
