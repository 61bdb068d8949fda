"""RAD-NeRF head field: 3-D grid -> ambient MLP -> tanh -> 2-D grid -> density MLP -> exp;
SH(direction) + geometry feature + identity code -> colour MLP -> sigmoid.

Drop-in for `RADNeRF` of modules/radnerfs/radnerf.py:11-129 (same hparams, sub-module names and
`forward / density / cal_cond_feat` signatures).  `forward` below is the op-by-op evaluation used by
impl="ops" and by callers that query the field directly; the fused renderer evaluates the same
arithmetic inside one kernel per march iteration.
"""
import torch

from .cond_encoder import MLP, AudioAttNet, AudioNet
from .encoders import get_encoder
from .renderer import NeRFRenderer
from .utils import trunc_exp

_COND_DIMS = {"esperanto": 44, "deepspeech": 29, "idexp_lm3d_normalized": 68 * 3}


class RADNeRF(NeRFRenderer):
    def __init__(self, hparams):
        super().__init__(hparams)
        self.hparams = hparams
        if hparams["cond_type"] not in _COND_DIMS:
            raise NotImplementedError()
        self.cond_in_dim = _COND_DIMS[hparams["cond_type"]]
        self.cond_out_dim = hparams["cond_out_dim"]
        self.cond_win_size = hparams["cond_win_size"]
        self.smo_win_size = hparams["smo_win_size"]
        self.cond_prenet = AudioNet(self.cond_in_dim, self.cond_out_dim, win_size=self.cond_win_size)
        self.with_att = hparams["with_att"]
        if self.with_att:
            self.cond_att_net = AudioAttNet(self.cond_out_dim, seq_len=self.smo_win_size)

        self.grid_type = hparams["grid_type"]
        self.grid_interpolation_type = hparams["grid_interpolation_type"]
        grid_kw = dict(num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=hparams["log2_hashmap_size"],
                       interpolation=self.grid_interpolation_type)
        self.position_embedder, self.position_embedding_dim = get_encoder(
            self.grid_type, input_dim=3, desired_resolution=hparams["desired_resolution"] * self.bound, **grid_kw)
        self.num_layers_ambient = hparams["num_layers_ambient"]
        self.hidden_dim_ambient = hparams["hidden_dim_ambient"]
        self.ambient_out_dim = hparams["ambient_out_dim"]
        self.ambient_net = MLP(self.position_embedding_dim + self.cond_out_dim, self.ambient_out_dim, self.hidden_dim_ambient,
                               self.num_layers_ambient)
        self.ambient_embedder, self.ambient_embedding_dim = get_encoder(
            self.grid_type, input_dim=self.ambient_out_dim, desired_resolution=hparams["desired_resolution"], **grid_kw)

        self.num_layers_sigma = hparams["num_layers_sigma"]
        self.hidden_dim_sigma = hparams["hidden_dim_sigma"]
        self.geo_feat_dim = hparams["geo_feat_dim"]
        self.sigma_net = MLP(self.position_embedding_dim + self.ambient_embedding_dim, 1 + self.geo_feat_dim, self.hidden_dim_sigma,
                             self.num_layers_sigma)

        self.num_layers_color = hparams["num_layers_color"]
        self.hidden_dim_color = hparams["hidden_dim_color"]
        self.direction_embedder, self.direction_embedding_dim = get_encoder("spherical_harmonics")
        self.color_net = MLP(self.direction_embedding_dim + self.geo_feat_dim + self.individual_embedding_dim, 3,
                             self.hidden_dim_color, self.num_layers_color)

    def cal_cond_feat(self, cond):
        """cond [smo_win, cond_win, C] (e.g. [5,1,204]) -> [cond_out_dim]."""
        if self.training and torch.is_grad_enabled() and self.cond_impl == "auto" and cond.is_cuda:
            from . import train_cond
            if train_cond.supported(self, cond):      # the encoder under autograd as ONE node (two launches instead of ~100)
                return train_cond.cond_feat_train(self, cond)
        feat = self.cond_prenet(cond)
        if self.with_att:
            feat = self.cond_att_net(feat)
        return feat

    #: training: "auto" = the condition encoder as one autograd node over two HIP launches (train_cond.py) whenever the kernel covers it;
    #: "ops" = the torch modules (the reference's op graph)
    cond_impl = "auto"

    def _geometry(self, position, cond_feat):
        M = position.shape[0]
        pos_feat = self.position_embedder(position, bound=self.bound)
        ambient_in = torch.cat([pos_feat, cond_feat.reshape(1, -1).expand(M, -1)], dim=1)
        ambient_pos = torch.tanh(self.ambient_net(ambient_in).float())
        ambient_feat = self.ambient_embedder(ambient_pos, bound=1)
        h = self.sigma_net(torch.cat([pos_feat, ambient_feat], dim=-1))
        return trunc_exp(h[..., 0]), h[..., 1:], ambient_pos

    def _fused_field_ok(self, position):
        """One-launch field (fused.field_forward) when nothing can ask for gradients and the kernels cover this model."""
        return (not torch.is_grad_enabled() and position.is_cuda and self.render_impl in ("auto", "fused")
                and self._pick_impl("auto", False, 1) == "fused")

    #: "auto": under autograd the field is ONE graph node (train_field.head_field: fused forward, hand-written backward) whenever the
    #: kernels cover this model; "ops": the reference's op-by-op torch graph.
    field_impl = "auto"
    #: under torch.autocast(float16): "f16" = the fused field on the f16 matrix pipe (train_field._HeadFieldAMP: the arithmetic of the
    #: reference's AMP training, base.yaml:49), "f32" = the exact-fp32 node even under autocast (custom_fwd(cast_inputs=float32), round 5)
    amp_field = "f16"
    amp_backward = "f16"      # the AMP node's dX chain: "f16" (k_field_backward16) or "f32" (the fp32 chain writing binary16 rows)

    def forward(self, position, direction, cond_feat, individual_code):
        if self._fused_field_ok(position):
            from .fused import field_forward
            return field_forward(self, position, direction, cond_feat, individual_code)
        if (torch.is_grad_enabled() and position.is_cuda and self.field_impl == "auto" and self.render_impl in ("auto", "fused")
                and self._pick_impl("auto", False, 1) == "fused"):
            from .train_field import head_field
            return head_field(self, position, direction, cond_feat, individual_code)
        sigma, geo_feat, ambient_pos = self._geometry(position, cond_feat)
        parts = [self.direction_embedder(direction), geo_feat]
        if individual_code is not None:
            parts.append(individual_code.reshape(1, -1).expand(position.shape[0], -1))
        color = torch.sigmoid(self.color_net(torch.cat(parts, dim=-1)))
        return sigma, color, ambient_pos

    def density(self, position, cond_feat, e=None):
        sigma, geo_feat, _ = self._geometry(position, cond_feat)
        return {"sigma": sigma, "geo_feat": geo_feat}
