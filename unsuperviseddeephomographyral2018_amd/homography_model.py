"""HomographyModel -- host-side mirror of /root/reference/code/homography_model.py.

Same namedtuple, same constructor signature, same stage methods (build_model, solve_DLT, transform,
build_losses, build_summaries) and the same result attributes (pred_h4p, H_mat, pred_I2, h_loss,
rec_loss, ssim_loss, l1_loss, l1_smooth_loss, ncc_loss, bounded_h_loss, num_fail), but eager on
torch tensors (NHWC logical layout, as the reference's dataloader emits) instead of a TF1 graph.

What runs where
  * VGG regressor (homography_model.py:107-133): stock PyTorch-ROCm (MIOpen / hipBLASLt), channels_last.
  * solve_DLT, transform and all six losses of build_losses (:136-166, :169-269, :286-352), values and the gradient
    of the one being trained on: the HIP C-ABI library via ops.py -- no torch math (the supervised h_loss that
    carries a gradient, :288, is one torch expression on [B,8]).
"""
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

# homography_model.py:12-22
homography_model_params = namedtuple('parameters',
                                     'mode,'
                                     'batch_size,'
                                     'patch_size,'
                                     'img_w,'
                                     'img_h,'
                                     'loss_type,'
                                     'use_batch_norm,'
                                     'augment_list,'
                                     'leftright_consistent_weight,'
                                     )

LOSS_TYPES = ('h_loss', 'rec_loss', 'ssim_loss', 'l1_loss', 'l1_smooth_loss', 'ncc_loss')


_GAUSS_CACHE = {}


def fspecial_gauss(size, sigma, device=None):
    """MATLAB fspecial('gaussian') (homography_model.py:24-40); only used as the stored ssim_window.  Cached per
    device: a host->device copy is not allowed while the step is being captured into a hipGraph."""
    key = (size, sigma, str(device))
    if key not in _GAUSS_CACHE:
        x, y = np.mgrid[-size // 2 + 1:size // 2 + 1, -size // 2 + 1:size // 2 + 1]
        g = np.exp(-((x ** 2 + y ** 2) / (2.0 * sigma ** 2)))
        _GAUSS_CACHE[key] = torch.tensor(g / g.sum(), dtype=torch.float32, device=device)
    return _GAUSS_CACHE[key]


class VGGRegressor(nn.Module):
    """_vgg (homography_model.py:107-133): 4 blocks of 2x(pad1 + conv3x3 VALID + ReLU), channels
    64,64 / 64,64 / 128,128 / 128,128, 2x2/2 max-pool after blocks 1-3, dropout(0.5) on conv4 and
    fc1 when training, flatten (NHWC order) -> FC1024 ReLU -> FC8.  slim defaults: Xavier-uniform
    weights, zero biases.  Optional BN keeps the reference's quirk: is_training is passed as the
    *decay* (:93), i.e. decay=1 -> moving statistics never move (momentum 0 here)."""

    def __init__(self, patch_size=128, use_batch_norm=False, fused_epilogue=True, dropout_p=0.5):
        super().__init__()
        # slim.dropout keep_prob 0.5 (:120-121,128); a test that compares two ways of computing ONE gradient sets 0
        self.dropout_p = dropout_p
        # conv GEMMs on stock MIOpen; bias + ReLU (+ their backward and the bias gradient) as one HIP pass each
        self.fused_epilogue = fused_epilogue
        chans = [(2, 64), (64, 64), (64, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128)]
        self.convs = nn.ModuleList([nn.Conv2d(i, o, 3, stride=1, padding=1) for i, o in chans])
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.bns = nn.ModuleList([nn.BatchNorm2d(o, eps=1e-3, momentum=0.0) for _, o in chans])
            for bn in self.bns:                      # slim.batch_norm default: center=True, scale=False
                bn.weight.requires_grad_(False)
        feat = patch_size // 8
        self.fc1 = nn.Linear(feat * feat * 128, 1024)
        self.fc2 = nn.Linear(1024, 8)
        for m in list(self.convs) + [self.fc1, self.fc2]:
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)

    def _fusable(self, x):
        return self.fused_epilogue and not self.use_batch_norm and x.is_cuda and x.dtype == torch.float32

    def _conv_pool(self, x, i):
        """second conv of a block followed by the 2x2/2 max-pool (homography_model.py:109-117)"""
        if self._fusable(x) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            return ops.conv_bias_relu_pool(x, self.convs[i].weight, self.convs[i].bias, 1)
        return F.max_pool2d(self._conv(x, i), 2, 2)

    def _conv(self, x, i):
        if self._fusable(x):
            return ops.conv_bias_relu(x, self.convs[i].weight, self.convs[i].bias, 1)
        x = F.relu(self.convs[i](x))
        if self.use_batch_norm:
            x = self.bns[i](x)
        return x

    def forward(self, model_input_nhwc):
        # [B,P,P,2] NHWC -> NCHW-logical view with channels_last strides (no copy)
        x = model_input_nhwc.permute(0, 3, 1, 2)
        x = self._conv_pool(self._conv(x, 0), 1)
        x = self._conv_pool(self._conv(x, 2), 3)
        x = self._conv_pool(self._conv(x, 4), 5)
        x = self._conv(self._conv(x, 6), 7)
        x = F.dropout(x, self.dropout_p, self.training)
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)          # slim.flatten of NHWC
        x = F.dropout(F.relu(self.fc1(x)), self.dropout_p, self.training)
        return self.fc2(x)


# TF keeps variables in the graph's variable scope; `reuse_variables=True` means "the tower shares the
# variables created by the first tower".  Eager torch needs an owner: this registry plays that role.
_VARIABLE_SCOPE = {}


def get_variables(scope='model'):
    """The nn.Module holding the variables of variable_scope('model') (for the optimizer / DP)."""
    return _VARIABLE_SCOPE[scope]


def reset_variables():
    _VARIABLE_SCOPE.clear()


class HomographyModel(object):
    def __init__(self, args, I1, I2, I1_aug, I2_aug, I_aug, I_prime_aug, h4p, gt, patch_indices,
                 reuse_variables=None, model_index=0, net=None, fused_patch=False, solve_f64=False, graph_tail=False,
                 h4p_offset=None, zero_nonfinite_grad=None):
        self.params = args
        self.mode = args.mode
        self.is_training = True if self.mode == 'train' else False
        self.I1 = I1
        self.I2 = I2
        self.I1_aug = I1_aug
        self.I2_aug = I2_aug
        # I and I_prime are augmented by default
        self.I = I_aug
        self.I_prime = I_prime_aug
        self.pts_1 = h4p
        self.gt = gt
        self.use_batch_norm = args.use_batch_norm
        self.patch_indices = patch_indices
        self.reuse_variables = reuse_variables
        self.model_collection = ['model_' + str(model_index)]
        # MI355X-side switches (not in the reference): fused patch kernel for the l1 path, f64 DLT solve
        self.fused_patch = fused_patch
        self.solve_f64 = solve_f64
        self.h4p_offset = h4p_offset        # [B,8] added to the regressor's output (bench / test hook; None = off)
        # A pair whose predicted corners are degenerate (collinear p2 -> singular 8x8 system -> theta = NaN) must not turn
        # every variable into NaN: in training its d loss / d pred_h4p is zeroed (UH_DLT_ZERO_NONFINITE_GRAD).  The reference
        # has no such guard (tf.matrix_solve raises on a singular system); default = on in train mode.
        self.zero_nonfinite_grad = (args.mode == 'train') if zero_nonfinite_grad is None else bool(zero_nonfinite_grad)
        # one library call / one hipGraph launch for solve_DLT + transform + l1 loss and their backward (l1_loss only)
        self.graph_tail = bool(graph_tail) and args.loss_type == 'l1_loss'
        if args.loss_type not in LOSS_TYPES:
            raise ValueError('===> Loss type does not exist! ' + str(args.loss_type))
        # Constants used for the spatial transformer (homography_model.py:63-72): live in ops.m_and_minv
        self.M, self.M_inv = ops.m_and_minv(self.params.img_w, self.params.img_h)
        # batch_indices_tensor (:74-76) is folded into the gather kernel (per-image base offset)
        self.ssim_window = fspecial_gauss(size=3, sigma=0.5, device=I_aug.device)

        if net is not None:
            self.net = net
        elif reuse_variables:
            if 'model' not in _VARIABLE_SCOPE:
                raise ValueError("reuse_variables=True but variable_scope('model') holds no variables yet")
            self.net = _VARIABLE_SCOPE['model']
        else:
            self.net = VGGRegressor(args.patch_size, args.use_batch_norm).to(I_aug.device)
            self.net = self.net.to(memory_format=torch.channels_last)
            _VARIABLE_SCOPE['model'] = self.net
        self.net.train(self.is_training)
        # with the supervised loss the DLT/warp only feed stop_gradient monitors (:286-296)
        self._hot_grad = args.loss_type != 'h_loss' and torch.is_grad_enabled()

        self.build_model()
        self.solve_DLT()
        self.transform()
        self.build_losses()
        self.build_summaries()

    # ---- homography_model.py:354-361 ------------------------------------------------------------
    def build_model(self):
        self.model_input = torch.cat([self.I1_aug, self.I2_aug], 3)
        self.pts_1_tile = self.pts_1.unsqueeze(2)                  # BATCH_SIZE x 8 x 1
        self._vgg()

    def _vgg(self):
        self.pred_h4p = self.net(self.model_input)                # BATCH_SIZE x 8
        if self.h4p_offset is not None:
            self.pred_h4p = self.pred_h4p + self.h4p_offset

    # ---- homography_model.py:169-250 (+ the theta fold of :254) ------------------------------------
    def solve_DLT(self):
        self._tail = None
        if self.graph_tail:
            with torch.set_grad_enabled(self._hot_grad):
                self._tail = ops.photometric_tail(self.pts_1, self.pred_h4p, self.I, self.I2_aug, self.patch_indices,
                                                  self.params.patch_size, fused_patch=self.fused_patch, graph=True,
                                                  solve_f64=self.solve_f64, zero_nonfinite_grad=self.zero_nonfinite_grad)
            self.H_mat = self._tail[2]
            return
        with torch.set_grad_enabled(self._hot_grad):
            self.H_mat, self._theta = ops.solve_dlt(self.pts_1, self.pred_h4p, self.params.img_w,
                                                    self.params.img_h, solve_f64=self.solve_f64,
                                                    zero_nonfinite_grad=self.zero_nonfinite_grad)

    # ---- homography_model.py:252-269 ----------------------------------------------------------------
    def transform(self):
        with torch.set_grad_enabled(self._hot_grad):
            self._transform()

    def _transform(self):
        P = self.params.patch_size
        if self._tail is not None:              # everything was produced by the tail call in solve_DLT()
            self._l1_fused, self.pred_I2 = self._tail[0], self._tail[1]
            if self._tail[3].warped is not None:
                self.warped_images = self._tail[3].warped
            return
        out_size = (self.params.img_h, self.params.img_w)
        if self.fused_patch and self.params.loss_type == 'l1_loss':
            # one kernel: sample -> gray -> patch -> |.| -> mean, and d/dtheta; warped frame never exists
            self._l1_fused, self.pred_I2 = ops.warp_patch_l1(self.I, self._theta, self.I2_aug,
                                                             self.patch_indices, P)
            return
        self._l1_fused = None
        self._mon = None
        if self.fused_patch:
            # the other photometric losses need a global norm before their gradient exists: sample the patch only
            # (no warped frame), let the loss kernels produce dPred, then the sparse warp backward
            self.pred_I2 = ops.warp_patch(self.I, self._theta, self.I2_aug, self.patch_indices, P)
            return
        # the reference's path: the full warped frame is materialised (:257), then gray + gather (:263-269) and the six
        # loss values (:286-352) -- as ONE autograd node (3 launches forward + backward; the loss gradient and the sparse
        # warp backward are one kernel, no 79 %-zero gradient frame, no dPred tensor)
        lt = self.params.loss_type
        train = lt if (torch.is_grad_enabled() and lt != 'h_loss') else None
        self.warped_images, self.pred_I2, self._mon = ops.warp_gather_losses(
            self.I, self._theta, self.patch_indices, P, self.I2_aug, self.pred_h4p, self.gt, train=train)

    # ---- homography_model.py:271-352 ----------------------------------------------------------------
    def build_losses(self):
        I2 = self.I2_aug
        lt = self.params.loss_type

        if self.params.mode == 'test':
            with torch.no_grad():
                batch_h_loss = torch.sqrt(torch.mean((self.pred_h4p - self.gt) ** 2, dim=1))
                h_loss_identity = torch.sqrt(torch.mean(self.gt ** 2, dim=1))
                is_failure = (batch_h_loss >= h_loss_identity).float()
                self.num_fail = torch.sum(is_failure)
                # If it is a fail, use identity matrix
                self.bounded_h_loss = torch.mean(batch_h_loss * (1 - is_failure) + is_failure * h_loss_identity)

        pred = self.pred_I2
        # ONE launch yields all six values (csrc/uh_losses.hip); the loss being trained on carries its gradient through
        # uh_patch_loss_backward (HIP), every other one is a stop_gradient monitor (homography_model.py:286-352).
        grad = torch.is_grad_enabled()
        train = lt if (grad and lt != 'h_loss' and not (lt == 'l1_loss' and self._l1_fused is not None)) else None
        mon = getattr(self, '_mon', None)           # already produced by transform() on the default path
        if mon is None:
            mon = ops.patch_losses(pred, I2, self.pred_h4p, self.gt, train=train)
        self.h_loss, self.rec_loss, self.ssim_loss = mon[5], mon[0], mon[1]
        self.l1_loss, self.l1_smooth_loss, self.ncc_loss = mon[2], mon[3], mon[4]
        if not grad:
            return
        if lt == 'h_loss':
            self.h_loss = torch.sqrt(torch.mean((self.pred_h4p - self.gt) ** 2))
        elif lt == 'l1_loss' and self._l1_fused is not None:
            # fused patch kernel / tail graph: loss and d loss / d theta came out of the sampling pass itself
            self.l1_loss = self._l1_fused

    @property
    def loss(self):
        """The tensor the trainer differentiates (homography_CNN_synthetic.py:251-265)."""
        return getattr(self, self.params.loss_type)

    # ---- homography_model.py:363-376 ----------------------------------------------------------------
    def build_summaries(self):
        # tf.summary.image(..., max_outputs=1): keep references to the first sample of each stream
        self.summaries = {
            'I': self.I[:1], 'I_prime': self.I_prime[:1] if self.I_prime is not None else None,
            'I1_aug': self.I1_aug[:1], 'I2_aug': self.I2_aug[:1], 'I1': self.I1[:1], 'I2': self.I2[:1],
            'pred_I2': self.pred_I2[:1].detach(),
        }
