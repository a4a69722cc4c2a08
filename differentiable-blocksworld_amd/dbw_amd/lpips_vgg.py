"""The perceptual criterion of the shipped configs: LPIPS with the VGG16 backbone, as the reference builds it in
src/model/loss.py:32-40 (`lpips.LPIPS(net='vgg')`, called with normalize=True, mean over the batch), written in plain torch so that it
runs on MIOpen convolutions next to the HIP render path -- outside of it, as SURVEY.md 8(a) A10 prescribes.

Neither the `lpips` package nor any weights exist in this environment, so REAL-WEIGHT PARITY IS UNPINNED.  What is pinned is the
architecture: oracle/lpips_ref.py restates the published forward of lpips 0.1.4 independently, tests/golden/lpips_random.npz freezes
it on seeded random weights, and this module reproduces that fixture -- forward and gradient, CPU and GPU -- when loaded with the same
weights (tests/test_lpips.py); the model's perceptual term is checked against the restatement too.  Sources of the constants:
  * the scaling layer constants and the layer taps (relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 of VGG16) are those of the published
    lpips 0.1.4 package (environment.yml:29), restated from its documentation;
  * `load_weights` takes the two state dicts a user has to bring: torchvision's `vgg16().features` ('0.weight', '0.bias', '2.weight', ...)
    and lpips' linear heads ('lin0.model.1.weight' ... 'lin4.model.1.weight', each (1, C, 1, 1), non-negative);
  * without weights the forward refuses to run unless `allow_random_init=True` (shape / throughput checks only).

Usage: `model.set_perceptual(LPIPSVGG().load_weights(vgg_sd, lin_sd).to(device))`."""
import torch
from torch import nn
import torch.nn.functional as F

# VGG16 `features`: index of every conv layer and its (in, out) channels; taps after the ReLU that follows the listed conv
_VGG16_CONVS = [(0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256),
                (17, 256, 512), (19, 512, 512), (21, 512, 512), (24, 512, 512), (26, 512, 512), (28, 512, 512)]
_POOL_BEFORE = {5, 10, 17, 24}                 # a 2x2 max-pool sits in front of these convs (features.4, 9, 16, 23)
_TAPS = {2: 0, 7: 1, 14: 2, 21: 3, 28: 4}      # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3
_CHANNELS = [64, 128, 256, 512, 512]


class LPIPSVGG(nn.Module):
    def __init__(self, allow_random_init=False):
        super().__init__()
        self.convs = nn.ModuleDict({str(i): nn.Conv2d(cin, cout, 3, padding=1) for i, cin, cout in _VGG16_CONVS})
        self.lins = nn.ModuleList([nn.Conv2d(c, 1, 1, bias=False) for c in _CHANNELS])
        self.register_buffer('shift', torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer('scale', torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))
        self.loaded = False
        self.allow_random_init = allow_random_init
        if allow_random_init:
            with torch.no_grad():
                for lin in self.lins:
                    lin.weight.abs_()                                # the published heads are non-negative
        for p in self.parameters():
            p.requires_grad = False                                  # loss.py:36-37

    def load_weights(self, vgg_features_state, lin_state):
        """vgg_features_state: state dict of torchvision `vgg16().features`; lin_state: lpips' `lin{k}.model.1.weight` tensors."""
        with torch.no_grad():
            for i, _, _ in _VGG16_CONVS:
                self.convs[str(i)].weight.copy_(vgg_features_state[f'{i}.weight'])
                self.convs[str(i)].bias.copy_(vgg_features_state[f'{i}.bias'])
            for k, lin in enumerate(self.lins):
                lin.weight.copy_(lin_state[f'lin{k}.model.1.weight'])
        self.loaded = True
        return self

    def features(self, x):
        x = (x - self.shift) / self.scale
        taps = []
        for i, _, _ in _VGG16_CONVS:
            if i in _POOL_BEFORE:
                x = F.max_pool2d(x, 2, 2)
            x = F.relu(self.convs[str(i)](x))
            if i in _TAPS:
                taps.append(x)
        return taps

    def forward(self, imgs, rec):
        """imgs, rec (B,3,H,W) in [0, 1] -> scalar: mean over the batch of sum_l mean_hw lin_l((n(f_l(a)) - n(f_l(b)))^2), n = unit
        normalisation along channels."""
        if not (self.loaded or self.allow_random_init):
            raise RuntimeError('LPIPSVGG has no weights: bring torchvision vgg16 features + lpips linear heads (load_weights)')
        a, b = self.features(imgs * 2 - 1), self.features(rec * 2 - 1)          # normalize=True
        total = 0
        for fa, fb, lin in zip(a, b, self.lins):
            na = fa / (fa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            nb = fb / (fb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            total = total + lin((na - nb) ** 2).mean((2, 3), keepdim=True)
        return total.mean()
