"""oracle/lpips_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the perceptual criterion the reference builds in src/model/loss.py:32-40:
    lpips.LPIPS(net='vgg', verbose=False)(imgs, rec, normalize=True).mean()
The algorithm lives in a third-party dependency that is ABSENT from /root/reference and from this image: lpips==0.1.4
(environment.yml:29), with the ImageNet weights of torchvision's vgg16 and the package's trained linear heads.  This file restates the
published forward of that release (lpips/lpips.py: LPIPS.forward, ScalingLayer, NetLinLayer, normalize_tensor, spatial_average;
lpips/pretrained_networks.py: vgg16 slices) as plain functions over a weight dictionary:

  1. normalize=True:            x <- 2 x - 1                                  (inputs in [0, 1])
  2. ScalingLayer (version 0.1): x <- (x - shift) / scale, shift = (-.030, -.088, -.188), scale = (.458, .448, .450)
  3. VGG16 `features` (torchvision configuration D: 64 64 M 128 128 M 256 256 256 M 512 512 512 M 512 512 512 M, 3x3 convs, padding 1,
     ReLU after every conv, 2x2/2 max-pool at M), tapped after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (slices [0:4], [4:9],
     [9:16], [16:23], [23:30] of `features`)
  4. normalize_tensor:          f <- f / (sqrt(sum_c f^2) + 1e-10)            per pixel, over channels
  5. per tap: lin_k((f0 - f1)^2) with lin_k a 1x1 convolution C_k -> 1 without bias (the Dropout in front of it is the identity in eval
     mode, which is how the package builds the model), spatial mean, summed over the five taps -> (N, 1, 1, 1)
  6. the reference takes .mean() over the batch.

PARITY UNPINNED for the real criterion: neither the package nor any of its weights exist offline, so nothing here can be checked against
lpips itself.  What this file pins is the ARCHITECTURE: tests/golden/lpips_random.npz (made by tests/golden/make_lpips_fixture.py from
this restatement with seeded random weights) freezes steps 1-6, and dbw_amd/lpips_vgg.py -- the module a user loads real weights into --
has to reproduce it, forward and gradient, with the same weights.  Only tests/ may import this file."""
import numpy as np
import torch
import torch.nn.functional as F

VGG16_D = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
TAP_AFTER_FEATURE_INDEX = (3, 8, 15, 22, 29)          # the ReLUs that end slices 1-5
LIN_CHANNELS = (64, 128, 256, 512, 512)
SHIFT = (-.030, -.088, -.188)
SCALE = (.458, .448, .450)


def feature_layout():
    """-> [(index in `features`, kind, cin, cout)] of torchvision's vgg16().features (conv / relu / pool)."""
    layers, cin, i = [], 3, 0
    for v in VGG16_D:
        if v == 'M':
            layers.append((i, 'pool', cin, cin)); i += 1
        else:
            layers.append((i, 'conv', cin, v)); layers.append((i + 1, 'relu', v, v)); i += 2
            cin = v
    return layers


def random_weights(seed, dtype=torch.float32):
    """Seeded stand-in for the absent weights, in the two state-dict layouts a user brings: torchvision's `vgg16().features`
    ('0.weight', '0.bias', '2.weight', ...) and lpips' heads ('lin0.model.1.weight' ... (1, C, 1, 1), non-negative like the trained
    ones).  He-scaled so that the activations neither vanish nor blow up through 13 layers."""
    g = torch.Generator().manual_seed(seed)
    vgg, lin = {}, {}
    for i, kind, cin, cout in feature_layout():
        if kind == 'conv':
            vgg[f'{i}.weight'] = (torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) * (2.0 / (9 * cin)) ** 0.5).to(dtype)
            vgg[f'{i}.bias'] = (torch.randn(cout, generator=g, dtype=torch.float64) * 0.05).to(dtype)
    for k, c in enumerate(LIN_CHANNELS):
        lin[f'lin{k}.model.1.weight'] = (torch.rand(1, c, 1, 1, generator=g, dtype=torch.float64) / c * 4.0).to(dtype)
    return vgg, lin


def vgg16_taps(x, vgg):
    taps = []
    for i, kind, _, _ in feature_layout():
        if kind == 'conv':
            x = F.conv2d(x, vgg[f'{i}.weight'], vgg[f'{i}.bias'], stride=1, padding=1)
        elif kind == 'relu':
            x = torch.clamp(x, min=0)
        else:
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        if i in TAP_AFTER_FEATURE_INDEX:
            taps.append(x)
    return taps


def lpips_vgg(in0, in1, vgg, lin, normalize=True):
    """-> (N, 1, 1, 1), as lpips.LPIPS(net='vgg').forward(in0, in1, normalize=normalize) returns it."""
    if normalize:
        in0, in1 = 2 * in0 - 1, 2 * in1 - 1
    shift = torch.tensor(SHIFT, dtype=in0.dtype)[None, :, None, None]
    scale = torch.tensor(SCALE, dtype=in0.dtype)[None, :, None, None]
    f0, f1 = vgg16_taps((in0 - shift) / scale, vgg), vgg16_taps((in1 - shift) / scale, vgg)
    val = 0
    for k in range(5):
        n0 = f0[k] / (torch.sqrt(torch.sum(f0[k] ** 2, dim=1, keepdim=True)) + 1e-10)
        n1 = f1[k] / (torch.sqrt(torch.sum(f1[k] ** 2, dim=1, keepdim=True)) + 1e-10)
        d = (n0 - n1) ** 2
        val = val + F.conv2d(d, lin[f'lin{k}.model.1.weight']).mean([2, 3], keepdim=True)
    return val


def lpips_loss(imgs, rec, vgg, lin):
    """src/model/loss.py:39-40"""
    return lpips_vgg(imgs, rec, vgg, lin, normalize=True).mean()
